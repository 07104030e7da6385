#!/bin/bash
# effective shader clock of one conv launch = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration  (tools/clock_probe.sh <tag> <pmc_one.py args>)
TAG=$1; shift
OUT=gpurun_out/clk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 tools/pmc_one.py "$@" 20 > $OUT/p.log 2>&1
python3 - <<PY
import csv, glob
kt = glob.glob("$OUT/p/**/*kernel_trace.csv", recursive=True)
cc = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "sg_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
        dur.setdefault(r["Kernel_Name"][:50], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
gui = {}
for r in csv.DictReader(open(cc[0])):
    if "sg_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui.setdefault(r["Kernel_Name"][:50], []).append(float(r["Counter_Value"]))
for k in dur:
    d = sorted(dur[k])[len(dur[k]) // 2]
    g = sorted(gui[k])[len(gui[k]) // 2]
    print(f"$TAG {k}: median {d:.1f} us, GUI_ACTIVE/8 = {g / 8:.0f} cycles -> {g / 8 / d / 1e3:.2f} GHz")
PY
