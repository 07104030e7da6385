"""Which Python lines make ops.as_nhwc COPY (sg_to_nhwc launches) / call concat_nhwc / slice_nhwc during one eager cgan step (diagnostic):
python tools/find_layout_copies.py"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from supervised_gan_amd import _lib, ops

ns = argparse.Namespace(n_update_G=2, skip_wasted_D_wgrad=False, no_d_streams=True, no_group=False)
torch.cuda.set_device(0)
m = bench.build_cgan(ns, 0)
ring = bench.synthetic_ring(4, 0, torch.device("cuda", 0))
for i in range(2):
    m.set_input(ring[i]); m.optimize_parameters()
seen = collections.Counter()
lib = _lib.lib()


def spy(name):
    real = getattr(lib, name)

    def f(*a):
        fr = [x for x in traceback.extract_stack()[:-1] if "tools/" not in x.filename]
        extra = f" strides(c,h,w)={a[1]},{a[2]},{a[3]} HxWxC={a[4]}x{a[5]}x{a[6]}" if name == "sgan_to_nhwc" else ""
        seen[(name, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-5:]) + extra)] += 1
        return real(*a)
    return f


class Proxy:
    def __getattr__(self, k):
        return spy(k) if k in ("sgan_to_nhwc", "sgan_concat_nhwc", "sgan_slice_nhwc") else getattr(lib, k)


_lib.lib = lambda: Proxy()
_real_as_nhwc = ops.as_nhwc


def as_nhwc_why(t):
    out = _real_as_nhwc(t)
    if out.data_ptr() != t.data_ptr() and t.dim() == 4 and t.stride(1) == 1:      # NHWC-strided, yet copied: why?
        ent = ops._VIEW_REGISTRY.get(t.data_ptr())
        fr = [x for x in traceback.extract_stack()[:-1] if "tools/" not in x.filename]
        why = "no registry entry" if ent is None else ("buffer object gone" if ent[0]() is None else f"entry C={ent[1]} shape={tuple(ent[0]().shape)}")
        seen[("as_nhwc miss: " + why + f" requires_grad={t.requires_grad} base={t._base is not None}",
              " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-7:]))] += 1
    return out


ops.as_nhwc = as_nhwc_why
m.set_input(ring[3]); m.optimize_parameters()
torch.cuda.synchronize()
for (n, w), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} {n:18s} {w}")
