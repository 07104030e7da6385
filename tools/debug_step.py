import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import sgan_oracle as O
from test_hip_step import build_model, step1_with_captures, real3

for name, kw in [("fcgan_step_full.npz", dict(n_update_G=2)), ("fcgan_step_full_nug1.npz", dict(n_update_G=1))]:
    g = np.load(os.path.join(ROOT, "tests/golden", name))
    cfg = O.FCGANConfig(**kw)
    m = build_model(cfg, int(g["n_init_noise_draws"]))
    cap = step1_with_captures(m, real3(cfg, 0))
    print(name, "fake crop rel", O.rel_err(cap["fake"][:, :, :64, :64], torch.as_tensor(g["step1/fake_crop"])),
          "lossD", np.abs(np.asarray(cap["loss_D"]) - g["step1/loss_D"]).max(), "lossG", abs(cap["loss_G"] - float(g["step1/loss_G"])))
    items = [(f"step1/gradD_{i}", k, v) for i, gd in enumerate(cap["gradD"]) for k, v in gd.items()] + \
            [("step1/gradG", k, v) for k, v in cap["gradG"].items()]
    for prefix, k, v in items:
        ref = g[f"{prefix}/sample/{k}"]; summ = g[f"{prefix}/summary/{k}"]
        flat = v.reshape(-1)
        smp = flat[torch.from_numpy(O.grad_sample_idx(flat.numel()))].double().numpy()
        l2 = np.linalg.norm(smp - ref) / (np.linalg.norm(ref) + 1e-30)
        mx = np.abs(smp - ref).max() / (summ[1] + 1e-30)
        print(f"  {prefix:14s} {k:18s} relL2 {l2:.2e}  max/absmax {mx:.2e}  |g|max {summ[1]:.2e}")
