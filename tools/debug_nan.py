"""Dev tool: run one train step on the small golden case and report NaN / mismatch per gradient key."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_parity as T
import golden_util as gu
g = gu.load("eval_full_ray")
eng = T._engine(g)
lc, sc = T._cfgs(g)
s = T._sample_hip(eng, g, sc)
R = g["depth_sample"].shape[0]
noise = g["draw_noise"].reshape(R, -1) * np.float32(g["noise_std"][0])
dbg = eng.train_step(s, lc, sc, noise=T._dev(noise), debug=True)
torch.cuda.synchronize()
for k in gu.params_of(g):
    v = eng.grad_view(k).cpu().numpy()
    print(k, v.shape, "nan", int(np.isnan(v).sum()), "absmax", float(np.nanmax(np.abs(v))))
    if v.ndim == 2 and np.isnan(v).any():
        r, c = np.where(np.isnan(v)); print("   nan rows", np.unique(r)[:20], "cols", np.unique(c)[:40], len(np.unique(c)))
ws = eng._ws
print("ws nan check skipped; spill stats:")
L = 6; HD = 256; TP = 64
nT = (R * sc.S + TP - 1) // TP
names = ["A%d" % i for i in range(L + 1)] + ["P%d" % i for i in range(L)] + ["GB%d" % i for i in range(L)] + ["INJ%d" % i for i in range(L)] + ["ZB%d" % i for i in range(L)]
stride = len(names) * TP * HD
sp = ws[: nT * stride * 2].view(torch.bfloat16).view(nT, len(names), TP * HD).float().cpu().numpy()
for k, nm in enumerate(names):
    x = sp[:, k]
    bad = ~np.isfinite(x)
    msg = ""
    if bad.any():
        t, i = np.where(bad)
        piece = i // 8; lane = piece % 64; c = piece // 64; e = i % 8
        msg = " BAD tiles %s chunk %s lane %s e %s" % (np.unique(t)[:8], np.unique(c), np.unique(lane)[:16], np.unique(e))
    print(nm, "absmax %.3g" % np.nanmax(np.abs(np.where(bad, 0, x))), "nbad", int(bad.sum()), msg)
