"""Dev tool: print the in-kernel phase timeline of one chain workgroup.  Needs the instrumented build:
    python tools/build_variants.py dbg="-DISDF_DEBUG_HOOKS=1"   ->  variants/lib_dbg.so   (used automatically)"""
import os, sys
os.environ["ISDF_DEBUG_TIMELINE"] = "1"
_dbg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "lib_dbg.so")
if not os.environ.get("ISDF_HIP_LIB") and os.path.exists(_dbg):
    os.environ["ISDF_HIP_LIB"] = _dbg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
from isdf_amd import synthetic
eng = Engine(NetConfig(transform=synthetic.bounds_transform(), fwd_operand=os.environ.get("ISDF_FWD_OPERAND", "fp16x2")), "cuda")
torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
cam = dict(synthetic.SCANNET_CAM)
d, n, T = synthetic.keyframes(5, cam, seed=1)
dev = lambda a: torch.as_tensor(a).cuda()
d, n, T = dev(d), dev(n), dev(T)
sc = SampleConfig(n_rays=int(os.environ.get("ISDF_TIMELINE_RAYS", "200")), **cam); lc = LossConfig()
idx = torch.arange(5, dtype=torch.int32, device="cuda")
s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
noise = torch.zeros(s["max_rays"], sc.S, device="cuda")
for _ in range(3):
    eng.train_step(s, lc, sc, noise=noise)
torch.cuda.synchronize()
raw = eng._ws[-4096:].view(torch.int64).cpu().numpy()
ts = raw[:128]; ts = ts[ts > 0]
t0 = ts[0]
print("n stamps", len(ts), "(s_memtime ticks = shader clock cycles)")
prev = t0
for i, t in enumerate(ts):
    print("%3d  t=%8d  d=%7d" % (i, t - t0, t - prev)); prev = t
ts7 = raw[384:512]; ts7 = ts7[ts7 > 0]
if len(ts7) == len(ts):
    print("last wave of the workgroup, same stamps (d = its own step; lag = behind wave 0):")
    prev7 = ts7[0]
    for i, t in enumerate(ts7):
        print("%3d  t=%8d  d=%7d  lag=%6d" % (i, t - t0, t - prev7, t - ts[i])); prev7 = t
# wall-clock start/end (100 MHz s_memrealtime) of every 4th workgroup
se = raw[128:128 + 2 * 190].reshape(-1, 2)
se = se[(se[:, 0] > 0) & (se[:, 1] > 0)]
if len(se):
    b = se[:, 0].min()
    st, en = (se[:, 0] - b) / 100.0, (se[:, 1] - b) / 100.0
    print("workgroups sampled %d: start us min/med/max %.1f %.1f %.1f  end us min/med/max %.1f %.1f %.1f  duration us min/med/max %.1f %.1f %.1f"
          % (len(se), st.min(), float(sorted(st)[len(st) // 2]), st.max(), en.min(), float(sorted(en)[len(en) // 2]), en.max(),
             (en - st).min(), float(sorted(en - st)[len(st) // 2]), (en - st).max()))
    idx4 = [4 * i for i in range(len(se))]
    grp = lambda lo, hi: [e for i, e in zip(idx4, en) if lo <= i < hi]
    nt = int(-(-int(s["n_valid"].item()) * sc.S // 64))
    second = nt - 256
    if second > 0:
        import numpy as np
        print("SUMMARY tiles %d: first-of-pair end %.1f us, lone end %.1f us, second-of-pair end %.1f us (max %.1f)" % (
            nt, np.mean(grp(0, second)), np.mean(grp(second, 256)), np.mean(grp(256, nt)), max(en)))
    if os.environ.get("TIMELINE_BRIEF"):
        sys.exit(0)
    k = list(se[:, 0]).index(se[:, 0].min())
    for i in range(0, len(se), max(1, len(se) // 24)):
        print("  wg %4d  start %7.1f  end %7.1f" % (4 * i, st[i], en[i]))
