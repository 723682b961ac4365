"""Dev tool: in-kernel stage timeline of one workgroup of the pair-tile chain kernel (instrumented build:
python tools/build_variants.py dbg="-DISDF_DEBUG_HOOKS=1" -> variants/lib_dbg.so, used automatically)."""
import os, sys
os.environ["ISDF_DEBUG_TIMELINE"] = "1"
os.environ["ISDF_CHAIN_PAIR"] = "1"
_dbg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "lib_dbg.so")
if not os.environ.get("ISDF_HIP_LIB") and os.path.exists(_dbg):
    os.environ["ISDF_HIP_LIB"] = _dbg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
from isdf_amd import synthetic
eng = Engine(NetConfig(transform=synthetic.bounds_transform()), "cuda")
torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
cam = dict(synthetic.SCANNET_CAM)
d, n, T = synthetic.keyframes(5, cam, seed=1)
dev = lambda a: torch.as_tensor(a).cuda()
d, n, T = dev(d), dev(n), dev(T)
sc = SampleConfig(n_rays=200, **cam); lc = LossConfig()
idx = torch.arange(5, dtype=torch.int32, device="cuda")
s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
noise = torch.zeros(s["max_rays"], sc.S, device="cuda")
for _ in range(3):
    eng.train_step(s, lc, sc, noise=noise)
torch.cuda.synchronize()
ts = eng._ws[-4096:-4096 + 1024].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
n = int((ts != 0).sum())
ts = ts[:n]
d = np.diff(ts) % (1 << 32)
print("stamps", n, "total cycles", int(d.sum()))
# layout: 0 start | PE,barrier -> 1 | A0 spill -> 2 | fwd prologue G -> 3, barrier -> 4 | then per stage: G, E, barrier
names = ["PE+barrier", "spill A0", "G prologue", "barrier"]
print(" ".join("%s=%d" % (a, b) for a, b in zip(names, d[:4])))
rest = d[4:]
i = 0; k = 0
tot = dict(G=0, E=0, B=0)
while i + 3 <= len(rest) and k < 200:
    g, e, b = rest[i:i + 3]
    print("stage %2d  E %6d  G %6d  barrier %6d" % (k, g, e, b))
    tot["E"] += g; tot["G"] += e; tot["B"] += b
    i += 3; k += 1
print("tail", rest[i:])
print("sums", tot)
