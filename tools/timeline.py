"""Dev tool: print the in-kernel phase timeline of one chain workgroup (ISDF_DEBUG_TIMELINE=1)."""
import os, sys
os.environ["ISDF_DEBUG_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
ts = eng._ws[-4096:].view(torch.int64).cpu().numpy()
ts = ts[ts > 0]
t0 = ts[0]
print("n stamps", len(ts), "(s_memtime ticks; 100 MHz constant clock => x10 ns)")
prev = t0
for i, t in enumerate(ts):
    print("%3d  t=%8d  d=%7d" % (i, t - t0, t - prev)); prev = t
