"""Dev tool: one training step through the one-tile chain kernel (default) and through the pair-tile
kernel (ISDF_CHAIN_PAIR=1) on identical inputs; reports the first differences per output, gradient tensor and spill tensor."""
import os, sys
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
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = SampleConfig(n_rays=nr, **cam); lc = LossConfig()
idx = torch.arange(5, dtype=torch.int32, device="cuda")
s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
noise = 0.01 * torch.randn(s["max_rays"], sc.S, device="cuda")
R = int(s["n_valid"].item()); P = R * sc.S
L = 6; names = ["A%d" % i for i in range(L + 1)] + ["S1_%d+%d" % (2 * i, 2 * i + 1) for i in range(L // 2)] + ["P%d" % i for i in range(L)] + ["GB%d" % i for i in range(L)] + \
    ["INJ%d" % i for i in range(L)] + ["ZB%d" % i for i in range(L)]
nT = (P + 63) // 64
def run(single):
    os.environ["ISDF_CHAIN_PAIR"] = "0" if single else "1"
    eng._ws = None
    ws = eng.workspace(s["max_rays"] * sc.S, True); ws.zero_()
    dbg = eng.train_step(s, lc, sc, noise=noise, debug=True)
    torch.cuda.synchronize()
    sp = ws[:nT * len(names) * 64 * 256 * 2].view(torch.bfloat16).float().view(nT, len(names), 64 * 256).cpu().numpy()
    return dict(sdf=dbg["sdf"].cpu().numpy(), sdf_grad=dbg["sdf_grad"].cpu().numpy(), tot=dbg["tot_loss_mat"].cpu().numpy(),
                red=eng.reduce_buf.clone().cpu().numpy(), sp=sp)
a, b = run(True), run(False)
rel = lambda x, y: float(np.linalg.norm(x - y) / (np.linalg.norm(x) + 1e-30))
print("points", P, "tiles", nT)
for k in ("sdf", "sdf_grad", "tot"):
    print("%-9s rel %.3e" % (k, rel(a[k], b[k])))
print("loss sums", a["red"][eng.n_params:eng.n_params + 5], b["red"][eng.n_params:eng.n_params + 5])
for k, (off, shp) in eng.slices.items():
    n_ = int(np.prod(shp))
    print("grad %-22s rel %.3e" % (k, rel(a["red"][off:off + n_], b["red"][off:off + n_])))
for ti, nm in enumerate(names):
    x, y = a["sp"][:, ti], b["sp"][:, ti]
    ev, od = rel(x[0::2], y[0::2]), rel(x[1::2], y[1::2])
    flag = "" if max(ev, od) < 2e-2 else "   <<<<<<"
    print("spill %-5s even tiles rel %.3e   odd tiles rel %.3e%s" % (nm, ev, od, flag))
