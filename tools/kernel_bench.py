"""Dev tool: time the chain kernel's modes (forward / forward+grad / train) and the other step kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
from isdf_amd import synthetic

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 27000
    eng = Engine(NetConfig(transform=synthetic.bounds_transform()), "cuda")
    torch.manual_seed(0)
    eng.params.normal_(0, 0.06); eng.pack()
    x = (torch.rand(npts, 3, device="cuda") * 4 - 2)
    print("points", npts)
    print("fwd only      %8.1f us" % timeit(lambda: eng.sdf_eval(x)))
    print("fwd+grad      %8.1f us" % timeit(lambda: eng.sdf_eval(x, want_grad=True)))
    cam = dict(synthetic.SCANNET_CAM)
    d, n, T = synthetic.keyframes(5, cam, seed=1)
    dev = lambda a: torch.as_tensor(a).cuda()
    d, n, T = dev(d), dev(n), dev(T)
    sc = SampleConfig(n_rays=npts // 27 // 5, **cam); lc = LossConfig()
    idx = torch.arange(5, dtype=torch.int32, device="cuda")
    s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
    print("sampler       %8.1f us" % timeit(lambda: eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)))
    noise = torch.zeros(s["max_rays"], sc.S, device="cuda")
    print("train_step    %8.1f us" % timeit(lambda: eng.train_step(s, lc, sc, noise=noise)))
    print("adamw+pack    %8.1f us" % timeit(lambda: eng.adamw()))
    print("frame_avg     %8.1f us" % timeit(lambda: eng.frame_avg(5)))

if __name__ == "__main__":
    main()
