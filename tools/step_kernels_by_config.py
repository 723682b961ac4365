"""Dev tool: HIP-event time of the step's kernels (chain, dW, tail) for a few network configurations -- the shapes bench.py has no flag
for (the realsense nets: eleven / nine octaves, three hidden layers per block).  usage: python tools/step_kernels_by_config.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
from isdf_amd import synthetic

cam = dict(synthetic.SCANNET_CAM)
d, n, T = synthetic.keyframes(5, cam, seed=1)
dev = lambda a: torch.as_tensor(a).cuda()
d, n, T = dev(d), dev(n), dev(T)
idx = torch.arange(5, dtype=torch.int32, device="cuda")
for name, kw in (("replicaCAD / scannet: 6 oct, 2 per block", dict()), ("realsense(_franka): 9 oct, 2 per block", dict(n_freqs=9)),
                 ("realsense_franka_offline: 11 oct, 3 per block", dict(n_freqs=11, blocks=3)),
                 ("(11 oct, 2 per block)", dict(n_freqs=11)), ("(9 oct, 3 per block)", dict(n_freqs=9, blocks=3)), ("16-bit spills", dict(spill_operand="16bit")),
                 ("bf16", dict(fwd_operand="bf16"))):
    eng = Engine(NetConfig(transform=synthetic.bounds_transform(), **kw), "cuda")
    torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
    sc = SampleConfig(n_rays=200, **cam); lc = LossConfig()
    s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
    opt = dict(lr=0.0013, weight_decay=0.012, betas=(0.9, 0.999), eps=1e-8)
    for _ in range(20):
        eng.train_step(s, lc, sc, noise_std=0.0, optim=opt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        eng.train_step(s, lc, sc, noise_std=0.0, optim=opt)
    e1.record(); torch.cuda.synchronize()
    print("%-46s chain + dW + tail %.1f us per step" % (name, e0.elapsed_time(e1) * 10))
