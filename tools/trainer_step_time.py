"""Dev tool: wall time of HipTrainer.step() itself (device-synchronised per step, as the reference's
metrics.start_timing/end_timing measure it, metrics.py:13-38) on the bench workload, next to the
pipelined rate bench.py reports.  Also prints a cProfile of the host side."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bench_support.standin_trainer import HipTrainer, FrameData
from isdf_amd import synthetic

cfg = bench.reference_config()
cam = dict(synthetic.REPLICA_CAM)
F = cfg["model"]["window_size"]
depth, normal, T = bench.make_keyframes(cam, F)
torch.manual_seed(1); np.random.seed(1)
tr = HipTrainer("cuda:0", cfg, incremental=True, inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=1)
dev = tr.device
tr.frames = FrameData(frame_id=np.arange(F), depth_batch=torch.from_numpy(depth).to(dev),
                      T_WC_batch=torch.from_numpy(T).to(dev), normal_batch=torch.from_numpy(normal).to(dev),
                      frame_avg_losses=torch.zeros(F, device=dev))
tr.noise_std = tr.noise_kf
for _ in range(30):
    tr.step()
N = 300
t0 = time.perf_counter(); ms = 0.0
for _ in range(N):
    losses, st = tr.step(); ms += st
wall = time.perf_counter() - t0
print("HipTrainer.step(): wall %.1f us/step (%.0f steps/s), device-timed part %.1f us" % (wall / N * 1e6, N / wall, ms / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    tr.step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
