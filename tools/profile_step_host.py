"""Dev tool: where the host time of the synchronised HipTrainer.step() goes (cProfile over 300 steps)."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bench_support.standin_trainer import HipTrainer, FrameData
from isdf_amd import synthetic
cfg = bench.reference_config()
cam = dict(synthetic.REPLICA_CAM)
depth, normal, T = bench.make_keyframes(cam, 5)
tr = HipTrainer("cuda:0", cfg, inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=1)
dev = tr.device
tr.frames = FrameData(frame_id=np.arange(5), depth_batch=torch.from_numpy(depth).to(dev), T_WC_batch=torch.from_numpy(T).to(dev),
                      normal_batch=torch.from_numpy(normal).to(dev), frame_avg_losses=torch.zeros(5, device=dev))
tr.noise_std = tr.noise_kf
for _ in range(300):
    tr.step()
t0 = time.perf_counter()
for _ in range(300):
    tr.step()
print("wall per step %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    tr.step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
