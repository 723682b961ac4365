"""Dev tool: where does the host time of a device-synchronised HipTrainer.step() go?  Wraps the two C-ABI launches of the step and the
stream synchronisation with perf_counter stamps (0.3 us each) and prints the median of every interval over 400 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bench_support.standin_trainer import HipTrainer, FrameData
from isdf_amd import synthetic, hot_path

cam = dict(synthetic.REPLICA_CAM)
cfg = bench.reference_config()
cfg["dataset"]["camera"] = {"w": cam["W"], "h": cam["H"], "fx": cam["fx"], "fy": cam["fy"], "cx": cam["cx"], "cy": cam["cy"]}
F = cfg["model"]["window_size"]
depth, normal, T = bench.make_keyframes(cam, F)
tr = HipTrainer("cuda", cfg, inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=1)
dev = tr.device
tr.frames = FrameData(frame_id=np.arange(F), depth_batch=torch.from_numpy(depth).to(dev), T_WC_batch=torch.from_numpy(T).to(dev),
                      normal_batch=torch.from_numpy(normal).to(dev), frame_avg_losses=torch.zeros(F, device=dev))
eng = tr.engine
for _ in range(300):
    tr.step()
stamps = []
pc = time.perf_counter


def wrap(fn, name):
    def w(*a):
        stamps.append((name + " in", pc()))
        r = fn(*a)
        stamps.append((name + " out", pc()))
        return r
    return w


eng.lib.isdf_sample_rays = wrap(eng.lib.isdf_sample_rays, "sample")
eng.lib.isdf_train_step_adamw = wrap(eng.lib.isdf_train_step_adamw, "step")
_sync = torch.cuda.Stream.synchronize


def sync(self):
    stamps.append(("sync in", pc()))
    _sync(self)
    stamps.append(("sync out", pc()))


torch.cuda.Stream.synchronize = sync
_rec = torch.cuda.Event.record


def rec(self, stream=None):
    stamps.append(("record in", pc()))
    _rec(self, stream) if stream is not None else _rec(self)
    stamps.append(("record out", pc()))


torch.cuda.Event.record = rec
_el = torch.cuda.Event.elapsed_time


def el(self, other):
    stamps.append(("elapsed in", pc()))
    r = _el(self, other)
    stamps.append(("elapsed out", pc()))
    return r


torch.cuda.Event.elapsed_time = el
rows = []
for _ in range(400):
    stamps.clear()
    t0 = pc()
    tr.step()
    t1 = pc()
    rows.append([("enter", t0)] + list(stamps) + [("return", t1)])
names = [n for n, _ in rows[0]]
assert all([n for n, _ in r] == names for r in rows), "the stamp sequence varies"
ts = np.array([[t for _, t in r] for r in rows])
d = np.diff(ts, axis=1) * 1e6
print("device-synchronised step: %.1f us mean, %.1f median" % ((ts[:, -1] - ts[:, 0]).mean() * 1e6, np.median(ts[:, -1] - ts[:, 0]) * 1e6))
for i in range(d.shape[1]):
    print("  %-12s -> %-12s  median %6.1f us   mean %6.1f" % (names[i], names[i + 1], np.median(d[:, i]), d[:, i].mean()))
