"""`engine.pinned_stream` (the step's launch stream, looked up once per step): the cached Stream object must follow the caller's
`torch.cuda.stream(...)` context -- the cache is keyed on the device's RAW current-stream handle, not on the device alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pinned_stream_follows_the_callers_stream_context():
    from isdf_amd import engine
    dev = torch.device("cuda", 0)
    with engine.pinned_stream(0) as st0:
        p0 = engine._stream(dev).value or 0
        assert st0.cuda_stream == p0 == torch.cuda.current_stream(0).cuda_stream
    side = torch.cuda.Stream(0)
    with torch.cuda.stream(side):
        with engine.pinned_stream(dev) as st1:                 # (device given as torch.device: the slow constructor path)
            assert st1.cuda_stream == side.cuda_stream == (engine._stream(dev).value or 0)
        with engine.pinned_stream(0) as st2:                   # cached object re-used while the raw handle is the same
            assert st2 is st1
    with engine.pinned_stream(0) as st3:                       # back on the default stream: the cache notices
        assert st3.cuda_stream == p0 and st3 is not st1
    assert engine._PINNED.get(0) is None                       # nothing stays pinned outside the contexts


def test_step_runs_on_the_callers_stream():
    """a whole HipTrainer.step() inside `torch.cuda.stream(side)`: the kernels go to `side` (an event recorded on `side` before the
    step and one after bracket a non-zero interval; the default stream stays idle)"""
    import numpy as np
    from bench_support.standin_trainer import HipTrainer, FrameData
    from isdf_amd import synthetic
    from tests.accuracy_experiment import config
    cam = dict(H=120, W=160, fx=150.0, fy=150.0, cx=79.5, cy=59.5)
    cfg = config(cam)
    cfg["sample"].update(n_rays=64)
    tr = HipTrainer("cuda:0", cfg, inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=3, virtual_step_ms=10.0)
    depth, normal, T = synthetic.keyframes(3, cam, seed=3, stride=60)
    dev = tr.device
    tr.frames = FrameData(frame_id=np.arange(3), depth_batch=torch.from_numpy(depth).to(dev), T_WC_batch=torch.from_numpy(T).to(dev),
                          normal_batch=torch.from_numpy(normal).to(dev), frame_avg_losses=torch.zeros(3, device=dev))
    tr.noise_std = tr.noise_kf
    tr.step()                                                   # default stream: buffers, plans
    before = tr.engine.params.clone()
    side = torch.cuda.Stream(0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        losses, ms = tr.step()
        e1.record(side)
    side.synchronize()
    assert e0.elapsed_time(e1) > 0.05                           # the step's kernels sat between the two events ON `side`
    assert np.isfinite(float(losses["total_loss"])) and not torch.equal(before, tr.engine.params)
