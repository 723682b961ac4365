"""The per-frame ingest behind graft() on hardware: `HotPath.get_data` lets the trainer's own `get_data`
run and answers its two geometry calls (transform.py:169-196, 215-270) with ONE launch of `isdf_estimate_normals`; `trainer.frames`
is swapped for the growing store.  The GPU box has no reference checkout, so the trainer here is the stand-in with a `get_data` of the
reference's shape (trainer.py:530-562: module-level `geometry.transform.*` calls, a store per frame) -- the same redirect the
reference's method gets (CPU, on the real reference: tests/test_graft_reference.py)."""
import types

import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.gpu

CALLS = {"pc": 0, "normals": 0}


def _eager_pc(depth, fx, fy, cx, cy):           # stands where transform.pointcloud_from_depth_torch stands upstream
    CALLS["pc"] += 1
    raise AssertionError("the eager point-cloud path ran on a grafted trainer")


def _eager_normals(pc):
    CALLS["normals"] += 1
    raise AssertionError("the eager normal estimation ran on a grafted trainer")


geometry = types.SimpleNamespace(transform=types.SimpleNamespace(pointcloud_from_depth_torch=_eager_pc,
                                                                 estimate_pointcloud_normals=_eager_normals))


class RefStore:
    """a store with torch.cat growth, as data_util.FrameData (data_util.py:84-102)"""

    def __init__(self, **kw):
        self.frame_id, self.depth_batch, self.T_WC_batch, self.normal_batch, self.frame_avg_losses = None, None, None, None, None
        self.__dict__.update(kw)

    def __len__(self):
        return 0 if self.frame_id is None else len(self.frame_id)

    def add_frame_data(self, data, replace):
        cat = lambda a, b: b if a is None else (np.concatenate((a, b)) if isinstance(b, np.ndarray) else torch.cat((a, b)))
        for k in ("frame_id", "depth_batch", "T_WC_batch", "normal_batch"):
            setattr(self, k, cat(getattr(self, k), getattr(data, k)))
        self.frame_avg_losses = cat(self.frame_avg_losses, torch.zeros(len(data), device=data.depth_batch.device))


def _trainer(cam, frames):
    from bench_support.standin_trainer import StandinTrainer
    from isdf_amd import synthetic
    from tests.accuracy_experiment import config

    class RefShapedTrainer(StandinTrainer):
        __module__ = __name__                   # `geometry` is looked up in the module that defines the trainer class

        def get_data(self, idxs):               # trainer.py:530-562, reduced to what the hot path reads
            out = RefStore()
            for idx in idxs:
                depth = torch.from_numpy(frames[0][idx][None]).float().to(self.device)
                T = torch.from_numpy(frames[1][idx][None]).float().to(self.device)
                data = RefStore(frame_id=np.array([idx]), depth_batch=depth, T_WC_batch=T)
                if self.do_normal:
                    pc = geometry.transform.pointcloud_from_depth_torch(depth[0], self.fx, self.fy, self.cx, self.cy)
                    data.normal_batch = geometry.transform.estimate_pointcloud_normals(pc)[None, :]
                out.add_frame_data(data, replace=False)
            return out
    tr = RefShapedTrainer("cuda", config(cam), None, True, inv_bounds_transform=synthetic.bounds_transform())
    tr.frames = RefStore()
    return tr


def test_grafted_get_data_runs_the_stencil_kernel_and_the_store_is_migrated():
    from isdf_amd import frame_store
    from isdf_amd.hot_path import graft
    g = gu.load("ingest_small")
    H, W, fx, fy, cx, cy = g["cam"]
    cam = dict(H=int(H), W=int(W), fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy))
    depth = np.stack([g["depth"], g["depth"][::-1].copy()])
    T = np.stack([np.eye(4, dtype=np.float32)] * 2)
    tr = _trainer(cam, (depth, T))
    tr.frames.add_frame_data(RefStore(frame_id=np.array([7]), depth_batch=torch.zeros(1, cam["H"], cam["W"], device="cuda"),
                                      T_WC_batch=torch.eye(4, device="cuda")[None], normal_batch=torch.zeros(1, cam["H"], cam["W"], 3, device="cuda")), False)
    graft(tr)
    assert isinstance(tr.frames, frame_store.FrameData) and len(tr.frames) == 1 and int(tr.frames.frame_id[0]) == 7    # migrated
    fd = tr.get_data([0])
    assert CALLS == {"pc": 0, "normals": 0} and tr._hip.ingest_launches == 1
    assert geometry.transform.estimate_pointcloud_normals is _eager_normals          # the redirect lives for the call only
    n = fd.normal_batch[0].cpu().numpy()
    ref = g["normals"]                                   # what the REAL reference computed for this depth image
    assert np.array_equal(np.isnan(n[..., 0]), np.isnan(ref[..., 0]))
    ok = ~np.isnan(ref[..., 0])
    assert (np.abs(n[ok] - ref[ok]).max(-1) < 1e-4).mean() > 0.999
    # ... and the frame lands in the growing store through the trainer's own add_frame; a second one appends without a re-allocation
    tr.last_is_keyframe = True
    tr.add_frame(fd)
    buf = tr.frames._back["depth_batch"]
    tr.last_is_keyframe = True
    tr.add_frame(tr.get_data([1]))
    assert len(tr.frames) == 3 and tr.frames._back["depth_batch"] is buf and tr._hip.ingest_launches == 2
    assert tr.frames.normal_batch.shape == (3, cam["H"], cam["W"], 3)
    # the hot path reads the migrated store (the fixture's 60 x 80 image does not tile into 8 x 8 loss blocks, which a full step wants:
    # loss.py:208-219; the sampler and the fused evaluation have no such condition)
    smp = tr.sample_points(tr.frames.depth_batch, tr.frames.T_WC_batch, tr.frames.normal_batch)
    assert smp["pc"].shape[1:] == (27, 3) and smp["pc"].shape[0] > 0 and smp["norm_sample"].shape[0] == smp["pc"].shape[0]
    assert torch.isfinite(tr.sdf_map(smp["pc"].reshape(-1, 3))).all()
