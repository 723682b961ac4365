"""The Python-side boundary (`isdf_amd.hot_path.graft`) applied to the REAL reference `Trainer`, imported
unmodified from /root/reference (build container only; skipped where the reference is absent), and to the
in-repo stand-in.  The HIP engine is replaced by the oracle-backed `tests.fake_engine.FakeEngine`, so what
is under test is the HOST logic of the binding:

  * ONE object owns the state the reference's drivers and its remaining methods read and write
    (train.py:102-136; trainer.py:574-650,1011-1014): clock, steps_since_frame, optim_frames,
    last_is_keyframe, noise_std, frames, active_idxs/active_pixels, frozen_sdf_map;
  * the grafted reference `Trainer.step` reproduces the reference's own `Trainer.step` trajectory
    (fixture `step_small_k7`: select_keyframes windows, quirk q4, AdamW) on the same seeds;
  * the reference's driver loop (train.py:86-136) ingests frames, promotes keyframes and follows the
    noise schedule through the grafted object.
"""
import io
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from bench_support.driver_loop import run_train_loop
from tests.fake_engine import FakeEngine

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "isdf")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref_mods():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    with contextlib.redirect_stdout(io.StringIO()):
        mods = mg.import_reference()
    return mg, mods


def _reference_trainer(mg, mods, g, window_size=5):
    """a reference Trainer on the fixture's frames / weights (object.__new__ + attributes, as make_golden.py)"""
    cam, net, sc, lc = gu.cam_of(g), g["net"], gu.sample_of(g), gu.loss_of(g)
    netd = dict(H=int(net[0]), B=int(net[1]), n_freqs=int(net[2]), scale_input=float(net[3]), scale_output=float(net[4]))
    lossd = dict(bounds_method=lc.bounds_method, loss_type=lc.loss_type, trunc_weight=lc.trunc_weight,
                 trunc_distance=lc.trunc_distance, eik_weight=lc.eik_weight, eik_apply_dist=lc.eik_apply_dist,
                 grad_weight=lc.grad_weight, orien_loss=lc.orien_loss)
    frames_np = (g["depth_batch"], g["normal_batch"], g["T_WC_batch"])
    with contextlib.redirect_stdout(io.StringIO()):
        tr = mg.build_trainer(mods, cam, netd, lossd, sc, frames_np, gu.params_of(g), g["bounds_T"],
                              float(g["noise_std"][0]), window_size=window_size)
    tr.fx, tr.fy, tr.cx, tr.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    return tr


@needs_ref
def test_grafted_reference_trainer_reproduces_reference_step_trajectory(ref_mods):
    mg, mods = ref_mods
    from isdf_amd.hot_path import graft, HotPath, FlatAdamW
    from isdf_amd.modules import SDFMapHIP
    g = gu.load("step_small_k7")
    tr = _reference_trainer(mg, mods, g, int(g["window_size"][0]))
    ref_cls = type(tr)
    tr.frames.frame_avg_losses = torch.from_numpy(g["frame_avg_losses0"].copy())
    with contextlib.redirect_stdout(io.StringIO()):
        out = graft(tr, rng="torch", engine_factory=FakeEngine)              # <- the INTEGRATION.md line
    assert out is tr and isinstance(tr, ref_cls) and isinstance(tr, HotPath)
    # the three replaced methods resolve to the HIP binding, everything else to the reference's own code
    assert type(tr).step is HotPath.step and type(tr).sample_points is HotPath.sample_points
    assert type(tr).sdf_eval_and_loss is HotPath.sdf_eval_and_loss
    assert type(tr).select_keyframes is ref_cls.select_keyframes and type(tr).add_data is ref_cls.add_data
    assert isinstance(tr.sdf_map, SDFMapHIP) and isinstance(tr.optimiser, FlatAdamW)
    assert list(tr.sdf_map.state_dict().keys()) == list(gu.params_of(g).keys())
    for k, v in gu.params_of(g).items():                                       # weights carried over
        assert np.array_equal(tr.sdf_map.state_dict()[k].numpy(), v), k
    assert tr.optimiser.param_groups[0]["lr"] == 0.0013 and tr.optimiser.param_groups[0]["weight_decay"] == 0.012

    seed = int(g["seed"][0])
    np.random.seed(seed); torch.manual_seed(seed)                              # as make_golden.run_step_case
    for s in range(int(g["n_steps"][0])):
        losses, step_time = tr.step()
        assert list(tr.active_idxs) == list(g["s%d/idxs" % s])               # select_keyframes: the reference's method
        for k in ["total_loss", "sdf_loss", "grad_loss", "eikonal_loss"]:
            ref = g["s%d/%s" % (s, k)][0]
            assert abs(float(losses[k]) - ref) < 1e-4 * abs(ref), (s, k, float(losses[k]), ref)
        np.testing.assert_allclose(tr.frames.frame_avg_losses.numpy(), g["s%d/frame_avg_losses" % s], rtol=5e-4, atol=1e-6)
        R = int(g["s%d/draw_U" % s].shape[0])
        assert tr.active_pixels["indices_b"].shape[0] == R and tr.active_pixels["indices_b"].dtype == torch.int64
    # state lives on the Trainer object the drivers hold
    assert tr.steps_since_frame == 3 and tr.tot_step_time > 0
    for k, v in tr.sdf_map.state_dict().items():
        assert gu.rel_err(v.numpy(), g["param_after/" + k]) < 2e-4, k
    st = tr.optimiser.state_dict()["state"]
    for i, k in enumerate(gu.params_of(g)):
        assert gu.rel_err(st[i]["exp_avg"].numpy(), g["exp_avg/" + k]) < 2e-3, k
        assert float(st[i]["step"]) == 3.0


@needs_ref
def test_graft_after_reference_steps_keeps_the_optimiser_state(ref_mods):
    """a reference Trainer that has ALREADY taken optimiser steps keeps exp_avg / exp_avg_sq / step through
    graft() -- one reference step + two grafted steps == three steps (fixture step_small_k7: windows, draws, AdamW)."""
    mg, mods = ref_mods
    from isdf_amd.hot_path import graft
    g = gu.load("step_small_k7")
    tr = _reference_trainer(mg, mods, g, int(g["window_size"][0]))
    tr.frames.frame_avg_losses = torch.from_numpy(g["frame_avg_losses0"].copy())
    seed = int(g["seed"][0])
    np.random.seed(seed); torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.step()                                                            # the reference's own step (torch AdamW)
        assert float(tr.optimiser.state_dict()["state"][0]["step"]) == 1.0
        graft(tr, rng="torch", engine_factory=FakeEngine)
    st = tr.optimiser.state_dict()["state"]
    assert float(st[0]["step"]) == 1.0 and float(st[0]["exp_avg"].abs().sum()) > 0   # carried over, not reset
    for s in range(1, int(g["n_steps"][0])):
        losses, _ = tr.step()
        assert list(tr.active_idxs) == list(g["s%d/idxs" % s])
        ref = g["s%d/total_loss" % s][0]
        assert abs(float(losses["total_loss"]) - ref) < 1e-4 * abs(ref), (s, float(losses["total_loss"]), ref)
    st = tr.optimiser.state_dict()["state"]
    for i, k in enumerate(gu.params_of(g)):
        assert gu.rel_err(tr.sdf_map.state_dict()[k].numpy(), g["param_after/" + k]) < 2e-4, k
        assert gu.rel_err(st[i]["exp_avg"].numpy(), g["exp_avg/" + k]) < 2e-3, k
        assert gu.rel_err(st[i]["exp_avg_sq"].numpy(), g["exp_avg_sq/" + k]) < 4e-3, k
        assert float(st[i]["step"]) == 3.0


@needs_ref
def test_integration_md_snippet_runs_on_the_reference_trainer(ref_mods, monkeypatch):
    """The code block INTEGRATION.md tells a maintainer to add is executed VERBATIM on a real reference Trainer
    (the engine stand-in is installed through the module hook, because this container has no GPU)."""
    import re
    import isdf_amd.hot_path as hp
    mg, mods = ref_mods
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python graft-snippet\n(.*?)```", md, re.S)
    assert len(blocks) == 1
    g = gu.load("step_small_k3")
    isdf_trainer = _reference_trainer(mg, mods, g)
    ref_cls = type(isdf_trainer)
    monkeypatch.setattr(hp, "ENGINE_FACTORY", FakeEngine)
    with contextlib.redirect_stdout(io.StringIO()):
        exec(blocks[0], {"isdf_trainer": isdf_trainer})
        np.random.seed(1); torch.manual_seed(1)
        losses, ms = isdf_trainer.step()
    assert isinstance(isdf_trainer, ref_cls) and isinstance(isdf_trainer, hp.HotPath)
    assert set(losses.keys()) == {"sdf_loss", "grad_loss", "eikonal_loss", "total_loss"}
    assert isdf_trainer.steps_since_frame == 1 and isdf_trainer.tot_step_time > 0
    "{:.6f}".format(losses["total_loss"]); losses["total_loss"].item()            # train.py:138,215


class _Dataset:
    """minimal scene_dataset for Trainer.get_data (trainer.py:530-562): dicts with image / depth / T"""

    def __init__(self, depth, T):
        self.depth, self.T = depth, T

    def __len__(self):
        return self.depth.shape[0]

    def __getitem__(self, i):
        H, W = self.depth.shape[1:]
        return {"image": np.zeros((H, W, 3), np.uint8), "depth": self.depth[i], "T": self.T[i]}


def _stream(n, H=48, W=64):
    rng = np.random.RandomState(5)
    depth, _, T = gu.synth_frames(rng, n, H, W, 60.0, 60.0, 31.5, 23.5)
    for i in range(n):   # a slowly moving camera (the fixture generator's poses jump too far per frame)
        T[i, :3, 3] = [0.005 * i, 0.0, 0.002 * i]
        T[i, :3, :3] = T[0, :3, :3]
    return depth, T


def _drive(tr, depth, T, n_steps, get_frame, first_frame_iters=60):
    log = []

    def on_step(t, losses, ms):
        log.append(dict(t=t, clock=tr.tot_step_time, ssf=tr.steps_since_frame, noise=tr.noise_std, K=len(tr.frames),
                        kf=tr.last_is_keyframe, total=float(losses["total_loss"])))
    with contextlib.redirect_stdout(io.StringIO()):
        # (the drivers optimise the first frame alone for 200 steps, train.py:125-127: 60 here -- every step runs the numpy oracle)
        n, ingests, losses = run_train_loop(tr, get_frame, depth.shape[0], n_steps, on_step=on_step, first_frame_iters=first_frame_iters)
    return n, ingests, log


def _check_schedule(tr, n, ingests, log, virtual_ms, fps, iters_per_frame, noise_kf, noise_frame):
    assert n == len(log) and len(ingests) >= 4, ingests
    # the virtual clock advanced on the trainer object, by the pinned step time
    assert abs(tr.tot_step_time - n * virtual_ms / 1000.0) < 1e-9
    # frame ids are a function of that clock (trainer.py:100-101) and therefore increase
    ids = [i for _, i in ingests]
    assert ids[0] == 0 and ids == sorted(ids) and len(set(ids)) == len(ids)
    for t, fid in ingests[1:]:
        assert fid == int(log[t - 1]["clock"] * fps)
    # steps_since_frame restarts at every ingest (add_frame, trainer.py:580) ...
    for t, _ in ingests:
        assert log[t]["ssf"] == 1
    # ... and the noise schedule reaches the kernels: noise_frame after add_frame, noise_kf once a frame is promoted
    noises = {round(r["noise"], 6) for r in log[ingests[1][0]:]}
    assert round(noise_frame, 6) in noises
    if any(r["kf"] for r in log[ingests[1][0]:]):
        assert round(noise_kf, 6) in noises
    assert len(tr.frames) >= 2 and len(tr.frames) <= len(ingests)
    assert np.isfinite(log[-1]["total"])


@needs_ref
def test_reference_driver_loop_on_grafted_reference_trainer(ref_mods):
    """train.py:86-136 replayed against ONE grafted reference Trainer: `get_data` / `add_frame` /
    `check_keyframe_latest` / `select_keyframes` / `get_latest_frame_id` are the reference's own methods."""
    mg, mods = ref_mods
    from isdf_amd.hot_path import graft
    from isdf_amd.modules import SDFMapHIP
    g = gu.load("step_small_k3")
    tr = _reference_trainer(mg, mods, g)
    FrameData = mods[6]
    depth, T = _stream(160)
    tr.frames = FrameData()
    tr.scene_dataset, tr.gt_traj, tr.live, tr.fps = _Dataset(depth, T), None, False, 30
    tr.n_rays, tr.n_rays_is_kf, tr.kf_dist_th, tr.kf_pixel_ratio = 20, 40, 0.1, 0.65
    tr.iters_per_kf, tr.iters_per_frame, tr.noise_kf, tr.noise_frame = 6, 3, 0.08, 0.04
    tr.last_is_keyframe, tr.optim_frames = False, 0
    np.random.seed(3); torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        graft(tr, rng="torch", virtual_step_ms=12.0, engine_factory=FakeEngine)
    n, ingests, log = _drive(tr, depth, T, 160, lambda i: tr.get_data([i]))
    _check_schedule(tr, n, ingests, log, 12.0, 30, 3, 0.08, 0.04)
    assert isinstance(tr.frozen_sdf_map, SDFMapHIP) and tr.frozen_sdf_map is not tr.sdf_map   # deepcopy at trainer.py:576
    assert tr.frozen_sdf_map.engine is not tr.sdf_map.engine
    assert tr.frames.normal_batch.shape[0] == len(tr.frames)
    # the ingest is bound (round 6): graft() swapped the keyframe store for the growing one ...
    from isdf_amd import frame_store
    assert isinstance(tr.frames, frame_store.FrameData) and not isinstance(tr.frames, FrameData)
    # ... and the reference's own get_data ran with its two geometry calls answered by the engine: one launch per ingested frame,
    assert tr._hip.ingest_launches == len(ingests) >= 4
    geo = mods[5] if hasattr(mods[5], "estimate_pointcloud_normals") else sys.modules["isdf.geometry.transform"]
    assert geo.estimate_pointcloud_normals.__module__ == "isdf.geometry.transform"      # (the redirect is undone after every call)
    # with the normals the reference computes for the same depth images (transform.py:169-196,215-270)
    for k, fid in enumerate(int(i) for i in tr.frames.frame_id):
        d = torch.from_numpy(depth[fid])
        ref = geo.estimate_pointcloud_normals(geo.pointcloud_from_depth_torch(d, tr.fx, tr.fy, tr.cx, tr.cy)).numpy()
        got = tr.frames.normal_batch[k].numpy()
        assert np.array_equal(np.isnan(ref), np.isnan(got)), fid
        ok = ~np.isnan(ref)
        assert np.abs(ref[ok] - got[ok]).max() < 2e-5, (fid, np.abs(ref[ok] - got[ok]).max())


def test_reference_driver_loop_on_hiptrainer_standin():
    """the same loop against HipTrainer (= graft applied to the in-repo stand-in; the object the GPU box runs)"""
    from bench_support.standin_trainer import HipTrainer
    from tests.accuracy_experiment import config
    cam = dict(H=48, W=64, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    cfg = config(cam)
    cfg["model"].update(hidden_feature_size=64, hidden_layers_block=1, iters_per_kf=6, iters_per_frame=3)
    cfg["sample"].update(n_rays=20, n_rays_is_kf=40)
    depth, T = _stream(160)
    np.random.seed(3); torch.manual_seed(3)
    tr = HipTrainer("cpu", cfg, inv_bounds_transform=gu.bounds_transform(), rng="philox", seed=3,
                    virtual_step_ms=12.0, engine_factory=FakeEngine)
    n, ingests, log = _drive(tr, depth, T, 160, lambda i: tr.make_frame(i, depth[i], T[i]))
    _check_schedule(tr, n, ingests, log, 12.0, 30, 3, 0.08, 0.04)
    # checkpoint while a NON-keyframe is being optimised, resume in a fresh trainer, keyframe test still works
    t = n
    while tr.last_is_keyframe or tr.steps_since_frame != tr.optim_frames:
        with contextlib.redirect_stdout(io.StringIO()):
            t, _, _ = run_train_loop(tr, lambda i: tr.make_frame(i, depth[i], T[i]), depth.shape[0], 1, t0=t)
        assert t < n + 200
    sd = tr.state_dict()
    assert sd["frozen_state_dict"] is not None
    tr2 = HipTrainer("cpu", cfg, inv_bounds_transform=gu.bounds_transform(), rng="philox", seed=3,
                     virtual_step_ms=12.0, engine_factory=FakeEngine)
    tr2.load_state_dict(sd)
    # every FrameData field is resumed: the host twins have the keyframe count, so the next ingest appends
    # / overwrites consistently instead of restarting them at length 1
    for k in ("depth_batch_np", "T_WC_batch_np", "im_batch_np"):
        a_, b_ = getattr(tr.frames, k, None), getattr(tr2.frames, k, None)
        assert (a_ is None) == (b_ is None) and (a_ is None or (len(a_) == len(tr.frames) and np.array_equal(a_, b_))), k
    assert len(tr2.frames) == len(tr.frames)
    with contextlib.redirect_stdout(io.StringIO()):
        a, b = tr.check_keyframe_latest(), tr2.check_keyframe_latest()
    assert a == b and tr.last_is_keyframe == tr2.last_is_keyframe
    for k, v in tr.frozen_sdf_map.state_dict().items():
        assert torch.equal(v, tr2.frozen_sdf_map.state_dict()[k])


def test_graft_refuses_cpu_device_and_unsupported_configs():
    from isdf_amd import _ffi
    from bench_support.standin_trainer import HipTrainer
    from tests.accuracy_experiment import config
    cam = dict(H=48, W=64, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    with pytest.raises(_ffi.IsdfError):
        HipTrainer("cpu", config(cam))                      # no CPU path: the product never routes around the kernels
    cfg = config(cam); cfg["loss"]["bounds_method"] = "normal"
    with pytest.raises(_ffi.IsdfError):
        HipTrainer("cpu", cfg, engine_factory=FakeEngine)
    cfg = config(cam); cfg["model"]["do_active"] = 1
    with pytest.raises(_ffi.IsdfError):
        HipTrainer("cpu", cfg, engine_factory=FakeEngine)


def test_optimiser_facade_resets_moments_for_an_empty_checkpoint():
    from bench_support.standin_trainer import HipTrainer
    from tests.accuracy_experiment import config
    cam = dict(H=48, W=64, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    cfg = config(cam); cfg["model"].update(hidden_feature_size=64, hidden_layers_block=1)
    tr = HipTrainer("cpu", cfg, engine_factory=FakeEngine)
    empty = tr.optimiser.state_dict()
    assert empty["state"] == {}
    eng = tr.sdf_map.engine
    eng.exp_avg.fill_(1.0); eng.exp_avg_sq.fill_(2.0); eng.opt_step = 7
    tr.optimiser.load_state_dict(empty)
    assert eng.opt_step == 0 and float(eng.exp_avg.abs().sum()) == 0 and float(eng.exp_avg_sq.abs().sum()) == 0
