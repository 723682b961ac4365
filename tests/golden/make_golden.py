#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE
implementation (imported read-only from /root/reference) on small seeded inputs.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py

The reference modules are imported unmodified; packages that are absent here and
untouched by the hot path (trimesh, cv2, open3d, ...) are stubbed with MagicMock
exactly as SURVEY.md section 8(c) describes.  A `Trainer` is built with
`object.__new__` and the attribute set the unmodified `Trainer.sample_points /
sdf_eval_and_loss / step` methods read.  Every random draw the reference makes
(`torch.randint/rand/normal/randn`, `np.random.choice`) is recorded so the
oracle and the HIP path can consume identical draws.
"""
import os
import sys
import copy
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"


STUB_ROOTS = ["trimesh", "cv2", "imgviz", "torchvision", "open3d", "pyglet", "skimage",
              "imageio", "git", "urdfpy", "rospy", "sensor_msgs", "geometry_msgs",
              "orb_slam3_ros_wrapper", "cv_bridge", "message_filters"]


class _StubFinder:
    """meta-path finder: any (sub)module of a package that is absent here
    resolves to a MagicMock (the hot path never touches them)."""

    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        pass


def import_reference():
    import importlib
    missing = []
    for name in STUB_ROOTS:
        try:
            importlib.import_module(name)
        except Exception:
            missing.append(name)
    sys.meta_path.insert(0, _StubFinder(missing))
    sys.path.insert(0, REF)
    from isdf.modules import trainer, sample, embedding, fc_map, loss  # noqa
    from isdf.geometry import transform  # noqa
    from isdf.datasets.data_util import FrameData  # noqa
    return trainer, sample, embedding, fc_map, loss, transform, FrameData


class DrawRecorder:
    """Wraps torch RNG entry points used on the hot path and logs what they return."""

    def __init__(self):
        self.log = []
        self._orig = {}

    def __enter__(self):
        for name in ["randint", "rand", "normal", "randn"]:
            self._orig[name] = getattr(torch, name)

            def wrap(*a, _n=name, **k):
                out = self._orig[_n](*a, **k)
                self.log.append((_n, out.detach().cpu().numpy().copy()))
                return out
            setattr(torch, name, wrap)
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)

    def pop_step(self, with_noise=True):
        """randint(h), randint(w), rand(U), normal(N_off)[, randn(noise)]"""
        names = ["randint", "randint", "rand", "normal"] + (["randn"] if with_noise else [])
        got = [self.log.pop(0) for _ in names]
        assert [g[0] for g in got] == names, [g[0] for g in got]
        return [g[1] for g in got]


from tests.golden_util import synth_frames, synth_frames_exact, frames_checksum, bounds_transform  # noqa: E402


def build_trainer(mods, cam, net, lossc, samplec, frames_np, params_np, transform_np,
                  noise_std, window_size=5, incremental=True, do_normal=True):
    trainer, sample, embedding, fc_map, loss, transform, FrameData = mods
    tr = object.__new__(trainer.Trainer)
    tr.device = "cpu"
    tr.incremental = incremental
    tr.window_size = window_size
    tr.do_normal = do_normal
    tr.H, tr.W = cam["H"], cam["W"]
    tr.n_rays = samplec["n_rays"]
    tr.dist_behind_surf = samplec["dist_behind_surf"]
    tr.n_strat_samples = samplec["n_strat"]
    tr.n_surf_samples = samplec["n_surf"]
    tr.min_depth = samplec["min_depth"]
    tr.dirs_C = transform.ray_dirs_C(1, cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"],
                                     cam["cy"], "cpu", depth_type="z")
    tr.eik_weight = lossc["eik_weight"]
    tr.grad_weight = lossc["grad_weight"]
    tr.noise_std = noise_std
    tr.bounds_method = lossc["bounds_method"]
    tr.trunc_distance = lossc["trunc_distance"]
    tr.loss_type = lossc["loss_type"]
    tr.orien_loss = lossc["orien_loss"]
    tr.eik_apply_dist = lossc["eik_apply_dist"]
    tr.trunc_weight = lossc["trunc_weight"]
    tr.loss_approx_factor = 8
    tr.frac_time_perception = 1.0
    tr.tot_step_time = 0.0
    tr.steps_since_frame = 0
    tr.cosSim = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)
    T_t = None if transform_np is None else torch.from_numpy(transform_np)
    pe = embedding.PostionalEncoding(min_deg=0, max_deg=net["n_freqs"] - 1,
                                     scale=net["scale_input"], transform=T_t)
    tr.sdf_map = fc_map.SDFMap(pe, hidden_size=net["H"], hidden_layers_block=net["B"],
                               scale_output=net["scale_output"])
    sd = {k: torch.from_numpy(v.copy()) for k, v in params_np.items()}
    tr.sdf_map.load_state_dict(sd)
    tr.optimiser = torch.optim.AdamW(tr.sdf_map.parameters(), lr=0.0013, weight_decay=0.012)
    depth, normal, T = frames_np
    K = depth.shape[0]
    tr.frames = FrameData(
        frame_id=np.arange(K), im_batch=torch.zeros(K, cam["H"], cam["W"], 3),
        depth_batch=torch.from_numpy(depth.copy()), T_WC_batch=torch.from_numpy(T.copy()),
        normal_batch=torch.from_numpy(normal.copy()),
        frame_avg_losses=torch.zeros(K))
    return tr


LOSS_DEFAULT = dict(bounds_method="ray", loss_type="L1", trunc_weight=5.38344020,
                    trunc_distance=0.29365022, eik_weight=0.268, eik_apply_dist=0.1,
                    grad_weight=0.018, orien_loss=False)
SAMPLE_DEFAULT = dict(n_rays=40, n_strat=19, n_surf=8, min_depth=0.07, dist_behind_surf=0.1)


def t2n(x):
    return None if x is None else x.detach().cpu().numpy().copy()


def run_eval_case(mods, name, net, lossc, samplec, F, cam, seed, noise_std, full_grads,
                  transform_on=True, exact_frames=False, with_normals=True, slim=False):
    """sample_points + sdf_eval_and_loss + backward on the reference; store everything.
    exact_frames: BASELINE-size keyframes from golden_util.synth_frames_exact -- NOT stored (65 MB at
    680x1200x5), the tests regenerate them from the seed and verify the checksum.
    with_normals=False: the reference's do_normal=False path (norm_batch None, needs grad_weight 0).
    slim: store no per-point tensors (same draws as a sibling fixture; losses + gradient digests only)."""
    import oracle.isdf_oracle as orc
    rng = np.random.RandomState(seed)
    if exact_frames:
        frames_np = synth_frames_exact(seed, F, cam["H"], cam["W"])
    else:
        frames_np = synth_frames(rng, F, cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    params_np = orc.init_params(net["H"], net["B"], net["n_freqs"], np.random.RandomState(seed + 100))
    Tb = bounds_transform(rng) if transform_on else None
    tr = build_trainer(mods, cam, net, lossc, samplec, frames_np, params_np, Tb, noise_std,
                       do_normal=with_normals)
    torch.manual_seed(seed)
    with DrawRecorder() as rec:
        sp = tr.sample_points(tr.frames.depth_batch, tr.frames.T_WC_batch,
                              norm_batch=tr.frames.normal_batch if with_normals else None)
        pc_in = sp["pc"].clone()
        total, losses, loss_approx, frame_avg_loss = tr.sdf_eval_and_loss(sp, do_avg_loss=True)
        ih, iw, U, N_off, noise = rec.pop_step(with_noise=True)
    # recompute sdf/sdf_grad with the same noise for storage
    mods_fc = mods[3]
    pc = pc_in.clone().requires_grad_()
    torch.manual_seed(0)
    raw_sdf = tr.sdf_map(pc, noise_std=None)
    sdf_nonoise = raw_sdf.detach()
    sdf_grad = mods_fc.gradient(pc, raw_sdf).detach()
    total.backward()
    grads = {k: t2n(p.grad) for k, p in tr.sdf_map.named_parameters()}
    if exact_frames:
        frames_kw = dict(frames_gen=np.array([F, seed], np.int64),
                         frames_sum=np.array(frames_checksum(*frames_np), np.int64))
    else:
        frames_kw = dict(depth_batch=frames_np[0], normal_batch=frames_np[1], T_WC_batch=frames_np[2])
    out = dict(
        with_normals=np.array([int(with_normals)]),
        cam=np.array([cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64),
        net=np.array([net["H"], net["B"], net["n_freqs"], net["scale_input"], net["scale_output"]], np.float64),
        seed=np.array([seed]), noise_std=np.array([noise_std], np.float64),
        has_transform=np.array([int(transform_on)]),
        bounds_T=Tb if Tb is not None else np.eye(4, dtype=np.float32),
        draw_indices_h=ih, draw_indices_w=iw, draw_U=U, draw_N_off=N_off,
        draw_noise=noise,  # raw N(0,1) draw; scaled by noise_std at use
        pc=t2n(sp["pc"]), z_vals=t2n(sp["z_vals"]), indices_b=t2n(sp["indices_b"]),
        indices_h=t2n(sp["indices_h"]), indices_w=t2n(sp["indices_w"]),
        dirs_C_sample=t2n(sp["dirs_C_sample"]), depth_sample=t2n(sp["depth_sample"]),
        T_WC_sample=t2n(sp["T_WC_sample"]), norm_sample=t2n(sp["norm_sample"]),
        sdf_nonoise=t2n(sdf_nonoise), sdf_grad=t2n(sdf_grad),
        total_loss=np.array([float(total)]),
        sdf_loss=np.array([losses["sdf_loss"]]),
        grad_loss=np.array([losses.get("grad_loss", np.nan)]),
        eikonal_loss=np.array([losses.get("eikonal_loss", np.nan)]),
        loss_approx=t2n(loss_approx), frame_avg_loss=t2n(frame_avg_loss),
    )
    if not with_normals:
        out.pop("norm_sample")
    if slim:
        for k in ("pc", "z_vals", "dirs_C_sample", "T_WC_sample", "norm_sample", "sdf_nonoise", "sdf_grad"):
            out.pop(k, None)
    elif exact_frames:
        out.pop("T_WC_sample")          # = T_WC_batch[indices_b]; 64 B/ray of redundancy
    out.update(frames_kw)
    for k, v in lossc.items():
        out["loss_" + k] = np.array([v]) if not isinstance(v, str) else np.array(v)
    for k, v in samplec.items():
        out["sample_" + k] = np.array([v], np.float64)
    if full_grads:
        for k, v in params_np.items():
            out["param/" + k] = v
        for k, v in grads.items():
            out["grad/" + k] = v
    else:
        prng = np.random.RandomState(1234)
        for k, v in grads.items():
            probe = prng.standard_normal(v.shape).astype(np.float64)
            out["gdig/" + k] = np.array([np.linalg.norm(v.astype(np.float64)),
                                         float((v.astype(np.float64) * probe).sum())])
            out["ghead/" + k] = v.reshape(-1)[:64].copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "R =", out["depth_sample"].shape[0], "total_loss =", float(total),
          {k: round(float(v), 6) if not torch.is_tensor(v) else float(v) for k, v in losses.items()})


def _digest(out, prefix, k, v, prng):
    v64 = v.astype(np.float64)
    probe = prng.standard_normal(v.shape)
    out[prefix + "dig/" + k] = np.array([np.linalg.norm(v64), float((v64 * probe).sum())])
    out[prefix + "head/" + k] = v.reshape(-1)[:64].copy()


def run_step_case(mods, name, net, lossc, samplec, K, cam, seed, noise_std, n_steps, window_size,
                  digest=False):
    """Unmodified Trainer.step (incl. select_keyframes when K > window) for n_steps.
    digest: default-size net -- the initial weights regenerate from the seed and the final parameters /
    AdamW moments are stored as (norm, probe dot, first 64 values) per tensor instead of in full."""
    import oracle.isdf_oracle as orc
    rng = np.random.RandomState(seed)
    frames_np = synth_frames(rng, K, cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    params_np = orc.init_params(net["H"], net["B"], net["n_freqs"], np.random.RandomState(seed + 100))
    Tb = bounds_transform(rng)
    tr = build_trainer(mods, cam, net, lossc, samplec, frames_np, params_np, Tb, noise_std,
                       window_size=window_size)
    tr.frames.frame_avg_losses = torch.from_numpy(
        np.random.RandomState(seed + 5).uniform(0.5, 1.5, K).astype(np.float32))
    fal0 = t2n(tr.frames.frame_avg_losses)
    np.random.seed(seed)
    torch.manual_seed(seed)
    out = dict(
        depth_batch=frames_np[0], normal_batch=frames_np[1], T_WC_batch=frames_np[2],
        cam=np.array([cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64),
        net=np.array([net["H"], net["B"], net["n_freqs"], net["scale_input"], net["scale_output"]], np.float64),
        seed=np.array([seed]), noise_std=np.array([noise_std], np.float64), bounds_T=Tb,
        frame_avg_losses0=fal0, window_size=np.array([window_size]), n_steps=np.array([n_steps]),
    )
    for k, v in lossc.items():
        out["loss_" + k] = np.array([v]) if not isinstance(v, str) else np.array(v)
    for k, v in samplec.items():
        out["sample_" + k] = np.array([v], np.float64)
    if not digest:
        for k, v in params_np.items():
            out["param/" + k] = v
    with DrawRecorder() as rec:
        for s in range(n_steps):
            losses, _ = tr.step()
            ih, iw, U, N_off, noise = rec.pop_step(with_noise=True)
            out["s%d/idxs" % s] = np.asarray(tr.active_idxs, np.int64)
            out["s%d/draw_indices_h" % s] = ih
            out["s%d/draw_indices_w" % s] = iw
            out["s%d/draw_U" % s] = U
            out["s%d/draw_N_off" % s] = N_off
            out["s%d/draw_noise" % s] = noise
            out["s%d/total_loss" % s] = np.array([float(losses["total_loss"])])
            out["s%d/sdf_loss" % s] = np.array([losses["sdf_loss"]])
            out["s%d/grad_loss" % s] = np.array([losses["grad_loss"]])
            out["s%d/eikonal_loss" % s] = np.array([losses["eikonal_loss"]])
            out["s%d/frame_avg_losses" % s] = t2n(tr.frames.frame_avg_losses)
            print(name, "step", s, "idxs", list(tr.active_idxs), "total", float(losses["total_loss"]))
    st = tr.optimiser.state_dict()["state"]
    names = [k for k, _ in tr.sdf_map.named_parameters()]
    if digest:
        prng = np.random.RandomState(4321)
        for i, (k, p) in enumerate(tr.sdf_map.named_parameters()):
            _digest(out, "param_after_", k, t2n(p) - params_np[k], prng)     # the UPDATE, not the weights
            _digest(out, "exp_avg_", k, t2n(st[i]["exp_avg"]), prng)
            _digest(out, "exp_avg_sq_", k, t2n(st[i]["exp_avg_sq"]), prng)
    else:
        for k, p in tr.sdf_map.named_parameters():
            out["param_after/" + k] = t2n(p)
        for i, k in enumerate(names):
            out["exp_avg/" + k] = t2n(st[i]["exp_avg"])
            out["exp_avg_sq/" + k] = t2n(st[i]["exp_avg_sq"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path)



TRAINED_SPECS = {
    # the default net on the analytic room: 480x640, replicaCAD.json loss / sample / optimiser settings, bounds transform on
    "default": dict(cam=dict(H=480, W=640, fx=577.87, fy=577.87, cx=319.5, cy=239.5),
                    net=dict(H=256, B=2, n_freqs=6, scale_input=0.05937489, scale_output=0.14),
                    loss=None, sample=dict(n_rays=200), transform=True, noise_std=0.08),
    # realsense_franka.json's constants (round 5): 720x1280, 9 PE octaves (E = 381), scale_input 0.4,
    # trunc_weight 30, trunc_distance 0.1, depth_range[0] 0.1, live mode = NO bounds transform, noise_kf 0.025
    "franka": dict(cam=dict(H=720, W=1280, fx=913.4483642578125, fy=913.4601440429688, cx=640.4678955078125, cy=359.1015319824219),
                   net=dict(H=256, B=2, n_freqs=9, scale_input=0.4, scale_output=0.14),
                   loss=dict(trunc_weight=30.0, trunc_distance=0.1), sample=dict(n_rays=120, min_depth=0.1), transform=False,
                   noise_std=0.025),
}


def _trained_setup(mods, seed, spec="default"):
    import oracle.isdf_oracle as orc
    from isdf_amd import synthetic
    sp = TRAINED_SPECS[spec]
    cam, net = sp["cam"], sp["net"]
    lossc = dict(LOSS_DEFAULT, **(sp["loss"] or {}))
    samplec = dict(SAMPLE_DEFAULT, **sp["sample"])
    frames_np = synthetic.keyframes(5, cam, seed=seed, stride=48, noise_std=0.01)
    params_np = orc.init_params(net["H"], net["B"], net["n_freqs"], np.random.RandomState(seed + 100))
    Tb = synthetic.bounds_transform() if sp["transform"] else None
    tr = build_trainer(mods, cam, net, lossc, samplec, frames_np, params_np, Tb, sp["noise_std"])
    return tr, cam, net, samplec, Tb, lossc, sp["noise_std"]


def trained_warmup(mods, seed, pre_steps, path, spec="default"):
    """child process of run_trained_case: `pre_steps` unmodified Trainer.step calls with flush-to-zero on"""
    tr = _trained_setup(mods, seed, spec)[0]
    np.random.seed(seed); torch.manual_seed(seed)
    for s in range(pre_steps):
        losses, _ = tr.step()
        if s % 20 == 0 or s == pre_steps - 1:
            print("warm-up step", s, "total", float(losses["total_loss"]), flush=True)
    torch.save(dict(model=tr.sdf_map.state_dict(), optim=tr.optimiser.state_dict(), frame_avg_losses=tr.frames.frame_avg_losses,
                    np_rng=np.random.get_state(), torch_rng=torch.get_rng_state()), path)


def bf16_round(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)


def run_trained_case(mods, name, seed=61, pre_steps=300, traj_steps=20, traj_rays=40, spec="default", grads_fp16=False):
    """Round 4: the default 6x256 network at TRAINED weights -- every gradient test before this one ran
    at random-initialised weights, where almost no unit of a Softplus(beta=100) layer is saturated.

    1. the UNMODIFIED reference `Trainer.step` x `pre_steps` on 5 keyframes of the analytic room (isdf_amd/synthetic.py, 480x640,
       replicaCAD.json loss / sample / optimiser settings).  Flush-to-zero is on for this warm-up only (9x faster on x86, SURVEY 6.1;
       it only decides WHICH trained state the fixture holds).
    2. eval batch at that state: `sample_points` + `sdf_eval_and_loss` + `backward()`; stored: the sampler's outputs (so the test needs
       no keyframes), the noise draw, sdf, d sdf/dx, loss terms, ALL gradients in full (signed-projection tests need them).
    3. trajectory: AdamW moments rounded to bfloat16 and loaded back (the fixture then holds the exact start state in half the
       bytes), `traj_steps` further unmodified `Trainer.step`s at `traj_rays` rays per keyframe; stored per step: the sampler outputs,
       the noise, the losses; the accumulated parameter update after 5 steps in full (float16), after 20 steps as digests.
       (Two fp32 implementations agree to 1e-7 for ~12 steps on this trajectory and part ways within the next 8: the loss is piecewise
       linear, so the first residual whose sign differs forks the runs.)"""
    import subprocess
    import tempfile
    trainer, sample, embedding, fc_map, loss, transform, FrameData = mods
    tr, cam, net, samplec, Tb, lossc, noise_std = _trained_setup(mods, seed, spec)
    # warm-up in a CHILD process with flush-to-zero set before its first torch op (worker threads inherit the mode of the thread
    # that creates them, so it cannot be switched off again inside one process); this process never enables it
    tmp = os.path.join(tempfile.gettempdir(), "isdf_trained_warmup_%s_%d_%d.pt" % (spec, seed, pre_steps))
    if not os.path.exists(tmp):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "_trained_warmup", str(seed), str(pre_steps), tmp, spec])
    ck = torch.load(tmp, weights_only=False)
    tr.sdf_map.load_state_dict(ck["model"])
    tr.optimiser.load_state_dict(ck["optim"])
    tr.frames.frame_avg_losses = ck["frame_avg_losses"]
    np.random.set_state(ck["np_rng"]); torch.set_rng_state(ck["torch_rng"])
    names = [k for k, _ in tr.sdf_map.named_parameters()]
    out = dict(cam=np.array([cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64),
               net=np.array([net["H"], net["B"], net["n_freqs"], net["scale_input"], net["scale_output"]], np.float64),
               seed=np.array([seed]), noise_std=np.array([noise_std], np.float64), has_transform=np.array([int(Tb is not None)]),
               bounds_T=Tb if Tb is not None else np.eye(4, dtype=np.float32),
               pre_steps=np.array([pre_steps]), traj_steps=np.array([traj_steps]), n_frames=np.array([5]),
               T_WC_batch=t2n(tr.frames.T_WC_batch))      # T_WC_sample = T_WC_batch[indices_b] (sample.py:63)
    for k, v in lossc.items():
        out["loss_" + k] = np.array([v]) if not isinstance(v, str) else np.array(v)
    for k, v in samplec.items():
        out["sample_" + k] = np.array([v], np.float64)
    for k, p in tr.sdf_map.named_parameters():
        out["param/" + k] = t2n(p)

    def sampler_outputs(prefix, sp, noise):
        _, dirs_W = transform.origin_dirs_W(sp["T_WC_sample"], sp["dirs_C_sample"])
        out[prefix + "pc"] = t2n(sp["pc"]); out[prefix + "z_vals"] = t2n(sp["z_vals"])
        out[prefix + "depth_sample"] = t2n(sp["depth_sample"]); out[prefix + "dirs_C_sample"] = t2n(sp["dirs_C_sample"])
        out[prefix + "dirs_W_sample"] = t2n(dirs_W); out[prefix + "norm_sample"] = t2n(sp["norm_sample"])
        for k in ("indices_b", "indices_h", "indices_w"):
            out[prefix + k] = t2n(sp[k]).astype(np.int16)
        out[prefix + "noise"] = (noise * np.float32(noise_std)).astype(np.float32)     # as added to raw: randn * noise_std (fc_map.py:106-108)

    # ---- 2. eval batch
    with DrawRecorder() as rec:
        sp = tr.sample_points(tr.frames.depth_batch, tr.frames.T_WC_batch, norm_batch=tr.frames.normal_batch)
        pc_in = sp["pc"].clone()
        total, losses, loss_approx, frame_avg_loss = tr.sdf_eval_and_loss(sp, do_avg_loss=True)
        ih, iw, U, N_off, noise = rec.pop_step(with_noise=True)
    sampler_outputs("eval/", sp, noise)
    pc = pc_in.clone().requires_grad_()
    raw_sdf = tr.sdf_map(pc, noise_std=None)
    out["eval/sdf_nonoise"] = t2n(raw_sdf)
    out["eval/sdf_grad"] = t2n(mods[3].gradient(pc, raw_sdf))
    total.backward()
    for k, p in tr.sdf_map.named_parameters():
        gq = t2n(p.grad)
        if grads_fp16:      # half the bytes: float16 mantissas on a per-tensor power-of-two scale (3e-4 rel-L2 of rounding; the bar is 1e-2)
            sc = np.float32(2.0 ** np.floor(np.log2(max(float(np.abs(gq).max()), 1e-30))))
            out["eval/grad16/" + k] = (gq / sc).astype(np.float16)
            out["eval/grad16_scale/" + k] = np.array([sc], np.float32)
            out["eval/grad_norm/" + k] = np.array([np.linalg.norm(gq.astype(np.float64))])
        else:
            out["eval/grad/" + k] = gq
        p.grad = None
    out["eval/total_loss"] = np.array([float(total)]); out["eval/sdf_loss"] = np.array([losses["sdf_loss"]])
    out["eval/grad_loss"] = np.array([losses["grad_loss"]]); out["eval/eikonal_loss"] = np.array([losses["eikonal_loss"]])
    out["eval/frame_avg_loss"] = t2n(frame_avg_loss)
    print(name, "eval batch R =", sp["pc"].shape[0], "total", float(total), losses, flush=True)

    if traj_steps == 0:
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) / 1e6, "MB")
        return
    # ---- 3. trajectory from a bf16-representable AdamW state
    st = tr.optimiser.state
    for k, p in tr.sdf_map.named_parameters():
        for mom in ("exp_avg", "exp_avg_sq"):
            q = bf16_round(t2n(st[p][mom]))
            st[p][mom].copy_(torch.from_numpy(q))
            out["adam/%s/%s" % (mom, k)] = (q.view(np.uint32) >> 16).astype(np.uint16)
    out["adam/step"] = np.array([float(st[next(iter(tr.sdf_map.parameters()))]["step"])])
    theta0 = {k: t2n(p) for k, p in tr.sdf_map.named_parameters()}
    tr.n_rays = traj_rays
    captured = []
    orig_sample_points = tr.sample_points

    def observing_sample_points(*a, **kw):       # the reference method runs unmodified; its return value is recorded
        o = orig_sample_points(*a, **kw)
        captured.append(o)
        return o
    tr.sample_points = observing_sample_points
    out["traj/frame_avg_losses0"] = t2n(tr.frames.frame_avg_losses)
    with DrawRecorder() as rec:
        for s in range(traj_steps):
            losses, _ = tr.step()
            ih, iw, U, N_off, noise = rec.pop_step(with_noise=True)
            sampler_outputs("traj/s%d/" % s, captured.pop(0), noise)
            out["traj/s%d/losses" % s] = np.array([losses["sdf_loss"], losses["grad_loss"], losses["eikonal_loss"],
                                                   float(losses["total_loss"])], np.float64)
            out["traj/s%d/frame_avg_losses" % s] = t2n(tr.frames.frame_avg_losses)
            if s + 1 == 5:      # the accumulated update while two correct implementations still agree (chaos sets in after ~15 steps)
                for k, p in tr.sdf_map.named_parameters():
                    out["traj/update5/" + k] = (t2n(p) - theta0[k]).astype(np.float16)
            print(name, "trajectory step", s, "total", float(losses["total_loss"]), flush=True)
    prng = np.random.RandomState(4321)
    for k, p in tr.sdf_map.named_parameters():
        _digest(out, "traj/update_", k, t2n(p) - theta0[k], prng)         # after all 20 steps: digest only (chaos-dominated by then)
        _digest(out, "traj/exp_avg_", k, t2n(st[p]["exp_avg"]), prng)
        _digest(out, "traj/exp_avg_sq_", k, t2n(st[p]["exp_avg_sq"]), prng)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


def run_ingest_case(mods, name, seed):
    """reference per-frame ingest (normals) and keyframe depth render on small seeded inputs"""
    trainer, sample, embedding, fc_map, loss, transform, FrameData = mods
    from isdf.modules import render
    rng = np.random.RandomState(seed)
    H, W = 60, 80
    cam = dict(fx=75.0, fy=75.0, cx=39.5, cy=29.5)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (2.0 + 0.8 * np.sin(0.13 * u) * np.cos(0.09 * v) + 0.02 * rng.standard_normal((H, W))).astype(np.float32)
    depth[rng.uniform(size=(H, W)) < 0.05] = 0.0
    depth[10:14, 20:30] = 0.0
    d_t = torch.from_numpy(depth)
    pc = transform.pointcloud_from_depth_torch(d_t, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    normals = transform.estimate_pointcloud_normals(pc)
    R, S = 120, 27
    z = np.sort(rng.uniform(0.1, 4.0, (R, S)).astype(np.float32), axis=1)
    sdf = (rng.uniform(0.5, 3.5, (R, 1)) - z + 0.05 * rng.standard_normal((R, S))).astype(np.float32)
    sdf[:7] = np.abs(sdf[:7]) + 0.01           # rays with no crossing
    sdf[7:12, :-1] = np.abs(sdf[7:12, :-1]) + 0.01; sdf[7:12, -1] = -0.2   # crossing only at the last sample
    rd = render.sdf_render_depth(torch.from_numpy(z), torch.from_numpy(sdf))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, depth=depth, cam=np.array([H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64),
                        pc=t2n(pc), normals=t2n(normals), z_sorted=z, sdf_sorted=sdf, render_depth=t2n(rd))
    print("wrote", path, "nan normals", int(torch.isnan(normals[..., 0]).sum()))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "_trained_warmup":
        torch.set_flush_denormal(True)           # before the first torch op of this (child) process
        torch.set_num_threads(8)
        trained_warmup(import_reference(), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "default")
        return
    torch.set_num_threads(4)
    mods = import_reference()
    if only == "ingest":
        run_ingest_case(mods, "ingest_small", 31)
        return
    full = dict(H=256, B=2, n_freqs=6, scale_input=0.05937489, scale_output=0.14)
    cam_replica = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)      # replicaCAD.json:10-17
    cam_scannet = dict(H=480, W=640, fx=577.87, fy=577.87, cx=319.5, cy=239.5)
    base_sample = dict(SAMPLE_DEFAULT, n_rays=200)                                  # replicaCAD.json:41-44
    cam_s = dict(H=48, W=64, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    small = dict(H=64, B=1, n_freqs=6, scale_input=0.05937489, scale_output=0.14)
    round2 = {
        # BASELINE.json configs[1]: 5 keyframes x 200 rays x 27 samples, 680x1200, default net, bounds "ray" ...
        "eval_base_680x1200_ray": lambda: run_eval_case(mods, "eval_base_680x1200_ray", full, LOSS_DEFAULT, base_sample,
                                                        5, cam_replica, 41, 0.25, False, exact_frames=True),
        # ... and "pc" (the supervision of the shipped results), same seed => same draws: losses + digests only
        "eval_base_680x1200_pc": lambda: run_eval_case(mods, "eval_base_680x1200_pc", full,
                                                       dict(LOSS_DEFAULT, bounds_method="pc"), base_sample, 5,
                                                       cam_replica, 41, 0.25, False, exact_frames=True, slim=True),
        # BASELINE.json configs[2] geometry: 480x640 (ScanNet-like intrinsics), same net
        "eval_base_480x640_ray": lambda: run_eval_case(mods, "eval_base_480x640_ray", full, LOSS_DEFAULT, base_sample,
                                                       5, cam_scannet, 42, 0.08, False, exact_frames=True),
        # default-size net through the unmodified Trainer.step x3, K=7 > window (select_keyframes, quirk q4)
        "step_full_k7": lambda: run_step_case(mods, "step_full_k7", full, LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=30),
                                              7, cam_s, 23, 0.08, 3, 5, digest=True),
        # oracle pins for configurations the GPU tests exercise: orien_loss, eikonal-only without normals
        "eval_small_orien": lambda: run_eval_case(mods, "eval_small_orien", small, dict(LOSS_DEFAULT, orien_loss=True),
                                                  SAMPLE_DEFAULT, 3, cam_s, 15, 0.08, True),
        "eval_small_eikonly": lambda: run_eval_case(mods, "eval_small_eikonly", small, dict(LOSS_DEFAULT, grad_weight=0.0),
                                                    SAMPLE_DEFAULT, 3, cam_s, 16, 0.08, True, with_normals=False),
    }
    if only == "round2":
        for fn in round2.values():
            fn()
        return
    if only in round2:
        round2[only]()
        return
    # ---- round 3: reference-generated pins for every network SHAPE the kernels are instantiated for, and for the
    # realsense*.json constants.  embedding.py:36-72 (n_freqs = n_embed_funcs + 1 octaves),
    # fc_map.py:77-92 (hidden_layers_block), realsense*.json (720x1280, scale_input 0.4 / 0.04, trunc_weight 30,
    # trunc_distance 0.1, dist_behind_surf 0.01, depth_range[0] 0.1 / 0.15; live modes have NO bounds transform)
    cam_rs = dict(H=720, W=1280, fx=636.1981811523438, fy=635.5728149414062, cx=633.679931640625, cy=372.60797119140625)       # realsense.json:10-17
    cam_franka = dict(H=720, W=1280, fx=913.4483642578125, fy=913.4601440429688, cx=640.4678955078125, cy=359.1015319824219)  # realsense_franka*.json
    loss_franka = dict(LOSS_DEFAULT, trunc_weight=30.0, trunc_distance=0.1)                     # realsense_franka.json:68-77
    net_b3 = lambda nf: dict(H=64, B=3, n_freqs=nf, scale_input=0.05937489, scale_output=0.14)
    round3 = {
        # (i) 64-wide, hidden_layers_block 3, n_freqs 9 / 10 / 11: every tensor + full gradients (oracle pins)
        "eval_small_b3_f9": lambda: run_eval_case(mods, "eval_small_b3_f9", net_b3(9), LOSS_DEFAULT, SAMPLE_DEFAULT, 3, cam_s, 51, 0.08, True),
        "eval_small_b3_f10": lambda: run_eval_case(mods, "eval_small_b3_f10", net_b3(10), LOSS_DEFAULT, SAMPLE_DEFAULT, 3, cam_s, 52, 0.08, True),
        "eval_small_b3_f11": lambda: run_eval_case(mods, "eval_small_b3_f11", net_b3(11), loss_franka, dict(SAMPLE_DEFAULT, dist_behind_surf=0.01), 3,
                                                   cam_s, 53, 0.025, True, transform_on=False),
        # (ii) BASELINE configs[4]: 8 x 512, n_freqs 10 (E = 423) -- the <512,512> instantiation; ~3 k points, digests
        "eval_wide_512": lambda: run_eval_case(mods, "eval_wide_512", dict(H=512, B=3, n_freqs=10, scale_input=0.05937489, scale_output=0.14),
                                               LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=24), 5, cam_s, 54, 0.08, False),
        # (iii) the three realsense configs at their own constants, 720x1280, identity PE transform: the <256,512> instantiation
        "eval_rs_realsense": lambda: run_eval_case(mods, "eval_rs_realsense", dict(H=256, B=2, n_freqs=9, scale_input=0.04, scale_output=0.14),
                                                   LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=60, min_depth=0.15), 5, cam_rs, 55, 0.25, False,
                                                   transform_on=False, exact_frames=True),
        "eval_rs_franka": lambda: run_eval_case(mods, "eval_rs_franka", dict(H=256, B=2, n_freqs=9, scale_input=0.4, scale_output=0.14),
                                                loss_franka, dict(SAMPLE_DEFAULT, n_rays=120, min_depth=0.1), 5, cam_franka, 56, 0.025, False,
                                                transform_on=False, exact_frames=True),
        "eval_rs_franka_offline": lambda: run_eval_case(mods, "eval_rs_franka_offline", dict(H=256, B=3, n_freqs=11, scale_input=0.04, scale_output=0.14),
                                                        loss_franka, dict(SAMPLE_DEFAULT, n_rays=120, min_depth=0.1, dist_behind_surf=0.01), 5,
                                                        cam_franka, 57, 0.025, False, transform_on=False, exact_frames=True),
    }
    if only == "round3":
        for fn in round3.values():
            fn()
        return
    if only in round3:
        round3[only]()
        return
    # ---- round 4: the paper's 4-hidden-layer net (hidden_layers_block = 1, fc_map.py:77-90) at width 256, and the default net
    # at TRAINED weights with a 20-step reference trajectory
    round4 = {
        "eval_b1_256": lambda: run_eval_case(mods, "eval_b1_256", dict(H=256, B=1, n_freqs=6, scale_input=0.05937489, scale_output=0.14),
                                             LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=24), 5, cam_s, 58, 0.08, False),
        "trained_default": lambda: run_trained_case(mods, "trained_default"),
        # hidden widths the tile kernels run zero-padded (NetLayout::H): 128 on the <256, 256> tile, 300 with 10 octaves on <512, 512>
        "eval_h128": lambda: run_eval_case(mods, "eval_h128", dict(H=128, B=2, n_freqs=6, scale_input=0.05937489, scale_output=0.14),
                                           LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=24), 5, cam_s, 59, 0.08, False),
        "eval_h300_f10": lambda: run_eval_case(mods, "eval_h300_f10", dict(H=300, B=2, n_freqs=10, scale_input=0.05937489, scale_output=0.14),
                                               LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=24), 5, cam_s, 60, 0.08, False),
    }
    if only == "round4":
        for fn in round4.values():
            fn()
        return
    if only in round4:
        round4[only]()
        return
    # ---- round 5: realsense_franka.json's network and loss constants at TRAINED weights (at random initialisation
    # the end-to-end gradient of this 9-octave net sits 1.6e-2 from the reference's for ANY 16-bit-operand path -- the non-smooth loss
    # turns forward rounding into flipped residual signs; at a trained state the bar is 1e-2)
    round5 = {
        "trained_franka": lambda: run_trained_case(mods, "trained_franka", seed=62, pre_steps=300, traj_steps=0, spec="franka", grads_fp16=True),
    }
    if only == "round5":
        for fn in round5.values():
            fn()
        return
    if only in round5:
        round5[only]()
        return
    cam_s = dict(H=48, W=64, fx=60.0, fy=60.0, cx=31.5, cy=23.5)
    small = dict(H=64, B=1, n_freqs=6, scale_input=0.05937489, scale_output=0.14)
    full = dict(H=256, B=2, n_freqs=6, scale_input=0.05937489, scale_output=0.14)
    # 1. small net, default loss, noise on, bounds transform on: every tensor + full grads
    run_eval_case(mods, "eval_small_ray", small, LOSS_DEFAULT, SAMPLE_DEFAULT, 3, cam_s, 11, 0.25, True)
    # 2. small net, bounds "pc", L2 loss
    lc = dict(LOSS_DEFAULT, bounds_method="pc", loss_type="L2")
    run_eval_case(mods, "eval_small_pc_l2", small, lc, SAMPLE_DEFAULT, 3, cam_s, 12, 0.08, True)
    # 3. small net, no eikonal/normal terms (no input gradient), identity transform
    lc = dict(LOSS_DEFAULT, eik_weight=0.0, grad_weight=0.0)
    run_eval_case(mods, "eval_small_nograd", small, lc, SAMPLE_DEFAULT, 2, cam_s, 13, 0.04, True,
                  transform_on=False)
    # 4. default-size net (6x256, E=255), digest of grads only (weights regenerate from seed)
    run_eval_case(mods, "eval_full_ray", full, LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=24), 5,
                  cam_s, 14, 0.25, False)
    # 5. unmodified Trainer.step x3 with K=7 keyframes > window 5 (select_keyframes, quirk q4)
    run_step_case(mods, "step_small_k7", small, LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=30), 7,
                  cam_s, 21, 0.08, 3, 5)
    # 6. K=3 <= window: all frames every step
    run_step_case(mods, "step_small_k3", small, LOSS_DEFAULT, dict(SAMPLE_DEFAULT, n_rays=30), 3,
                  cam_s, 22, 0.04, 2, 5)
    # 7. next tier: per-frame normal estimation, keyframe depth render
    run_ingest_case(mods, "ingest_small", 31)
    # 8. round 2: BASELINE-size fixtures, default-net Trainer.step, extra oracle pins
    for fn in round2.values():
        fn()
    for fn in round3.values():
        fn()
    for fn in round4.values():
        fn()
    for fn in round5.values():
        fn()


if __name__ == "__main__":
    main()
