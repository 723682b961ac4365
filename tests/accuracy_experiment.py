#!/usr/bin/env python3
"""SDF-accuracy experiment (test infrastructure; not collected by pytest).

The second half of BASELINE.json's metric is "SDF L1 vs GT".  The ReplicaCAD / ScanNet
sequences and GT grids are absent (download-only), so this runs the analogous experiment on
the synthetic analytic room (isdf_amd/synthetic.py, closed-form GT SDF) with a PINNED
schedule (the reference's frame schedule depends on measured step time, so a
faster step changes the experiment): a new keyframe every `--steps-per-kf` optimisation
steps, window of 5 keyframes chosen by `select_keyframes` (trainer.py:652-674), replicaCAD
loss/sample/optimiser settings on the 480x640 (ScanNet-like) camera.

Metric (eval_pts.py:332-400 / trainer.py:1819-1866 analogue): mean |sdf_pred - sdf_gt| over
points sampled one-per-ray uniformly in [min_depth, depth + 0.1] along rays of all keyframes
seen (visible region), plus the surface-only L1 (points at the measured depth).

    python tests/accuracy_experiment.py --backend hip  --seeds 1 2 3      # on the MI355X
    python tests/accuracy_experiment.py --backend port --seeds 1          # reference op chain, CPU
    python tests/accuracy_experiment.py --backend port --device cuda ...  # the same op chain as PyTorch-ROCm eager, fp32: THE CONTROL
    python tests/accuracy_experiment.py --native-clock --backend hip|port [--device cuda] [--max-steps 20000]
        # BASELINE configs[2] / SURVEY 8d Metric 2: the reference driver loop (train.py:86-136) with the MEASURED step time as
        # the virtual clock over the 600-frame 480x640 sequence, L1 vs GT at the reference's eval cadence (eval_freq_s = 1)

`tools/accuracy_stats.py a.json b.json` reports paired per-seed differences and Welch's t-test between two result files.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isdf_amd import synthetic  # noqa: E402


def config(cam):
    return {
        "dataset": {"camera": {"w": cam["W"], "h": cam["H"], "fx": cam["fx"], "fy": cam["fy"], "cx": cam["cx"],
                               "cy": cam["cy"]}},
        "optimiser": {"lr": 0.0013, "weight_decay": 0.012},
        "sample": {"n_rays": 200, "n_rays_is_kf": 400, "n_strat_samples": 19, "n_surf_samples": 8,
                   "depth_range": [0.07, 12.0], "dist_behind_surf": 0.1},
        "model": {"do_active": 0, "frac_time_perception": 1.0, "scale_output": 0.14, "noise_std": 0.25,
                  "noise_kf": 0.08, "noise_frame": 0.04, "window_size": 5, "hidden_layers_block": 2,
                  "hidden_feature_size": 256, "iters_per_kf": 60, "iters_per_frame": 10,
                  "kf_dist_th": 0.1, "kf_pixel_ratio": 0.65,
                  "embedding": {"scale_input": 0.05937489, "n_embed_funcs": 5}},
        "loss": {"bounds_method": "ray", "loss_type": "L1", "trunc_weight": 5.38344020,
                 "trunc_distance": 0.29365022, "eik_weight": 0.268, "eik_apply_dist": 0.1,
                 "grad_weight": 0.018, "orien_loss": 0},
    }


def eval_points(depth, T, cam, rng, n_per_frame=16000):
    """visible-region evaluation points: one uniform sample per ray in [0.07, depth+0.1]"""
    dc = synthetic.dirs_C(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    pts, surf = [], []
    for f in range(depth.shape[0]):
        ok = np.argwhere(depth[f] > 0)
        sel = ok[rng.choice(len(ok), n_per_frame, replace=False)]
        d = depth[f][sel[:, 0], sel[:, 1]].astype(np.float64)
        dw = dc[sel[:, 0], sel[:, 1]] @ T[f][:3, :3].astype(np.float64).T
        z = rng.uniform(0.07, d + 0.1)
        pts.append(T[f][:3, 3] + dw * z[:, None])
        surf.append(T[f][:3, 3] + dw * d[:, None])
    return np.concatenate(pts), np.concatenate(surf)


def _kf_job(args):
    seed, n, path = args
    cam = dict(synthetic.SCANNET_CAM)
    depth, normal, T = synthetic.keyframes(n, cam, seed=seed, stride=48, noise_std=0.01)
    np.savez(path, depth=depth, normal=normal, T=T)
    return seed


def prepare_keyframes(seeds, n):
    """The pinned-schedule inputs of every seed (numpy ray casting + the reference's normal estimation: 0.55 s per keyframe),
    generated in parallel BEFORE any device is touched and cached on local disk, so that several runs over the same seeds on
    one (billed) GPU box pay for them once."""
    import multiprocessing as mp
    import tempfile
    root = os.path.join(tempfile.gettempdir(), "isdf_kf_cache")
    os.makedirs(root, exist_ok=True)
    paths = {sd: os.path.join(root, "kf_%d_%d.npz" % (sd, n)) for sd in seeds}
    todo = [(sd, n, paths[sd]) for sd in seeds if not os.path.exists(paths[sd])]
    if todo:
        with mp.get_context("fork").Pool(min(len(todo), max(1, (os.cpu_count() or 2) // 2), 16)) as pool:
            pool.map(_kf_job, todo)

    def load(sd):
        z = np.load(paths[sd])
        return z["depth"], z["normal"], z["T"]
    return load


FWD_OPERAND = "fp16x2"   # --fwd-operand: the HIP path's operand mode for the pinned-schedule runs
BWD_OPERAND = None       # --bwd-operand: fp16 | bf16 (None: the forward mode's default)
PAIRED = False           # --paired-draws: both backends draw from torch's global generators in the reference's order
DEVICE = "cpu"           # --device: where the port backend runs ("cuda" = PyTorch-ROCm eager, the fp32 control on the same GPU)


def run_hip(seed, depth, normal, T, cam, steps_per_kf):
    from bench_support.standin_trainer import HipTrainer, FrameData
    np.random.seed(seed); torch.manual_seed(seed)
    tr = HipTrainer("cuda", config(cam), inv_bounds_transform=synthetic.bounds_transform(), rng="torch" if PAIRED else "philox",
                    seed=seed, fwd_operand=FWD_OPERAND, bwd_operand=BWD_OPERAND)
    dev = tr.device
    if PAIRED:      # the same initial network as the control: torch's xavier draw under the same seed (PortNet's constructor order)
        from oracle import torch_port as tp
        torch.manual_seed(seed)
        net0 = tp.PortNet(256, 2, 6, 0.05937489, 0.14, synthetic.bounds_transform())
        tr.sdf_map.load_state_dict(net0.state_dict())
    t_train = 0.0
    for k in range(depth.shape[0]):
        fd = FrameData(frame_id=np.array([k]), depth_batch=torch.from_numpy(depth[k:k + 1]).to(dev),
                       T_WC_batch=torch.from_numpy(T[k:k + 1]).to(dev),
                       normal_batch=torch.from_numpy(normal[k:k + 1]).to(dev))
        tr.last_is_keyframe = True          # pinned schedule: every ingested frame is a keyframe
        tr.frames.add_frame_data(fd, replace=False)
        tr.noise_std = tr.noise_kf
        for _ in range(steps_per_kf):
            losses, ms = tr.step()
            t_train += ms
    return (lambda p: tr.sdf_map(torch.from_numpy(p.astype(np.float32)).to(dev)).cpu().numpy()), \
        float(losses["total_loss"]), t_train / 1e3


def run_hip_reference_schedule(seed, cam, n_steps=1200, virtual_step_ms=20.0, n_frames=600, quiet=True):
    """The reference driver's frame scheduling (train.py:86-136, restated in bench_support/driver_loop.py) on the
    synthetic 30 fps stream: after `optim_frames` steps on the latest frame, `check_keyframe_latest` (keyframe
    test on the frozen net, trainer.py:586-650) decides whether it becomes a keyframe; the next frame id is
    int(tot_step_time * fps) (trainer.py:100).  The virtual clock advances by `virtual_step_ms` per step
    (pinned) instead of the measured step time so the schedule is reproducible."""
    import contextlib, io
    from bench_support.standin_trainer import HipTrainer
    from bench_support.driver_loop import run_train_loop
    np.random.seed(seed); torch.manual_seed(seed)
    tr = HipTrainer("cuda", config(cam), inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=seed,
                    virtual_step_ms=virtual_step_ms)
    traj = synthetic.trajectory(n_frames)
    rng = np.random.RandomState(seed)
    seen = {}

    def frame(i):
        if i not in seen:
            seen[i] = synthetic.render_depth(traj[i], cam, rng, noise_std=0.01)
        return tr.make_frame(i, seen[i], traj[i])
    sink = io.StringIO()
    with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
        n, ingests, losses = run_train_loop(tr, frame, n_frames, n_steps)
    ids = [int(i) for i in tr.frames.frame_id]
    depth = np.stack([seen[i] for i in ids]); T = np.stack([traj[i] for i in ids])
    fn = lambda p: tr.sdf_map(torch.from_numpy(p.astype(np.float32)).to(tr.device)).cpu().numpy()
    return fn, float(losses["total_loss"]), depth, T, ids, n


def run_port_reference_schedule(seed, cam, n_steps=1200, virtual_step_ms=20.0, n_frames=600, threads=None):
    """CONTROL for run_hip_reference_schedule: the reference's op chain on torch CPU (oracle/torch_port.PortTrainer)
    under the same driver loop, stream, seeds and pinned virtual clock."""
    import contextlib, io
    import oracle.isdf_oracle as orc
    from oracle.torch_port import PortTrainer
    from bench_support.driver_loop import run_train_loop
    if threads:
        torch.set_num_threads(threads)
    np.random.seed(seed); torch.manual_seed(seed)
    torch.set_flush_denormal(True)
    tr = PortTrainer(config(cam), cam, synthetic.bounds_transform(), seed, virtual_step_ms, device=DEVICE)
    traj = synthetic.trajectory(n_frames)
    rng = np.random.RandomState(seed)
    seen = {}

    def frame(i):
        if i not in seen:
            seen[i] = synthetic.render_depth(traj[i], cam, rng, noise_std=0.01)
        pc = orc.pointcloud_from_depth(seen[i], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        nrm = orc.estimate_pointcloud_normals(pc).astype(np.float32)
        return (i, torch.from_numpy(seen[i]), torch.from_numpy(traj[i]), torch.from_numpy(nrm))
    with contextlib.redirect_stdout(io.StringIO()):
        n, ingests, losses = run_train_loop(tr, frame, n_frames, n_steps)
    ids = [int(i) for i in tr.frame_id]
    depth = np.stack([seen[i] for i in ids]); T = np.stack([traj[i] for i in ids])

    def fn(p):
        with torch.no_grad():
            return tr.net(torch.from_numpy(p.astype(np.float32)).to(tr.device)).cpu().numpy()
    return fn, float(losses["total_loss"]), depth, T, ids, n


def run_port(seed, depth, normal, T, cam, steps_per_kf):
    from oracle import torch_port as tp
    np.random.seed(seed); torch.manual_seed(seed)
    torch.set_flush_denormal(True)
    dev = torch.device(DEVICE)
    net = tp.PortNet(256, 2, 6, 0.05937489, 0.14, synthetic.bounds_transform()).to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=0.0013, weight_decay=0.012)
    gen = None if PAIRED else torch.Generator().manual_seed(seed)
    c = config(cam)
    sc = dict(n_rays=200, n_strat=19, n_surf=8, min_depth=0.07, dist_behind_surf=0.1)
    lc = dict(trunc_distance=c["loss"]["trunc_distance"], loss_type="L1", trunc_weight=c["loss"]["trunc_weight"],
              eik_apply_dist=0.1, eik_weight=c["loss"]["eik_weight"], grad_weight=c["loss"]["grad_weight"])
    d, n, Tt = torch.from_numpy(depth).to(dev), torch.from_numpy(normal).to(dev), torch.from_numpy(T).to(dev)
    fal = torch.zeros(0, device=dev)
    t0 = time.perf_counter()
    for k in range(depth.shape[0]):
        fal = torch.cat((fal, torch.zeros(1, device=dev)))
        K = k + 1
        for _ in range(steps_per_kf):
            if K > 5:       # select_keyframes, trainer.py:652-674
                p = (fal[:-2] / fal[:-2].sum()).cpu().numpy()
                idxs = [*np.random.choice(np.arange(0, K - 2), size=3, replace=False, p=p), K - 2, K - 1]
            else:
                idxs = list(range(K))
            # quirk q4: normals come from the un-windowed batch with window-local indices
            losses, fa = tp.train_step(net, opt, d[idxs], Tt[idxs], n[:len(idxs)], cam, sc, lc, 0.08, gen)
            fal[idxs] = fa
    el = time.perf_counter() - t0
    def fn(p):
        with torch.no_grad():
            return net(torch.from_numpy(p.astype(np.float32)).to(dev)).cpu().numpy()
    return fn, losses["total_loss"], el



# ---------------------------------------------------------------------------------------------------------------
# Native-clock sequence run (BASELINE configs[2], SURVEY 8d Metric 2): the reference's driver loop with the MEASURED
# step time as the virtual clock (trainer.py:100-101,1011-1013), over the whole 600-frame 480x640 stream.
# ---------------------------------------------------------------------------------------------------------------
def render_depth_torch(T_WC, cam, device):
    """synthetic.raycast / render_depth restated with torch float64 ops so that the 600 frames of the stream render in
    under a second on the GPU box (numpy: ~0.15 s per frame of billed GPU-box time).  Clean z-depth, 0 beyond 12 m."""
    H, W = cam["H"], cam["W"]
    f64 = dict(dtype=torch.float64, device=device)
    c = torch.arange(W, **f64)[None, :].expand(H, W)
    r = torch.arange(H, **f64)[:, None].expand(H, W)
    dc = torch.stack(((c - cam["cx"]) / cam["fx"], (r - cam["cy"]) / cam["fy"], torch.ones_like(c)), -1)
    Tm = torch.as_tensor(np.asarray(T_WC, np.float64), **f64)
    d = dc @ Tm[:3, :3].T
    o = Tm[:3, 3]
    lo, hi = torch.as_tensor(synthetic.ROOM_LO, **f64), torch.as_tensor(synthetic.ROOM_HI, **f64)
    inf = torch.full_like(d, float("inf"))
    tw = torch.where(d > 0, (hi - o) / d, torch.where(d < 0, (lo - o) / d, inf))
    t = tw.min(-1).values
    a = (d * d).sum(-1)
    for cc, rr in synthetic.SPHERES:
        oc = o - torch.as_tensor(cc, **f64)
        b = 2 * (d * oc).sum(-1)
        c0 = (oc * oc).sum(-1) - rr * rr
        disc = b * b - 4 * a * c0
        ts = (-b - torch.sqrt(torch.clamp(disc, min=0))) / (2 * a)
        t = torch.where((disc > 0) & (ts > 1e-6), torch.minimum(t, ts), t)
    for blo, bhi in synthetic.BOXES:
        t1, t2 = (torch.as_tensor(blo, **f64) - o) / d, (torch.as_tensor(bhi, **f64) - o) / d
        tn = torch.minimum(t1, t2).max(-1).values
        tf = torch.maximum(t1, t2).min(-1).values
        t = torch.where((tn <= tf) & (tn > 1e-6), torch.minimum(t, tn), t)
    depth = t.to(torch.float32)
    depth[depth > 12.0] = 0.0
    return depth


class FrameStream:
    """The synthetic 30 fps stream for the native-clock runs: clean depth of every frame rendered once (shared by all
    seeds and backends), per-(seed, frame) sensor noise (1 cm) + 2 % invalid pixels added on the fly, normals from
    the HIP ingest stencil for BOTH backends (input preparation, identical for the product and the control)."""

    def __init__(self, cam, n_frames, device, eval_stride=4):
        self.cam, self.n, self.dev = cam, n_frames, torch.device(device)
        self.traj = synthetic.trajectory(n_frames)
        self.clean = torch.stack([render_depth_torch(self.traj[i], cam, self.dev) for i in range(n_frames)])
        self.eval_stride = eval_stride
        self._norm_eng = None

    def noisy(self, seed, i):
        """sensor noise (1 cm) + 2 % invalid pixels of frame i for this seed, drawn ON the device (a host-side draw of 2 x 307 k
        values kept the GPU idle for ~10 ms per ingested frame: call 1 of round 4)"""
        g = torch.Generator(device=self.dev).manual_seed(seed * 1000003 + i)
        nz = torch.randn(self.cam["H"], self.cam["W"], generator=g, device=self.dev) * 0.01
        drop = torch.rand(self.cam["H"], self.cam["W"], generator=g, device=self.dev) < 0.02
        d = self.clean[i]
        d = torch.where((d > 0) & ~drop, d + nz, torch.zeros_like(d))
        d[d > 12.0] = 0.0
        return d

    def normals(self, depth):
        if self.dev.type != "cuda":      # plumbing checks on a GPU-less host
            return torch.from_numpy(synthetic.estimate_normals(depth.numpy(), self.cam))
        from isdf_amd.engine import Engine, NetConfig, SampleConfig
        if self._norm_eng is None:
            self._norm_eng = Engine(NetConfig(), self.dev)
        c = self.cam
        sc = SampleConfig(H=c["H"], W=c["W"], fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"])
        return self._norm_eng.estimate_normals(depth, sc)

    def eval_l1(self, sdf_fn, t_virtual, gen, samples=200000):
        """trainer.eval_sdf(visible_region=True) (trainer.py:1819-1900): `samples` rays spread over the frames of the stream
        seen so far (here every `eval_stride`-th of them), one uniform sample per ray in [min_depth, depth + 0.1]; mean |sdf - gt|.
        Returns (visible-region L1, surface L1)."""
        n_seen = max(1, min(self.n, int(t_virtual * 30)))
        ids = torch.arange(0, n_seen, self.eval_stride)
        per = samples // len(ids)
        H, W, c = self.cam["H"], self.cam["W"], self.cam
        fi = ids.repeat_interleave(per).to(self.dev)
        ih = torch.randint(0, H, (len(fi),), generator=gen).to(self.dev)
        iw = torch.randint(0, W, (len(fi),), generator=gen).to(self.dev)
        d = self.clean[fi, ih, iw]
        ok = d > 0
        fi, ih, iw, d = fi[ok], ih[ok], iw[ok], d[ok].double()
        Tm = torch.as_tensor(self.traj, dtype=torch.float64, device=self.dev)[fi]
        dC = torch.stack(((iw.double() - c["cx"]) / c["fx"], (ih.double() - c["cy"]) / c["fy"], torch.ones_like(d)), -1)
        dW = (Tm[:, :3, :3] * dC[:, None, :]).sum(-1)
        u = torch.rand(len(d), generator=gen, dtype=torch.float64).to(self.dev)
        z = 0.07 + u * (d + 0.1 - 0.07)
        pts = Tm[:, :3, 3] + dW * z[:, None]
        surf = Tm[:, :3, 3] + dW * d[:, None]
        with torch.no_grad():
            pv = sdf_fn(pts.float()).double().cpu().numpy()
            sv = sdf_fn(surf.float()).double().cpu().numpy()
        return (float(np.abs(pv - synthetic.gt_sdf(pts.cpu().numpy())).mean()),
                float(np.abs(sv - synthetic.gt_sdf(surf.cpu().numpy())).mean()))


def run_native_clock(backend, seed, stream, max_steps, extra_opt_steps=400, eval_freq_s=1.0, virtual_step_ms=None):
    """train.py:86-279 with the product (or the eager control) behind `Trainer.step`: frame id = int(tot_step_time * fps), the
    clock advances by the measured, device-synchronised step time, evaluation every `eval_freq_s` of VIRTUAL time, the
    run ends `extra_opt_steps` after the stream does (train.py:30,114) or at `max_steps` (replicaCAD.json "steps": 20000).
    virtual_step_ms: pin the clock instead (the `--reference-schedule` mode: same loop, reproducible schedule)."""
    import contextlib, io
    from bench_support.driver_loop import run_train_loop
    cam = stream.cam
    np.random.seed(seed); torch.manual_seed(seed)
    if backend == "hip":
        from bench_support.standin_trainer import HipTrainer
        tr = HipTrainer("cuda", config(cam), inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=seed,
                        fwd_operand=FWD_OPERAND, bwd_operand=BWD_OPERAND, virtual_step_ms=virtual_step_ms)
        sdf_fn = lambda p: tr.sdf_map(p)
        frames_of = lambda: [int(i) for i in tr.frames.frame_id]

        def frame(i):
            d = stream.noisy(seed, i)
            return tr.make_frame(i, d, stream.traj[i])
    else:
        from oracle.torch_port import PortTrainer
        tr = PortTrainer(config(cam), cam, synthetic.bounds_transform(), seed, virtual_step_ms, device=DEVICE)
        sdf_fn = lambda p: tr.net(p.to(tr.device))
        frames_of = lambda: [int(i) for i in tr.frame_id]

        def frame(i):
            d = stream.noisy(seed, i)
            return (i, d, torch.from_numpy(stream.traj[i]), stream.normals(d))
    gen = torch.Generator(device="cpu").manual_seed(7000 + seed)
    curve, state = [], dict(last_eval=0.0, step_ms=[], pos=[])

    def on_step(t, losses, step_ms):
        state["step_ms"].append(step_ms)
        state["pos"].append(tr.steps_since_frame)       # 1 = the first step after a frame ingest
        tt = tr.tot_step_time
        if tt - state["last_eval"] > eval_freq_s:                     # train.py:245-248
            state["last_eval"] = tt - tt % eval_freq_s
            l1, l1s = stream.eval_l1(sdf_fn, tt, gen)
            curve.append(dict(t=round(tt, 3), step=t, l1_visible_m=round(l1, 5), l1_surface_m=round(l1s, 5),
                              keyframes=len(frames_of()) - 1))
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        n, ingests, losses = run_train_loop(tr, frame, stream.n, max_steps, on_step=on_step)
        ended = n < max_steps
        if ended:                                                     # end of sequence: train.py:113-117
            for k in range(min(extra_opt_steps, max_steps - n)):
                losses, ms = tr.step()
                state["step_ms"].append(ms)
            n += min(extra_opt_steps, max_steps - n)
    wall = time.perf_counter() - t0
    l1, l1s = stream.eval_l1(sdf_fn, tr.tot_step_time, gen)
    ms = np.array(state["step_ms"])
    pos = np.array(state["pos"] + [0] * (len(ms) - len(state["pos"])))
    by_pos = {int(k): round(float(ms[pos == k].mean()), 4) for k in (1, 2, 3, 5, 10) if (pos == k).any()}
    pct = {q: round(float(np.percentile(ms, q)), 4) for q in (10, 50, 90, 99)}
    return dict(step_ms_percentiles=pct, step_ms_mean_by_steps_since_ingest=by_pos, backend=backend if backend == "hip" else "port-%s" % DEVICE,
                schedule="native-clock" if virtual_step_ms is None else "reference-driver@%gms" % virtual_step_ms, seed=seed, steps=int(n),
                max_steps=max_steps, reached_end_of_sequence=bool(ended), virtual_seconds=round(float(tr.tot_step_time), 3),
                wall_seconds=round(wall, 2), mean_step_ms=round(float(ms.mean()), 4), median_step_ms=round(float(np.median(ms)), 4),
                frames_ingested=len(ingests), keyframes_kept=len(frames_of()) - 1, keyframe_ids=frames_of(),
                l1_visible_m=round(l1, 5), l1_surface_m=round(l1s, 5),
                final_total_loss=round(float(losses["total_loss"]), 5), curve=curve)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="hip", choices=["hip", "port"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--keyframes", type=int, default=12)
    ap.add_argument("--steps-per-kf", type=int, default=60)
    ap.add_argument("--out", default=None)
    ap.add_argument("--reference-schedule", action="store_true",
                    help="reference driver frame scheduling + keyframe test (train.py:86-136) instead of the pinned schedule")
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--virtual-step-ms", type=float, default=20.0)
    ap.add_argument("--fwd-operand", default="fp16x2", choices=["fp16x2", "fp16", "bf16", "fp16x2_full"])
    ap.add_argument("--bwd-operand", default=None, choices=["fp16", "bf16"],
                    help="HIP backend: operand / spill type of the second-order sweeps and dW (default: fp16 with an fp16-family forward)")
    ap.add_argument("--device", default="cpu", help="device of the port backend (cuda = PyTorch-ROCm eager fp32: the control)")
    ap.add_argument("--paired-draws", action="store_true",
                    help="pinned schedule: both backends start from the SAME initial network and draw pixels / offsets / noise from "
                         "torch's global generators in the reference's order (HipTrainer rng='torch'; port gen=None): identical "
                         "random streams, so what differs between the two runs is the arithmetic")
    ap.add_argument("--native-clock", action="store_true",
                    help="reference driver loop with the MEASURED step time as the virtual clock over the whole stream")
    ap.add_argument("--frames", type=int, default=600, help="length of the 30 fps stream (native-clock mode)")
    ap.add_argument("--max-steps", type=int, default=1000000,
                    help="native-clock mode: optimisation-step cap (replicaCAD.json trainer.steps is 20000)")
    a = ap.parse_args()
    global FWD_OPERAND, DEVICE, PAIRED, BWD_OPERAND
    FWD_OPERAND, DEVICE, PAIRED, BWD_OPERAND = a.fwd_operand, a.device, a.paired_draws, a.bwd_operand
    cam = dict(synthetic.SCANNET_CAM)
    res = []
    if a.native_clock or a.reference_schedule:
        # one driver loop (train.py:86-136) for both: native clock = measured step time, reference schedule = pinned clock
        stream = FrameStream(cam, a.frames, "cuda" if torch.cuda.is_available() else "cpu")
        for seed in a.seeds:
            r = run_native_clock(a.backend, seed, stream, a.max_steps if a.native_clock else a.steps,
                                 extra_opt_steps=400 if a.native_clock else 0,
                                 virtual_step_ms=None if a.native_clock else a.virtual_step_ms)
            print(json.dumps({k: v for k, v in r.items() if k != "curve"}), flush=True)
            res.append(r)
        l1 = np.array([r["l1_visible_m"] for r in res])
        summary = dict(backend=res[0]["backend"], n=len(res), l1_visible_mean=round(float(l1.mean()), 5),
                       l1_visible_sd=round(float(l1.std()), 5),
                       l1_surface_mean=round(float(np.mean([r["l1_surface_m"] for r in res])), 5),
                       steps_mean=float(np.mean([r["steps"] for r in res])),
                       keyframes_mean=float(np.mean([r["keyframes_kept"] for r in res])),
                       mean_step_ms=float(np.mean([r["mean_step_ms"] for r in res])))
        print(json.dumps(summary), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(dict(runs=res, summary=summary), f, indent=1)
        return
    load_kf = prepare_keyframes(a.seeds, a.keyframes)
    for seed in a.seeds:
        depth, normal, T = load_kf(seed)
        fn, last_loss, t = (run_hip if a.backend == "hip" else run_port)(seed, depth, normal, T, cam, a.steps_per_kf)
        rng = np.random.RandomState(1000 + seed)
        pts, surf = eval_points(depth, T, cam, rng)
        with torch.no_grad():
            l1 = float(np.abs(fn(pts) - synthetic.gt_sdf(pts)).mean())
            l1s = float(np.abs(fn(surf) - synthetic.gt_sdf(surf)).mean())
        r = dict(backend=a.backend if a.backend == "hip" else "port-%s" % a.device, seed=seed, keyframes=a.keyframes,
                 steps=a.keyframes * a.steps_per_kf,
                 l1_visible_m=round(l1, 5), l1_surface_m=round(l1s, 5), final_total_loss=round(float(last_loss), 5),
                 train_seconds=round(t, 2))
        print(json.dumps(r), flush=True)
        res.append(r)
    l1 = np.array([r["l1_visible_m"] for r in res])
    summary = dict(backend=a.backend, n=len(res), l1_visible_mean=round(float(l1.mean()), 5),
                   l1_visible_sd=round(float(l1.std()), 5),
                   l1_surface_mean=round(float(np.mean([r["l1_surface_m"] for r in res])), 5))
    print(json.dumps(summary), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(runs=res, summary=summary), f, indent=1)


if __name__ == "__main__":
    main()
