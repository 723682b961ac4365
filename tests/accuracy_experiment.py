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


FWD_OPERAND = "fp16x2"   # --fwd-operand: the HIP path's operand mode for the pinned-schedule runs


def run_hip(seed, depth, normal, T, cam, steps_per_kf):
    from isdf_amd.trainer import HipTrainer, FrameData
    np.random.seed(seed); torch.manual_seed(seed)
    tr = HipTrainer("cuda", config(cam), inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=seed,
                    fwd_operand=FWD_OPERAND)
    dev = tr.device
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
    """The reference driver's frame scheduling (train.py:86-136, restated in tests/driver_loop.py) on the
    synthetic 30 fps stream: after `optim_frames` steps on the latest frame, `check_keyframe_latest` (keyframe
    test on the frozen net, trainer.py:586-650) decides whether it becomes a keyframe; the next frame id is
    int(tot_step_time * fps) (trainer.py:100).  The virtual clock advances by `virtual_step_ms` per step
    (pinned) instead of the measured step time so the schedule is reproducible."""
    import contextlib, io
    from isdf_amd.trainer import HipTrainer
    from tests.driver_loop import run_train_loop
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
    from tests.driver_loop import run_train_loop
    if threads:
        torch.set_num_threads(threads)
    np.random.seed(seed); torch.manual_seed(seed)
    torch.set_flush_denormal(True)
    tr = PortTrainer(config(cam), cam, synthetic.bounds_transform(), seed, virtual_step_ms)
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
    with torch.no_grad():
        fn = lambda p: tr.net(torch.from_numpy(p.astype(np.float32))).detach().numpy()
    return fn, float(losses["total_loss"]), depth, T, ids, n


def run_port(seed, depth, normal, T, cam, steps_per_kf):
    from oracle import torch_port as tp
    np.random.seed(seed); torch.manual_seed(seed)
    torch.set_flush_denormal(True)
    net = tp.PortNet(256, 2, 6, 0.05937489, 0.14, synthetic.bounds_transform())
    opt = torch.optim.AdamW(net.parameters(), lr=0.0013, weight_decay=0.012)
    gen = torch.Generator().manual_seed(seed)
    c = config(cam)
    sc = dict(n_rays=200, n_strat=19, n_surf=8, min_depth=0.07, dist_behind_surf=0.1)
    lc = dict(trunc_distance=c["loss"]["trunc_distance"], loss_type="L1", trunc_weight=c["loss"]["trunc_weight"],
              eik_apply_dist=0.1, eik_weight=c["loss"]["eik_weight"], grad_weight=c["loss"]["grad_weight"])
    d, n, Tt = torch.from_numpy(depth), torch.from_numpy(normal), torch.from_numpy(T)
    fal = torch.zeros(0)
    t0 = time.perf_counter()
    for k in range(depth.shape[0]):
        fal = torch.cat((fal, torch.zeros(1)))
        K = k + 1
        for _ in range(steps_per_kf):
            if K > 5:       # select_keyframes, trainer.py:652-674
                p = (fal[:-2] / fal[:-2].sum()).numpy()
                idxs = [*np.random.choice(np.arange(0, K - 2), size=3, replace=False, p=p), K - 2, K - 1]
            else:
                idxs = list(range(K))
            # quirk q4: normals come from the un-windowed batch with window-local indices
            losses, fa = tp.train_step(net, opt, d[idxs], Tt[idxs], n[:len(idxs)], cam, sc, lc, 0.08, gen)
            fal[idxs] = fa
    el = time.perf_counter() - t0
    with torch.no_grad():
        return (lambda p: net(torch.from_numpy(p.astype(np.float32))).numpy()), losses["total_loss"], el


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
    a = ap.parse_args()
    global FWD_OPERAND
    FWD_OPERAND = a.fwd_operand
    cam = dict(synthetic.SCANNET_CAM)
    res = []
    if a.reference_schedule:
        for seed in a.seeds:
            run = run_hip_reference_schedule if a.backend == "hip" else run_port_reference_schedule
            fn, last, depth, T, ids, n = run(seed, cam, a.steps, a.virtual_step_ms)
            pts, surf = eval_points(depth, T, cam, np.random.RandomState(1000 + seed), n_per_frame=8000)
            r = dict(backend=a.backend, schedule="reference", seed=seed, steps=n, virtual_step_ms=a.virtual_step_ms,
                     keyframe_ids=ids,
                     l1_visible_m=round(float(np.abs(fn(pts) - synthetic.gt_sdf(pts)).mean()), 5),
                     l1_surface_m=round(float(np.abs(fn(surf) - synthetic.gt_sdf(surf)).mean()), 5),
                     final_total_loss=round(last, 5))
            print(json.dumps(r), flush=True)
            res.append(r)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(dict(runs=res), f, indent=1)
        return
    for seed in a.seeds:
        depth, normal, T = synthetic.keyframes(a.keyframes, cam, seed=seed, stride=48, noise_std=0.01)
        fn, last_loss, t = (run_hip if a.backend == "hip" else run_port)(seed, depth, normal, T, cam, a.steps_per_kf)
        rng = np.random.RandomState(1000 + seed)
        pts, surf = eval_points(depth, T, cam, rng)
        with torch.no_grad():
            l1 = float(np.abs(fn(pts) - synthetic.gt_sdf(pts)).mean())
            l1s = float(np.abs(fn(surf) - synthetic.gt_sdf(surf)).mean())
        r = dict(backend=a.backend, seed=seed, keyframes=a.keyframes, steps=a.keyframes * a.steps_per_kf,
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
