"""Helpers shared by the tests: load a golden fixture (produced by the real
reference, `tests/golden/make_golden.py`) into oracle-shaped configs."""
import os
import numpy as np

import oracle.isdf_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    if "depth_batch" not in g and "frames_gen" in g:
        # BASELINE-size keyframes (680x1200x5: 65 MB) are not committed: they are regenerated from
        # the seed by `synth_frames_exact` (IEEE-exact operations only) and verified by checksum
        cam = cam_of(g)
        F, seed = int(g["frames_gen"][0]), int(g["frames_gen"][1])
        d, n, T = synth_frames_exact(seed, F, cam["H"], cam["W"])
        assert frames_checksum(d, n, T) == [int(v) for v in g["frames_sum"]], "regenerated keyframes differ"
        g["depth_batch"], g["normal_batch"], g["T_WC_batch"] = d, n, T
        if "T_WC_sample" not in g and "indices_b" in g:
            g["T_WC_sample"] = T[g["indices_b"]]          # sample.py:63 (dropped from the fixture: redundant)
    return g


def synth_frames(rng, F, H, W, fx, fy, cx, cy):
    """Small posed-depth keyframes: smooth depth 1-4 m, ~4% invalid (0), unit
    normals with a NaN border + a few NaN pixels (as the reference's normal
    estimator leaves them).  Used by tests/golden/make_golden.py for the SMALL fixtures (the
    frames are stored in the fixture)."""
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = np.empty((F, H, W), np.float32)
    normal = np.empty((F, H, W, 3), np.float32)
    T = np.empty((F, 4, 4), np.float32)
    for f in range(F):
        depth[f] = (2.5 + 1.2 * np.sin(0.11 * u + f) * np.cos(0.07 * v - 0.5 * f)
                    + 0.3 * rng.standard_normal((H, W))).astype(np.float32)
        depth[f][rng.uniform(size=(H, W)) < 0.04] = 0.0
        n = rng.standard_normal((H, W, 3)).astype(np.float32)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        n[:2] = np.nan; n[-2:] = np.nan; n[:, :2] = np.nan; n[:, -2:] = np.nan
        n[rng.uniform(size=(H, W)) < 0.02] = np.nan
        normal[f] = n
        a = 0.3 * f
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        b = 0.1 * f
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        T[f] = np.eye(4)
        T[f, :3, :3] = Rm @ Rx
        T[f, :3, 3] = [0.5 * f - 1.0, 0.1 * f, 0.2 * f]
    return depth, normal, T


def bounds_transform(rng=None):
    a, b = 0.4, -0.25
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = (Rz @ Ry).astype(np.float32)
    T[:3, 3] = [0.3, -0.2, 0.1]
    return T


def _rot_exact(t):
    """(cos, sin) of an angle from the rational parametrisation t = tan(angle/2): only + - * /"""
    return (1.0 - t * t) / (1.0 + t * t), 2.0 * t / (1.0 + t * t)


def synth_frames_exact(seed, F, H, W):
    """Full-size synthetic keyframes that regenerate BIT-IDENTICALLY on any IEEE-754 host: only
    + - * / sqrt on float64, integer arithmetic, and RandomState.random_sample (integer -> double).
    Depth: a smooth polynomial surface 1.5-3.6 m + uniform noise, ~4 % invalid (0); normals: random
    unit vectors with a 2-pixel NaN border and ~2 % NaN pixels; poses: rational rotations."""
    rng = np.random.RandomState(seed)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = np.empty((F, H, W), np.float32)
    normal = np.empty((F, H, W, 3), np.float32)
    T = np.empty((F, 4, 4), np.float32)
    for f in range(F):
        a = (u / W - 0.5 + 0.07 * f)
        b = (v / H - 0.5 - 0.05 * f)
        surf = 2.8 + 1.2 * a * b - 0.9 * a * a + 0.6 * b * b * (1.0 - a)
        depth[f] = (surf + 0.25 * (rng.random_sample((H, W)) - 0.5)).astype(np.float32)
        depth[f][rng.random_sample((H, W)) < 0.04] = 0.0
        n = rng.random_sample((H, W, 3)) - 0.5
        n = n / np.sqrt((n * n).sum(-1, keepdims=True))
        n = n.astype(np.float32)
        n[:2] = np.nan; n[-2:] = np.nan; n[:, :2] = np.nan; n[:, -2:] = np.nan
        n[rng.random_sample((H, W)) < 0.02] = np.nan
        normal[f] = n
        c1, s1 = _rot_exact(0.15 * f)
        c2, s2 = _rot_exact(0.05 * f)
        Ry = np.array([[c1, 0, s1], [0, 1, 0], [-s1, 0, c1]])
        Rx = np.array([[1, 0, 0], [0, c2, -s2], [0, s2, c2]])
        T[f] = np.eye(4)
        T[f, :3, :3] = Ry @ Rx
        T[f, :3, 3] = [0.5 * f - 1.0, 0.1 * f, 0.2 * f]
    return depth, normal, T


def frames_checksum(depth, normal, T):
    """order-dependent integer checksum of the raw bit patterns (uint32 words)"""
    out = []
    for a in (depth, normal, T):
        w = np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).ravel()
        k = (np.arange(w.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        out.append(int((w * k).sum() % np.uint64(2 ** 61 - 1)))
    return out


def with_normals(g):
    """False for fixtures of the reference's do_normal=False path (norm_batch None, trainer.py:316-318)"""
    return "with_normals" not in g or bool(g["with_normals"][0])


def cam_of(g):
    H, W, fx, fy, cx, cy = g["cam"]
    return dict(H=int(H), W=int(W), fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy))


def net_of(g, dtype=np.float32):
    H, B, nf, si, so = g["net"]
    has_T = int(g["has_transform"][0]) if "has_transform" in g else 1
    T = g["bounds_T"].astype(dtype) if has_T else None
    return orc.NetCfg(int(H), int(B), int(nf), float(si), float(so), T)


def loss_of(g):
    return orc.LossCfg(
        bounds_method=str(g["loss_bounds_method"]), loss_type=str(g["loss_loss_type"]),
        trunc_weight=float(g["loss_trunc_weight"][0]),
        trunc_distance=float(g["loss_trunc_distance"][0]),
        eik_weight=float(g["loss_eik_weight"][0]),
        eik_apply_dist=float(g["loss_eik_apply_dist"][0]),
        grad_weight=float(g["loss_grad_weight"][0]), orien_loss=bool(g["loss_orien_loss"][0]))


def sample_of(g):
    return dict(n_rays=int(g["sample_n_rays"][0]), n_strat=int(g["sample_n_strat"][0]),
                n_surf=int(g["sample_n_surf"][0]), min_depth=float(g["sample_min_depth"][0]),
                dist_behind_surf=float(g["sample_dist_behind_surf"][0]))


def params_of(g, prefix="param/"):
    p = {k[len(prefix):]: v.copy() for k, v in g.items() if k.startswith(prefix)}
    if not p:  # full-size case: weights regenerate deterministically from the seed
        H, B, nf = int(g["net"][0]), int(g["net"][1]), int(g["net"][2])
        p = orc.init_params(H, B, nf, np.random.RandomState(int(g["seed"][0]) + 100))
    return p


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ---- `trained_default` (round 4): the default net at trained weights, sampler outputs stored instead of keyframes ----------
def trained_batch(g, prefix):
    """One batch of the `trained_default` fixture (prefix "eval/" or "traj/s<k>/"): the reference sampler's outputs in the
    oracle's argument order + the scaled noise it added."""
    ib = g[prefix + "indices_b"].astype(np.int64)
    return dict(pc=g[prefix + "pc"], z_vals=g[prefix + "z_vals"], depth_sample=g[prefix + "depth_sample"],
                dirs_C_sample=g[prefix + "dirs_C_sample"], dirs_W_sample=g[prefix + "dirs_W_sample"],
                T_WC_sample=g["T_WC_batch"][ib], norm_sample=g[prefix + "norm_sample"], noise=g[prefix + "noise"],
                indices_b=ib, indices_h=g[prefix + "indices_h"].astype(np.int64), indices_w=g[prefix + "indices_w"].astype(np.int64))


def trained_eval_grads(g, names):
    """reference gradients of a trained-state fixture's eval batch: stored in full (`trained_default`) or as float16 mantissas on a
    per-tensor power-of-two scale (`trained_franka`: 3e-4 rel-L2 of storage rounding, make_golden.run_trained_case grads_fp16)"""
    out = {}
    for k in names:
        if "eval/grad/" + k in g:
            out[k] = g["eval/grad/" + k].astype(np.float64)
        else:
            out[k] = g["eval/grad16/" + k].astype(np.float64) * float(g["eval/grad16_scale/" + k][0])
    return out


def trained_adam_state(g, names):
    """the trajectory's start state: AdamW moments stored as bfloat16 bit patterns (they ARE the exact start state: the
    reference run loaded these rounded values, make_golden.run_trained_case)"""
    up = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
    return {"step": int(g["adam/step"][0]),
            "exp_avg": {k: up(g["adam/exp_avg/" + k]) for k in names},
            "exp_avg_sq": {k: up(g["adam/exp_avg_sq/" + k]) for k in names}}


def signed_projection(got, ref):
    """<got - ref, ref> / |ref|^2: the part of the error that is ALONG the reference (a scale bias); a rel-L2 bound cannot
    tell it from noise, and unlike noise it does not average out over optimisation steps"""
    a, b = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float((a - b) @ b / max(b @ b, 1e-300))
