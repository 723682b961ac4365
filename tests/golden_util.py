"""Helpers shared by the tests: load a golden fixture (produced by the real
reference, `tests/golden/make_golden.py`) into oracle-shaped configs."""
import os
import numpy as np

import oracle.isdf_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


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
