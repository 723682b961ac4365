"""Synthetic posed-depth stream with a closed-form ground-truth SDF.

Stand-in for the ReplicaCAD / ScanNet sequences, which are download-only
(`/root/reference/data/download_apt_2_nav.sh:8`) and absent here:
an analytic room -- axis-aligned box interior 6 x 3 x 5 m, three spheres and two
floor boxes -- seen from a smooth Lissajous trajectory, rendered to z-depth by
analytic ray casting, with the reference's depth conventions (0 = invalid,
far-clipped like `image_transforms.DepthFilter`, `image_transforms.py:29-38`)
and per-pixel normals estimated exactly as the reference ingests them
(8-neighbour cross products on the camera-frame point cloud,
`isdf/geometry/transform.py:169-196,215-270`).  numpy only; this is input
generation, not the hot path.
"""
import numpy as np

ROOM_LO = np.array([0.0, 0.0, 0.0])
ROOM_HI = np.array([6.0, 3.0, 5.0])            # y is "down" in camera terms; the floor is y = 3
SPHERES = [(np.array([1.5, 2.5, 1.5]), 0.5), (np.array([4.2, 2.5, 3.6]), 0.5), (np.array([3.0, 1.2, 4.3]), 0.5)]
BOXES = [(np.array([4.0, 2.2, 0.6]), np.array([5.2, 3.0, 1.6])), (np.array([0.6, 2.4, 3.2]), np.array([1.8, 3.0, 4.4]))]


def gt_sdf(p):
    """Closed-form signed distance of world points p [...,3] (positive in free space)."""
    p = np.asarray(p, np.float64)
    d = np.minimum(p - ROOM_LO, ROOM_HI - p).min(-1)          # distance to the nearest wall (inside)
    for c, r in SPHERES:
        d = np.minimum(d, np.linalg.norm(p - c, axis=-1) - r)
    for lo, hi in BOXES:
        q = np.maximum(lo - p, p - hi)
        outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
        inside = np.minimum(q.max(-1), 0.0)
        d = np.minimum(d, outside + inside)
    return d


def bounds_transform():
    """inv_bounds_transform stand-in: world -> room-centred frame (trainer.py:419-425)."""
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = -(ROOM_LO + ROOM_HI) / 2
    return T


def look_at(eye, target, up=np.array([0.0, -1.0, 0.0])):
    """Camera-to-world pose, OpenCV/Replica convention (x right, y down, z forward)."""
    z = target - eye
    z = z / np.linalg.norm(z)
    x = np.cross(z, up)               # right-handed with y = z cross x pointing "down"
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, eye
    return T.astype(np.float32)


def trajectory(n_frames, fps=30.0):
    """600-pose style Lissajous path at ~1.5 m height looking at a moving target."""
    t = np.arange(n_frames) / fps
    eye = np.stack([3.0 + 1.6 * np.sin(0.31 * t), 1.5 + 0.25 * np.sin(0.53 * t),
                    2.5 + 1.2 * np.sin(0.47 * t + 0.8)], axis=-1)
    tgt = np.stack([3.0 + 2.4 * np.cos(0.23 * t + 0.4), 1.9 + 0.5 * np.sin(0.19 * t),
                    2.5 + 2.0 * np.sin(0.29 * t + 2.0)], axis=-1)
    return np.stack([look_at(e, g) for e, g in zip(eye, tgt)])


def dirs_C(H, W, fx, fy, cx, cy):
    c, r = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    return np.stack(((c - cx) / fx, (r - cy) / fy, np.ones_like(c)), -1)


def raycast(o, d):
    """z-depth parameter t of the first hit of rays o + t d (d has camera z == 1)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        tw = np.where(d > 0, (ROOM_HI - o) / d, np.where(d < 0, (ROOM_LO - o) / d, np.inf))
        t = tw.min(-1)
        a = (d * d).sum(-1)
        for c, r in SPHERES:
            oc = o - c
            b = 2 * (d * oc).sum(-1)
            c0 = (oc * oc).sum(-1) - r * r
            disc = b * b - 4 * a * c0
            ts = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
            t = np.where((disc > 0) & (ts > 1e-6), np.minimum(t, ts), t)
        for lo, hi in BOXES:
            t1, t2 = (lo - o) / d, (hi - o) / d
            tn, tf = np.nanmax(np.minimum(t1, t2), -1), np.nanmin(np.maximum(t1, t2), -1)
            t = np.where((tn <= tf) & (tn > 1e-6), np.minimum(t, tn), t)
    return t


def render_depth(T_WC, cam, rng=None, invalid_frac=0.02, noise_std=0.0, max_depth=12.0):
    H, W = cam["H"], cam["W"]
    dc = dirs_C(H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    dw = dc @ T_WC[:3, :3].astype(np.float64).T
    depth = raycast(T_WC[:3, 3].astype(np.float64), dw)
    if rng is not None and noise_std > 0:
        depth = depth + noise_std * rng.standard_normal(depth.shape)
    depth = depth.astype(np.float32)
    depth[depth > max_depth] = 0.0
    if rng is not None and invalid_frac > 0:
        depth[rng.uniform(size=depth.shape) < invalid_frac] = 0.0
    return depth


def estimate_normals(depth, cam):
    """Camera-frame normals as the reference ingests them: point cloud from depth
    (transform.py:169-196; 0-depth pixels stay 0, i.e. the point at the camera
    origin, as in the reference) then, per pixel, the best of 8 neighbour pairs at
    stride 2 by summed edge length, cross product, normalise (transform.py:215-270)."""
    H, W = depth.shape
    d = 2
    c, r = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    z = depth
    pts = np.stack((z * (c - np.float32(cam["cx"])) / np.float32(cam["fx"]),
                    z * (r - np.float32(cam["cy"])) / np.float32(cam["fy"]), z), -1)
    P = np.full((H + 2 * d, W + 2 * d, 3), np.nan, np.float32)
    P[d:-d, d:-d] = pts
    look = [(-d, 0), (-d, d), (0, d), (d, d), (d, 0), (d, -d), (0, -d), (-d, -d)]

    def shifted(k):
        dy, dx = look[k]
        return P[d + dy:d + dy + H, d + dx:d + dx + W]
    p1 = pts
    best = np.full((H, W), np.inf, np.float32)
    nrm = np.full((H, W, 3), np.nan, np.float32)
    for k in range(8):
        p2, p3 = shifted(k), shifted((k + 2) % 8)
        diff = np.linalg.norm(p2 - p1, axis=-1) + np.linalg.norm(p3 - p1, axis=-1)
        diff = np.where(np.isnan(diff), np.inf, diff)
        take = diff < best                     # argmin keeps the first minimum
        n = np.cross(p2 - p1, p3 - p1)
        nrm = np.where(take[..., None], n, nrm)
        best = np.where(take, diff, best)
    with np.errstate(invalid="ignore", divide="ignore"):
        nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
    return nrm.astype(np.float32)


def keyframes(n, cam, seed=1, stride=40, noise_std=0.0):
    """n keyframes (depth [n,H,W], normals [n,H,W,3], T_WC [n,4,4]) `stride` frames apart."""
    rng = np.random.RandomState(seed)
    T = trajectory(n * stride)[::stride][:n]
    depth = np.stack([render_depth(T[i], cam, rng, noise_std=noise_std) for i in range(n)])
    normal = np.stack([estimate_normals(depth[i], cam) for i in range(n)])
    return depth, normal, T


REPLICA_CAM = dict(H=680, W=1200, fx=600.0, fy=600.0, cx=599.5, cy=339.5)   # replicaCAD.json:10-17
SCANNET_CAM = dict(H=480, W=640, fx=577.87, fy=577.87, cx=319.5, cy=239.5)
