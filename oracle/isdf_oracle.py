"""CPU oracle for the iSDF training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the per-step algorithm of the reference
(`isdf/modules/trainer.py:951-1016` and the functions it calls), written from
the reference's behaviour, with the second-order backward derived by hand
(the reference gets it from autograd, `fc_map.py:12-22` + `trainer.py:981`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product path (`isdf_amd/`) never does.

PINNING: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4 / 8c).  The oracle is pinned instead against outputs of the
reference itself, generated in the build container by importing
`/root/reference/isdf` (script: `tests/golden/make_golden.py`, fixtures:
`tests/golden/*.npz`); `tests/test_oracle_golden.py` checks every function below
against those fixtures.  Third-party arithmetic on the path (torch Softplus,
CosineSimilarity, AdamW; torch==2.10.0, not under /root/reference) is restated
from its documented behaviour and pinned through the same fixtures.

All functions take/return numpy arrays; `dtype` selects float32 (default,
mirrors the reference) or float64 (tight checks of the hand derivation).
"""
import numpy as np

# ----------------------------------------------------------------------------
# Network description
# ----------------------------------------------------------------------------

# 21 icosahedron directions, `isdf/modules/embedding.py:40-62` (3 x 21 after .T)
ICO_DIRS = np.array([
    0.8506508, 0, 0.5257311,
    0.809017, 0.5, 0.309017,
    0.5257311, 0.8506508, 0,
    1, 0, 0,
    0.809017, 0.5, -0.309017,
    0.8506508, 0, -0.5257311,
    0.309017, 0.809017, -0.5,
    0, 0.5257311, -0.8506508,
    0.5, 0.309017, -0.809017,
    0, 1, 0,
    -0.5257311, 0.8506508, 0,
    -0.309017, 0.809017, -0.5,
    0, 0.5257311, 0.8506508,
    -0.309017, 0.809017, 0.5,
    0.309017, 0.809017, 0.5,
    0.5, 0.309017, 0.809017,
    0.5, -0.309017, 0.809017,
    0, 0, 1,
    -0.5, 0.309017, 0.809017,
    -0.809017, 0.5, 0.309017,
    -0.809017, 0.5, -0.309017,
], dtype=np.float64).reshape(-1, 3).T  # [3, 21]

SOFTPLUS_BETA = 100.0      # fc_map.py:54
SOFTPLUS_THRESHOLD = 20.0  # torch.nn.Softplus default


def layer_names(hidden_layers_block):
    """state_dict prefixes of the hidden layers in forward order
    (`fc_map.py:77-90`): in_layer, mid1.*, cat_layer, mid2.*"""
    B = hidden_layers_block
    return (["in_layer.0"] + ["mid1.%d.0" % i for i in range(B)]
            + ["cat_layer.0"] + ["mid2.%d.0" % i for i in range(B)])


def embedding_size(n_freqs):
    """`embedding.py:66`: 2 * 21 * n_freqs + 3"""
    return 2 * ICO_DIRS.shape[1] * n_freqs + 3


def init_params(hidden_size, hidden_layers_block, n_freqs, rng):
    """Random parameters with the shapes (and roughly the scale) of
    `SDFMap.__init__` (`fc_map.py:63-92`): xavier-normal weights
    (`fc_map.py:58-60`), uniform(+-1/sqrt(fan_in)) biases (torch.nn.Linear
    default).  `rng` is a numpy RandomState: deterministic across platforms, so
    fixtures need not store full-size weights."""
    E, H, B = embedding_size(n_freqs), hidden_size, hidden_layers_block
    names = layer_names(B)
    fan_in = [E] + [H] * B + [H + E] + [H] * B
    p = {}
    for n, fi in zip(names, fan_in):
        std = np.sqrt(2.0 / (fi + H))
        p[n + ".weight"] = (rng.standard_normal((H, fi)) * std).astype(np.float32)
        p[n + ".bias"] = rng.uniform(-1, 1, H).astype(np.float32) / np.float32(np.sqrt(fi))
    p["out_alpha.weight"] = (rng.standard_normal((1, H)) * np.sqrt(2.0 / (H + 1))).astype(np.float32)
    p["out_alpha.bias"] = rng.uniform(-1, 1, 1).astype(np.float32) / np.float32(np.sqrt(H))
    return p


# ----------------------------------------------------------------------------
# Geometry  (isdf/geometry/transform.py)
# ----------------------------------------------------------------------------

def ray_dirs_C(H, W, fx, fy, cx, cy, dtype=np.float32):
    """`transform.py:13-33`, depth_type 'z': ((c-cx)/fx, (r-cy)/fy, 1) -> [H,W,3]"""
    c = np.arange(W, dtype=dtype)[None, :].repeat(H, 0)
    r = np.arange(H, dtype=dtype)[:, None].repeat(W, 1)
    x = (c - dtype(cx)) / dtype(fx)
    y = (r - dtype(cy)) / dtype(fy)
    return np.stack((x, y, np.ones_like(x)), axis=2)


def origin_dirs_W(T_WC, dirs_C):
    """`transform.py:36-41`: dirs_W = R_WC dirs_C, origins = T_WC[:, :3, 3]"""
    R = T_WC[:, :3, :3]
    dirs_W = (R * dirs_C[:, None, :]).sum(-1)
    return T_WC[:, :3, 3], dirs_W


# ----------------------------------------------------------------------------
# Sampling  (isdf/modules/sample.py)
# ----------------------------------------------------------------------------

def sample_pixels_indices_b(n_rays, n_frames):
    """`sample.py:18-19`: arange(F).repeat_interleave(n_rays)"""
    return np.repeat(np.arange(n_frames, dtype=np.int64), n_rays)


def get_batch_data(depth_batch, T_WC_batch, dirs_C, indices_b, indices_h, indices_w,
                   norm_batch=None):
    """`sample.py:24-74`: gather depth/normals at the drawn pixels, keep rays
    with depth != 0 and a non-NaN normal (stable, order-preserving compaction),
    gather pose and camera-frame direction."""
    depth_sample = depth_batch[indices_b, indices_h, indices_w]
    valid = depth_sample != 0
    norm_sample = None
    if norm_batch is not None:
        norm_sample = norm_batch[indices_b, indices_h, indices_w, :]
        valid = np.logical_and(valid, ~np.isnan(norm_sample[:, 0]))
        norm_sample = norm_sample[valid]
    depth_sample = depth_sample[valid]
    indices_b, indices_h, indices_w = indices_b[valid], indices_h[valid], indices_w[valid]
    T_WC_sample = T_WC_batch[indices_b]
    dirs_C_sample = dirs_C[indices_h, indices_w, :]
    return dict(dirs_C_sample=dirs_C_sample, depth_sample=depth_sample, norm_sample=norm_sample,
                T_WC_sample=T_WC_sample, indices_b=indices_b, indices_h=indices_h,
                indices_w=indices_w, valid=valid)


def stratified_sample(min_depth, max_depth, n_bins, U):
    """`sample.py:77-128`, tensor max_depth / scalar min_depth branch.
    U: the `torch.rand(n_rays, n_bins)` draw."""
    dt = U.dtype.type
    sample_range = (max_depth - dt(min_depth))[:, None]
    lin = np.linspace(0, 1, n_bins + 1, dtype=U.dtype)[None, :]
    bin_limits = lin.repeat(U.shape[0], 0) * sample_range + dt(min_depth)
    bin_length = sample_range / dt(n_bins)
    return bin_limits[:, :-1] + U * bin_length


def sample_along_rays(T_WC_sample, min_depth, max_depth, n_strat, n_surf, dirs_C_sample,
                      gt_depth, U, N_off):
    """`sample.py:131-178`.  U: rand(R, n_strat) draw; N_off: the
    `torch.normal(zeros(R, n_surf-1), 0.1)` draw (already scaled by 0.1).
    Column order: [surface, n_surf-1 near-surface, n_strat stratified]."""
    origins, dirs_W = origin_dirs_W(T_WC_sample, dirs_C_sample)
    z_vals = stratified_sample(min_depth, max_depth, n_strat, U)
    if n_surf > 0:
        near = gt_depth[:, None] + N_off
        near = np.clip(near, U.dtype.type(min_depth), max_depth[:, None])
        z_vals = np.concatenate((gt_depth[:, None], near, z_vals), axis=1)
    pc = origins[:, None, :] + dirs_W[:, None, :] * z_vals[:, :, None]
    return pc, z_vals


# ----------------------------------------------------------------------------
# Positional encoding  (isdf/modules/embedding.py)
# ----------------------------------------------------------------------------

def _pe_core(x, transform, scale, n_freqs):
    """`embedding.py:95-111` + `scale_input :12-22` + `transform_3D_grid`
    (`transform.py:287-304`).  x: [P,3].  Returns x' [P,3], xb [P,21*n_freqs]."""
    dt = x.dtype.type
    xs = x
    if transform is not None:
        T = transform.astype(x.dtype)
        xs = x @ T[:3, :3].T + T[:3, 3]
    xs = xs * dt(scale)
    proj = xs @ ICO_DIRS.astype(x.dtype)               # [P,21]
    freq = (2.0 ** np.arange(n_freqs)).astype(x.dtype)  # 2**linspace(0, n-1, n)
    xb = (proj[:, :, None] * freq).reshape(x.shape[0], -1)  # index d*n_freqs+f
    return xs, xb


def positional_encoding(x, transform, scale, n_freqs):
    xs, xb = _pe_core(x, transform, scale, n_freqs)
    half_pi = x.dtype.type(0.5 * np.pi)
    return np.concatenate([xs, np.sin(xb), np.sin(xb + half_pi)], axis=-1)


def pe_vjp(x, Eg, transform, scale, n_freqs):
    """g = (d emb / d x)^T Eg  -> [P,3]   (what autograd does for `fc_map.gradient`)."""
    dt = x.dtype.type
    xs, xb = _pe_core(x, transform, scale, n_freqs)
    nD = ICO_DIRS.shape[1]
    freq = (2.0 ** np.arange(n_freqs)).astype(x.dtype)
    half_pi = dt(0.5 * np.pi)
    n = nD * n_freqs
    w = (Eg[:, 3:3 + n] * np.cos(xb) + Eg[:, 3 + n:3 + 2 * n] * np.cos(xb + half_pi))
    w = (w.reshape(-1, nD, n_freqs) * freq).sum(-1)          # [P,21]
    gxs = Eg[:, :3] + w @ ICO_DIRS.astype(x.dtype).T         # d/d x'
    gxs = gxs * dt(scale)
    if transform is not None:
        gxs = gxs @ transform[:3, :3].astype(x.dtype)         # R^T applied to column vec
    return gxs


def pe_jvp(x, v, transform, scale, n_freqs):
    """(d emb / d x) v  -> [P,E]   (tangent of the embedding along v [P,3])."""
    dt = x.dtype.type
    xs, xb = _pe_core(x, transform, scale, n_freqs)
    vs = v
    if transform is not None:
        vs = v @ transform[:3, :3].astype(x.dtype).T
    vs = vs * dt(scale)
    nD = ICO_DIRS.shape[1]
    freq = (2.0 ** np.arange(n_freqs)).astype(x.dtype)
    half_pi = dt(0.5 * np.pi)
    pv = ((vs @ ICO_DIRS.astype(x.dtype))[:, :, None] * freq).reshape(x.shape[0], -1)
    return np.concatenate([vs, np.cos(xb) * pv, np.cos(xb + half_pi) * pv], axis=-1)


# ----------------------------------------------------------------------------
# Softplus(beta=100, threshold=20) and derivatives (torch semantics)
# ----------------------------------------------------------------------------

def softplus(z):
    dt = z.dtype.type
    bz = z * dt(SOFTPLUS_BETA)
    with np.errstate(over="ignore"):
        soft = np.log1p(np.exp(np.minimum(bz, dt(SOFTPLUS_THRESHOLD)))) / dt(SOFTPLUS_BETA)
    return np.where(bz > dt(SOFTPLUS_THRESHOLD), z, soft)


def softplus_d1(z):
    """softplus_backward: beta*z > threshold ? 1 : e/(e+1), e = exp(beta z)"""
    dt = z.dtype.type
    bz = z * dt(SOFTPLUS_BETA)
    e = np.exp(np.minimum(bz, dt(SOFTPLUS_THRESHOLD)))
    return np.where(bz > dt(SOFTPLUS_THRESHOLD), dt(1), e / (e + dt(1)))


def softplus_d2(z):
    """softplus_double_backward: beta * s(1-s) * [beta z < threshold]"""
    dt = z.dtype.type
    bz = z * dt(SOFTPLUS_BETA)
    with np.errstate(over="ignore"):
        s = dt(1) / (dt(1) + np.exp(-bz))
    return np.where(bz < dt(SOFTPLUS_THRESHOLD), dt(SOFTPLUS_BETA) * s * (dt(1) - s), dt(0))


# ----------------------------------------------------------------------------
# SDF MLP forward + input gradient  (isdf/modules/fc_map.py)
# ----------------------------------------------------------------------------

class NetCfg:
    def __init__(self, hidden_size=256, hidden_layers_block=2, n_freqs=6,
                 scale_input=0.05937489, scale_output=0.14, transform=None):
        self.H, self.B, self.n_freqs = hidden_size, hidden_layers_block, n_freqs
        self.scale_input, self.scale_output, self.transform = scale_input, scale_output, transform
        self.E = embedding_size(n_freqs)
        self.names = layer_names(hidden_layers_block)
        self.cat = hidden_layers_block + 1  # index of cat_layer in names


def _forward_cache(params, cfg, x):
    dt = x.dtype
    emb = positional_encoding(x, cfg.transform, cfg.scale_input, cfg.n_freqs)
    A, Z, I = [], [], []
    a = emb
    for li, n in enumerate(cfg.names):
        inp = np.concatenate([a, emb], axis=-1) if li == cfg.cat else a
        z = inp @ params[n + ".weight"].astype(dt).T + params[n + ".bias"].astype(dt)
        a = softplus(z)
        I.append(inp); Z.append(z); A.append(a)
    raw = a @ params["out_alpha.weight"].astype(dt).T + params["out_alpha.bias"].astype(dt)
    return emb, I, Z, A, raw[:, 0]


def sdf_forward(params, cfg, x, noise=None):
    """`SDFMap.forward` (`fc_map.py:94-111`).  x [P,3]; noise: the
    `torch.randn(raw.shape) * noise_std` term (already scaled) or None."""
    raw = _forward_cache(params, cfg, x)[4]
    if noise is not None:
        raw = raw + noise
    return raw * x.dtype.type(cfg.scale_output)


def _input_grad_cache(params, cfg, x, Z):
    """Reverse sweep for d sdf / d x (`fc_map.gradient`, `fc_map.py:12-22`).
    Returns per-layer Q (d sdf/d a_l), Pm (d sdf/d z_l), S1, and Eg (d sdf/d emb)."""
    dt = x.dtype
    L = len(cfg.names)
    q = np.broadcast_to(params["out_alpha.weight"].astype(dt) * dt.type(cfg.scale_output),
                        (x.shape[0], cfg.H)).copy()
    Q, Pm, S1 = [None] * L, [None] * L, [None] * L
    Eg = np.zeros((x.shape[0], cfg.E), dt)
    for li in range(L - 1, -1, -1):
        s1 = softplus_d1(Z[li])
        p = q * s1
        Q[li], Pm[li], S1[li] = q, p, s1
        g = p @ params[cfg.names[li] + ".weight"].astype(dt)
        if li == cfg.cat:
            q = g[:, :cfg.H]
            Eg = Eg + g[:, cfg.H:]
        elif li == 0:
            Eg = Eg + g
        else:
            q = g
    return Q, Pm, S1, Eg


def sdf_forward_grad(params, cfg, x, noise=None):
    """(sdf [P], sdf_grad [P,3]) as `trainer.py:789-793` computes them."""
    emb, I, Z, A, raw = _forward_cache(params, cfg, x)
    _, _, _, Eg = _input_grad_cache(params, cfg, x, Z)
    g = pe_vjp(x, Eg, cfg.transform, cfg.scale_input, cfg.n_freqs)
    if noise is not None:
        raw = raw + noise
    return raw * x.dtype.type(cfg.scale_output), g


# ----------------------------------------------------------------------------
# Loss  (isdf/modules/loss.py, trainer.py:795-840)
# ----------------------------------------------------------------------------

class LossCfg:
    def __init__(self, bounds_method="ray", loss_type="L1", trunc_weight=5.38344020,
                 trunc_distance=0.29365022, eik_weight=0.268, eik_apply_dist=0.1,
                 grad_weight=0.018, orien_loss=False):
        self.bounds_method, self.loss_type = bounds_method, loss_type
        self.trunc_weight, self.trunc_distance = trunc_weight, trunc_distance
        self.eik_weight, self.eik_apply_dist = eik_weight, eik_apply_dist
        self.grad_weight, self.orien_loss = grad_weight, orien_loss


def bounds_ray(depth_sample, z_vals, dirs_C_sample, T_WC_sample):
    """`loss.py:13-22,48-53`: bounds = |dirs_C| (depth - z); grad_vec = -dirs_W
    repeated over the S-1 non-surface samples."""
    b = (depth_sample[:, None] - z_vals) * np.linalg.norm(dirs_C_sample, axis=-1)[:, None]
    _, dirs_W = origin_dirs_W(T_WC_sample, dirs_C_sample)
    grad = -np.repeat(dirs_W[:, None, :], z_vals.shape[1] - 1, axis=1)
    return b, grad


def bounds_pc(pc, z_vals, depth_sample):
    """`loss.py:56-89`: distance to the closest surface point of the batch,
    negative behind the surface; grad_vec = unit vector from that point."""
    surf = pc[:, 0]
    R, S = z_vals.shape
    dists = np.empty((R, S), pc.dtype)
    closest = np.empty((R, S), np.int64)
    for r in range(R):  # row-blocked to bound memory
        d = np.linalg.norm(pc[r][:, None, :] - surf[None, :, :], axis=-1)
        closest[r] = d.argmin(-1)
        dists[r] = d.min(-1)
    behind = z_vals > depth_sample[:, None]
    b = np.where(behind, -dists, dists)
    diff = pc - surf[closest]
    g = diff[:, 1:]
    with np.errstate(invalid="ignore", divide="ignore"):
        g = g / np.linalg.norm(g, axis=-1)[..., None]
    g = np.where(behind[:, 1:, None], -g, g)
    return b, g


def _cos_sim(x, y, eps=1e-6):
    """torch.nn.CosineSimilarity(dim=-1, eps): sum(x/max(|x|,eps) * y/max(|y|,eps))"""
    dt = x.dtype.type
    xn = np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), dt(eps))
    yn = np.maximum(np.linalg.norm(y, axis=-1, keepdims=True), dt(eps))
    return ((x / xn) * (y / yn)).sum(-1)


def loss_terms(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc):
    """Forward value of every loss matrix, `trainer.py:803-836` + `loss.py:122-205`.
    sdf/bounds [R,S]; sdf_grad [R,S,3]; grad_vec [R,S-1,3]; norm_sample [R,3].
    Returns dict of matrices and scalar means (what `losses` logs)."""
    dt = sdf.dtype.type
    free = bounds > dt(lc.trunc_distance)
    fs = np.maximum(np.maximum(sdf - bounds, dt(0)), np.exp(dt(-5.0) * sdf) - dt(1))
    tr = sdf - bounds
    mat = np.where(free, fs, tr)
    sdf_loss = np.abs(mat) if lc.loss_type == "L1" else np.square(mat)
    sdf_loss = np.where(free, sdf_loss, sdf_loss * dt(lc.trunc_weight))
    out = {"free": free, "sdf_loss_mat": sdf_loss, "sdf_loss": sdf_loss.mean()}
    tot = sdf_loss
    if lc.grad_weight != 0:
        gv = grad_vec.copy()
        nanrow = np.isnan(gv[..., 0])
        rr = np.where(nanrow)[0]
        gv[nanrow] = norm_sample[rr]
        surf = dt(1) - _cos_sim(sdf_grad[:, 0], norm_sample)
        rest = dt(1) - _cos_sim(gv, sdf_grad[:, 1:])
        gl = np.concatenate((surf[:, None], rest), axis=1)
        if lc.orien_loss:
            gl = (gl > 1).astype(sdf.dtype)
        out["grad_loss_mat"], out["grad_loss"] = gl, gl.mean()
        tot = tot + dt(lc.grad_weight) * gl
    if lc.eik_weight != 0:
        eik = np.abs(np.linalg.norm(sdf_grad, axis=-1) - dt(1))
        eik = np.where(bounds < dt(lc.eik_apply_dist), dt(0), eik) * dt(lc.eik_weight)
        out["eik_loss_mat"], out["eikonal_loss"] = eik, eik.mean()
        tot = tot + eik
    out["tot_loss_mat"], out["total_loss"] = tot, tot.mean()
    return out


def loss_adjoints(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc):
    """d total_loss / d sdf  [R,S]  and  d total_loss / d sdf_grad  [R,S,3]
    (hand derivative of `loss_terms`; includes the 1/(R*S) of the mean)."""
    dt = sdf.dtype.type
    N = dt(sdf.size)
    free = bounds > dt(lc.trunc_distance)
    m1 = np.maximum(sdf - bounds, dt(0))
    m2 = np.exp(dt(-5.0) * sdf) - dt(1)
    v_fs = np.maximum(m1, m2)
    dv_fs = np.where(m1 >= m2, (sdf > bounds).astype(sdf.dtype), dt(-5.0) * np.exp(dt(-5.0) * sdf))
    v = np.where(free, v_fs, sdf - bounds)
    dv = np.where(free, dv_fs, dt(1))
    if lc.loss_type == "L1":
        ds = np.sign(v) * dv
    else:
        ds = dt(2) * v * dv
    ds = np.where(free, ds, ds * dt(lc.trunc_weight)) / N
    dg = np.zeros_like(sdf_grad)
    if sdf_grad is not None:
        gn = np.linalg.norm(sdf_grad, axis=-1, keepdims=True)
        with np.errstate(invalid="ignore", divide="ignore"):
            n = np.where(gn > 0, sdf_grad / gn, dt(0))
        if lc.eik_weight != 0:
            mask = (bounds >= dt(lc.eik_apply_dist)).astype(sdf.dtype)[..., None]
            dg = dg + dt(lc.eik_weight) * mask * np.sign(gn - dt(1)) * n
        if lc.grad_weight != 0 and not lc.orien_loss:
            gv = grad_vec.copy()
            nanrow = np.isnan(gv[..., 0])
            gv[nanrow] = norm_sample[np.where(nanrow)[0]]
            y = np.concatenate((norm_sample[:, None, :], gv), axis=1)       # target dirs [R,S,3]
            yh = y / np.maximum(np.linalg.norm(y, axis=-1, keepdims=True), dt(1e-6))
            gc = np.maximum(gn, dt(1e-6))
            xh = sdf_grad / gc
            cos = (xh * yh).sum(-1, keepdims=True)
            # d/dx [ x/max(|x|,eps) . yh ]  (for |x| > eps: (yh - cos * x/|x|)/|x|)
            dcos = np.where(gn > dt(1e-6), (yh - cos * n) / gc, yh / gc)
            dg = dg - dt(lc.grad_weight) * dcos
        dg = dg / N
    return ds, dg


def frame_avg(tot_loss_mat, indices_b, indices_h, indices_w, n_frames, H, W, factor=8):
    """`loss.py:208-240`: scatter per-ray loss sums into a zero [F,H,W] image
    (duplicate pixels: last ray wins, counted once), 8x8 block means over the
    sampled pixels, frame mean of the 64 block values."""
    full = np.zeros((n_frames, H, W), tot_loss_mat.dtype)
    mask = np.zeros((n_frames, H, W), tot_loss_mat.dtype)
    full[indices_b, indices_h, indices_w] = tot_loss_mat.sum(-1)  # numpy: last write wins
    mask[indices_b, indices_h, indices_w] = 1
    hb, wb = H // factor, W // factor
    la = full.reshape(-1, factor, hb, factor, wb).sum(axis=(2, 4))
    ac = mask.reshape(-1, factor, hb, factor, wb).sum(axis=(2, 4))
    ac[ac == 0] = 1.0
    la = la / ac
    return la, la.sum(axis=(1, 2)) / (factor * factor)


# ----------------------------------------------------------------------------
# Full step: loss + parameter gradients (hand-derived double backward), AdamW
# ----------------------------------------------------------------------------

def loss_and_grads(params, cfg, lc, pc, z_vals, depth_sample, dirs_C_sample, T_WC_sample,
                   norm_sample, noise=None, want_intermediates=False, adjoints_from=None):
    """`Trainer.sdf_eval_and_loss` (`trainer.py:768-868`) + `total_loss.backward()`
    (`trainer.py:981`).  pc [R,S,3].  Returns (terms dict, grads dict) where
    grads has the 14 state_dict keys.

    Derivation (row-vector convention, per point; l = hidden layer index):
      forward     z_l = I_l W_l^T + b_l, a_l = sp(z_l), raw = a_L w_out^T + b_out
      1st reverse q_L = so*w_out; p_l = q_l*sp'(z_l); G_l = p_l W_l  (-> q_{l-1}, Eg)
                  g = J_pe^T Eg
      adjoint of the 1st reverse runs upward like a JVP along gbar:
                  Gb_1 = J_pe gbar;  u_l = Gb_l W_l^T;  dW_l += p_l^T Gb_l
                  qb_l = u_l*sp'(z_l);  zb_l(inj) = u_l*q_l*sp''(z_l)
      ordinary reverse with the injected term:
                  ab_L = sbar*so*w_out; zb_l = ab_l*sp'(z_l) + zb_l(inj)
                  dW_l += zb_l^T I_l; db_l = sum zb_l; ab_{l-1} = (zb_l W_l)[:, :H]

    adjoints_from = (sdf, sdf_grad): evaluate the loss adjoints (sbar, gbar) at THESE outputs instead of this
    function's own (the backward pass is linear in the adjoints; the loss is not smooth -- L1 / eikonal signs,
    free-space branch -- so a checker that wants to judge a 16-bit implementation's BACKWARD arithmetic feeds it the
    adjoints that implementation actually used; tests/test_gpu_parity.py).
    """
    R, S = z_vals.shape
    dt = pc.dtype
    x = pc.reshape(-1, 3)
    L = len(cfg.names)
    so = dt.type(cfg.scale_output)
    do_grad = lc.eik_weight != 0 or lc.grad_weight != 0

    emb, I, Z, A, raw = _forward_cache(params, cfg, x)
    if noise is not None:
        raw = raw + noise.reshape(-1)
    sdf = (raw * so).reshape(R, S)
    sdf_grad = None
    if do_grad:
        Q, Pm, S1, Eg = _input_grad_cache(params, cfg, x, Z)
        sdf_grad = pe_vjp(x, Eg, cfg.transform, cfg.scale_input, cfg.n_freqs).reshape(R, S, 3)
    else:
        S1 = [softplus_d1(z) for z in Z]

    if lc.bounds_method == "ray":
        bounds, grad_vec = bounds_ray(depth_sample, z_vals, dirs_C_sample, T_WC_sample)
    elif lc.bounds_method == "pc":
        bounds, grad_vec = bounds_pc(pc, z_vals, depth_sample)
    else:
        raise ValueError("bounds_method 'normal' is unusable in the reference (loss.py:29)")

    terms = loss_terms(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc)
    terms.update(sdf=sdf, sdf_grad=sdf_grad, bounds=bounds, grad_vec=grad_vec)
    if adjoints_from is None:
        adj_sdf, adj_grad = sdf, sdf_grad
    else:
        adj_sdf = np.asarray(adjoints_from[0], dt).reshape(R, S)
        adj_grad = None if adjoints_from[1] is None else np.asarray(adjoints_from[1], dt).reshape(R, S, 3)
    sbar, gbar = loss_adjoints(adj_sdf, adj_grad if do_grad else np.zeros((R, S, 3), dt), bounds,
                               grad_vec, norm_sample, lc)
    sbar = sbar.reshape(-1)

    grads = {}
    w_out = params["out_alpha.weight"].astype(dt)
    d_wout = (sbar * so) @ A[-1]
    inj = [None] * L
    if do_grad:
        Eb = pe_jvp(x, gbar.reshape(-1, 3), cfg.transform, cfg.scale_input, cfg.n_freqs)
        qb = None
        for li, n in enumerate(cfg.names):
            W = params[n + ".weight"].astype(dt)
            Gb = Eb if li == 0 else (np.concatenate([qb, Eb], -1) if li == cfg.cat else qb)
            u = Gb @ W.T
            grads[n + ".weight"] = Pm[li].T @ Gb
            qb = u * S1[li]
            inj[li] = u * Q[li] * softplus_d2(Z[li])
            if want_intermediates:
                terms.setdefault("Gb", []).append(Gb)
        d_wout = d_wout + so * qb.sum(0)
    ab = (sbar * so)[:, None] * w_out
    for li in range(L - 1, -1, -1):
        n = cfg.names[li]
        W = params[n + ".weight"].astype(dt)
        zb = ab * S1[li]
        if inj[li] is not None:
            zb = zb + inj[li]
        gW = zb.T @ I[li]
        grads[n + ".weight"] = grads[n + ".weight"] + gW if n + ".weight" in grads else gW
        grads[n + ".bias"] = zb.sum(0)
        if want_intermediates:
            terms.setdefault("Zb", [None] * L)[li] = zb
        if li > 0:
            ab = (zb @ W)[:, :cfg.H]
    grads["out_alpha.weight"] = d_wout[None, :]
    grads["out_alpha.bias"] = np.array([(sbar * so).sum()], dt)
    if want_intermediates:
        terms.update(A=A, Pm=Pm if do_grad else None, emb=emb, sbar=sbar, gbar=gbar)
    return terms, grads


def adamw_step(params, grads, state, lr=0.0013, weight_decay=0.012, betas=(0.9, 0.999),
               eps=1e-8):
    """torch.optim.AdamW.step (single-tensor path), constructed at
    `trainer.py:435-439`.  state: {'step': int, 'exp_avg': {}, 'exp_avg_sq': {}}.
    Updates params/state in place (float32 arithmetic like torch)."""
    state["step"] += 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k, p in params.items():
        g = grads[k].astype(np.float32)
        m = state["exp_avg"].setdefault(k, np.zeros_like(p))
        v = state["exp_avg_sq"].setdefault(k, np.zeros_like(p))
        p *= np.float32(1.0 - lr * weight_decay)
        m *= np.float32(b1); m += np.float32(1.0 - b1) * g
        v *= np.float32(b2); v += np.float32(1.0 - b2) * g * g
        denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
        p -= np.float32(lr / bc1) * (m / denom)
    return params


def new_adam_state():
    return {"step": 0, "exp_avg": {}, "exp_avg_sq": {}}


def train_step(params, state, cfg, lc, frames, cam, sample_cfg, draws, optim_cfg=None):
    """One `Trainer.step` (`trainer.py:951-1016`) on a fixed keyframe window with
    injected random draws.

    frames: dict(depth_batch [F,H,W], T_WC_batch [F,4,4], normal_batch [F,H,W,3])
    cam:    dict(H,W,fx,fy,cx,cy)
    sample_cfg: dict(n_rays,n_strat,n_surf,min_depth,dist_behind_surf)
    draws:  dict(indices_h [F*n], indices_w [F*n], U [>=R,n_strat], N_off [>=R,n_surf-1],
                 noise [>=R,S] or None)  -- rows beyond the valid-ray count R are ignored
    Returns dict with sample tensors, loss terms, grads; params/state updated in place.
    """
    F = frames["depth_batch"].shape[0]
    dirs_C = ray_dirs_C(cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    ib = sample_pixels_indices_b(sample_cfg["n_rays"], F)
    bd = get_batch_data(frames["depth_batch"], frames["T_WC_batch"], dirs_C, ib,
                        draws["indices_h"], draws["indices_w"], frames.get("normal_batch"))
    R = bd["depth_sample"].shape[0]
    max_depth = bd["depth_sample"] + np.float32(sample_cfg["dist_behind_surf"])
    pc, z_vals = sample_along_rays(bd["T_WC_sample"], sample_cfg["min_depth"], max_depth,
                                   sample_cfg["n_strat"], sample_cfg["n_surf"],
                                   bd["dirs_C_sample"], bd["depth_sample"],
                                   draws["U"][:R], draws["N_off"][:R])
    noise = None if draws.get("noise") is None else draws["noise"][:R]
    terms, grads = loss_and_grads(params, cfg, lc, pc, z_vals, bd["depth_sample"],
                                  bd["dirs_C_sample"], bd["T_WC_sample"], bd["norm_sample"],
                                  noise=noise)
    la, fa = frame_avg(terms["tot_loss_mat"], bd["indices_b"], bd["indices_h"], bd["indices_w"],
                       F, cam["H"], cam["W"])
    adamw_step(params, grads, state, **(optim_cfg or {}))
    out = dict(bd)
    out.update(terms)
    out.update(pc=pc, z_vals=z_vals, grads=grads, loss_approx=la, frame_avg_loss=fa)
    return out


# ----------------------------------------------------------------------------
# Per-frame ingest and keyframe test ("next" tier)
# ----------------------------------------------------------------------------

def pointcloud_from_depth(depth, fx, fy, cx, cy):
    """`transform.pointcloud_from_depth_torch` (`transform.py:169-196`), 'z' depth:
    (z (c-cx)/fx, z (r-cy)/fy, z); NaN depth stays NaN, 0 depth gives the origin."""
    H, W = depth.shape
    c = np.arange(W, dtype=np.float32)[None, :]
    r = np.arange(H, dtype=np.float32)[:, None]
    z = depth.astype(np.float32)
    x = z * (c - np.float32(cx)) / np.float32(fx)
    y = z * (r - np.float32(cy)) / np.float32(fy)
    return np.stack((x, y, z + 0 * x), axis=-1).astype(np.float32)


def estimate_pointcloud_normals(points):
    """`transform.estimate_pointcloud_normals` (`transform.py:215-270`): for each
    pixel the neighbour pair (k, k+2 mod 8) at stride 2 with the smallest summed
    edge length (NaN -> inf, first minimum wins), cross product, normalise."""
    H, W = points.shape[:2]
    d = 2
    P = np.full((H + 2 * d, W + 2 * d, 3), np.nan, np.float32)
    P[d:-d, d:-d] = points
    look = [(-d, 0), (-d, d), (0, d), (d, d), (d, 0), (d, -d), (0, -d), (-d, -d)]
    sh = lambda k: P[d + look[k][0]:d + look[k][0] + H, d + look[k][1]:d + look[k][1] + W]
    p1 = points
    best = None
    nrm = None
    for k in range(8):
        p2, p3 = sh(k), sh((k + 2) % 8)
        diff = np.linalg.norm(p2 - p1, axis=-1) + np.linalg.norm(p3 - p1, axis=-1)
        diff = np.where(np.isnan(diff), np.float32(np.inf), diff)
        n = np.cross(p2 - p1, p3 - p1)
        if best is None:
            best, nrm = diff, n
        else:
            take = diff < best
            nrm = np.where(take[..., None], n, nrm)
            best = np.where(take, diff, best)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)


def sdf_render_depth(z_vals, sdf):
    """`render.sdf_render_depth` (`render.py:12-35`) on z-sorted samples: depth =
    z + sdf at the FIRST sample with sdf < 0 (sample 0 if none is negative), and 0
    where that sample is the last one."""
    n = sdf.shape[1]
    inside = sdf < 0
    mul = inside * np.arange(n, 0, -1)[None, :]
    ix = mul.argmax(axis=1)
    ar = np.arange(z_vals.shape[0])
    depths = z_vals[ar, ix] + sdf[ar, ix]
    depths[ix == n - 1] = 0.0
    return depths


def keyframe_ratio(z_vals, sdf, depth_sample, kf_dist_th):
    """`Trainer.is_keyframe` (`trainer.py:597-609`): sort by z, render, fraction of rays whose
    relative depth error is below kf_dist_th."""
    order = np.argsort(z_vals, axis=-1, kind="stable")
    zs = np.take_along_axis(z_vals, order, -1)
    ss = np.take_along_axis(sdf, order, -1)
    view = sdf_render_depth(zs, ss)
    loss = np.abs(view - depth_sample) / depth_sample
    return float((loss < kf_dist_th).sum() / loss.shape[0]), view
