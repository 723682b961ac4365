"""Would fp8 (e4m3) be enough for the two COMPENSATION products of the fp16x2 forward?
    W x ~= W_h x_h + [ W_l x_h + W_h x_l ]          (W_h = fp16(W), W_l = fp16(W - W_h), the same for the activation x)
The bracket is 2^-11 of the first product; today it is two more fp16 MFMA passes (a compensated K = 256 stage is 96 MFMA slots instead of
32).  v_mfma_scale_f32_32x32x64_f8f6f4 runs at twice the fp16 rate and takes a power-of-two scale per operand block, so the bracket as
ONE fp8 pass over K = 512 ([W_l 2^s | W_h] . [x_h ; x_l 2^s], scale 2^-s) would cost one fp16-equivalent pass instead of two.
Numpy model of the forward at BASELINE size against the reference fixtures (dev tooling; imports the test oracle).
Run: python tools/studies/fp8_compensation_study.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.isdf_oracle as orc
from tests import golden_util as gu


def f16(x): return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def e4m3(x):
    """round to nearest even onto OCP fp8 e4m3 (3 mantissa bits, normals 2^-6 .. 448, subnormal step 2^-9, saturating)"""
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -6)))          # exponent of the binade (subnormals share 2^-6's)
    step = 2.0 ** (e - 3)
    return (np.sign(x) * np.round(a / step) * step).astype(np.float32)   # np.round: half to even


def block_scale(x, axis, block=32):
    """power-of-two scale per block of 32 along the contraction axis, as the E8M0 scale operand of the scaled MFMA: the block's largest
    magnitude lands in e4m3's top binade"""
    x = np.moveaxis(np.asarray(x, np.float32), axis, -1)
    n = x.shape[-1]
    if n % block:                                             # (the embedding is 255 wide: the kernels pad it to 256)
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (block - n % block,), np.float32)], -1)
    shp = x.shape
    xb = x.reshape(shp[:-1] + (shp[-1] // block, block))
    m = np.abs(xb).max(-1, keepdims=True)
    s = 2.0 ** (8 - np.ceil(np.log2(np.maximum(m, 1e-30))))   # max * s in (128, 256]
    q = e4m3(xb * s) / s
    return np.moveaxis(q.reshape(shp)[..., :n], -1, axis).astype(np.float32)


def forward(params, cfg, x, mode):
    """mode: 'f16' plain fp16 operands | 'x2' today's compensated forward | 'x2_fp8' the bracket in block-scaled e4m3 |
    'x2_fp8_noscale' e4m3 with ONE scale 2^12 for every residual (no block scales)"""
    L = len(cfg.names); H = cfg.H
    emb = orc.positional_encoding(x, cfg.transform, cfg.scale_input, cfg.n_freqs)
    af = None
    for li, n in enumerate(cfg.names):
        W = params[n + '.weight'].astype(np.float32)
        Wh = f16(W); Wl = f16(W - Wh)
        if li == 0: inp = emb
        elif li == cfg.cat: inp = np.concatenate([af, emb], -1)
        else: inp = af
        xh = f16(inp); xl = f16(inp - xh)
        z = xh @ Wh.T
        if mode != 'f16' and li >= cfg.cat:
            # chain.hip: the cat layer adds W_l[:, HD:] emb only; the layers past it W_l a and W a_l
            if li == cfg.cat:
                cols = slice(H, None)
                pairs = [(Wl[:, cols], xh[:, cols])]
            else:
                pairs = [(Wl, xh), (Wh, xl)]
            for Wc, xc in pairs:
                if mode == 'x2':
                    z = z + xc @ Wc.T
                elif mode == 'x2_fp8':
                    z = z + block_scale(xc, 1) @ block_scale(Wc, 1).T
                else:
                    s = np.float32(4096.0)
                    big_w = np.abs(Wc).max() > 1e-3; big_x = np.abs(xc).max() > 1e-2      # which operand is the residual
                    wq = e4m3(Wc if big_w else Wc * s) / (1 if big_w else s)
                    xq = e4m3(xc if big_x else xc * s) / (1 if big_x else s)
                    z = z + xq @ wq.T
        z = z + params[n + '.bias']
        af = orc.softplus(z)
    raw = af @ params['out_alpha.weight'][0] + params['out_alpha.bias'][0]
    return raw * np.float32(cfg.scale_output)


if __name__ == '__main__':
    names = sys.argv[1:] or ['eval_base_680x1200_ray', 'eval_base_480x640_ray', 'eval_full_ray']
    for name in names:
        g = gu.load(name)
        cfg, params = gu.net_of(g), gu.params_of(g)
        x = g['pc'].reshape(-1, 3).astype(np.float32)
        ref = g['sdf_nonoise'].reshape(-1)
        print(name, '(%d points; sdf rel-L2 and max |d| / max |sdf| against the REFERENCE; the bar is 1e-3)' % x.shape[0])
        for mode, label in (('f16', 'plain fp16 operands'), ('x2', 'fp16x2 as shipped (bracket in fp16)'),
                            ('x2_fp8', 'bracket in e4m3, power-of-two scale per 32 (the scaled MFMA)'),
                            ('x2_fp8_noscale', 'bracket in e4m3, one scale 2^12 for every residual')):
            sdf = forward(params, cfg, x, mode)
            print('  %-66s %.2e   %.2e' % (label, gu.rel_err(sdf, ref), np.abs(sdf - ref).max() / np.abs(ref).max()))
