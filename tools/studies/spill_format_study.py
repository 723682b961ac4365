"""Round-6 study (dev tooling, imports the test oracle): which of the chain kernel's SPILLED tensors tolerate fewer bits.

The chain kernel parks four tensor families per hidden layer in HBM for the dW kernel: the layer input I_l (= A_l),
P_l = q_l s'_l, GB_l (the adjoint entering the layer in the upward sweep) and ZB_l (d loss / d z_l) -- 25 tensors x 512 B per
point in fp16, written once, read once by dW and 26 times more by the chain's own later sweeps.  VERDICT r5 item 1 asks whether
8-bit formats fit the parity bars (worst gradient tensor <= 3e-3 of the reference's autograd at BASELINE size).  This is the numpy
model of the kernel's numerics (every GEMM operand rounded to fp16, fp32 accumulation; tools/studies/operand_precision_model.py)
with ONE more knob: the format each family is rounded to ON ITS WAY TO THE dW CONTRACTION, and separately the format of what the
chain's own sweeps re-read (s' source, P / GB for the injected term).

    python tools/studies/spill_format_study.py [fixture]        (default eval_base_680x1200_ray: 25 k points)
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.isdf_oracle as orc
from tests import golden_util as gu


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def bf16(x):
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    r = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return r.astype(np.uint32).view(np.float32)


def minifloat(x, mant, emin, emax_val):
    """round-to-nearest-even to a float with `mant` explicit mantissa bits, smallest normal 2^emin (gradual underflow below),
    saturating at emax_val."""
    x = np.asarray(x, np.float32).astype(np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, emin)
    q = np.exp2(e - mant)
    r = np.round(a / q) * q          # numpy rounds half to even
    r = np.minimum(r, emax_val)
    return (np.sign(x) * r).astype(np.float32)


def e4m3(x, scale=1.0):   # OCP e4m3fn: 3 mantissa bits, min normal 2^-6, max 448
    return minifloat(np.asarray(x, np.float32) * np.float32(scale), 3, -6, 448.0) / np.float32(scale)


def e5m2(x, scale=1.0):
    return minifloat(np.asarray(x, np.float32) * np.float32(scale), 2, -14, 57344.0) / np.float32(scale)


def fp12(x):              # hypothetical 1-5-6 format (what a packed 12-bit store would hold)
    return minifloat(x, 6, -14, 65504.0)


def tile_scaled(fn, x, rows=64):
    """fn applied with one power-of-two scale per `rows` points (a chain tile): max |x| of the tile maps to 256"""
    x = np.asarray(x, np.float32)
    out = np.empty_like(x)
    for r0 in range(0, x.shape[0], rows):
        blk = x[r0:r0 + rows]
        m = np.abs(blk).max()
        s = 1.0 if m == 0 else 2.0 ** np.floor(np.log2(256.0 / m))
        out[r0:r0 + rows] = fn(blk, s)
    return out


def row_normalised(fn, sc):          # GB is linear in the point's loss adjoint gbar: one scale per POINT (row) takes its magnitude out
    def f(v):
        v = np.asarray(v, np.float32)
        m = np.abs(v).max(axis=1, keepdims=True)
        rs = np.where(m > 0, np.exp2(-np.floor(np.log2(np.maximum(m, 1e-30)))), 1.0).astype(np.float32)   # row max -> [1, 2)
        return fn(v * rs, sc) / rs
    return f


def emul(params, cfg, lc, pc, z_vals, depth_sample, dirs_C_sample, T_WC_sample, norm_sample, noise, dw=None, own=None):
    """The shipped kernel's numerics (fp16 operands everywhere, fp32 accumulate, s' re-derived from the fp16 activation, injected
    term rebuilt from GB and P) with per-family formats:
      dw[fam]  : rounding applied to family fam in {'A','P','GB','ZB'} before the dW contraction (default fp16)
      own[fam] : rounding of what the chain's own sweeps re-read: 'S1' (the s' source: None = from the fp16 activation, else a
                 function applied to s' itself, i.e. a stored s' side tensor), 'P', 'GB' (operands of the injected term)"""
    dw = dict(dict(A=f16, P=f16, GB=f16, ZB=f16, keep={}), **(dw or {}))
    own = dict(dict(S1=None, P=f16, GB=f16), **(own or {}))
    keep = dw['keep']          # {'P': {layers}, 'GB': {layers}}: layers whose tensor of that family stays fp16 (dW operand and own re-read)
    fP = lambda li, v, d: f16(v) if li in keep.get('P', ()) else d['P'](v)
    fG = lambda li, v, d: f16(v) if li in keep.get('GB', ()) else d['GB'](v)
    hp = f16
    R, S = z_vals.shape
    x = pc.reshape(-1, 3).astype(np.float32)
    L = len(cfg.names); so = np.float32(cfg.scale_output); H = cfg.H
    Wb = {n: hp(params[n + '.weight']) for n in cfg.names}
    emb = orc.positional_encoding(x, cfg.transform, cfg.scale_input, cfg.n_freqs)
    embb = hp(emb)
    A, S1x, Z = [], [], []
    a = embb
    for li, n in enumerate(cfg.names):
        inp = np.concatenate([a, embb], -1) if li == cfg.cat else a
        z = inp @ Wb[n].T + params[n + '.bias']
        af = orc.softplus(z)
        Z.append(z); S1x.append(orc.softplus_d1(z))
        a = hp(af); A.append(a)
    w_out = params['out_alpha.weight'][0]
    raw = af @ w_out + params['out_alpha.bias'][0]
    sdf = ((raw + noise.reshape(-1)) * so).reshape(R, S)

    def s1(li):
        v = np.where(Z[li] * 100 > 20, np.float32(1), np.float32(1) - np.exp(np.float32(-100) * A[li]))
        return v if own['S1'] is None else own['S1'](v)
    q = np.broadcast_to(w_out * so, (x.shape[0], H)).astype(np.float32)
    P = [None] * L; Pf = [None] * L
    Eg = np.zeros((x.shape[0], cfg.E), np.float32)
    for li in range(L - 1, -1, -1):
        s = S1x[li] if li == L - 1 else s1(li)
        p = q * s
        Pf[li] = p; P[li] = hp(p)
        g = P[li] @ Wb[cfg.names[li]]
        if li == cfg.cat: q = g[:, :H]; Eg = Eg + g[:, H:]
        elif li == 0: Eg = Eg + g
        else: q = g
    sdf_grad = orc.pe_vjp(x, Eg, cfg.transform, cfg.scale_input, cfg.n_freqs).reshape(R, S, 3)
    bounds, grad_vec = orc.bounds_ray(depth_sample, z_vals, dirs_C_sample, T_WC_sample)
    terms = orc.loss_terms(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc)
    sbar, gbar = orc.loss_adjoints(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc)
    N = np.float32(sdf.size)
    sbar = sbar.reshape(-1) * N; gbar = gbar * N
    Ebf = orc.pe_jvp(x, gbar.reshape(-1, 3), cfg.transform, cfg.scale_input, cfg.n_freqs)
    Eb = hp(Ebf)
    grads = {}
    qb = None; qbf = None
    GBf = [None] * (L + 1)     # GB[li+1] = u_li s'_li in fp32, GB[0] = Ebar
    for li, n in enumerate(cfg.names):
        Gb = Eb if li == 0 else (np.concatenate([qb, Eb], -1) if li == cfg.cat else qb)
        # (Ebar is rebuilt by the dW kernel from gbar since round 6: always the 16-bit value)
        Gb_dw = f16(Ebf) if li == 0 else (np.concatenate([fG(li, qbf, dw), f16(Ebf)], -1) if li == cfg.cat else fG(li, qbf, dw))
        u = Gb @ Wb[n].T
        grads[n + '.weight'] = fP(li, Pf[li], dw).T @ Gb_dw
        qbf = u * s1(li)
        GBf[li + 1] = qbf
        qb = hp(qbf)
    d_wout = so * qbf.sum(0)
    ab = (sbar * so)[:, None] * w_out[None, :]
    d_wout = d_wout + (sbar * so) @ orc.softplus(Z[-1])
    for li in range(L - 1, -1, -1):
        n = cfg.names[li]
        s = s1(li)
        # injected term rebuilt from the re-read GB[li+1] = u s' and P[li] = q s' (top layer: from registers, fp32)
        if li == L - 1:
            inj = np.float32(100) * GBf[li + 1] * Pf[li] * (1 - s) / np.maximum(s, 1e-30)
        else:
            inj = np.float32(100) * fG(li + 1, GBf[li + 1], own) * fP(li, Pf[li], own) * (1 - s) / np.maximum(s, 1e-30)
        inj = np.where(Z[li] * 100 < 20, inj, 0)
        zb = ab * s + inj
        zbb = hp(zb)
        Ain = (A[li - 1] if li > 0 else emb)      # fp32-exact source of the input operand for the dW rounding
        Ain_f = (np.concatenate([dw['A'](A[li - 1]), dw['A'](emb)], -1) if li == cfg.cat else (dw['A'](A[li - 1]) if li > 0 else dw['A'](emb)))
        grads[n + '.weight'] = grads[n + '.weight'] + dw['ZB'](zb).T @ Ain_f
        grads[n + '.bias'] = zb.sum(0)
        if li > 0: ab = (zbb @ Wb[n])[:, :H]
    grads['out_alpha.weight'] = d_wout[None]
    grads['out_alpha.bias'] = np.array([(sbar * so).sum()])
    for k in grads: grads[k] = grads[k] / N
    return terms, sdf, sdf_grad, grads


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'eval_base_680x1200_ray'
    g = gu.load(name)
    cfg, lc, params = gu.net_of(g), gu.loss_of(g), gu.params_of(g)
    if name.startswith('trained'):       # trained-state fixtures store the eval batch under eval/
        b = gu.trained_batch(g, 'eval/')
        noise = b['noise'].reshape(b['z_vals'].shape)
        args = (b['pc'], b['z_vals'], b['depth_sample'], b['dirs_C_sample'], b['T_WC_sample'], b['norm_sample'])
    else:
        noise = g['draw_noise'].reshape(g['z_vals'].shape) * np.float32(g['noise_std'][0])
        args = (g['pc'], g['z_vals'], g['depth_sample'], g['dirs_C_sample'], g['T_WC_sample'], g['norm_sample'])
    t0, g0 = orc.loss_and_grads(params, cfg, lc, *args, noise=noise)
    print('fixture %s: %d points; reference = fp32 oracle (pinned to the reference fixtures)' % (name, args[0].shape[0] * args[0].shape[1]))
    quick = '--quick' in sys.argv

    def report(label, **kw):
        t, sdf, sg, gr = emul(params, cfg, lc, *args, noise, **kw)
        errs = {k: gu.rel_err(gr[k], g0[k]) for k in g0}
        wk = max((k for k in errs if k.endswith('weight')), key=lambda k: errs[k])
        allv = np.concatenate([gr[k].ravel() for k in g0]); ref = np.concatenate([g0[k].ravel() for k in g0])
        print('%-72s worst dW %.2e (%s)  all-params %.2e  signed %+.1e' % (label, errs[wk], wk, gu.rel_err(allv, ref), gu.signed_projection(allv, ref)), flush=True)
    ts = lambda fn: (lambda v: tile_scaled(fn, v))
    fx = lambda fn, sc: (lambda v: fn(v, sc))
    report('shipped: fp16 everywhere')
    # the candidate designs: P and GB stored ONCE in an 8-bit format -- what dW contracts AND what the reverse sweep rebuilds the
    # injected term from
    report('P, GB e5m2, no scale (dW operands and the chain\'s own re-reads)', dw=dict(P=e5m2, GB=e5m2), own=dict(P=e5m2, GB=e5m2))
    report('P, GB e4m3, fixed scales 2^8 / 2^4 (dW and own)', dw=dict(P=fx(e4m3, 256.), GB=fx(e4m3, 16.)), own=dict(P=fx(e4m3, 256.), GB=fx(e4m3, 16.)))
    report('P, GB e4m3, per-tile scale (dW and own)', dw=dict(P=ts(e4m3), GB=ts(e4m3)), own=dict(P=ts(e4m3), GB=ts(e4m3)))
    report('P e4m3 x 2^8, GB e5m2 unscaled (dW and own)', dw=dict(P=fx(e4m3, 256.), GB=e5m2), own=dict(P=fx(e4m3, 256.), GB=e5m2))

    report('P e4m3 x 2^8, GB e4m3 per-point normalised x 2^6 (dW and own)', dw=dict(P=fx(e4m3, 256.), GB=row_normalised(e4m3, 64.)),
           own=dict(P=fx(e4m3, 256.), GB=row_normalised(e4m3, 64.)))
    if '--layers' in sys.argv:
        Lh = len(cfg.names); cat = cfg.cat
        e = fx(e4m3, 256.); gn = row_normalised(e4m3, 64.)
        def per_tensor(label, **kw):
            t, sdf, sg, gr = emul(params, cfg, lc, *args, noise, **kw)
            print('%-60s ' % label + ' '.join('%s %.1e' % (k.split('.weight')[0][-8:], gu.rel_err(gr[k], g0[k])) for k in g0 if k.endswith('weight')), flush=True)
        per_tensor('fp16')
        per_tensor('P, GB e4m3 all layers', dw=dict(P=e, GB=gn), own=dict(P=e, GB=gn))
        per_tensor('P fp16, GB e4m3', dw=dict(GB=gn), own=dict(GB=gn))
        per_tensor('P e4m3, GB fp16', dw=dict(P=e), own=dict(P=e))
        per_tensor('P e4m3 except layers 0 and cat; GB e4m3', dw=dict(P=e, GB=gn, keep=dict(P={0, cat})), own=dict(P=e, GB=gn))
        per_tensor('P e4m3 except 0, cat; GB e4m3 except 1, cat+1', dw=dict(P=e, GB=gn, keep=dict(P={0, cat}, GB={1, cat + 1})), own=dict(P=e, GB=gn))
        per_tensor('dW ONLY: P, GB e4m3 (chain re-reads fp16 / a stored 16-bit INJ)', dw=dict(P=e, GB=gn))
        per_tensor('dW ONLY: GB e4m3, P fp16', dw=dict(GB=gn))
        per_tensor('own ONLY: P, GB e4m3 (dW operands fp16)', own=dict(P=e, GB=gn))
        per_tensor('P, GB bf16', dw=dict(P=bf16, GB=bf16), own=dict(P=bf16, GB=bf16))
        per_tensor('P, GB fp12', dw=dict(P=fp12, GB=fp12), own=dict(P=fp12, GB=fp12))
        sys.exit(0)
    if not quick:
        report('dW operands all bf16', dw=dict(A=bf16, P=bf16, GB=bf16, ZB=bf16))
        report('dW operands all fp12 (1-5-6)', dw=dict(A=fp12, P=fp12, GB=fp12, ZB=fp12))
        for fam in ('A', 'P', 'GB', 'ZB'):
            report('dW %-2s in e4m3 (tile scale), rest fp16' % fam, dw={fam: ts(e4m3)})
        report('dW A in e5m2', dw=dict(A=e5m2))
        report('dW ZB in e5m2', dw=dict(ZB=e5m2))
        report('dW all four in e4m3 (tile scale)', dw={k: ts(e4m3) for k in ('A', 'P', 'GB', 'ZB')})
        report("chain's own re-reads: s' as unorm8", own=dict(S1=lambda v: np.round(v * 255) / np.float32(255)))
        report("chain's own re-reads: P, GB (injected term) in e4m3", own=dict(P=ts(e4m3), GB=ts(e4m3)))
        report("chain's own re-reads: P, GB in e5m2", own=dict(P=e5m2, GB=e5m2))
