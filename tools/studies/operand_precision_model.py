"""Operand-precision study for DESIGN.md (dev tooling, imports the test oracle).

Models the HIP chain kernel's numerics in numpy -- every GEMM operand rounded to
a 16-bit type, fp32 accumulation, s' recomputed from the stored activation,
the injected second-order term stored in 16 bits -- and compares against the
fp32 oracle on the full-size net (fixture eval_full_ray).
Run: python tools/studies/operand_precision_model.py
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.isdf_oracle as orc
from tests import golden_util as gu

def bf(x):
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    r = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return r.astype(np.uint32).view(np.float32)

def emul(params, cfg, lc, pc, z_vals, depth_sample, dirs_C_sample, T_WC_sample, norm_sample, noise,
         s1_from_a=True, inj_bf16=True, hp=lambda x: x):
    """kernel numerics model: every GEMM operand rounded to bf16 (hp = operand rounding fn)."""
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
    raw = af @ w_out + params['out_alpha.bias'][0]      # fp32 VALU dot on unrounded a_L
    sdf = ((raw + noise.reshape(-1)) * so).reshape(R, S)
    def s1(li):
        if not s1_from_a: return S1x[li]
        bz = Z[li] * 100
        return np.where(bz > 20, np.float32(1), np.float32(1) - np.exp(np.float32(-100) * A[li]))
    # bwd1
    q = np.broadcast_to(w_out * so, (x.shape[0], H)).astype(np.float32)
    P = [None] * L; Q = [None]*L
    Eg = np.zeros((x.shape[0], cfg.E), np.float32)
    for li in range(L - 1, -1, -1):
        s = S1x[li] if li == L - 1 else s1(li)
        p = q * s; Q[li] = q
        P[li] = hp(p)
        g = P[li] @ Wb[cfg.names[li]]
        if li == cfg.cat: q = g[:, :H]; Eg = Eg + g[:, H:]
        elif li == 0: Eg = Eg + g
        else: q = g
    sdf_grad = orc.pe_vjp(x, Eg, cfg.transform, cfg.scale_input, cfg.n_freqs).reshape(R, S, 3)
    bounds, grad_vec = orc.bounds_ray(depth_sample, z_vals, dirs_C_sample, T_WC_sample)
    terms = orc.loss_terms(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc)
    sbar, gbar = orc.loss_adjoints(sdf, sdf_grad, bounds, grad_vec, norm_sample, lc)
    N = np.float32(sdf.size)
    sbar = sbar.reshape(-1) * N; gbar = gbar * N   # unnormalised per-point adjoints (scale at AdamW)
    Eb = hp(orc.pe_jvp(x, gbar.reshape(-1, 3), cfg.transform, cfg.scale_input, cfg.n_freqs))
    grads = {}
    qb = None; inj = [None]*L
    for li, n in enumerate(cfg.names):
        Gb = Eb if li == 0 else (np.concatenate([qb, Eb], -1) if li == cfg.cat else qb)
        u = Gb @ Wb[n].T
        grads[n + '.weight'] = P[li].T @ Gb
        s = s1(li)
        qbf = u * s
        # inj = u*q*s'' = 100*u*p*(1-s')
        pj = Q[li] * s
        ij = np.float32(100) * u * pj * (1 - s)
        ij = np.where(Z[li]*100 < 20, ij, 0)
        inj[li] = hp(ij) if inj_bf16 else ij
        qb_last = qbf
        qb = hp(qbf)
    d_wout = so * qb_last.sum(0)
    ab = (sbar * so)[:, None] * w_out[None, :]
    d_wout = d_wout + (sbar * so) @ orc.softplus(Z[-1])
    for li in range(L - 1, -1, -1):
        n = cfg.names[li]
        zb = ab * s1(li) + inj[li]
        zbb = hp(zb)
        inp = (np.concatenate([A[li-1], embb], -1) if li == cfg.cat else (A[li-1] if li > 0 else embb))
        grads[n + '.weight'] = grads[n + '.weight'] + zbb.T @ inp
        grads[n + '.bias'] = zb.sum(0)
        if li > 0: ab = (zbb @ Wb[n])[:, :H]
    grads['out_alpha.weight'] = d_wout[None]
    grads['out_alpha.bias'] = np.array([(sbar*so).sum()])
    for k in grads: grads[k] = grads[k] / N
    return terms, sdf, sdf_grad, grads


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
def split2(x):  # hi+lo bf16 pair ~ 16 mantissa bits
    hi = bf(x); lo = bf(np.asarray(x, np.float32) - hi); return hi + lo


if __name__ == '__main__':
    g = gu.load('eval_full_ray')
    cfg, lc, params = gu.net_of(g), gu.loss_of(g), gu.params_of(g)
    noise = g['draw_noise'].reshape(g['z_vals'].shape) * np.float32(g['noise_std'][0])
    args = (g['pc'], g['z_vals'], g['depth_sample'], g['dirs_C_sample'], g['T_WC_sample'], g['norm_sample'])
    t0, g0 = orc.loss_and_grads(params, cfg, lc, *args, noise=noise)
    def report(label, t, sdf, sg, gr):
        worst = max(gu.rel_err(gr[k], g0[k]) for k in g0)
        print('%-34s sdf %.2e (max %.2e) grad %.2e (max %.2e) | loss tot %.1e sdf %.1e grad %.1e eik %.1e | dW worst relL2 %.2e' % (
          label, gu.rel_err(sdf, t0['sdf']), np.abs(sdf-t0['sdf']).max()/np.abs(t0['sdf']).max(),
          gu.rel_err(sg, t0['sdf_grad']), np.abs(sg-t0['sdf_grad']).max()/np.abs(t0['sdf_grad']).max(),
          *[abs(t[k]-t0[k])/abs(t0[k]) for k in ['total_loss','sdf_loss','grad_loss','eikonal_loss']], worst))
    report('bf16', *emul(params, cfg, lc, *args, noise, hp=bf))
    report('fp16', *emul(params, cfg, lc, *args, noise, hp=f16))
    report('split2 (bf16 hi+lo everywhere)', *emul(params, cfg, lc, *args, noise, hp=split2))
    # bf16-representable weights: oracle and kernel share them
    pb = {k: (bf(v) if k.endswith('weight') and not k.startswith('out') else v) for k, v in params.items()}
    t0b, g0b = orc.loss_and_grads(pb, cfg, lc, *args, noise=noise)
    t, sdf, sg, gr = emul(pb, cfg, lc, *args, noise, hp=bf)
    print('bf16 kernel, bf16-representable weights: sdf %.2e grad %.2e' % (gu.rel_err(sdf, t0b['sdf']), gu.rel_err(sg, t0b['sdf_grad'])))
