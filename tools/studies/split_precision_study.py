"""Which operands need a hi+lo fp16 split for sdf / d sdf/dx to meet 1e-3 vs the reference at BASELINE size?
(dev tooling; numpy model of the chain kernel's forward + first reverse sweep; imports the test oracle)
Run: python tools/studies/split_precision_study.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.isdf_oracle as orc
from tests import golden_util as gu

def f16(x): return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
def bf(x):
    x = np.ascontiguousarray(x, np.float32); u = x.view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)
def x2(x):
    h = f16(x); return h + f16(np.asarray(x, np.float32) - h)
ident = lambda x: np.asarray(x, np.float32)
Q = {'f16': f16, 'x2': x2, 'bf': bf, 'f32': ident}

def model(params, cfg, x, wfwd, afwd, wbwd, pbwd, s1src, emb_q='f16'):
    """wfwd[li], afwd[li] (rounding of the INPUT activation of layer li), wbwd[li], pbwd[li] (p operand), s1src in {'exact','bf','f16'}"""
    L = len(cfg.names); so = np.float32(cfg.scale_output); H = cfg.H
    emb = orc.positional_encoding(x, cfg.transform, cfg.scale_input, cfg.n_freqs)
    Z = []; A = []
    af = None
    for li, n in enumerate(cfg.names):
        qa = Q[afwd[li]]
        if li == 0: inp = qa(emb)
        elif li == cfg.cat: inp = np.concatenate([qa(af), qa(emb)], -1)
        else: inp = qa(af)
        z = inp @ Q[wfwd[li]](params[n + '.weight']).T + params[n + '.bias']
        af = orc.softplus(z); Z.append(z); A.append(af)
    w_out = params['out_alpha.weight'][0]
    raw = af @ w_out + params['out_alpha.bias'][0]
    sdf = raw * so
    def s1(li):
        if s1src == 'exact' or li == L - 1: return orc.softplus_d1(Z[li])
        a = Q[s1src](A[li])
        return np.where(Z[li] * 100 > 20, np.float32(1), np.float32(1) - np.exp(np.float32(-100) * a))
    q = np.broadcast_to(w_out * so, (x.shape[0], H)).astype(np.float32)
    Eg = np.zeros((x.shape[0], cfg.E), np.float32)
    for li in range(L - 1, -1, -1):
        p = Q[pbwd[li]](q * s1(li))
        g = p @ Q[wbwd[li]](params[cfg.names[li] + '.weight'])
        if li == cfg.cat: q = g[:, :H]; Eg = Eg + g[:, H:]
        elif li == 0: Eg = Eg + g
        else: q = g
    sg = orc.pe_vjp(x, Eg, cfg.transform, cfg.scale_input, cfg.n_freqs)
    return sdf, sg

if __name__ == '__main__':
    names = sys.argv[1:] or ['eval_base_680x1200_ray', 'eval_base_480x640_ray', 'eval_full_ray']
    for name in names:
        g = gu.load(name)
        cfg, params = gu.net_of(g), gu.params_of(g)
        x = g['pc'].reshape(-1, 3).astype(np.float32)
        ref_sdf = g['sdf_nonoise'].reshape(-1); ref_g = g['sdf_grad'].reshape(-1, 3)
        L = len(cfg.names)
        def run(label, wf, af_, wb, pb, s1src):
            sdf, sg = model(params, cfg, x, wf, af_, wb, pb, s1src)
            print('  %-58s sdf %.2e   dsdf/dx %.2e' % (label, gu.rel_err(sdf, ref_sdf), gu.rel_err(sg, ref_g)))
        print(name, 'L =', L)
        A = lambda v: [v] * L
        def some(v, idx, base='f16'): return [v if i in idx else base for i in range(L)]
        run('fp32 everything (oracle check)', A('f32'), A('f32'), A('f32'), A('f32'), 'exact')
        run('f16 all, s1 from bf16 a (today)', A('f16'), A('f16'), A('f16'), A('f16'), 'bf')
        run('f16 all, s1 exact', A('f16'), A('f16'), A('f16'), A('f16'), 'exact')
        run('f16 all, s1 from f16 a', A('f16'), A('f16'), A('f16'), A('f16'), 'f16')
        run('W x2 layers 3-5 fwd', some('x2', (3,4,5)), A('f16'), A('f16'), A('f16'), 'bf')
        run('W x2 all fwd', A('x2'), A('f16'), A('f16'), A('f16'), 'bf')
        run('W x2 all fwd + act x2 all', A('x2'), A('x2'), A('f16'), A('f16'), 'bf')
        run('W x2 layers 2-5 fwd', some('x2', (2,3,4,5)), A('f16'), A('f16'), A('f16'), 'bf')
        run('W x2 all fwd+bwd, s1 bf', A('x2'), A('f16'), A('x2'), A('f16'), 'bf')
        run('W x2 all fwd+bwd, s1 f16', A('x2'), A('f16'), A('x2'), A('f16'), 'f16')
        run('W x2 all fwd+bwd, s1 exact', A('x2'), A('f16'), A('x2'), A('f16'), 'exact')
        run('W x2 all fwd+bwd, act+p x2, s1 exact', A('x2'), A('x2'), A('x2'), A('x2'), 'exact')
        run('W x2 3-5 fwd+bwd, s1 f16', some('x2', (3,4,5)), A('f16'), some('x2', (3,4,5)), A('f16'), 'f16')
        run('W x2 3-5 fwd+bwd, s1 exact', some('x2', (3,4,5)), A('f16'), some('x2', (3,4,5)), A('f16'), 'exact')
        run('W f16, act x2 all', A('f16'), A('x2'), A('f16'), A('f16'), 'bf')
