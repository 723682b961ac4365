"""numpy model of the chain kernel's FORWARD numerics (TEST INFRASTRUCTURE): every GEMM operand -- weights,
embedding, activations -- rounded to the 16-bit MFMA operand type, fp32 accumulation, fp32 element-wise math,
the output layer as an fp32 dot product (isdf_amd/csrc/chain.hip).  It separates "the kernel computes what its
design says" (HIP vs this model: accumulation order and transcendental approximations only) from the operand
rounding floor of that design (this model vs the fp32 reference)."""
import numpy as np

import oracle.isdf_oracle as orc


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def bf16(x):
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    r = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return r.astype(np.uint32).view(np.float32)


def forward(params, cfg, x, operand="fp16"):
    """operand "fp16x2": the compensated forward (isdf_amd/csrc/chain.hip, OPER 2) -- the cat layer adds W_lo[:, H:] emb,
    the layers past it W_lo x and W x_lo (W_lo = fp16(W - fp16(W)), x_lo = fp16(x - fp16(x))).
    "fp16x2_full" (OPER 3): those three products in EVERY layer, the embedding included."""
    x2 = operand == "fp16x2"
    full = operand == "fp16x2_full"
    q = bf16 if operand == "bf16" else f16
    x = np.asarray(x, np.float32).reshape(-1, 3)
    ef = orc.positional_encoding(x, cfg.transform, cfg.scale_input, cfg.n_freqs)
    e = q(ef)
    e_lo = f16(ef - e)
    a, a_lo = e, e_lo
    for li, n in enumerate(cfg.names):
        inp = np.concatenate([a, e], -1) if li == cfg.cat else a
        W = params[n + ".weight"]
        z = inp @ q(W).T
        if full:
            inp_lo = np.concatenate([a_lo, e_lo], -1) if li == cfg.cat else a_lo
            z = z + inp @ f16(W - f16(W)).T + inp_lo @ f16(W).T
        elif x2 and li == cfg.cat:      # residual of the embedding columns only
            z = z + e @ f16(W - f16(W))[:, cfg.H:].T
        elif x2 and li > cfg.cat:
            z = z + inp @ f16(W - f16(W)).T + f16(af - a) @ f16(W).T
        z = z + params[n + ".bias"]
        af = orc.softplus(z)
        a = q(af)
        a_lo = f16(af - a)
    raw = af @ params["out_alpha.weight"][0] + params["out_alpha.bias"][0]
    return raw * np.float32(cfg.scale_output)
