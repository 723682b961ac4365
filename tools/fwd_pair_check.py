"""Dev tool: the pair-tile forward kernel (csrc/fwd_pair.hip) against chain.hip's one-tile MODE 0 -- bit for bit while both use
the same Softplus arithmetic (the kernel's first two versions: 66 of 66 arrays identical), to fp32 rounding of the epilogue since
fwd_pair.hip evaluates Softplus on the base-2 image of the pre-activation (chain_dev.h softplus_x): one flipped operand rounding of an activation moves sdf by
~1e-5 (fp16) / ~1e-4 (bf16); bars on max |d|: bf16 5e-4, fp16 1e-4, fp16x2 5e-5 (the sdf scale is 0.14).

    bash tools/ab_fwd_variants.sh                                                   # variants/lib_onetile.so: MODE 0 on the one-tile kernel (+ lib_dbg.so)
    python tools/fwd_pair_check.py --dump /tmp/a.npz                                # in-tree library
    ISDF_HIP_LIB=$PWD/variants/lib_onetile.so python tools/fwd_pair_check.py --dump /tmp/b.npz
    python tools/fwd_pair_check.py --compare /tmp/a.npz /tmp/b.npz

Sizes cover the ragged cases (1, 63 .. 129 points, a half-filled last pair), operand modes bf16 / fp16 / fp16x2, a narrower net
(zero-padded hidden width), explicit noise and the in-kernel Philox noise; --time N also times N points per mode."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def dump(path, time_n):
    import torch
    from isdf_amd.engine import Engine, NetConfig
    from isdf_amd import synthetic
    out = {}
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    big = ((torch.rand(300000, 3, generator=g) - 0.5) * torch.tensor([6.0, 3.0, 5.0])).cuda()
    nz = (torch.randn(300000, generator=g) * 0.01).cuda()
    for op in ("bf16", "fp16", "fp16x2"):
        for hidden in (256, 200):
            eng = Engine(NetConfig(hidden=hidden, transform=synthetic.bounds_transform(), fwd_operand=op), "cuda")
            torch.manual_seed(3); eng.params.normal_(0, 0.06); eng.pack()
            for n in (1, 63, 64, 65, 127, 128, 129, 1000, 27000, 32897, 300000):   # (256 CUs: 32897 points = 257 pairs + 1 point, the first size a persistent workgroup takes a second pair at)
                out["%s_h%d_n%d" % (op, hidden, n)] = eng.sdf_eval(big[:n]).cpu().numpy()
            out["%s_h%d_noise" % (op, hidden)] = eng.sdf_eval(big[:5000], noise=nz[:5000]).cpu().numpy()
            if time_n and hidden == 256:
                x = ((torch.rand(time_n, 3, generator=g) - 0.5) * torch.tensor([6.0, 3.0, 5.0])).cuda()
                for _ in range(3): eng.sdf_eval(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): eng.sdf_eval(x)
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 10 * 1e-3
                print("%-7s %d points: %.3f ms  %.1f TFLOP/s  %.4f of the MFMA peak" % (op, time_n, t * 1e3, 2 * 458496 * time_n / t / 1e12,
                                                                                       2 * 458496 * time_n / t / 2.5e15), flush=True)
    np.savez(path, **out)
    print("wrote", path, len(out), "arrays")


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        same = np.array_equal(x, y)
        d = float(np.abs(x - y).max()) if x.shape == y.shape else float("nan")
        if not same:
            bad += 1
        print("%-24s %s  max|d| %.3e  (n %d, nan %d)" % (k, "bit-identical" if same else "DIFFERENT    ", d, x.size, int(np.isnan(x).sum())))
    bars, fail = {"bf16": 5e-4, "fp16": 1e-4, "fp16x2": 5e-5}, 0
    for op, bar in bars.items():
        worst = max(float(np.abs(A[k] - B[k]).max()) for k in A.files if k.startswith(op + "_"))
        print("RESULT %-6s max |d| %.3e (bar %.0e; sdf scale 0.14)" % (op, worst, bar))
        fail += worst > bar
    print("RESULT: %d of %d arrays differ bitwise" % (bad, len(A.files)))
    return fail


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump"); ap.add_argument("--compare", nargs=2); ap.add_argument("--time", type=int, default=0)
    a = ap.parse_args()
    if a.dump:
        dump(a.dump, a.time)
    if a.compare:
        sys.exit(1 if compare(*a.compare) else 0)
