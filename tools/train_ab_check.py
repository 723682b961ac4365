"""Dev tool: one training step of two libraries on identical inputs, output by output (A/B of a kernel change).

    python tools/train_ab_check.py --dump /tmp/a.npz                                   # in-tree library
    ISDF_HIP_LIB=$PWD/variants/lib_prev.so python tools/train_ab_check.py --dump /tmp/b.npz
    python tools/train_ab_check.py --compare /tmp/a.npz /tmp/b.npz

Dumps sdf, d sdf / d x, the loss sums and the whole gradient buffer of one step on the BASELINE-size workload (fixed sampler draws,
zero noise) for the default and the plain-fp16 operand modes, and the same for the 8x512 net; --compare reports bit identity and
rel-L2 per array."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def dump(path):
    import torch
    from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
    from isdf_amd import synthetic
    out = {}
    cam = dict(synthetic.SCANNET_CAM)
    d, n, T = synthetic.keyframes(5, cam, seed=1)
    dev = lambda a: torch.as_tensor(a).cuda()
    d, n, T = dev(d), dev(n), dev(T)
    cases = [("default", {}), ("fp16", dict(fwd_operand="fp16")), ("bf16", dict(fwd_operand="bf16")),
             ("wide", dict(hidden=512, blocks=3, n_freqs=10)),
             ("nf11", dict(n_freqs=11)), ("nf9b3", dict(n_freqs=9, blocks=3))]      # the realsense nets' shapes (eleven / nine octaves)
    if "spill_operand" in NetConfig.__dataclass_fields__:      # the spill formats of the default net, side by side
        cases += [("spill16", dict(spill_operand="16bit")), ("spill_e4m3", dict(spill_operand="e4m3")),
                  ("spill_e4m3_gb", dict(spill_operand="e4m3_gb"))]
    for tag, kw in cases:
        eng = Engine(NetConfig(transform=synthetic.bounds_transform(), **kw), "cuda")
        torch.manual_seed(0); eng.params.normal_(0, 0.06 if tag != "wide" else 0.04); eng.pack()
        sc = SampleConfig(n_rays=200, **cam); lc = LossConfig()
        idx = torch.arange(5, dtype=torch.int32, device="cuda")
        s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=0)
        noise = torch.zeros(s["max_rays"], sc.S, device="cuda")
        dbg = eng.train_step(s, lc, sc, noise=noise, debug=True)
        torch.cuda.synchronize()
        R = int(s["n_valid"].item())
        out[tag + "/sdf"] = dbg["sdf"][:R].cpu().numpy()
        out[tag + "/sdf_grad"] = dbg["sdf_grad"][:R].cpu().numpy()
        out[tag + "/loss_sums"] = eng.loss_sums().cpu().numpy()
        out[tag + "/grad"] = eng.reduce_buf[:eng.n_params].cpu().numpy()
    np.savez(path, **out)
    print("wrote", path, len(out), "arrays")


def compare(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        x, y = A[k].astype(np.float64), B[k].astype(np.float64)
        same = np.array_equal(A[k], B[k])
        rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
        print("%-20s %s  rel-L2 %.3e  max|d| %.3e  (n %d, nan %d)" % (k, "bit-identical" if same else "different    ", rel, float(np.abs(x - y).max()), x.size, int(np.isnan(x).sum())))


def modes(a):
    A = np.load(a)
    ref = A["spill16/grad"].astype(np.float64)
    for tag in ("spill_e4m3", "spill_e4m3_gb", "default"):
        x = A[tag + "/grad"].astype(np.float64)
        print("%-16s vs spill16: grad rel-L2 %.3e  sign differs %.4f %%  | sdf identical %s" % (
            tag, np.linalg.norm(x - ref) / np.linalg.norm(ref), 100 * np.mean(np.sign(x) != np.sign(ref)),
            np.array_equal(A[tag + "/sdf"], A["spill16/sdf"])))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump"); ap.add_argument("--compare", nargs=2); ap.add_argument("--modes")
    a = ap.parse_args()
    if a.dump:
        dump(a.dump)
    if a.compare:
        compare(*a.compare)
    if a.modes:
        modes(a.modes)
