"""Dev probe (GPU): which points of a 27 000-point sdf_eval come out wrong, and by how much, when a chain-kernel build
misbehaves with two workgroups per CU.  Written in round 4, when every change to how the epilogues fetch the biases broke the
>= 422-tile cases non-deterministically while <= 256 tiles stayed clean; it led to the `v_pk_fma_f32 ... op_sel:[0,1,0]` finding
(isdf_amd/isa_lint.py rule 1, profiles/r04_pk_fma_opsel_erratum.txt).  For every wrong point the error is matched against the
per-wave partial sums of the output layer (wave w owns hidden units 32w .. 32w+31); with the `partk` diagnostic build (tile t
reports the partial of wave t % 8 alone, --partk) against the single terms w_out[u] * a[u] of that wave.

    ISDF_HIP_LIB=variants/lib_x.so python tests/fwd_race_probe.py [--reps 20] [--n 27000] [--partk]

A clean library prints `bad runs 0/20` twice.  Test infrastructure: uses the oracle for the expected values."""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import golden_util as gu
from tests.test_gpu_parity import _engine, _dev
from oracle import isdf_oracle as orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=27000)
    ap.add_argument("--fwd-operand", default="fp16x2")
    ap.add_argument("--partk", action="store_true",
                    help="the library is the `partk` diagnostic build: with want_grad, tile t reports the output-layer partial of wave t % 8 alone")
    args = ap.parse_args()
    g = gu.load("eval_full_ray")
    n = args.n
    x = np.random.RandomState(n).uniform(-3, 3, (n, 3)).astype(np.float32)
    cfg, params = gu.net_of(g), gu.params_of(g)
    emb, I, Z, A, raw = orc._forward_cache(params, cfg, x)
    so = np.float32(cfg.scale_output)
    ref = raw * so
    _, refg = orc.sdf_forward_grad(params, cfg, x)
    partial = (A[-1] * params["out_alpha.weight"][0]).reshape(n, 8, 32).sum(-1) * so      # [n, wave]
    eng = _engine(g, args.fwd_operand)
    xd = _dev(x)
    out = {"lib": os.environ.get("ISDF_HIP_LIB", "in-tree"), "n": n, "reps": args.reps, "modes": {}}
    for mode, want_grad in (("sdf_only", False), ("sdf_and_grad", True)):
        bad_runs, all_bad = 0, {}
        gerr_max = 0.0
        for r in range(args.reps):
            res = eng.sdf_eval(xd, want_grad=want_grad)
            sdf = (res[0] if want_grad else res).cpu().numpy()
            if want_grad:
                gerr_max = max(gerr_max, float(np.abs(res[1].cpu().numpy() - refg).max()))
            err = sdf - ref
            bad = np.nonzero(np.abs(err) > 1.5e-3 * 0.14 * 3)[0]          # clean runs sit below 1e-3 of the output scale
            if len(bad):
                bad_runs += 1
            for i in bad:
                all_bad.setdefault(int(i), []).append(float(err[i]))
        rows = []
        for i in sorted(all_bad)[:4000]:
            e = np.mean(all_bad[i])
            miss = np.abs(e + partial[i])            # a partial that never arrived
            dbl = np.abs(e - partial[i])             # ... or arrived twice
            rows.append(dict(idx=i, tile=i // 64, lane=i % 64, times=len(all_bad[i]), err=e, spread=float(np.ptp(all_bad[i])),
                             miss_wave=int(miss.argmin()), miss_resid=float(miss.min()), dbl_wave=int(dbl.argmin()), dbl_resid=float(dbl.min())))
        tiles = sorted({r["tile"] for r in rows})
        summary = dict(bad_runs=bad_runs, bad_points=len(all_bad), bad_tiles=len(tiles), first_tiles=tiles[:24],
                       tiles_ge_256=sum(t >= 256 for t in tiles), grad_abs_err_max=gerr_max,
                       lanes_hist=np.bincount([r["lane"] for r in rows], minlength=64).tolist() if rows else [],
                       miss_wave_hist=np.bincount([r["miss_wave"] for r in rows if r["miss_resid"] < 3e-4], minlength=8).tolist() if rows else [],
                       dbl_wave_hist=np.bincount([r["dbl_wave"] for r in rows if r["dbl_resid"] < 3e-4], minlength=8).tolist() if rows else [],
                       explained_missing=sum(r["miss_resid"] < 3e-4 for r in rows), explained_doubled=sum(r["dbl_resid"] < 3e-4 for r in rows),
                       sample=rows[:12])
        out["modes"][mode] = summary
        print("%-28s %-13s bad runs %d/%d, bad points %d in %d tiles (%d with tile >= 256), grad err max %.2e; missing-partial explains %d, doubled %d"
              % (out["lib"][-28:], mode, bad_runs, args.reps, len(all_bad), len(tiles), summary["tiles_ge_256"], gerr_max,
                 summary["explained_missing"], summary["explained_doubled"]))
        if rows:
            print("   lanes", summary["lanes_hist"])
            print("   miss wave hist", summary["miss_wave_hist"], "dbl", summary["dbl_wave_hist"], "first tiles", tiles[:16])
            for r in rows[:6]:
                print("   ", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()})
    if args.partk:
        contrib = (A[-1] * params["out_alpha.weight"][0]) * so                  # [n, 256] per-unit contributions
        k_of = (np.arange(n) // 64) % 8
        want = partial[np.arange(n), k_of]
        hist_feat, hist_grp, unexplained, nbad = np.zeros(32, int), np.zeros(4, int), 0, 0
        shown = 0
        for r in range(args.reps):
            got = eng.sdf_eval(xd, want_grad=True)[0].cpu().numpy()
            err = got - want
            for i in np.nonzero(np.abs(err) > 3e-4)[0]:
                nbad += 1
                c = contrib[i, 32 * k_of[i]: 32 * k_of[i] + 32]                     # this wave's 32 units
                # unit u = 16 qp + 4 hi + (e & 3) + 8 (e >> 2): the lane half `hi` of block qp holds 8 of them
                grp = np.array([[c[[16 * qp + 4 * hi + (e & 3) + 8 * (e >> 2) for e in range(8)]].sum() for hi in range(2)] for qp in range(2)]).reshape(-1)
                f_best, g_best = np.abs(err[i] + c).argmin(), np.abs(err[i] + grp).argmin()
                if abs(err[i] + grp[g_best]) < 1e-4 * max(1, abs(grp[g_best]) / 1e-3):
                    hist_grp[g_best] += 1
                elif abs(err[i] + c[f_best]) < 5e-5:
                    hist_feat[f_best] += 1
                else:
                    unexplained += 1
                    if shown < 8:
                        shown += 1
                        print("    point %d (tile %d lane %d wave %d): got %.6f want %.6f err %.6f; groups(qp,hi) %s" %
                              (i, i // 64, i % 64, k_of[i], got[i], want[i], err[i], np.round(grp, 6).tolist()))
        print("partk: %d wrong partials over %d runs; missing (qp,hi) group hist %s; missing single unit hist %s; unexplained %d"
              % (nbad, args.reps, hist_grp.tolist(), hist_feat.tolist(), unexplained))
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
