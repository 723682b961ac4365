#!/usr/bin/env python3
"""Systematic-gradient-error probe at TRAINED weights (test infrastructure; not collected by pytest).

Why: a kernel change (round 3, commit d490710) passed every rel-L2 / cosine gradient bound at random-initialised
weights and still moved the trained SDF accuracy.  A rel-L2 bound cannot see a small error that has the SAME sign step
after step; the signed projection  <g_hip - g_ref, g_ref> / |g_ref|^2  can, and a trained Softplus(beta = 100) network
(most units saturated, few in transition) is where the backward formulas are exercised differently from random init.

Reference: the reference's op chain as PyTorch-ROCm eager autograd (oracle/torch_port.py) on the SAME GPU in FLOAT64
(truth) and float32 (what a legitimate fp32 implementation scatters), on the sampler outputs and the noise tensor the
HIP step consumes.

    python tests/grad_bias_probe.py --make-weights gpurun_out/probe_w.pt          # eager fp32 training, snapshots
    ISDF_HIP_LIB=variants/lib_x.so python tests/grad_bias_probe.py --weights gpurun_out/probe_w.pt --out x.json

Per snapshot and tensor: rel-L2, cosine and signed projection of the HIP gradient against the float64 one (and of the
float32 eager gradient, the noise floor); per-unit bias-gradient errors of the top hidden layer; then `--traj N` steps of
AdamW from the snapshot (moments included) on identical samples: the signed projection of the accumulated parameter UPDATE.
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isdf_amd import synthetic  # noqa: E402
from oracle import torch_port as tp  # noqa: E402

LC = dict(trunc_distance=0.29365022, loss_type="L1", trunc_weight=5.38344020, eik_apply_dist=0.1, eik_weight=0.268,
          grad_weight=0.018)
SC = dict(n_rays=200, n_strat=19, n_surf=8, min_depth=0.07, dist_behind_surf=0.1)


def make_weights(path, seed, keyframes, steps_per_kf, snaps):
    """pinned-schedule eager fp32 training (tests/accuracy_experiment.run_port) with parameter / AdamW snapshots"""
    cam = dict(synthetic.SCANNET_CAM)
    from tests.accuracy_experiment import prepare_keyframes
    depth, normal, T = prepare_keyframes([seed], keyframes)(seed)      # the accuracy experiment's inputs of this seed
    dev = torch.device("cuda")
    np.random.seed(seed); torch.manual_seed(seed)
    net = tp.PortNet(256, 2, 6, 0.05937489, 0.14, synthetic.bounds_transform()).to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=0.0013, weight_decay=0.012)
    gen = torch.Generator().manual_seed(seed)
    d, n, Tt = torch.from_numpy(depth).to(dev), torch.from_numpy(normal).to(dev), torch.from_numpy(T).to(dev)
    fal = torch.zeros(0, device=dev)
    out = dict(depth=depth, normal=normal, T=T, cam=cam, snaps={})
    step = 0

    def snap():
        st = opt.state_dict()["state"]
        out["snaps"][step] = dict(
            params={k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
            exp_avg=[st[i]["exp_avg"].cpu().clone() for i in range(len(st))],
            exp_avg_sq=[st[i]["exp_avg_sq"].cpu().clone() for i in range(len(st))],
            opt_step=step, K=int(fal.numel()))
    for k in range(keyframes):
        fal = torch.cat((fal, torch.zeros(1, device=dev)))
        K = k + 1
        for _ in range(steps_per_kf):
            if step in snaps and step > 0:
                snap()
            if K > 5:
                p = (fal[:-2] / fal[:-2].sum()).cpu().numpy()
                idxs = [*np.random.choice(np.arange(0, K - 2), size=3, replace=False, p=p), K - 2, K - 1]
            else:
                idxs = list(range(K))
            losses, fa = tp.train_step(net, opt, d[idxs], Tt[idxs], n[:len(idxs)], cam, SC, LC, 0.08, gen)
            fal[idxs] = fa
            step += 1
    snap()
    torch.save(out, path)
    print("wrote", path, "snapshots at", sorted(out["snaps"]))


def port_sample_from_hip(s):
    R = int(s["n_valid"].item())
    return dict(pc=s["pc"][:R], z=s["z_vals"][:R], depth=s["depth_sample"][:R], dC=s["dirs_C_sample"][:R],
                dW=s["dirs_W_sample"][:R], normals=s["norm_sample"][:R], ib=s["indices_b"][:R], ih=s["indices_h"][:R],
                iw=s["indices_w"][:R]), R


def ref_grads(net, s, noise, dtype):
    """gradient of the mean loss w.r.t. every parameter, reference op chain, eager autograd in `dtype`"""
    n2 = copy.deepcopy(net).to(dtype)
    s2 = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in s.items()}
    total, losses, tot = tp.loss_step(n2, s2, LC, None, None, noise=noise.to(dtype))
    total.backward()
    return {k: p.grad.detach().double() for k, p in n2.named_parameters()}, float(total.detach()), losses


def stats(g, r):
    g, r = g.double().reshape(-1), r.double().reshape(-1)
    rr = float(r @ r)
    return dict(rel_l2=float((g - r).norm() / r.norm()), cos=float(g @ r / (g.norm() * r.norm())),
                signed_proj=float((g - r) @ r / rr))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-weights", default=None)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keyframes", type=int, default=24)
    ap.add_argument("--steps-per-kf", type=int, default=100)
    ap.add_argument("--snaps", type=int, nargs="+", default=[300, 1000, 2400])
    ap.add_argument("--batches", type=int, default=4, help="independent sample batches per snapshot")
    ap.add_argument("--traj", type=int, default=50, help="AdamW steps from each snapshot on identical samples")
    ap.add_argument("--fwd-operand", default="fp16x2")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.make_weights:
        make_weights(a.make_weights, a.seed, a.keyframes, a.steps_per_kf, set(a.snaps))
        return
    from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig
    from isdf_amd import _ffi
    W = torch.load(a.weights, weights_only=False)
    cam = W["cam"]
    dev = torch.device("cuda")
    depth, normal, T = (torch.from_numpy(W[k]).to(dev) for k in ("depth", "normal", "T"))
    eng = Engine(NetConfig(transform=synthetic.bounds_transform(), fwd_operand=a.fwd_operand), dev)
    sc = SampleConfig(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    lc = LossConfig()
    net = tp.PortNet(256, 2, 6, 0.05937489, 0.14, synthetic.bounds_transform()).to(dev)
    names = [k for k, _ in net.named_parameters()]
    report = dict(lib=os.environ.get("ISDF_HIP_LIB", "isdf_amd/libisdf_hip.so"), fwd_operand=a.fwd_operand, snapshots={})
    for step in sorted(W["snaps"]):
        sn = W["snaps"][step]
        net.load_state_dict(sn["params"])
        K = sn["K"]
        fidx = torch.arange(max(0, K - 5), K, dtype=torch.int32, device=dev)
        nidx = torch.arange(len(fidx), dtype=torch.int32, device=dev)      # quirk q4, as the training runs do
        rows = []
        acc = {k: [] for k in ("hip", "f32")}
        top_bias = []
        for b in range(a.batches):
            eng.load_params(sn["params"])
            s = eng.sample(depth, T, normal, fidx, nidx, sc, seed=1000 + step, offset=b + 1)
            ps, R = port_sample_from_hip(s)
            noise = torch.randn(R, sc.S, generator=torch.Generator().manual_seed(step * 100 + b)).to(dev) * 0.08
            g64, tot64, l64 = ref_grads(net, ps, noise, torch.float64)
            g32, tot32, _ = ref_grads(net, ps, noise, torch.float32)
            dbg = eng.train_step(s, lc, sc, noise=noise, debug=True)
            cnt = float(eng.loss_sums()[_ffi.LS_COUNT].item())
            gh = {k: (eng.grad_view(k).double() / cnt).clone() for k in names}
            ls = eng.loss_sums().cpu().numpy()
            row = dict(batch=b, R=R, total_loss_ref64=tot64, total_loss_hip=float(ls[_ffi.LS_TOTAL] / ls[_ffi.LS_COUNT]),
                       tensors={k: dict(hip=stats(gh[k], g64[k]), f32=stats(g32[k], g64[k])) for k in names})
            allh = torch.cat([gh[k].reshape(-1) for k in names]); all64 = torch.cat([g64[k].reshape(-1) for k in names])
            all32 = torch.cat([g32[k].reshape(-1) for k in names])
            row["all"] = dict(hip=stats(allh, all64), f32=stats(all32, all64))
            rows.append(row)
            acc["hip"].append(allh - all64); acc["f32"].append(all32 - all64)
            kb = names[-4]      # top hidden layer's bias (mid2.<B-1>.0.bias): d b = sum over points of zbar_L
            top_bias.append(((gh[kb] - g64[kb]) / g64[kb].abs().clamp_min(g64[kb].abs().max() * 1e-3)).cpu().numpy())
        # is the error the SAME from batch to batch (bias) or independent (noise)?  correlation of the error vectors
        def coherence(errs):
            e = torch.stack(errs)
            m = e.mean(0)
            return float(m.norm() ** 2 * len(errs) / (e.norm() ** 2))     # 1/len... = 1 for identical errors, ~1/B for independent ones
        tb = np.stack(top_bias)
        snap = dict(batches=rows, error_coherence=dict(hip=coherence(acc["hip"]), f32=coherence(acc["f32"])),
                    top_bias_rel_err=dict(mean=float(tb.mean()), median=float(np.median(tb)), rms=float(np.sqrt((tb ** 2).mean()))))
        # ---- trajectory: `traj` AdamW steps from this snapshot (moments included), identical samples and noise
        if a.traj > 0:
            net.load_state_dict(sn["params"])
            opt = torch.optim.AdamW(net.parameters(), lr=0.0013, weight_decay=0.012)
            st = {"state": {i: dict(step=torch.tensor(float(sn["opt_step"])), exp_avg=sn["exp_avg"][i].to(dev).clone(),
                                    exp_avg_sq=sn["exp_avg_sq"][i].to(dev).clone()) for i in range(len(names))},
                  "param_groups": opt.state_dict()["param_groups"]}
            opt.load_state_dict(st)
            eng.load_params(sn["params"])
            flat = lambda lst: torch.cat([t.reshape(-1) for t in lst]).to(dev)
            eng.exp_avg.copy_(flat(sn["exp_avg"])); eng.exp_avg_sq.copy_(flat(sn["exp_avg_sq"])); eng.opt_step = sn["opt_step"]
            theta0 = eng.params.double().clone()
            curve = []
            for t in range(a.traj):
                s = eng.sample(depth, T, normal, fidx, nidx, sc, seed=5000 + step, offset=t + 1)
                ps, R = port_sample_from_hip(s)
                noise = torch.randn(R, sc.S, generator=torch.Generator().manual_seed(step * 1000 + t)).to(dev) * 0.08
                total, losses, tot = tp.loss_step(net, {k: v.clone() for k, v in ps.items()}, LC, None, None, noise=noise)
                total.backward(); opt.step()
                for p in net.parameters():
                    p.grad = None
                eng.train_step(s, lc, sc, noise=noise)
                eng.adamw()
                if (t + 1) % 10 == 0 or t + 1 == a.traj:
                    th_ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double()
                    th_hip = eng.params.double()
                    du_r, du_h = th_ref - theta0, th_hip - theta0
                    curve.append(dict(step=t + 1, update_rel_l2=float((du_h - du_r).norm() / du_r.norm()),
                                      update_signed_proj=float((du_h - du_r) @ du_r / (du_r @ du_r)),
                                      update_cos=float(du_h @ du_r / (du_h.norm() * du_r.norm())),
                                      loss_ref=float(total), loss_hip=float((eng.loss_sums()[_ffi.LS_TOTAL] / eng.loss_sums()[_ffi.LS_COUNT]).item())))
            snap["trajectory"] = curve
        report["snapshots"][step] = snap
        al = [r["all"] for r in rows]
        print("step %5d  all-params: hip rel-L2 %.2e signed %.2e (f32 eager %.1e / %.1e)  coherence hip %.2f f32 %.2f  top-bias rel err mean %+.2e rms %.2e"
              % (step, np.mean([x["hip"]["rel_l2"] for x in al]), np.mean([x["hip"]["signed_proj"] for x in al]),
                 np.mean([x["f32"]["rel_l2"] for x in al]), np.mean([x["f32"]["signed_proj"] for x in al]),
                 snap["error_coherence"]["hip"], snap["error_coherence"]["f32"], snap["top_bias_rel_err"]["mean"],
                 snap["top_bias_rel_err"]["rms"]), flush=True)
        worst = sorted(((np.mean([abs(r["tensors"][k]["hip"]["signed_proj"]) for r in rows]), k) for k in names), reverse=True)[:4]
        print("           largest |signed projection|:", ", ".join("%s %.2e" % (k, v) for v, k in worst), flush=True)
        if a.traj > 0:
            c = snap["trajectory"][-1]
            print("           %d-step update: rel-L2 %.2e signed %+.2e cos %.6f" % (c["step"], c["update_rel_l2"],
                                                                                   c["update_signed_proj"], c["update_cos"]), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
