"""Dev tool: gradient accuracy of the chain kernel against the oracle evaluated at the KERNEL's own parameters, before and
after AdamW steps (separates the kernel's error from trajectory divergence).  Used for the sigma'-as-unorm8 decision."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import tests.test_gpu_parity as T
from tests import golden_util as gu
import oracle.isdf_oracle as orc
g = gu.load("eval_full_ray")
eng = T._engine(g); lc, sc = T._cfgs(g)
cfg = gu.net_of(g); lco, cam = gu.loss_of(g), gu.cam_of(g)
F = g["depth_batch"].shape[0]
rng = np.random.RandomState(99)
for it in range(4):
    R0 = F * sc.n_rays
    draws = dict(indices_h=rng.randint(0, cam["H"], R0).astype(np.int64), indices_w=rng.randint(0, cam["W"], R0).astype(np.int64),
                 U=rng.uniform(size=(R0, sc.n_strat)).astype(np.float32), N_off=(0.1 * rng.standard_normal((R0, sc.n_surf - 1))).astype(np.float32))
    noise = (0.04 * rng.standard_normal((R0, sc.S))).astype(np.float32)
    idx = torch.arange(F, dtype=torch.int32, device="cuda")
    s = eng.sample(T._dev(g["depth_batch"]), T._dev(g["T_WC_batch"]), T._dev(g["normal_batch"]), idx, idx, sc,
                   draws={k: T._dev(v) for k, v in draws.items()}, want_T=True)
    R = int(s["n_valid"].item())
    params = {k: eng.param_view(k).cpu().numpy().copy() for k in eng.slices}
    eng.train_step(s, lc, sc, noise=T._dev(noise[:R]))
    torch.cuda.synchronize()
    N = R * sc.S
    Tw = g["T_WC_batch"][s["indices_b"][:R].cpu().numpy()]
    terms, grads = orc.loss_and_grads(params, cfg, lco, s["pc"][:R].cpu().numpy(), s["z_vals"][:R].cpu().numpy(), s["depth_sample"][:R].cpu().numpy(),
                                      s["dirs_C_sample"][:R].cpu().numpy(), Tw, s["norm_sample"][:R].cpu().numpy(), noise=noise[:R])
    errs = {k: gu.rel_err((eng.grad_view(k).cpu().numpy().astype(np.float64) / N).reshape(-1), grads[k].astype(np.float64).reshape(-1)) for k in grads}
    wk = max(errs, key=errs.get)
    print("step %d (kernel's own parameters): worst dW rel-L2 %.3e (%s)  in_layer.w %.3e  cat.w %.3e  out.w %.3e" % (it + 1, errs[wk], wk, errs["in_layer.0.weight"], errs["cat_layer.0.weight"], errs["out_alpha.weight"]))
    eng.adamw()
