"""Dev tool: per-tensor gradient errors of the <256,512> instantiation for several (blocks, n_freqs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle.isdf_oracle as orc
from tests import golden_util as gu
from tests.test_gpu_parity import _cfgs, _sample_hip, _dev
from isdf_amd.engine import Engine, NetConfig
g = gu.load("eval_full_ray")
for blocks, nf in [(2, 9), (3, 9), (2, 11), (3, 11), (2, 10), (2, 12)]:
    params = orc.init_params(256, blocks, nf, np.random.RandomState(70 + nf))
    net = NetConfig(hidden=256, blocks=blocks, n_freqs=nf, scale_input=0.05937489, scale_output=0.14, transform=g["bounds_T"])
    eng = Engine(net, "cuda"); eng.load_params(params)
    cfg = orc.NetCfg(256, blocks, nf, 0.05937489, 0.14, g["bounds_T"])
    lc, sc = _cfgs(g)
    s_ = _sample_hip(eng, g, sc)
    R = g["depth_sample"].shape[0]
    noise = g["draw_noise"].reshape(R, -1) * np.float32(0.08)
    eng.train_step(s_, lc, sc, noise=_dev(noise))
    terms, grads = orc.loss_and_grads(params, cfg, gu.loss_of(g), g["pc"], g["z_vals"], g["depth_sample"],
                                      g["dirs_C_sample"], g["T_WC_sample"], g["norm_sample"], noise=noise)
    N = R * g["z_vals"].shape[1]
    print("blocks %d n_freqs %d E %d" % (blocks, nf, net.emb))
    for k in grads:
        got = (eng.grad_view(k).cpu().numpy().astype(np.float64) / N)
        ref = grads[k].astype(np.float64)
        line = "   %-22s rel %.2e" % (k, gu.rel_err(got, ref))
        if k in ("in_layer.0.weight", "cat_layer.0.weight"):
            E = net.emb; off = 0 if k.startswith("in") else 256
            half = 21 * nf
            for f in range(nf):   # error per frequency band (sin part and cos part columns of that frequency)
                cols = [off + 3 + d * nf + f for d in range(21)] + [off + 3 + half + d * nf + f for d in range(21)]
                line += " f%d:%.1e" % (f, gu.rel_err(got[:, cols], ref[:, cols]))
        print(line)
