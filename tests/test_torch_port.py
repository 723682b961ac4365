"""Pin oracle/torch_port.py (the timed CPU baseline of bench.py) against the
reference-generated fixtures: same losses, same parameter gradients."""
import numpy as np
import torch

from oracle import torch_port as tp
from tests import golden_util as gu


def test_port_matches_reference_fixture():
    g = gu.load("eval_small_ray")
    H, B, nf, si, so = g["net"]
    net = tp.PortNet(int(H), int(B), int(nf), float(si), float(so), g["bounds_T"])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gu.params_of(g).items()})
    lcf = gu.loss_of(g)
    lc = dict(trunc_distance=lcf.trunc_distance, loss_type=lcf.loss_type, trunc_weight=lcf.trunc_weight,
              eik_apply_dist=lcf.eik_apply_dist, eik_weight=lcf.eik_weight, grad_weight=lcf.grad_weight)
    T = torch.from_numpy(g["T_WC_sample"]); dC = torch.from_numpy(g["dirs_C_sample"])
    s = dict(pc=torch.from_numpy(g["pc"]), z=torch.from_numpy(g["z_vals"]),
             depth=torch.from_numpy(g["depth_sample"]), dC=dC,
             dW=(T[:, :3, :3] * dC[:, None, :]).sum(-1), normals=torch.from_numpy(g["norm_sample"]))
    noise = torch.from_numpy(g["draw_noise"].reshape(g["z_vals"].shape) * np.float32(g["noise_std"][0]))
    total, losses, _ = tp.loss_step(net, s, lc, None, None, noise=noise)
    assert abs(float(total) - g["total_loss"][0]) < 1e-5 * abs(g["total_loss"][0])
    for k in ("sdf_loss", "grad_loss", "eikonal_loss"):
        assert abs(losses[k] - g[k][0]) < 1e-5 * abs(g[k][0]), k
    total.backward()
    for k, p in net.named_parameters():
        assert gu.rel_err(p.grad.numpy(), g["grad/" + k]) < 1e-4, k


def test_port_matches_reference_on_the_default_net_at_baseline_size():
    """The timed CPU / eager-GPU baseline of bench.py is THIS module on the default 6x256 net: pin it there too
    (`eval_base_480x640_ray`: the unmodified reference on 5 x 200 rays x 27 samples; losses + gradient digests)."""
    g = gu.load("eval_base_480x640_ray")
    H, B, nf, si, so = g["net"]
    assert (int(H), int(B), int(nf)) == (256, 2, 6)
    net = tp.PortNet(int(H), int(B), int(nf), float(si), float(so), g["bounds_T"])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in gu.params_of(g).items()})
    lcf = gu.loss_of(g)
    lc = dict(trunc_distance=lcf.trunc_distance, loss_type=lcf.loss_type, trunc_weight=lcf.trunc_weight,
              eik_apply_dist=lcf.eik_apply_dist, eik_weight=lcf.eik_weight, grad_weight=lcf.grad_weight)
    T = torch.from_numpy(g["T_WC_sample"]); dC = torch.from_numpy(g["dirs_C_sample"])
    s = dict(pc=torch.from_numpy(g["pc"]), z=torch.from_numpy(g["z_vals"]), depth=torch.from_numpy(g["depth_sample"]), dC=dC,
             dW=(T[:, :3, :3] * dC[:, None, :]).sum(-1), normals=torch.from_numpy(g["norm_sample"]),
             ib=torch.from_numpy(g["indices_b"]), ih=torch.from_numpy(g["indices_h"]), iw=torch.from_numpy(g["indices_w"]))
    noise = torch.from_numpy(g["draw_noise"].reshape(g["z_vals"].shape) * np.float32(g["noise_std"][0]))
    total, losses, tot = tp.loss_step(net, s, lc, None, None, noise=noise)
    assert abs(float(total) - g["total_loss"][0]) < 1e-5 * abs(g["total_loss"][0])
    for k in ("sdf_loss", "grad_loss", "eikonal_loss"):
        assert abs(losses[k] - g[k][0]) < 1e-5 * abs(g[k][0]), k
    cam = gu.cam_of(g)
    la, fa = tp.frame_avg_step(tot, s, 5, cam["H"], cam["W"])
    np.testing.assert_allclose(fa.numpy(), g["frame_avg_loss"], rtol=1e-4, atol=1e-7)
    total.backward()
    prng = np.random.RandomState(1234)
    for k, p in net.named_parameters():
        v = p.grad.numpy().astype(np.float64)
        probe = prng.standard_normal(v.shape)
        nrm, dot = g["gdig/" + k]
        assert abs(np.linalg.norm(v) - nrm) < 1e-4 * nrm, k
        assert abs((v * probe).sum() - dot) < 1e-4 * nrm * np.sqrt(v.size), k
        np.testing.assert_allclose(v.reshape(-1)[:64], g["ghead/" + k], rtol=0, atol=1e-4 * np.abs(g["ghead/" + k]).max())


def test_port_full_step_runs_and_learns():
    torch.manual_seed(0)
    net = tp.PortNet(64, 1)
    opt = torch.optim.AdamW(net.parameters(), lr=0.0013, weight_decay=0.012)
    gen = torch.Generator().manual_seed(1)
    g = gu.load("eval_small_ray")
    cam, sc = gu.cam_of(g), gu.sample_of(g)
    lcf = gu.loss_of(g)
    lc = dict(trunc_distance=lcf.trunc_distance, loss_type="L1", trunc_weight=lcf.trunc_weight,
              eik_apply_dist=lcf.eik_apply_dist, eik_weight=lcf.eik_weight, grad_weight=lcf.grad_weight)
    d, T, n = (torch.from_numpy(g[k]) for k in ("depth_batch", "T_WC_batch", "normal_batch"))
    first = last = None
    for i in range(6):
        losses, fa = tp.train_step(net, opt, d, T, n, cam, sc, lc, 0.04, gen)
        first = first or losses["total_loss"]
        last = losses["total_loss"]
        assert fa.shape == (d.shape[0],)
    assert np.isfinite(last) and last < first
