"""torch-CPU port of one reference training step -- TEST INFRASTRUCTURE, used only as
the timed `cpu_baseline` of bench.py (kind "port") and as a second checker.

The reference runs this path as PyTorch eager ops + autograd on whatever device
it is given (`isdf/modules/trainer.py:951-1016`); it cannot travel to the GPU
box (Python package with absent dependencies), so its op chain is restated here
in the same order with the same torch primitives (Linear, Softplus(beta=100),
autograd.grad(create_graph=True), CosineSimilarity(eps=1e-6), AdamW), which is
what "the reference's CPU PyTorch path timed on the same box's host cores"
means.  Pinned in tests/test_torch_port.py against the golden fixtures.
"""
import numpy as np
import torch

from .isdf_oracle import ICO_DIRS, layer_names


class PortNet(torch.nn.Module):
    """SDFMap + PostionalEncoding (`fc_map.py:63-111`, `embedding.py:24-111`)."""

    def __init__(self, hidden=256, blocks=2, n_freqs=6, scale_input=0.05937489, scale_output=0.14,
                 transform=None):
        super().__init__()
        self.hidden, self.blocks, self.n_freqs = hidden, blocks, n_freqs
        self.scale_input, self.scale_output = scale_input, scale_output
        # buffers, so that .to(device) moves them (the eager-GPU baseline of bench.py runs this module on "cuda")
        self.register_buffer("transform", None if transform is None else torch.as_tensor(np.asarray(transform), dtype=torch.float32),
                             persistent=False)
        self.register_buffer("dirs", torch.tensor(ICO_DIRS, dtype=torch.float32), persistent=False)
        E = 2 * 21 * n_freqs + 3
        sp = lambda i, o: torch.nn.Sequential(torch.nn.Linear(i, o), torch.nn.Softplus(beta=100))
        self.in_layer = sp(E, hidden)
        self.mid1 = torch.nn.Sequential(*[sp(hidden, hidden) for _ in range(blocks)])
        self.cat_layer = sp(hidden + E, hidden)
        self.mid2 = torch.nn.Sequential(*[sp(hidden, hidden) for _ in range(blocks)])
        self.out_alpha = torch.nn.Linear(hidden, 1)
        for m in self.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_normal_(m.weight)       # fc_map.py:58-60

    def embed(self, x):
        if self.transform is not None:                        # transform.py:287-304
            x = x @ self.transform[:3, :3].T + self.transform[:3, 3]
        x = x * self.scale_input
        freq = 2.0 ** torch.linspace(0, self.n_freqs - 1, self.n_freqs, device=x.device)
        proj = x @ self.dirs
        xb = (proj[..., None] * freq).reshape(*proj.shape[:-1], -1)
        return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * np.pi], -1))], -1)

    def forward(self, x, noise=None):
        e = self.embed(x)
        h = self.mid1(self.in_layer(e))
        h = self.mid2(self.cat_layer(torch.cat((h, e), -1)))
        raw = self.out_alpha(h)
        if noise is not None:
            raw = raw + noise[..., None]
        return (raw * self.scale_output).squeeze(-1)


def sample_step(depth, T_WC, normals, cam, sc, gen):
    """sample_pixels + get_batch_data + sample_along_rays (`sample.py:11-178`).  The draws come from the (CPU)
    generator `gen` and are moved to the data's device -- what the reference itself does for the surface offsets
    (sample.py:160-162); on "cuda" the rest of the chain then runs as eager device ops like upstream.
    gen None: draw exactly as the reference does -- torch's GLOBAL generators, pixel indices and stratified offsets on the data's
    device (sample.py:15-16,123), the surface offsets on the CPU (sample.py:160-162) -- so that a run seeded with
    torch.manual_seed consumes the same random stream as `HipTrainer(rng="torch")` (paired accuracy runs)."""
    F, H, W = depth.shape
    n = sc["n_rays"]
    dev = depth.device
    if gen is None:
        ih = torch.randint(0, H, (F * n,), device=dev)
        iw = torch.randint(0, W, (F * n,), device=dev)
    else:
        ih = torch.randint(0, H, (F * n,), generator=gen).to(dev)
        iw = torch.randint(0, W, (F * n,), generator=gen).to(dev)
    ib = torch.arange(F, device=dev).repeat_interleave(n)
    d = depth[ib, ih, iw]
    nm = normals[ib, ih, iw]
    ok = (d != 0) & ~torch.isnan(nm[:, 0])
    d, nm, ib, ih, iw = d[ok], nm[ok], ib[ok], ih[ok], iw[ok]
    Tm = T_WC[ib]
    dC = torch.stack(((iw.float() - cam["cx"]) / cam["fx"], (ih.float() - cam["cy"]) / cam["fy"],
                      torch.ones_like(d)), -1)
    dW = (Tm[:, :3, :3] * dC[:, None, :]).sum(-1)
    R = d.shape[0]
    maxd = d + sc["dist_behind_surf"]
    rng = (maxd - sc["min_depth"])[:, None]
    lim = torch.linspace(0, 1, sc["n_strat"] + 1, device=dev)[None, :].repeat(R, 1) * rng + sc["min_depth"]
    U = torch.rand(R, sc["n_strat"], device=dev) if gen is None else torch.rand(R, sc["n_strat"], generator=gen).to(dev)
    z = lim[:, :-1] + U * (rng / sc["n_strat"])
    off = torch.normal(torch.zeros(R, sc["n_surf"] - 1), 0.1, generator=gen).to(dev)
    near = torch.clamp(d[:, None] + off, torch.full((R, 1), sc["min_depth"], device=dev), maxd[:, None])
    z = torch.cat((d[:, None], near, z), 1)
    pc = Tm[:, None, :3, 3] + dW[:, None, :] * z[:, :, None]
    return dict(pc=pc, z=z, depth=d, dC=dC, dW=dW, normals=nm, ib=ib, ih=ih, iw=iw)


def loss_step(net, s, lc, noise_std, gen, noise=None):
    """`Trainer.sdf_eval_and_loss` with bounds_method "ray" (`trainer.py:768-868`).
    noise: optional pre-drawn, pre-scaled noise (parity tests)."""
    pc = s["pc"].clone().requires_grad_()
    if noise is None and noise_std is not None:
        noise = (torch.randn(pc.shape[:-1], device=pc.device) if gen is None else
                 torch.randn(pc.shape[:-1], generator=gen).to(pc.device)) * noise_std
    sdf = net(pc, noise)
    g = torch.autograd.grad(sdf, pc, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
    bounds = s["dC"].norm(dim=-1)[:, None] * (s["depth"][:, None] - s["z"])
    gvec = -s["dW"][:, None, :].repeat(1, s["z"].shape[1] - 1, 1)
    free = bounds > lc["trunc_distance"]
    fs = torch.max(torch.relu(sdf - bounds), torch.exp(-5.0 * sdf) - 1.0)
    mat = torch.where(free, fs, sdf - bounds)
    mat = mat.abs() if lc["loss_type"] == "L1" else mat.square()
    mat = torch.where(free, mat, mat * lc["trunc_weight"])
    cos = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)
    gl = torch.cat(((1 - cos(g[:, 0], s["normals"]))[:, None], 1 - cos(gvec, g[:, 1:])), 1)
    eik = (g.norm(2, dim=-1) - 1).abs()
    eik = torch.where(bounds < lc["eik_apply_dist"], torch.zeros_like(eik), eik) * lc["eik_weight"]
    tot = mat + lc["grad_weight"] * gl + eik
    return tot.mean(), dict(sdf_loss=mat.mean().item(), grad_loss=gl.mean().item(),
                            eikonal_loss=eik.mean().item()), tot


def frame_avg_step(tot, s, F, H, W):
    """`loss.frame_avg` (`loss.py:208-240`) incl. its dense [F,H,W] scatter images."""
    full = torch.zeros(F, H, W, device=tot.device)
    mask = torch.zeros(F, H, W, device=tot.device)
    full[s["ib"], s["ih"], s["iw"]] = tot.sum(-1).detach()
    mask[s["ib"], s["ih"], s["iw"]] = 1
    la = full.view(-1, 8, H // 8, 8, W // 8).sum(dim=(2, 4))
    ac = mask.view(-1, 8, H // 8, 8, W // 8).sum(dim=(2, 4))
    ac[ac == 0] = 1.0
    la = la / ac
    return la, la.sum(dim=(1, 2)) / 64


def train_step(net, opt, depth, T_WC, normals, cam, sc, lc, noise_std, gen):
    s = sample_step(depth, T_WC, normals, cam, sc, gen)
    total, losses, tot = loss_step(net, s, lc, noise_std, gen)
    _, fa = frame_avg_step(tot, s, depth.shape[0], cam["H"], cam["W"])
    total.backward()
    opt.step()
    for p in net.parameters():
        p.grad = None
    losses["total_loss"] = total.item()
    return losses, fa


class PortTrainer:
    """The reference `Trainer` as the drivers see it (train.py:86-136), on torch CPU and in-memory frames --
    TEST INFRASTRUCTURE: the accuracy CONTROL for the HIP path under the reference's own frame schedule
    (tests/accuracy_experiment.py --backend port --reference-schedule).  Method names, state fields and decision
    logic follow trainer.py:574-674,951-1016; the arithmetic is the op chain above."""

    def __init__(self, cfg, cam, transform, seed, virtual_step_ms, fps=30, device="cpu"):
        """virtual_step_ms None: the clock advances by the MEASURED (device-synchronised) step time, like upstream
        (metrics.py:13-38, trainer.py:1011-1013).  device "cuda": the same op chain as PyTorch-ROCm eager."""
        m, s, lo = cfg["model"], cfg["sample"], cfg["loss"]
        self.cam, self.fps, self.virtual_step_ms = cam, fps, virtual_step_ms
        self.device = torch.device(device)
        self.sc = dict(n_rays=s["n_rays"], n_strat=s["n_strat_samples"], n_surf=s["n_surf_samples"],
                       min_depth=s["depth_range"][0], dist_behind_surf=s["dist_behind_surf"])
        self.n_rays_is_kf = s["n_rays_is_kf"]
        self.lc = dict(trunc_distance=lo["trunc_distance"], loss_type=lo["loss_type"], trunc_weight=lo["trunc_weight"],
                       eik_apply_dist=lo["eik_apply_dist"], eik_weight=lo["eik_weight"], grad_weight=lo["grad_weight"])
        self.window_size, self.iters_per_kf, self.iters_per_frame = m["window_size"], m["iters_per_kf"], m["iters_per_frame"]
        self.noise_std, self.noise_kf, self.noise_frame = m["noise_std"], m["noise_kf"], m["noise_frame"]
        self.kf_dist_th, self.kf_pixel_ratio = m["kf_dist_th"], m["kf_pixel_ratio"]
        self.frac_time_perception = m["frac_time_perception"]
        self.net = PortNet(m["hidden_feature_size"], m["hidden_layers_block"], m["embedding"]["n_embed_funcs"] + 1,
                           m["embedding"]["scale_input"], m["scale_output"], transform).to(self.device)
        self.opt = torch.optim.AdamW(self.net.parameters(), lr=cfg["optimiser"]["lr"],
                                     weight_decay=cfg["optimiser"]["weight_decay"])
        self.gen = torch.Generator().manual_seed(seed)
        self.tot_step_time, self.steps_since_frame, self.optim_frames, self.last_is_keyframe = 0.0, 0, 0, False
        self.frame_id, self.depth, self.T, self.normals = [], None, None, None
        self.fal = torch.zeros(0, device=self.device)
        self.frozen = None
        self.n_steps_done = 0

    def get_latest_frame_id(self):
        return int(self.tot_step_time * self.fps)

    def add_frame(self, frame):
        """frame = (id, depth [H,W], T_WC [4,4], normals [H,W,3]) as torch tensors"""
        import copy
        if self.last_is_keyframe:
            self.frozen = copy.deepcopy(self.net)
        fid, d, T, n = frame
        d, T, n = d.to(self.device), T.to(self.device), n.to(self.device)
        replace = self.last_is_keyframe is False and len(self.frame_id) > 0
        if replace:
            self.frame_id[-1] = fid; self.depth[-1] = d; self.T[-1] = T; self.normals[-1] = n; self.fal[-1] = 0.0
        else:
            cat = lambda a, b: b[None] if a is None else torch.cat((a, b[None]))
            self.frame_id.append(fid)
            self.depth, self.T, self.normals = cat(self.depth, d), cat(self.T, T), cat(self.normals, n)
            self.fal = torch.cat((self.fal, torch.zeros(1, device=self.device)))
        self.steps_since_frame, self.last_is_keyframe = 0, False
        self.optim_frames, self.noise_std = self.iters_per_frame, self.noise_frame

    def is_keyframe(self):
        from .isdf_oracle import keyframe_ratio
        sc = dict(self.sc, n_rays=self.n_rays_is_kf, dist_behind_surf=0.8)
        s = sample_step(self.depth[-1:], self.T[-1:], self.normals[-1:], self.cam, sc, self.gen)
        with torch.no_grad():
            noise = None if self.noise_std is None else \
                (torch.randn(s["pc"].shape[:-1], generator=self.gen) * self.noise_std).to(self.device)
            sdf = self.frozen(s["pc"], noise)
        ratio, _ = keyframe_ratio(s["z"].cpu().numpy(), sdf.cpu().numpy(), s["depth"].cpu().numpy(), self.kf_dist_th)
        return ratio < self.kf_pixel_ratio

    def check_keyframe_latest(self):
        add_new_frame = False
        if self.last_is_keyframe:
            add_new_frame = True
        else:
            self.last_is_keyframe = bool(self.is_keyframe())
            if self.tot_step_time - self.frame_id[-2] / 30. > 5.:
                self.last_is_keyframe = True
            if self.last_is_keyframe:
                self.optim_frames, self.noise_std = self.iters_per_kf, self.noise_kf
            else:
                add_new_frame = True
        return add_new_frame

    def step(self):
        import time
        native = self.virtual_step_ms is None
        if native:                     # metrics.start_timing: device-synchronised wall clock
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
        K = len(self.frame_id)
        if K > self.window_size:       # select_keyframes, trainer.py:652-674
            p = (self.fal[:-2] / self.fal[:-2].sum()).cpu().numpy()
            idxs = [*np.random.choice(np.arange(0, K - 2), size=self.window_size - 2, replace=False, p=p), K - 2, K - 1]
        else:
            idxs = list(range(K))
        # quirk q4: normals come from the un-windowed batch with window-local indices
        losses, fa = train_step(self.net, self.opt, self.depth[idxs], self.T[idxs], self.normals[:len(idxs)], self.cam,
                                self.sc, self.lc, self.noise_std, self.gen)
        self.fal[idxs] = fa
        if native:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            step_ms = (time.perf_counter() - t0) * 1000.0
        else:
            step_ms = self.virtual_step_ms
        self.tot_step_time += (1 / self.frac_time_perception) * step_ms / 1000.0
        self.steps_since_frame += 1
        self.n_steps_done += 1
        return losses, step_ms
