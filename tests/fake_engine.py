"""CPU stand-in for `isdf_amd.engine.Engine` backed by the ORACLE -- TEST INFRASTRUCTURE for the host logic
(`isdf_amd.hot_path`: state ownership, window indirection, frame schedule, data-parallel protocol) on
hosts without a GPU.  Same method surface and buffer contract as the real engine (flat fp32 parameter /
moment buffers, capacity-sized sampler outputs with a device `n_valid`, SUMS in `reduce_buf`); the arithmetic
is `oracle/isdf_oracle.py`.  Never imported by the product."""
import numpy as np
import torch

import oracle.isdf_oracle as orc


def _as_index(ix):
    return [int(v) for v in ix] if isinstance(ix, (tuple, list)) else ix.long()


class FakeEngine:
    def __init__(self, net, device="cpu"):
        self.net, self.device = net, torch.device("cpu")
        shapes = net.param_shapes()
        self.slices, off = {}, 0
        for k, shp in shapes:
            self.slices[k] = (off, shp)
            off += int(np.prod(shp))
        self.n_params = off
        self.params = torch.zeros(off)
        self.exp_avg, self.exp_avg_sq = torch.zeros(off), torch.zeros(off)
        self.reduce_buf, self.opt_step = None, 0
        self.reduce_extra, self.reduce_floats = 0, 0     # caller-owned tail of the all-reduce message (data parallel)
        self.calls = []

    # ---- parameters
    def param_view(self, key):
        off, shp = self.slices[key]
        return self.params[off:off + int(np.prod(shp))].view(*shp)

    def load_params(self, state):
        for k, (off, shp) in self.slices.items():
            v = torch.as_tensor(np.asarray(state[k].detach().cpu() if torch.is_tensor(state[k]) else state[k]),
                                dtype=torch.float32).reshape(-1)
            self.params[off:off + v.numel()].copy_(v)

    def pack(self):
        pass

    def _np_params(self):
        return {k: self.param_view(k).numpy() for k in self.slices}

    def _cfg(self):
        n = self.net
        return orc.NetCfg(n.hidden, n.blocks, n.n_freqs, n.scale_input, n.scale_output,
                          None if n.transform is None else np.asarray(n.transform, np.float32))

    # ---- sampler (capacity-sized outputs, rows >= n_valid undefined)
    def sample(self, depth_batch, T_WC_batch, normal_batch, frame_idx, normal_idx, sc, draws=None, seed=0, offset=0,
               want_T=False, reuse=False):
        if isinstance(frame_idx, (tuple, list)):       # the step loop passes the window inline (engine.Engine.sample)
            frame_idx = torch.as_tensor(list(frame_idx), dtype=torch.int32)
            normal_idx = None if normal_idx is None else torch.as_tensor(list(normal_idx), dtype=torch.int32)
        F = int(frame_idx.numel())
        R0, S = F * sc.n_rays, sc.S
        fi = frame_idx.long().numpy()
        depth = depth_batch.numpy()[fi]
        T = T_WC_batch.numpy()[fi]
        normal = None if normal_batch is None else normal_batch.numpy()[normal_idx.long().numpy()]
        if draws is None:   # stands in for the in-kernel Philox stream
            rng = np.random.RandomState((int(seed) * 1000003 + int(offset)) % (2 ** 31))
            draws = dict(indices_h=torch.from_numpy(rng.randint(0, sc.H, R0)), indices_w=torch.from_numpy(rng.randint(0, sc.W, R0)),
                         U=torch.from_numpy(rng.uniform(size=(R0, sc.n_strat)).astype(np.float32)),
                         N_off=torch.from_numpy((0.1 * rng.standard_normal((R0, max(sc.n_surf - 1, 0)))).astype(np.float32)))
        dirs_C = orc.ray_dirs_C(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
        ib = orc.sample_pixels_indices_b(sc.n_rays, F)
        bd = orc.get_batch_data(depth, T, dirs_C, ib, draws["indices_h"].numpy(), draws["indices_w"].numpy(), normal)
        R = bd["depth_sample"].shape[0]
        max_depth = bd["depth_sample"] + np.float32(sc.dist_behind_surf)
        pc, z = orc.sample_along_rays(bd["T_WC_sample"], sc.min_depth, max_depth, sc.n_strat, sc.n_surf,
                                      bd["dirs_C_sample"], bd["depth_sample"], draws["U"].numpy()[:R],
                                      draws["N_off"].numpy()[:R])

        def pad(a, fill=0):
            out = np.full((R0,) + a.shape[1:], fill, a.dtype)
            out[:R] = a
            return torch.from_numpy(out)
        _, dirs_W = orc.origin_dirs_W(bd["T_WC_sample"], bd["dirs_C_sample"])
        out = dict(n_valid=torch.tensor([R], dtype=torch.int32), indices_b=pad(bd["indices_b"]), indices_h=pad(bd["indices_h"]),
                   indices_w=pad(bd["indices_w"]), depth_sample=pad(bd["depth_sample"]), dirs_C_sample=pad(bd["dirs_C_sample"]),
                   norm_sample=None if normal is None else pad(bd["norm_sample"]),
                   T_WC_sample=pad(bd["T_WC_sample"]) if want_T else None, dirs_W_sample=pad(dirs_W.astype(np.float32)),
                   z_vals=pad(z), pc=pad(pc), max_rays=R0, S=S, n_frames=F)
        out["_T_WC_sample"] = bd["T_WC_sample"]
        self.calls.append("sample")
        return out

    # ---- training step: reduce_buf = [grad SUMS | loss sums(8) | block_loss | block_cnt]
    def train_step(self, smp, lc, sc, noise=None, debug=False, prof_events=None, noise_std=0.0, noise_seed=0,
                   noise_offset=0, optim=None, surf_group=None, extra_slot=0, extra_value=0.0):
        R = int(smp["n_valid"].item())
        F, S = smp["n_frames"], smp["S"]
        if noise is None and noise_std:
            rng = np.random.RandomState((int(noise_seed) * 7919 + int(noise_offset)) % (2 ** 31))
            noise = torch.from_numpy((noise_std * rng.standard_normal((R, S))).astype(np.float32))
        lco = orc.LossCfg(lc.bounds_method, lc.loss_type, lc.trunc_weight, lc.trunc_distance, lc.eik_weight,
                          lc.eik_apply_dist, lc.grad_weight, lc.orien_loss)
        T = smp.get("_T_WC_sample")
        if T is None:       # rebuilt from world directions: only rotation*dirs_C and the translation matter
            raise RuntimeError("FakeEngine.train_step needs a sample produced by FakeEngine.sample")
        terms, grads = orc.loss_and_grads(self._np_params(), self._cfg(), lco, smp["pc"][:R].numpy(), smp["z_vals"][:R].numpy(),
                                          smp["depth_sample"][:R].numpy(), smp["dirs_C_sample"][:R].numpy(), T,
                                          None if smp["norm_sample"] is None else smp["norm_sample"][:R].numpy(),
                                          noise=None if noise is None else noise[:R].numpy())
        N = R * S
        nred = self.n_params + 8 + 2 * F * 64
        self.reduce_floats = nred
        self.reduce_buf = torch.zeros(nred + self.reduce_extra)
        if self.reduce_extra:
            self.reduce_buf[nred + extra_slot] = extra_value     # caller-owned tail: one slot per rank
        self._F = F
        for k, (off, shp) in self.slices.items():
            self.reduce_buf[off:off + int(np.prod(shp))] = torch.from_numpy((grads[k].astype(np.float64) * N).astype(np.float32).reshape(-1))
        ls = self.reduce_buf[self.n_params:self.n_params + 8]
        ls[0], ls[3], ls[4] = float(terms["sdf_loss"]) * N, float(terms["total_loss"]) * N, float(N)
        ls[1] = float(terms.get("grad_loss", 0.0)) * N if lc.grad_weight != 0 else 0.0
        ls[2] = float(terms.get("eikonal_loss", 0.0)) * N if lc.eik_weight != 0 else 0.0
        la, fa = orc.frame_avg(terms["tot_loss_mat"], smp["indices_b"][:R].numpy(), smp["indices_h"][:R].numpy(),
                               smp["indices_w"][:R].numpy(), F, sc.H, sc.W)
        self._la, self._fa = torch.from_numpy(la.astype(np.float32)), torch.from_numpy(fa.astype(np.float32))
        # block bins as SUMS / counts, like the kernels write them (this rank's rays; last ray on a pixel wins)
        hb, wb = sc.H // 8, sc.W // 8
        ray = terms["tot_loss_mat"].sum(-1)
        bl, bc = np.zeros((F, 8, 8), np.float32), np.zeros((F, 8, 8), np.float32)
        pix = list(zip(smp["indices_b"][:R].tolist(), smp["indices_h"][:R].tolist(), smp["indices_w"][:R].tolist()))
        last = {p: r for r, p in enumerate(pix)}
        for r, (b, h, w) in enumerate(pix):
            if last[(b, h, w)] == r:
                bl[b, h // hb, w // wb] += ray[r]; bc[b, h // hb, w // wb] += 1
        o = self.n_params + 8
        self.reduce_buf[o:o + 64 * F] = torch.from_numpy(bl.reshape(-1))
        self.reduce_buf[o + 64 * F:o + 128 * F] = torch.from_numpy(bc.reshape(-1))
        self.calls.append("train_step")
        dbg = {}
        if optim is not None:
            if optim.get("frame_avg_out") is not None:
                optim["frame_avg_out"][_as_index(optim["frame_avg_index"])] = self._fa
                dbg["loss_approx"] = self._la
            self.adamw(lr=optim.get("lr", 0.0013), weight_decay=optim.get("weight_decay", 0.012),
                       betas=optim.get("betas", (0.9, 0.999)), eps=optim.get("eps", 1e-8))
        return dbg

    def train_step_finish(self, n_frames, optim):
        """isdf_train_step_finish: AdamW on the (all-reduced) sums + frame averages from the (all-reduced) bins"""
        o = self.n_params + 8
        bl = self.reduce_buf[o:o + 64 * n_frames].view(n_frames, 64)
        bc = self.reduce_buf[o + 64 * n_frames:o + 128 * n_frames].view(n_frames, 64).clone()
        bc[bc == 0] = 1.0
        la = bl / bc
        if optim.get("frame_avg_out") is not None:
            optim["frame_avg_out"][_as_index(optim["frame_avg_index"])] = la.sum(1) / 64.0
        self.adamw(lr=optim.get("lr", 0.0013), weight_decay=optim.get("weight_decay", 0.012),
                   betas=optim.get("betas", (0.9, 0.999)), eps=optim.get("eps", 1e-8))
        self.calls.append("train_step_finish")
        return {"loss_approx": la.view(n_frames, 8, 8)}

    def loss_sums(self):
        return self.reduce_buf[self.n_params:self.n_params + 8]

    def grad_view(self, key):
        off, shp = self.slices[key]
        return self.reduce_buf[off:off + int(np.prod(shp))].view(*shp)

    def frame_avg(self, n_frames, out=None, index=None):
        if out is None:
            return self._la, self._fa
        out[index.long()] = self._fa
        return self._la, out

    def adamw(self, lr=0.0013, weight_decay=0.012, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, use_device_count=True):
        self.opt_step += 1
        cnt = float(self.reduce_buf[self.n_params + 4]) if use_device_count else 1.0
        g = self.reduce_buf[:self.n_params] * (grad_scale / cnt)
        p, m, v = self.params, self.exp_avg, self.exp_avg_sq
        p.mul_(1 - lr * weight_decay)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** self.opt_step, 1 - betas[1] ** self.opt_step
        p.addcdiv_(m, v.sqrt() / np.sqrt(bc2) + eps, value=-lr / bc1)
        self.calls.append("adamw")

    # ---- inference / keyframe test / ingest
    def sdf_eval(self, pts, noise=None, want_grad=False):
        x = pts.reshape(-1, 3).detach().numpy().astype(np.float32)
        nz = None if noise is None else noise.reshape(-1).numpy()
        if want_grad:
            sdf, g = orc.sdf_forward_grad(self._np_params(), self._cfg(), x, noise=nz)
            return torch.from_numpy(sdf.astype(np.float32)).view(pts.shape[:-1]), torch.from_numpy(g.astype(np.float32)).view(*pts.shape[:-1], 3)
        sdf = orc.sdf_forward(self._np_params(), self._cfg(), x, noise=nz)
        return torch.from_numpy(sdf.astype(np.float32)).view(pts.shape[:-1])

    def render_depth(self, z_vals, sdf, depth_sample=None, kf_dist_th=0.1, n_valid=None):
        R = z_vals.shape[0] if n_valid is None else int(n_valid.item())
        z, s = z_vals[:R].numpy(), sdf[:R].numpy()
        if depth_sample is None:
            order = np.argsort(z, axis=1, kind="stable")
            view = orc.sdf_render_depth(np.take_along_axis(z, order, 1), np.take_along_axis(s, order, 1))
            return torch.from_numpy(view), torch.zeros(1, dtype=torch.int32)
        ratio, view = orc.keyframe_ratio(z, s, depth_sample[:R].numpy(), kf_dist_th)
        return torch.from_numpy(view), torch.tensor([int(round(ratio * R))], dtype=torch.int32)

    def estimate_normals(self, depth, sc):
        pc = orc.pointcloud_from_depth(depth.numpy(), sc.fx, sc.fy, sc.cx, sc.cy)
        return torch.from_numpy(orc.estimate_pointcloud_normals(pc).astype(np.float32))
