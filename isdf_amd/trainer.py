"""Host mirror of the reference `Trainer`'s training hot path
(`isdf/modules/trainer.py`): same method names, argument meaning, return
contracts and side effects for

    step()               trainer.py:951-1016
    sample_points()      trainer.py:683-766
    sdf_eval_and_loss()  trainer.py:768-868
    select_keyframes()   trainer.py:652-674
    add_frame()/add_data trainer.py:564-582
    FrameData            isdf/datasets/data_util.py:11-102

with every numeric operation executed by the HIP kernels behind the C ABI
(include/isdf_hip.h).  Config parsing follows `Trainer.set_params`
(trainer.py:157-333) on the reference's JSON schema.  What the reference does
around this path (datasets, visualisation, evaluation, ROS) is out of scope
 and is not mirrored.  There is no CPU fallback.
"""
import json
import time

import numpy as np
import torch

from . import _ffi, dp
from .engine import LossConfig, SampleConfig
from .modules import PositionalEncodingHIP, SDFMapHIP


class FrameData:
    """Keyframe store (`data_util.FrameData`, isdf/datasets/data_util.py:11-81): same fields and the same
    `add_frame_data(data, replace)` contract (append, or overwrite the last slot when the previous frame was
    not promoted to a keyframe), but the device batches live in pre-allocated buffers that grow
    geometrically instead of being re-built with `torch.cat` on every frame (data_util.py:84-102 copies the
    whole keyframe set -- ~13 MB per keyframe at 680x1200 -- each time a frame arrives).
    The public attributes stay plain tensors: views of the first len(self) rows of the backing buffers."""

    def __init__(self, frame_id=None, depth_batch=None, T_WC_batch=None, normal_batch=None,
                 frame_avg_losses=None, im_batch=None):
        self.frame_id = frame_id
        self.im_batch = im_batch
        self.depth_batch = depth_batch
        self.T_WC_batch = T_WC_batch
        self.normal_batch = normal_batch
        self.frame_avg_losses = frame_avg_losses
        self._back = {}          # field name -> backing tensor (capacity >= len)

    def __len__(self):
        return 0 if self.frame_id is None else len(self.frame_id)

    def __deepcopy__(self, memo):   # snapshots carry only the live rows
        import copy
        out = FrameData()
        for k in ("frame_id", "im_batch", "depth_batch", "T_WC_batch", "normal_batch", "frame_avg_losses"):
            v = getattr(self, k)
            setattr(out, k, None if v is None else (v.copy() if isinstance(v, np.ndarray) else v.clone()))
        return out

    def _expand(self, name, batch, data, replace):
        if data is None:
            return batch
        if batch is None:
            if isinstance(data, np.ndarray):
                return data
            batch = data[:0]
        elif replace:
            batch[-1] = data[0]
            return batch
        if isinstance(data, np.ndarray):     # frame ids: a few bytes
            return np.concatenate((batch, data))
        n, k = batch.shape[0], data.shape[0]
        back = getattr(self, "_back", None)
        if back is None:
            back = self._back = {}
        buf = back.get(name)
        if (buf is None or buf.data_ptr() != batch.data_ptr() or buf.shape[0] < n + k or buf.dtype != data.dtype
                or buf.device != data.device or buf.shape[1:] != data.shape[1:]):
            cap = max(2 * (n + k), 8)        # geometric growth: amortised O(1) copies per keyframe
            buf = torch.empty((cap,) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
            if n:
                buf[:n] = batch
            back[name] = buf
        buf[n:n + k] = data
        return buf[:n + k]

    def add_frame_data(self, data, replace):
        """data_util.py:45-78"""
        n_new = len(data)
        self.frame_id = self._expand("frame_id", self.frame_id, data.frame_id, replace)
        self.im_batch = self._expand("im_batch", self.im_batch, data.im_batch, replace)
        self.depth_batch = self._expand("depth_batch", self.depth_batch, data.depth_batch, replace)
        self.T_WC_batch = self._expand("T_WC_batch", self.T_WC_batch, data.T_WC_batch, replace)
        self.normal_batch = self._expand("normal_batch", self.normal_batch, data.normal_batch, replace)
        empty = torch.zeros([n_new], device=data.depth_batch.device)
        self.frame_avg_losses = self._expand("frame_avg_losses", self.frame_avg_losses, empty, replace)


class FlatAdamW:
    """Facade with the `torch.optim.AdamW` surface the drivers touch
    (`state_dict()`, `param_groups`, `step()`; trainer.py:435-439, train.py:213);
    the update itself is the fused flat HIP kernel."""

    def __init__(self, sdf_map, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        self.sdf_map = sdf_map
        self.param_groups = [dict(params=list(sdf_map.parameters()), lr=lr, betas=betas, eps=eps,
                                  weight_decay=weight_decay, amsgrad=False)]

    def step(self):
        g = self.param_groups[0]
        self.sdf_map.engine.adamw(lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"])

    def state_dict(self):
        eng = self.sdf_map.engine
        state = {}
        if eng.opt_step > 0:
            for i, (k, (off, shp)) in enumerate(eng.slices.items()):
                n = int(np.prod(shp))
                state[i] = dict(step=torch.tensor(float(eng.opt_step)),
                                exp_avg=eng.exp_avg[off:off + n].view(*shp).clone(),
                                exp_avg_sq=eng.exp_avg_sq[off:off + n].view(*shp).clone())
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(eng.slices)))
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        eng = self.sdf_map.engine
        for i, (k, (off, shp)) in enumerate(eng.slices.items()):
            if i in sd["state"]:
                n = int(np.prod(shp))
                eng.exp_avg[off:off + n].copy_(sd["state"][i]["exp_avg"].reshape(-1))
                eng.exp_avg_sq[off:off + n].copy_(sd["state"][i]["exp_avg_sq"].reshape(-1))
                eng.opt_step = int(float(sd["state"][i]["step"]))


class LazyLosses(dict):
    """`losses` of `Trainer.step`: keys sdf_loss / grad_loss / eikonal_loss (floats)
    and total_loss (0-d tensor; callers use '{:.6f}'.format and .item(),
    train.py:138,215).  Backed by ONE 5-float device->host copy made on first use
    instead of the reference's three `.item()` syncs (loss.py:187-200)."""

    def __init__(self, loss_sums_dev, has_grad, has_eik):
        super().__init__()
        self._dev, self._done = loss_sums_dev, False
        self._keys = ["sdf_loss"] + (["grad_loss"] if has_grad else []) + (["eikonal_loss"] if has_eik else []) \
            + ["total_loss"]

    def _fill(self):
        if self._done:
            return
        ls = self._dev.detach().cpu()
        n = max(float(ls[_ffi.LS_COUNT]), 1.0)
        idx = dict(sdf_loss=_ffi.LS_SDF, grad_loss=_ffi.LS_GRAD, eikonal_loss=_ffi.LS_EIK)
        for k in self._keys[:-1]:
            dict.__setitem__(self, k, float(ls[idx[k]]) / n)
        dict.__setitem__(self, "total_loss", (self._dev[_ffi.LS_TOTAL] / self._dev[_ffi.LS_COUNT]).detach())
        self._done = True

    def __getitem__(self, k):
        self._fill(); return dict.__getitem__(self, k)

    def keys(self):
        self._fill(); return dict.keys(self)

    def items(self):
        self._fill(); return dict.items(self)

    def __iter__(self):
        self._fill(); return dict.__iter__(self)

    def __contains__(self, k):
        return k in self._keys

    def __len__(self):
        return len(self._keys)


class HipTrainer:
    def __init__(self, device, config, incremental=True, inv_bounds_transform=None, rng="philox",
                 seed=1, dist_group=None, fix_normal_window=False, fwd_operand="fp16"):
        """config: path to / dict with the reference's JSON schema (replicaCAD.json).
        rng: "philox" (in-kernel, no host sync) or "torch" (draw with torch in the
        reference's order and shapes -- parity mode, one host sync per step)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _ffi.IsdfError("HipTrainer needs a HIP device; the reference's own Python path is the CPU path")
        if isinstance(config, str):
            with open(config) as f:
                config = json.load(f)
        self.config = config
        self.incremental = incremental
        self.rng, self.seed, self.dist_group = rng, int(seed), dist_group
        self.fix_normal_window = fix_normal_window
        self.inv_bounds_transform = inv_bounds_transform
        self.tot_step_time = 0.0
        self.steps_since_frame = 0
        self.last_is_keyframe = False
        self.optim_frames = 0
        self.active_idxs = None
        self.active_pixels = None
        self._step_count = 0
        self.frames = FrameData()
        self.set_params()
        self.load_networks(fwd_operand)

    # ---- trainer.py:157-333 (hot-path subset) ---------------------------------
    def set_params(self):
        c = self.config
        cam = c["dataset"]["camera"]
        self.fx, self.fy, self.cx, self.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        self.H, self.W = cam["h"], cam["w"]
        m = c["model"]
        self.scale_output = m["scale_output"]
        self.noise_std, self.noise_kf, self.noise_frame = m["noise_std"], m["noise_kf"], m["noise_frame"]
        self.window_size = m["window_size"]
        self.hidden_layers_block, self.hidden_feature_size = m["hidden_layers_block"], m["hidden_feature_size"]
        self.frac_time_perception = m["frac_time_perception"]
        self.iters_per_kf, self.iters_per_frame = m["iters_per_kf"], m["iters_per_frame"]
        self.kf_dist_th, self.kf_pixel_ratio = m.get("kf_dist_th", 0.1), m.get("kf_pixel_ratio", 0.65)
        self.scale_input = m["embedding"]["scale_input"]
        self.n_embed_funcs = m["embedding"]["n_embed_funcs"]
        lo = c["loss"]
        self.bounds_method = lo["bounds_method"]
        assert self.bounds_method in ["ray", "normal", "pc"]
        self.loss_type = lo["loss_type"]
        assert self.loss_type in ["L1", "L2"]
        self.trunc_weight, self.trunc_distance = lo["trunc_weight"], lo["trunc_distance"]
        self.eik_weight, self.eik_apply_dist = lo["eik_weight"], lo["eik_apply_dist"]
        self.grad_weight, self.orien_loss = lo["grad_weight"], bool(lo["orien_loss"])
        self.do_normal = self.bounds_method == "normal" or self.grad_weight != 0
        self.learning_rate, self.weight_decay = c["optimiser"]["lr"], c["optimiser"]["weight_decay"]
        s = c["sample"]
        self.min_depth, self.max_depth = s["depth_range"]
        self.dist_behind_surf, self.n_rays = s["dist_behind_surf"], s["n_rays"]
        self.n_rays_is_kf = s.get("n_rays_is_kf", 400)
        self.n_strat_samples, self.n_surf_samples = s["n_strat_samples"], s["n_surf_samples"]
        self.loss_approx_factor = 8

    def load_networks(self, fwd_operand="fp16"):
        """trainer.py:419-439"""
        pe = PositionalEncodingHIP(min_deg=0, max_deg=self.n_embed_funcs, scale=self.scale_input,
                                   transform=self.inv_bounds_transform)
        self.sdf_map = SDFMapHIP(pe, hidden_size=self.hidden_feature_size,
                                 hidden_layers_block=self.hidden_layers_block, scale_output=self.scale_output,
                                 device=self.device, fwd_operand=fwd_operand)
        self.optimiser = FlatAdamW(self.sdf_map, lr=self.learning_rate, weight_decay=self.weight_decay)
        self.engine = self.sdf_map.engine

    def _loss_cfg(self):
        return LossConfig(self.bounds_method, self.loss_type, self.trunc_weight, self.trunc_distance,
                          self.eik_weight, self.eik_apply_dist, self.grad_weight, self.orien_loss)

    def _sample_cfg(self, n_rays=None, dist_behind_surf=None, n_strat=None, n_surf=None):
        return SampleConfig(n_rays=self.n_rays if n_rays is None else n_rays,
                            n_strat=self.n_strat_samples if n_strat is None else n_strat,
                            n_surf=self.n_surf_samples if n_surf is None else n_surf,
                            min_depth=self.min_depth,
                            dist_behind_surf=self.dist_behind_surf if dist_behind_surf is None else dist_behind_surf,
                            H=self.H, W=self.W, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy)

    # ---- frames (trainer.py:564-582) ----------------------------------------------
    def add_data(self, data, replace=False):
        replace = self.last_is_keyframe is False
        self.frames.add_frame_data(data, replace)

    def add_frame(self, frame_data):
        if self.last_is_keyframe:
            import copy
            self.frozen_sdf_map = copy.deepcopy(self.sdf_map)
        self.add_data(frame_data)
        self.steps_since_frame = 0
        self.last_is_keyframe = False
        self.optim_frames = self.iters_per_frame
        self.noise_std = self.noise_frame

    # ---- per-frame ingest (trainer.py:530-562) -----------------------------------------
    def make_frame(self, frame_id, depth, T_WC, im=None):
        """`Trainer.get_data` for one frame already in memory: device tensors + normals from
        the HIP stencil kernel (reference: pointcloud_from_depth_torch + estimate_pointcloud_normals,
        trainer.py:553-557).  depth [H,W] metres (0 = invalid), T_WC [4,4]."""
        depth = torch.as_tensor(depth, dtype=torch.float32).to(self.device)[None, ...]
        T = torch.as_tensor(T_WC, dtype=torch.float32).to(self.device)[None, ...]
        normals = None
        if self.do_normal:
            normals = self.engine.estimate_normals(depth[0], self._sample_cfg())[None, ...]
        return FrameData(frame_id=np.array([frame_id]), depth_batch=depth, T_WC_batch=T, normal_batch=normals,
                         im_batch=im)

    # ---- keyframe test (trainer.py:586-650) ------------------------------------------------
    def is_keyframe(self, T_WC, depth_gt):
        sample_pts = self.sample_points(depth_gt, T_WC, n_rays=self.n_rays_is_kf, dist_behind_surf=0.8)
        s = sample_pts["_raw"]
        pc = s["pc"]
        noise = None
        if self.noise_std is not None:
            noise = torch.randn(pc.shape[:-1], device=self.device) * self.noise_std
        sdf = self.frozen_sdf_map.engine.sdf_eval(pc, noise=noise)           # frozen net, no grad (trainer.py:594-595)
        view, below = self.engine.render_depth(s["z_vals"], sdf, s["depth_sample"], self.kf_dist_th,
                                               n_valid=s["n_valid"])
        n = int(s["n_valid"].item())
        below_th_prop = float(below.item()) / max(n, 1)
        is_keyframe = below_th_prop < self.kf_pixel_ratio
        print("Proportion of loss below threshold", below_th_prop, "for KF should be less than",
              self.kf_pixel_ratio, " ---> is keyframe:", is_keyframe)
        return is_keyframe

    def check_keyframe_latest(self):
        """returns whether or not to add a new frame (trainer.py:622-650)."""
        add_new_frame = False
        if self.last_is_keyframe:
            add_new_frame = True
        else:
            T_WC = self.frames.T_WC_batch[-1].unsqueeze(0)
            depth_gt = self.frames.depth_batch[-1].unsqueeze(0)
            self.last_is_keyframe = self.is_keyframe(T_WC, depth_gt)
            time_since_kf = self.tot_step_time - self.frames.frame_id[-2] / 30.
            if time_since_kf > 5.:
                print("More than 5 seconds since last kf, so add new")
                self.last_is_keyframe = True
            if self.last_is_keyframe:
                self.optim_frames = self.iters_per_kf
                self.noise_std = self.noise_kf
            else:
                add_new_frame = True
        return add_new_frame

    def select_keyframes(self):
        """trainer.py:652-674: last two keyframes + (window-2) drawn without
        replacement with p ~ frame_avg_losses (numpy global RNG, as the reference)."""
        n_frames = len(self.frames)
        limit = n_frames - 2
        denom = self.frames.frame_avg_losses[:-2].sum()
        loss_dist = (self.frames.frame_avg_losses[:-2] / denom).cpu().numpy()
        rand_ints = np.random.choice(np.arange(0, limit), size=self.window_size - 2, replace=False, p=loss_dist)
        last = n_frames - 1
        return [*rand_ints, last - 1, last]

    # ---- checkpoint / resume ---------------------------------------------
    def state_dict(self):
        """Everything needed for a true resume.  The reference saves only model + optimiser
        (train.py:207-219) and restores only the model (trainer.py:441-444); keyframes, RNG
        position and the virtual clock are lost there."""
        fr = self.frames
        return {
            "model_state_dict": {k: v.detach().clone() for k, v in self.sdf_map.state_dict().items()},
            "optimizer_state_dict": self.optimiser.state_dict(),
            "frames": {k: (None if getattr(fr, k) is None else
                           (getattr(fr, k).copy() if isinstance(getattr(fr, k), np.ndarray) else getattr(fr, k).clone()))
                       for k in ("frame_id", "depth_batch", "T_WC_batch", "normal_batch", "frame_avg_losses")},
            "clock": dict(tot_step_time=self.tot_step_time, steps_since_frame=self.steps_since_frame,
                          last_is_keyframe=self.last_is_keyframe, optim_frames=self.optim_frames,
                          noise_std=self.noise_std, step_count=self._step_count),
            "rng": dict(draw_count=getattr(self, "_draw_count", 0), noise_count=getattr(self, "_noise_count", 0),
                        seed=self.seed, numpy=np.random.get_state(), torch=torch.get_rng_state(),
                        torch_cuda=torch.cuda.get_rng_state(self.device)),
        }

    def load_state_dict(self, sd):
        self.sdf_map.load_state_dict(sd["model_state_dict"])
        self.optimiser.load_state_dict(sd["optimizer_state_dict"])
        f = sd["frames"]
        self.frames = FrameData(frame_id=f["frame_id"], depth_batch=f["depth_batch"], T_WC_batch=f["T_WC_batch"],
                                normal_batch=f["normal_batch"], frame_avg_losses=f["frame_avg_losses"])
        c = sd["clock"]
        self.tot_step_time, self.steps_since_frame = c["tot_step_time"], c["steps_since_frame"]
        self.last_is_keyframe, self.optim_frames = c["last_is_keyframe"], c["optim_frames"]
        self.noise_std, self._step_count = c["noise_std"], c["step_count"]
        r = sd["rng"]
        self._draw_count, self._noise_count, self.seed = r["draw_count"], r["noise_count"], r["seed"]
        np.random.set_state(r["numpy"]); torch.set_rng_state(r["torch"])
        torch.cuda.set_rng_state(r["torch_cuda"], self.device)

    # ---- sampling (trainer.py:683-766) ---------------------------------------------
    def _draws_torch(self, F, sc, n_valid_fn):
        """Reference draw order/shapes/devices: randint(h), randint(w) on the
        training device, rand(R, n_strat) on the device, normal(0, 0.1, (R, n_surf-1))
        on the CPU generator (sample.py:15-16,123,160-162)."""
        total = sc.n_rays * F
        ih = torch.randint(0, sc.H, (total,), device=self.device)
        iw = torch.randint(0, sc.W, (total,), device=self.device)
        R = n_valid_fn(ih, iw)
        U = torch.rand(R, sc.n_strat, device=self.device)
        N_off = torch.normal(torch.zeros(R, max(sc.n_surf - 1, 0)), 0.1).to(self.device)
        return dict(indices_h=ih, indices_w=iw, U=U, N_off=N_off)

    def _sample(self, depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc):
        eng = self.engine
        if self.rng == "torch":
            def n_valid(ih, iw):   # the reference learns R from its boolean-mask compaction (a sync)
                ib = torch.arange(frame_idx.numel(), device=self.device).repeat_interleave(sc.n_rays)
                d = depth_batch[frame_idx.long()[ib], ih, iw]
                ok = d != 0
                if norm_batch is not None:
                    ok &= ~torch.isnan(norm_batch[normal_idx.long()[ib], ih, iw, 0])
                return int(ok.sum().item())
            draws = self._draws_torch(frame_idx.numel(), sc, n_valid)
            return eng.sample(depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc, draws=draws,
                              want_T=True)
        rank = 0 if self.dist_group is None else torch.distributed.get_rank(self.dist_group)
        self._draw_count = getattr(self, "_draw_count", 0) + 1
        return eng.sample(depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc,
                          seed=dp.rank_seed(self.seed, rank), offset=self._draw_count, want_T=True)

    def sample_points(self, depth_batch, T_WC_batch, norm_batch=None, active_loss_approx=None, n_rays=None,
                      dist_behind_surf=None, n_strat_samples=None, n_surf_samples=None, _idx=None):
        if active_loss_approx is not None:
            raise Exception('Active sampling not currently supported.')
        sc = self._sample_cfg(n_rays, dist_behind_surf, n_strat_samples, n_surf_samples)
        if _idx is None:
            ar = torch.arange(depth_batch.shape[0], dtype=torch.int32, device=self.device)
            frame_idx, normal_idx = ar, ar
        else:
            frame_idx, normal_idx = _idx
        s = self._sample(depth_batch.contiguous(), T_WC_batch.contiguous(),
                         None if norm_batch is None else norm_batch.contiguous(), frame_idx, normal_idx, sc)
        R = int(s["n_valid"].item()) if self.rng == "torch" else None
        cut = (lambda t: t) if R is None else (lambda t: None if t is None else t[:R])
        out = {
            "depth_batch": depth_batch, "pc": cut(s["pc"]), "z_vals": cut(s["z_vals"]),
            "indices_b": cut(s["indices_b"]), "indices_h": cut(s["indices_h"]), "indices_w": cut(s["indices_w"]),
            "dirs_C_sample": cut(s["dirs_C_sample"]), "depth_sample": cut(s["depth_sample"]),
            "T_WC_sample": cut(s["T_WC_sample"]), "norm_sample": cut(s["norm_sample"]),
            "binary_masks": None,   # the dense [F,H,W] mask image is never materialised (sample.py:58-61)
            "_raw": s, "_sc": sc,
        }
        return out

    # ---- loss + backward (trainer.py:768-868, 981) -----------------------------------
    def sdf_eval_and_loss(self, sample, do_avg_loss=True, fused_optim=False, frame_avg_dst=None, _want_total=True):
        """fused_optim: also apply the optimiser step inside the same native call (single-GPU fast path,
        isdf_train_step_adamw); the caller must then NOT call self.optimiser.step().
        frame_avg_dst = (store, index): write frame f's average loss to store[index[f]] inside the native call
        (what trainer.py:979 does with the returned vector); frame_avg_loss is then returned as None."""
        s, sc = sample["_raw"], sample["_sc"]
        noise = None
        if self.noise_std is not None:   # fc_map.py:106-108 (drawn even for 0)
            if self.rng == "torch":
                R = sample["pc"].shape[0]
                noise = torch.randn(R, sc.S, device=self.device) * self.noise_std
        kw = {}
        if noise is None and self.noise_std is not None:   # philox mode: noise drawn inside the kernel
            self._noise_count = getattr(self, "_noise_count", 0) + 1
            rank = 0 if self.dist_group is None else torch.distributed.get_rank(self.dist_group)
            kw = dict(noise_std=self.noise_std, noise_seed=dp.rank_seed(self.seed, rank),
                      noise_offset=self._noise_count)
        if fused_optim:
            if self.dist_group is not None:
                raise ValueError("fused_optim is the single-GPU path: the gradient all-reduce sits before the update")
            g = self.optimiser.param_groups[0]
            kw["optim"] = dict(lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"])
            if do_avg_loss and frame_avg_dst is not None:   # frames.frame_avg_losses[idxs] = ... inside the launch
                kw["optim"].update(frame_avg_out=frame_avg_dst[0], frame_avg_index=frame_avg_dst[1])
        if self.dist_group is not None and self.bounds_method == "pc":
            kw["surf_group"] = self.dist_group
        dbg = self.engine.train_step(s, self._loss_cfg(), sc, noise=noise, **kw)
        if self.dist_group is not None:   # sums over ranks; AdamW divides by the reduced count
            dp.allreduce_(self.engine.reduce_buf, self.dist_group)
        ls = self.engine.loss_sums()
        losses = LazyLosses(ls, self.grad_weight != 0, self.eik_weight != 0)
        # the reference returns the graph-attached mean loss; here backward is already done, so step() skips the
        # (one tiny launch) division and callers get it through losses['total_loss'] on demand
        total_loss = ls[_ffi.LS_TOTAL] / ls[_ffi.LS_COUNT] if _want_total else None
        loss_approx = frame_avg_loss = None
        if do_avg_loss and "loss_approx" in dbg:      # written by the fused tail, straight into frame_avg_dst
            loss_approx = dbg["loss_approx"]
            frame_avg_loss = None                         # already in frame_avg_dst (no gather launch for nothing)
        elif do_avg_loss and frame_avg_dst is not None:   # two-call / data-parallel path: scatter inside isdf_frame_avg
            loss_approx, _ = self.engine.frame_avg(s["n_frames"], out=frame_avg_dst[0], index=frame_avg_dst[1])
            frame_avg_loss = None
        elif do_avg_loss:
            loss_approx, frame_avg_loss = self.engine.frame_avg(s["n_frames"])
        return total_loss, losses, loss_approx, frame_avg_loss

    # ---- step (trainer.py:951-1016) -----------------------------------------------------
    def step(self):
        torch.cuda.synchronize()                       # metrics.start_timing (metrics.py:13-22)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()

        K = self.frames.T_WC_batch.shape[0]
        if len(self.frames) > self.window_size and self.incremental:
            idxs = self.select_keyframes()
        else:
            idxs = np.arange(K)
        self.active_idxs = idxs
        # device copies of the window indices are cached while the window does not change (between keyframe
        # selections the reference rebuilds them every step: one H2D copy + one arange launch in front of the
        # first kernel of a device-synchronised step)
        key = (tuple(int(i) for i in idxs), bool(self.fix_normal_window))
        cache = getattr(self, "_idx_cache", None)
        if cache is None or cache[0] != key:
            fidx = torch.as_tensor(np.asarray(idxs), dtype=torch.int32, device=self.device)
            # reference quirk q4: normals are read from the UN-windowed normal_batch with
            # window-local indices (trainer.py:956,969); fix_normal_window=True uses idxs.
            nidx = fidx if self.fix_normal_window else torch.arange(len(idxs), dtype=torch.int32, device=self.device)
            self._idx_cache = cache = (key, fidx, nidx)
        _, fidx, nidx = cache
        norm_batch = self.frames.normal_batch if self.do_normal else None
        sample_pts = self.sample_points(self.frames.depth_batch, self.frames.T_WC_batch, norm_batch=norm_batch,
                                        _idx=(fidx, nidx))
        self.active_pixels = {k: sample_pts[k] for k in ("indices_b", "indices_h", "indices_w")}

        fused = self.dist_group is None and getattr(self, "fuse_optimiser", True)
        dst = (self.frames.frame_avg_losses, fidx) if self.frames.frame_avg_losses.is_contiguous() else None
        total_loss, losses, active_loss_approx, frame_avg_loss = self.sdf_eval_and_loss(sample_pts, True, fused, dst, _want_total=False)
        if dst is None:
            self.frames.frame_avg_losses[fidx.long()] = frame_avg_loss   # trainer.py:979
        if not fused:
            self.optimiser.step()                       # backward is fused into sdf_eval_and_loss
        self._step_count += 1

        torch.cuda.synchronize()                       # metrics.end_timing (metrics.py:25-38)
        end.record()
        torch.cuda.synchronize()
        step_time = start.elapsed_time(end)
        self.tot_step_time += (1 / self.frac_time_perception) * (step_time / 1000.0)
        self.steps_since_frame += 1
        return losses, step_time
