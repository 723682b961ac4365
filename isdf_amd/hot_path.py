"""The drop-in boundary on the Python side: `graft(trainer)` re-binds the three hot-path methods of a
reference `isdf.modules.trainer.Trainer` INSTANCE to the HIP kernels behind the C ABI
(include/isdf_hip.h), in place:

    Trainer.step                trainer.py:951-1016
    Trainer.sample_points       trainer.py:683-766
    Trainer.sdf_eval_and_loss   trainer.py:768-868   (+ total_loss.backward(), trainer.py:981)
    Trainer.is_keyframe         trainer.py:586-620   (fused sampler -> frozen net -> depth render)
    Trainer.get_data            trainer.py:530-562   (the reference's own method runs; its two geometry calls -- depth -> point cloud ->
                                                      8-neighbour normals, transform.py:169-196,215-270 -- are ONE stencil launch)

ONE object owns the state: every attribute the reference's drivers and its own remaining methods read or
write -- `tot_step_time`, `steps_since_frame`, `optim_frames`, `last_is_keyframe`, `noise_std`, `frames`,
`active_idxs`, `active_pixels`, `frozen_sdf_map`, `sdf_map`, `optimiser` (train.py:102-136,
trainer.py:574-650,1011-1014) -- stays on the `Trainer` instance; the methods here only use `self.<name>`
with the reference's names.  `trainer.sdf_map` becomes an `SDFMapHIP` (a real nn.Module with the
reference's state_dict keys whose parameters are views of one flat buffer), `trainer.optimiser` a facade
with the `torch.optim.AdamW` surface, `trainer.frames` an `isdf_amd.frame_store.FrameData` (the reference's fields and
`add_frame_data` contract on geometrically growing buffers instead of one `torch.cat` of the whole keyframe set per frame,
data_util.py:84-102; the existing keyframes migrate).  Everything else (`add_frame`, `add_data`,
`check_keyframe_latest`, `select_keyframes`, evaluation, visualisation) keeps running as the
reference's own code on the same object.

Where the reference is not importable (the GPU box, bench.py), `bench_support/standin_trainer.py` (test / bench infrastructure, outside
this package) restates the driver-side methods; its `HipTrainer` is `graft()` applied to that stand-in -- the same code path.

There is no CPU fallback: without the HIP library or a HIP device `graft` raises.
"""
import copy
import ctypes
import os
import sys
import types

import numpy as np
import torch

from . import _ffi, dp, frame_store
from .engine import LossConfig, NetConfig, SampleConfig
from .modules import PositionalEncodingHIP, SDFMapHIP


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

    def zero_grad(self, set_to_none=True):
        pass   # gradients never live on the parameters (they are sums in the engine's reduce buffer)

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
        if not sd["state"]:          # a checkpoint taken before the first step: fresh moments
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_(); eng.opt_step = 0
        for i, (k, (off, shp)) in enumerate(eng.slices.items()):
            if i in sd["state"]:
                n = int(np.prod(shp))
                eng.exp_avg[off:off + n].copy_(sd["state"][i]["exp_avg"].reshape(-1))
                eng.exp_avg_sq[off:off + n].copy_(sd["state"][i]["exp_avg_sq"].reshape(-1))
                eng.opt_step = int(float(sd["state"][i]["step"]))
        for k in ("lr", "weight_decay", "betas", "eps"):
            if sd.get("param_groups") and k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]


_CUDA_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)       # the current device index without torch.cuda's Python layers


class StepLosses(dict):
    """`losses` of `Trainer.step`: keys sdf_loss / grad_loss / eikonal_loss (floats) and total_loss (0-d tensor;
    callers use '{:.6f}'.format and .item(), train.py:138,215).  Built from ONE 8-float device->host copy that
    rides on the step's closing synchronisation instead of the reference's three `.item()` syncs
    (loss.py:187-200); the values are snapshotted at construction, so a later step cannot change them."""

    def __init__(self, loss_sums_host, has_grad, has_eik):
        ls = loss_sums_host.tolist()        # (iterating a tensor goes through unbind(): ~20 us)
        n = max(ls[_ffi.LS_COUNT], 1.0)
        super().__init__()
        self["sdf_loss"] = ls[_ffi.LS_SDF] / n
        if has_grad:
            self["grad_loss"] = ls[_ffi.LS_GRAD] / n
        if has_eik:
            self["eikonal_loss"] = ls[_ffi.LS_EIK] / n
        self["total_loss"] = torch.scalar_tensor(ls[_ffi.LS_TOTAL] / n)   # (0-d fp32 like torch.tensor(float), at half its cost)


class _LazyCut(dict):
    """`self.active_pixels` (trainer.py:970-974, read by keyframe_vis :1162-1175): the sampler's index tensors
    cut to the valid-ray count on first access, so `step()` itself never waits for the count."""

    def __init__(self, raw):
        super().__init__()
        self._raw = raw

    def _fill(self):
        if self._raw is not None:
            R = int(self._raw["n_valid"].item())
            for k in ("indices_b", "indices_h", "indices_w"):
                dict.__setitem__(self, k, self._raw[k][:R])
            self._raw = None

    def __getitem__(self, k):
        self._fill(); return dict.__getitem__(self, k)

    def keys(self):
        self._fill(); return dict.keys(self)

    def items(self):
        self._fill(); return dict.items(self)

    def values(self):
        self._fill(); return dict.values(self)

    def __iter__(self):
        self._fill(); return dict.__iter__(self)

    def __len__(self):
        return 3

    def __contains__(self, k):
        return k in ("indices_b", "indices_h", "indices_w")


class _BackwardDone(torch.autograd.Function):
    """`total_loss` of `sdf_eval_and_loss`: the reference returns a graph-attached scalar and the caller runs
    `total_loss.backward(); self.optimiser.step()` (trainer.py:981-982).  Here the backward pass has already
    run inside the native call (gradient sums sit in the engine's reduce buffer), so `.backward()` is a no-op
    and the same two caller lines keep working."""

    @staticmethod
    def forward(ctx, value, anchor):
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return None, None


class HotPath:
    """Mix-in holding the replaced methods.  `self` is the Trainer (reference or stand-in)."""

    # ------------------------------------------------------------------ helpers
    def _loss_cfg(self):
        return LossConfig(self.bounds_method, self.loss_type, self.trunc_weight, self.trunc_distance,
                          self.eik_weight, self.eik_apply_dist, self.grad_weight, bool(self.orien_loss))

    def _sample_cfg(self, n_rays=None, dist_behind_surf=None, n_strat=None, n_surf=None):
        return SampleConfig(n_rays=self.n_rays if n_rays is None else n_rays,
                            n_strat=self.n_strat_samples if n_strat is None else n_strat,
                            n_surf=self.n_surf_samples if n_surf is None else n_surf,
                            min_depth=self.min_depth,
                            dist_behind_surf=self.dist_behind_surf if dist_behind_surf is None else dist_behind_surf,
                            H=self.H, W=self.W, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy)

    @property
    def engine(self):
        return self.sdf_map.engine

    def _rank(self):
        g = self._hip.dist_group
        return 0 if g is None else torch.distributed.get_rank(g)

    # ------------------------------------------------------------------ sampling (trainer.py:683-766)
    def _draws_torch(self, F, sc, n_valid_fn):
        """Reference draw order / shapes / devices: randint(h), randint(w) on the training device,
        rand(R, n_strat) on the device, normal(0, 0.1, (R, n_surf-1)) on the CPU generator
        (sample.py:15-16,123,160-162)."""
        dev = self._hip.device
        total = sc.n_rays * F
        ih = torch.randint(0, sc.H, (total,), device=dev)
        iw = torch.randint(0, sc.W, (total,), device=dev)
        R = n_valid_fn(ih, iw)
        U = torch.rand(R, sc.n_strat, device=dev)
        N_off = torch.normal(torch.zeros(R, max(sc.n_surf - 1, 0)), 0.1).to(dev)
        return dict(indices_h=ih, indices_w=iw, U=U, N_off=N_off)

    def _sample_raw(self, depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc, want_T, shared_key=False,
                    reuse=False):
        """sampler launch; returns the engine's capacity-sized output dict (rows >= n_valid are undefined).
        shared_key: use the rank-INDEPENDENT Philox key (keyframe test under data parallelism: every rank
        must reach the same decision)."""
        eng, hip = self.engine, self._hip
        if hip.rng == "torch":
            as_t = lambda ix: torch.as_tensor(list(ix), dtype=torch.long, device=hip.device) if isinstance(ix, (tuple, list)) else ix.long()
            fi_t, ni_t = as_t(frame_idx), (None if normal_idx is None else as_t(normal_idx))

            def n_valid(ih, iw):   # the reference learns R from its boolean-mask compaction (a sync)
                ib = torch.arange(fi_t.numel(), device=ih.device).repeat_interleave(sc.n_rays)
                d = depth_batch[fi_t[ib], ih, iw]
                ok = d != 0
                if norm_batch is not None:
                    ok &= ~torch.isnan(norm_batch[ni_t[ib], ih, iw, 0])
                return int(ok.sum().item())
            draws = self._draws_torch(fi_t.numel(), sc, n_valid)
            return eng.sample(depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc, draws=draws,
                              want_T=want_T, reuse=reuse)
        hip.draw_count += 1
        seed = hip.seed if shared_key else dp.rank_seed(hip.seed, self._rank())
        return eng.sample(depth_batch, T_WC_batch, norm_batch, frame_idx, normal_idx, sc,
                          seed=seed, offset=hip.draw_count, want_T=want_T, reuse=reuse)

    def sample_points(self, depth_batch, T_WC_batch, norm_batch=None, active_loss_approx=None, n_rays=None,
                      dist_behind_surf=None, n_strat_samples=None, n_surf_samples=None, _shared_key=False):
        """Same 11 named tensors as the reference (trainer.py:753-766), cut to the R valid rays (one host
        sync for R, exactly where the reference's boolean-mask compaction has one, sample.py:39-55).
        `binary_masks` is None: the dense [F,H,W] mask image is never materialised (sample.py:58-61)."""
        if active_loss_approx is not None:
            raise Exception('Active sampling not currently supported.')
        sc = self._sample_cfg(n_rays, dist_behind_surf, n_strat_samples, n_surf_samples)
        dev = self._hip.device
        ar = torch.arange(depth_batch.shape[0], dtype=torch.int32, device=dev)
        s = self._sample_raw(depth_batch.contiguous(), T_WC_batch.contiguous(),
                             None if norm_batch is None else norm_batch.contiguous(), ar,
                             None if norm_batch is None else ar, sc, want_T=True, shared_key=_shared_key)
        R = int(s["n_valid"].item())

        def cut(t):
            return None if t is None else t[:R]
        return {
            "depth_batch": depth_batch, "pc": cut(s["pc"]), "z_vals": cut(s["z_vals"]),
            "indices_b": cut(s["indices_b"]), "indices_h": cut(s["indices_h"]), "indices_w": cut(s["indices_w"]),
            "dirs_C_sample": cut(s["dirs_C_sample"]), "depth_sample": cut(s["depth_sample"]),
            "T_WC_sample": cut(s["T_WC_sample"]), "norm_sample": cut(s["norm_sample"]),
            "binary_masks": None,
            "_raw": s, "_sc": sc,
        }

    # ------------------------------------------------------------------ loss + backward (trainer.py:768-868, 981)
    def _raw_from_public(self, sample):
        """A `sample` dict that did not come from our sample_points (a caller assembled the reference's 11
        tensors itself): rebuild what the step kernel needs."""
        dev = self._hip.device
        pc = sample["pc"].detach().to(dev, torch.float32).contiguous()
        R, S = pc.shape[0], pc.shape[1]
        T = sample["T_WC_sample"].to(dev, torch.float32)
        dC = sample["dirs_C_sample"].to(dev, torch.float32).contiguous()
        dW = (T[:, :3, :3] * dC[:, None, :]).sum(-1).contiguous()          # transform.py:36-41
        i64 = lambda t: t.to(dev, torch.int64).contiguous()
        F = int(sample["depth_batch"].shape[0])
        s = dict(n_valid=torch.tensor([R], dtype=torch.int32, device=dev), pc=pc,
                 z_vals=sample["z_vals"].to(dev, torch.float32).contiguous(),
                 depth_sample=sample["depth_sample"].to(dev, torch.float32).contiguous(),
                 dirs_C_sample=dC, dirs_W_sample=dW,
                 norm_sample=None if sample.get("norm_sample") is None else sample["norm_sample"].to(dev, torch.float32).contiguous(),
                 indices_b=i64(sample["indices_b"]), indices_h=i64(sample["indices_h"]), indices_w=i64(sample["indices_w"]),
                 max_rays=R, S=S, n_frames=F)
        sc = self._sample_cfg(n_strat=S - self.n_surf_samples)
        return s, sc

    def _step_kernels(self, s, sc, fused_optim, frame_avg_dst):
        """sampler outputs -> reduce buffer (+ optimiser when fused).  Returns (loss_approx, frame_avg_loss)."""
        hip, eng = self._hip, self.engine
        noise, kw = None, {}
        if self.noise_std is not None:   # fc_map.py:106-108 (drawn even for 0)
            if hip.rng == "torch":
                R = int(s["n_valid"].item())
                noise = torch.randn(R, s["S"], device=hip.device) * self.noise_std
            else:                        # philox mode: drawn inside the kernel
                hip.noise_count += 1
                kw = dict(noise_std=self.noise_std, noise_seed=dp.rank_seed(hip.seed, self._rank()),
                          noise_offset=hip.noise_count)
        if fused_optim:
            g = self.optimiser.param_groups[0]
            kw["optim"] = dict(lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"])
            if frame_avg_dst is not None:   # frames.frame_avg_losses[idxs] = ... inside the launch
                kw["optim"].update(frame_avg_out=frame_avg_dst[0], frame_avg_index=frame_avg_dst[1])
        if hip.dist_group is not None and self.bounds_method == "pc":
            kw["surf_group"] = hip.dist_group
        if hip.clock_slots:              # this rank's previous step time rides in the message's tail (one slot per rank)
            kw.update(extra_slot=hip.rank, extra_value=hip.prev_step_ms)
        if hip.prof_events is not None:  # bench.py: HIP events around this step's kernels (a ctypes array of four hipEvent_t), once
            kw["prof_events"], hip.prof_events = hip.prof_events, None
        split = hip.dist_group is not None and hip.overlap_allreduce and not fused_optim
        if split:
            if hip.split_event is None:
                hip.split_event, hip.comm_stream = dp.new_split_event(hip.device), torch.cuda.Stream(hip.device)
            kw["split_event"] = hip.split_event
        dbg = eng.train_step(s, self._loss_cfg(), sc, noise=noise, **kw)
        if split:                        # the message in two parts, the first one beside the closing reduction's second launch
            dp.allreduce_split_(eng.reduce_buf, eng.reduce_split, hip.split_event, hip.comm_stream, hip.dist_group)
        elif hip.dist_group is not None:   # sums over ranks; AdamW divides by the reduced count
            if hip.rccl is not None:       # THE collective of the step: RCCL on the step's own stream, enqueued by the C library
                eng.allreduce_direct(hip.rccl)
            else:                          # (gloo / a torch without the communicator handle: the framework's side-stream form)
                dp.allreduce_(eng.reduce_buf, hip.dist_group)
        return dbg

    def sdf_eval_and_loss(self, sample, do_avg_loss=True):
        """(total_loss, losses, loss_approx, frame_avg_loss) as the reference.  The backward pass has already
        run when this returns (gradient sums are in the engine's reduce buffer): the caller's
        `total_loss.backward()` is a no-op and `self.optimiser.step()` applies the update."""
        if "_raw" in sample:
            s, sc = sample["_raw"], sample["_sc"]
        else:
            s, sc = self._raw_from_public(sample)
        self._step_kernels(s, sc, fused_optim=False, frame_avg_dst=None)
        eng = self.engine
        ls = eng.loss_sums().detach().cpu()
        losses = StepLosses(ls, self.grad_weight != 0, self.eik_weight != 0)
        anchor = next(iter(self.sdf_map.parameters()))
        total_loss = _BackwardDone.apply((eng.loss_sums()[_ffi.LS_TOTAL] / eng.loss_sums()[_ffi.LS_COUNT]).detach(), anchor)
        loss_approx = frame_avg_loss = None
        if do_avg_loss:
            loss_approx, frame_avg_loss = eng.frame_avg(s["n_frames"])
        return total_loss, losses, loss_approx, frame_avg_loss

    # ------------------------------------------------------------------ step (trainer.py:951-1016)
    def _timing_start(self, st):
        """metrics.start_timing (metrics.py:13-22): device-synchronised, CUDA(HIP) events.  `st` is the stream all of
        the step's work is launched on; synchronising it is the reference's device synchronisation for this path
        (torch.cuda.synchronize() itself costs ~5 us of host time per call in index / env lookups)."""
        if st is not None:
            st.synchronize()
            ev = self._hip.timing_events          # ONE pair of events re-recorded every step: no event creation on the step path
            if ev is None:
                ev = self._hip.timing_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(st)
            return ev
        import time
        return time.perf_counter(), None

    def _timing_end(self, st, start, end):
        """metrics.end_timing (metrics.py:25-38), milliseconds.  The reference synchronises, records `end`, synchronises again;
        here `end` is recorded IN STREAM ORDER behind the step's last launch and the stream is synchronised once: the same
        interval on the device (start of the step's first kernel's queue slot to the end of its last kernel) without the host's
        wake-up latency of the first synchronisation inside it, and one device round trip less per step."""
        if st is not None:
            end.record(st)
            st.synchronize()
            return start.elapsed_time(end)
        import time
        return (time.perf_counter() - start) * 1000.0

    def step(self):
        dev = self._hip.device
        if dev.type != "cuda":
            return self._step(None)
        from .engine import pinned_stream
        if _CUDA_GET_DEVICE is not None and dev.index is not None and _CUDA_GET_DEVICE() == dev.index:
            with pinned_stream(dev.index) as st:      # already the current device: no device switch around the step (~3 us of host time)
                return self._step(st)
        with torch.cuda.device(dev), pinned_stream(dev) as st:
            return self._step(st)

    def _step(self, st):
        hip = self._hip
        start, end = self._timing_start(st)

        K = self.frames.T_WC_batch.shape[0]
        if len(self.frames) > self.window_size and self.incremental:
            # from here on select_keyframes reads frames.frame_avg_losses EVERY step: a store that can keep them in pinned host
            # memory (isdf_amd.frame_store; not the reference's FrameData) is asked to, once -- the closing launch then writes
            # them zero-copy and the draw below never touches the device
            if hip.device.type == "cuda" and self.frames.frame_avg_losses.device.type == "cuda":
                to_host = getattr(self.frames, "losses_to_host", None)
                if to_host is not None:
                    to_host()
            idxs = self._select_window()               # the reference's select_keyframes; replicated under data parallelism
        else:
            idxs = np.arange(K)
        self.active_idxs = idxs
        # The window: up to 8 keyframes travel INLINE as kernel arguments (sampler, step tail) -- select_keyframes re-draws it on
        # the host every step once K > window_size, and a device copy per step (two H2D copies + new call plans) cost ~80 us of
        # the synchronised step in that regime.  Longer windows (non-incremental runs over many frames) use cached device tensors.
        if hip.inline_window and len(idxs) <= _ffi.MAX_INLINE_FRAMES:
            fidx = tuple(idxs.tolist()) if isinstance(idxs, np.ndarray) else tuple(int(i) for i in idxs)   # (ndarray.tolist(): Python ints, 0.2 us)
            # reference quirk q4: normals are read from the UN-windowed normal_batch with
            # window-local indices (trainer.py:956,969); fix_normal_window=True uses idxs.
            nidx = fidx if hip.fix_normal_window else tuple(range(len(fidx)))
        else:
            key = (tuple(int(i) for i in idxs), bool(hip.fix_normal_window))
            if hip.idx_cache is None or hip.idx_cache[0] != key:
                fidx = torch.as_tensor(np.asarray(idxs), dtype=torch.int32, device=hip.device)
                nidx = fidx if hip.fix_normal_window else torch.arange(len(idxs), dtype=torch.int32, device=hip.device)
                hip.idx_cache = (key, fidx, nidx)
            _, fidx, nidx = hip.idx_cache
        norm_batch = self.frames.normal_batch if self.do_normal else None
        sc = self._sample_cfg()
        # no depth_batch[idxs] copy (trainer.py:965: 16 MB at 680x1200x5): the sampler takes the window indices
        s = self._sample_raw(self.frames.depth_batch, self.frames.T_WC_batch, norm_batch, fidx,
                             nidx if norm_batch is not None else None, sc, want_T=False, reuse=True)
        self.active_pixels = _LazyCut(s)

        fal = self.frames.frame_avg_losses
        fused = hip.dist_group is None and hip.fuse_optimiser
        # the kernels write the window's averages straight into the keyframe store: a device tensor (the reference's FrameData), or
        # pinned host memory (isdf_amd.frame_store: select_keyframes then never touches the device)
        direct = fal.is_contiguous() and fal.dtype == torch.float32 and (fal.device == hip.device or self._pinned(fal))
        dbg = self._step_kernels(s, sc, fused, (fal, fidx) if direct else None)
        eng = self.engine
        if not fused and direct and hip.fuse_optimiser:
            # two-call / data-parallel form: ONE closing launch = AdamW on the (all-reduced) gradient sums + operand
            # repack + `frames.frame_avg_losses[idxs] = frame_avg_loss` from the reduced bins (trainer.py:979-982)
            g = self.optimiser.param_groups[0]
            eng.train_step_finish(len(idxs), dict(lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"],
                                                  frame_avg_out=fal, frame_avg_index=fidx))
        else:
            if not fused or not direct:                # separate launches (caller-owned frame_avg_losses layout, or
                if direct:                             # fuse_optimiser=False: the reference's call sequence)
                    eng.frame_avg(len(idxs), out=fal, index=fidx)
                else:
                    _, fa = eng.frame_avg(len(idxs))
                    self.frames.frame_avg_losses[[int(i) for i in idxs]] = fa      # trainer.py:979
            if not fused:
                self.optimiser.step()                  # AdamW on the (all-reduced) gradient sums
        # `losses`: the step's last launch stored the (reduced) loss sums -- and, data parallel, the ranks' step times -- in
        # pinned host memory (eng.mailbox); they are valid after the closing synchronisation below.  No copy command.
        if getattr(eng, "mailbox", None) is None:      # engines without a host mailbox (tests' CPU stand-in)
            hip.loss_host.copy_(eng.loss_sums(), non_blocking=True)
            mailbox = hip.loss_host
        else:
            mailbox = eng.mailbox
            if hip.dist_group is not None and not (not fused and direct and hip.fuse_optimiser):
                # reference call sequence (separate AdamW launch): the reduced sums were not mirrored by a closing launch
                mailbox[:8].copy_(eng.loss_sums(), non_blocking=True)
                if hip.clock_slots:
                    mailbox[8:].copy_(eng.reduce_buf[eng.reduce_floats:], non_blocking=True)

        step_time = self._timing_end(st, start, end)
        clock_ms = step_time
        if hip.virtual_step_ms is not None:            # pinned schedule (parity / accuracy runs)
            clock_ms = step_time = float(hip.virtual_step_ms)
        elif hip.clock_slots:
            # ONE virtual clock for all ranks (the frame schedule is a function of it) WITHOUT a second collective: every
            # rank's time of the PREVIOUS step came back in the tail of this step's all-reduce message; the clock advances
            # by the slowest rank's, one step late (step 0 advances it by 0).  This step's own time goes out with the next.
            clock_ms = float(mailbox[8:8 + hip.clock_slots].max()) if mailbox.numel() >= 8 + hip.clock_slots else \
                float(eng.reduce_buf[eng.reduce_floats:].max())
            hip.prev_step_ms = float(np.float32(step_time))
        losses = StepLosses(mailbox, self.grad_weight != 0, self.eik_weight != 0)   # (reads the first 8 floats; a slice is 1.7 us)
        hip.step_count += 1
        self.tot_step_time += (1 / self.frac_time_perception) * (clock_ms / 1000.0)
        self.steps_since_frame += 1
        return losses, step_time

    def _pinned(self, t):
        """t is a pinned host tensor the HIP device can write (is_pinned() is a driver query: asked once per buffer)"""
        hip = self._hip
        if t.device.type != "cpu" or hip.device.type != "cuda":
            return False
        key = (t.data_ptr(), t.numel())
        if getattr(hip, "pinned_key", None) != key:
            hip.pinned_key, hip.pinned_ok = key, bool(t.is_pinned())
        return hip.pinned_ok

    def _select_window(self):
        """The reference's window draw (`select_keyframes`, trainer.py:652-674) runs unchanged.  Under data parallelism
        every rank must pick the SAME window, and it does without a collective: the inputs are bit-identical on every rank
        (`frames.frame_avg_losses` comes from the all-reduced bins) and `np.random.choice` draws from a private stream
        that graft() seeded identically everywhere -- swapped in for the call, so nothing else that touches numpy's
        global generator on one rank can de-synchronise it."""
        hip = self._hip
        if hip.dist_group is None:
            return self.select_keyframes()
        saved = np.random.get_state()
        np.random.set_state(hip.window_rng_state)
        try:
            return self.select_keyframes()
        finally:
            hip.window_rng_state = np.random.get_state()
            np.random.set_state(saved)

    # ------------------------------------------------------------------ keyframe test (trainer.py:586-620)
    def is_keyframe(self, T_WC, depth_gt):
        """sampler (n_rays_is_kf rays, 0.8 m behind the surface) -> frozen net forward -> per-ray z sort +
        first-zero-crossing depth render + below-threshold count, all in HIP kernels; the decision and the
        printed line are the reference's.  Rank-independent draws under data parallelism."""
        hip = self._hip
        sample_pts = self.sample_points(depth_gt, T_WC, n_rays=self.n_rays_is_kf, dist_behind_surf=0.8,
                                        _shared_key=hip.dist_group is not None)
        s = sample_pts["_raw"]
        pc = s["pc"]
        n = int(sample_pts["pc"].shape[0])           # R valid rays (sample_points already synchronised on it)
        noise = None
        if self.noise_std is not None:               # drawn on the CUT tensor like upstream (trainer.py:594-595,
            noise = torch.zeros(pc.shape[:-1], device=hip.device)     # fc_map.py:106): the generator advances by R x S
            noise[:n] = torch.randn(n, pc.shape[1], device=hip.device) * self.noise_std
        sdf = self.frozen_sdf_map.engine.sdf_eval(pc, noise=noise)           # frozen net, no grad (trainer.py:594-595)
        view, below = self.engine.render_depth(s["z_vals"], sdf, s["depth_sample"], self.kf_dist_th,
                                               n_valid=s["n_valid"])
        # no valid ray: upstream divides 0 by 0 -> NaN -> `NaN < ratio` is False (trainer.py:606-609)
        below_th_prop = float(below.item()) / n if n > 0 else float("nan")
        is_keyframe = below_th_prop < self.kf_pixel_ratio
        print("Proportion of loss below threshold", below_th_prop, "for KF should be less than",
              self.kf_pixel_ratio, " ---> is keyframe:", is_keyframe)
        return is_keyframe

    # ------------------------------------------------------------------ per-frame ingest (trainer.py:530-562)
    def get_data(self, idxs):
        """The reference's `get_data` runs unchanged (dataset access, host twins, `T_WC_gt`, its own `FrameData` per frame); for the
        duration of the call its two geometry functions are answered by the stencil kernel: `pointcloud_from_depth_torch`
        (transform.py:169-196) hands over the depth image, `estimate_pointcloud_normals` (transform.py:215-270: eight shifted
        cross products through a 70 MB index tensor, ~10 eager kernels) becomes ONE launch of `isdf_estimate_normals` -- same
        camera-frame normals, NaN where the reference leaves NaN (fixture `ingest_small`).  With a 0.3 ms step and
        iters_per_frame = 10 a frame arrives every ~3 ms of stepping: the eager chain was a visible share of that."""
        geo = self._hip.geometry_transform
        if not self.do_normal or geo is None:      # (no normals wanted, or not the reference's trainer module)
            return super().get_data(idxs)
        eng, hip = self.engine, self._hip

        class _DepthAsPointCloud:          # what the first call hands to the second
            def __init__(self, depth, fx, fy, cx, cy):
                self.depth, self.k = depth, (fx, fy, cx, cy)

        def pointcloud_from_depth(depth, fx, fy, cx, cy):
            return _DepthAsPointCloud(depth, fx, fy, cx, cy)

        def estimate_normals(pc):
            if not isinstance(pc, _DepthAsPointCloud):      # some other caller's point cloud: the reference's own arithmetic
                return saved[1](pc)
            fx, fy, cx, cy = pc.k
            hip.ingest_launches += 1
            return eng.estimate_normals(pc.depth, SampleConfig(H=int(pc.depth.shape[0]), W=int(pc.depth.shape[1]),
                                                               fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy)))
        saved = (geo.pointcloud_from_depth_torch, geo.estimate_pointcloud_normals)
        geo.pointcloud_from_depth_torch, geo.estimate_pointcloud_normals = pointcloud_from_depth, estimate_normals
        try:
            return super().get_data(idxs)
        finally:
            geo.pointcloud_from_depth_torch, geo.estimate_pointcloud_normals = saved

    # ------------------------------------------------------------------ data parallel
    def check_keyframe_latest(self):
        """The reference's decision logic (trainer.py:622-650) runs unchanged; under data parallelism rank 0's
        outcome is broadcast so the keyframe sets (and with them the all-reduce message size) cannot diverge."""
        add_new_frame = super().check_keyframe_latest()
        hip = self._hip
        if hip.dist_group is not None:
            v = dp.broadcast_floats([float(add_new_frame), float(self.last_is_keyframe), float(self.optim_frames),
                                     -1.0 if self.noise_std is None else float(self.noise_std)], hip.dist_group, hip.device)
            add_new_frame, self.last_is_keyframe, self.optim_frames = bool(v[0]), bool(v[1]), int(v[2])
            self.noise_std = None if v[3] < 0 else v[3]
        return add_new_frame

    def add_frame(self, frame_data):
        """New frame: under data parallelism rank 0's depth / pose / normals are broadcast once per FRAME
        (1.2-3.3 MB depth + 3.7-9.8 MB normals), then the reference's add_frame runs."""
        hip = self._hip
        if hip.dist_group is not None:
            for k in ("depth_batch", "T_WC_batch", "normal_batch"):
                t = getattr(frame_data, k, None)
                if torch.is_tensor(t):
                    torch.distributed.broadcast(t, dp.src_rank(hip.dist_group), group=hip.dist_group)
        return super().add_frame(frame_data)

    # ------------------------------------------------------------------ checkpoint / resume
    def hip_state_dict(self):
        """Everything needed for a true resume.  The reference saves only model + optimiser
        (train.py:207-219) and restores only the model (trainer.py:441-444); keyframes, RNG
        position, the frozen keyframe-test network and the virtual clock are lost there."""
        fr, hip = self.frames, self._hip
        frozen = getattr(self, "frozen_sdf_map", None)
        return {
            "model_state_dict": {k: v.detach().clone() for k, v in self.sdf_map.state_dict().items()},
            "optimizer_state_dict": self.optimiser.state_dict(),
            "frozen_state_dict": None if frozen is None else {k: v.detach().clone() for k, v in frozen.state_dict().items()},
            # EVERY field of the keyframe store (data_util.py:11-43): the device batches the hot path reads AND the host
            # twins / tracked / ground-truth poses / frame count the reference's visualisation and evaluation index
            # (trainer.py:1021-1067,1152-1157,1228-1231) -- a resumed run appends to all of them consistently
            "frames": {k: (None if getattr(fr, k, None) is None else
                           (getattr(fr, k).copy() if isinstance(getattr(fr, k), np.ndarray) else
                            (getattr(fr, k).clone() if torch.is_tensor(getattr(fr, k)) else copy.deepcopy(getattr(fr, k)))))
                       for k in FRAME_FIELDS if hasattr(fr, k)},
            "clock": dict(tot_step_time=self.tot_step_time, steps_since_frame=self.steps_since_frame,
                          last_is_keyframe=self.last_is_keyframe, optim_frames=self.optim_frames,
                          noise_std=self.noise_std, step_count=hip.step_count,
                          prev_step_ms=hip.prev_step_ms),     # data parallel: the step time still to be added to the clock
            "rng": dict(draw_count=hip.draw_count, noise_count=hip.noise_count, seed=hip.seed,
                        window=getattr(hip, "window_rng_state", None),
                        numpy=np.random.get_state(), torch=torch.get_rng_state(),
                        torch_cuda=torch.cuda.get_rng_state(hip.device) if hip.device.type == "cuda" else None),
        }

    def load_hip_state_dict(self, sd):
        hip = self._hip
        self.sdf_map.load_state_dict(sd["model_state_dict"])
        self.optimiser.load_state_dict(sd["optimizer_state_dict"])
        if sd.get("frozen_state_dict") is not None:
            self.frozen_sdf_map = copy.deepcopy(self.sdf_map)
            self.frozen_sdf_map.load_state_dict(sd["frozen_state_dict"])
        f = sd["frames"]
        fr = type(self.frames)()
        for k, v in f.items():
            setattr(fr, k, v)
        # checkpoints written before the host twins were saved: rebuild them from the device batches so that the next
        # add_frame_data appends at the right length
        for twin, src in (("im_batch_np", "im_batch"), ("depth_batch_np", "depth_batch"), ("T_WC_batch_np", "T_WC_batch")):
            if twin not in f and getattr(fr, src, None) is not None and hasattr(fr, twin):
                t = getattr(fr, src).detach()
                if twin == "im_batch_np":     # the host twin is the raw 0..255 image, the device batch is float in [0, 1]
                    t = (t * 255).round().to(torch.uint8)          # (trainer.py:536-547)
                setattr(fr, twin, t.cpu().numpy())
        fal = getattr(fr, "frame_avg_losses", None)
        if fal is not None and fal.device.type == "cpu" and hip.device.type == "cuda" and not fal.is_pinned():
            fr.frame_avg_losses = fal.pin_memory()      # the ring store keeps them in pinned host memory (frame_store.py)
        self.frames = fr
        c = sd["clock"]
        self.tot_step_time, self.steps_since_frame = c["tot_step_time"], c["steps_since_frame"]
        self.last_is_keyframe, self.optim_frames = c["last_is_keyframe"], c["optim_frames"]
        self.noise_std, hip.step_count = c["noise_std"], c["step_count"]
        hip.prev_step_ms = float(c.get("prev_step_ms", 0.0))
        r = sd["rng"]
        hip.draw_count, hip.noise_count, hip.seed = r["draw_count"], r["noise_count"], r["seed"]
        hip.idx_cache = None
        if r.get("window") is not None:
            hip.window_rng_state = r["window"]
        np.random.set_state(r["numpy"]); torch.set_rng_state(r["torch"])
        if r.get("torch_cuda") is not None:
            torch.cuda.set_rng_state(r["torch_cuda"], hip.device)


# data_util.FrameData's fields (isdf/datasets/data_util.py:11-43) incl. the reference's frame counter
FRAME_FIELDS = ("frame_id", "im_batch", "im_batch_np", "depth_batch", "depth_batch_np", "T_WC_batch", "T_WC_batch_np",
                "normal_batch", "frame_avg_losses", "T_WC_track", "T_WC_gt", "count")

ENGINE_FACTORY = None      # host-logic tests on GPU-less machines install an oracle-backed stand-in here; never set by the product


UNSUPPORTED_HINT = ("isdf_amd hot path: %s (the reference's own Python path is the CPU / fallback path; "
                    "this build ships no second implementation)")


def unsupported_reason(trainer):
    """None, or why the kernels cannot take this trainer's CONFIGURATION (such configs keep running on the reference's
    own Python path -- `can_graft` says no, nothing raises): `bounds_method="normal"` (loss.py:29), `do_active` (trainer.py:718), a
    loss_type other than L1 / L2 (loss.py:104), a network shape the tile kernels are not instantiated for (`isdf_check_net`)."""
    if getattr(trainer, "bounds_method", "ray") not in ("ray", "pc"):
        return "bounds_method %r (only 'ray' and 'pc' can run upstream, loss.py:29)" % (trainer.bounds_method,)
    if getattr(trainer, "do_active", False):
        return "active sampling (rejected by the reference itself, trainer.py:718)"
    if getattr(trainer, "loss_type", "L1") not in ("L1", "L2"):
        return "loss_type %r (loss.py:104)" % (trainer.loss_type,)
    net = getattr(trainer, "sdf_map", None)
    if net is not None and not isinstance(net, SDFMapHIP) and ENGINE_FACTORY is None and os.path.exists(_ffi.LIB_PATH):
        try:
            pe = net.positional_encoding
            c = NetConfig(hidden=net.out_alpha.in_features, blocks=len(net.mid1), n_freqs=int(pe.n_freqs)).to_c()
        except AttributeError:
            return "trainer.sdf_map is not an isdf.modules.fc_map.SDFMap"
        rc = _ffi.lib().isdf_check_net(ctypes.byref(c))
        if rc != 0:
            return "network shape (hidden %d, %d blocks, %d octaves): %s" % (c.hidden, c.blocks, c.n_freqs,
                                                                              _ffi.lib().isdf_error_string(int(rc)).decode())
    return None


def can_graft(trainer):
    """True when `graft(trainer)` can bind the kernels: the trainer sits on a HIP device, libisdf_hip.so is built and the
    configuration is one the kernels take (`unsupported_reason`).  The guard INTEGRATION.md puts in front of graft(): wherever it says
    no -- a CPU-only host (BASELINE configs[0]), `bounds_method="normal"`, `do_active`, an uninstantiated network shape -- the
    reference's own Python path keeps running; this package ships no second implementation, and graft() itself raises there."""
    if unsupported_reason(trainer) is not None:
        return False
    if ENGINE_FACTORY is not None:          # host-logic tests with a stand-in engine
        return True
    return torch.device(trainer.device).type == "cuda" and torch.cuda.is_available() and os.path.exists(_ffi.LIB_PATH)


def graft(trainer, rng="philox", seed=1, dist_group=None, fix_normal_window=False, fwd_operand="fp16x2",
          fuse_optimiser=True, virtual_step_ms=None, engine_factory=None, overlap_allreduce=False, bwd_operand=None, spill_operand=None, migrate_frames=True):
    """Re-bind the hot path of `trainer` (an `isdf.modules.trainer.Trainer` or a `StandinTrainer`) to the HIP
    kernels, IN PLACE, and return it.

    rng: "philox" (in-kernel draws, no host sync inside step) or "torch" (draw with torch in the reference's
         order, shapes and devices: parity mode, one host sync per step like the reference).
    dist_group: torch.distributed process group for ray-sharded data parallelism (one all-reduce per step);
         weights are broadcast from rank 0 here so every rank starts from the same network.
    virtual_step_ms: if set, the virtual clock advances by this much per step instead of the measured step time
         (the frame schedule is a function of measured time, trainer.py:100-101,1011-1013; pin it to compare runs).
    bwd_operand: "fp16" | "bf16" | None (default for the forward mode): operand / spill type of the second-order sweeps and dW.
    spill_operand: None (auto) | "16bit" | "e4m3": storage of the spilled P / GB tensors (engine.NetConfig.spill_operand).
    overlap_allreduce: data parallel only -- the closing reduction in two launches and the all-reduce in two parts, the first
         one on a side stream beside the second launch (dp.allreduce_split_); two collectives per step instead of one.
    migrate_frames: replace `trainer.frames` (the reference's FrameData: one torch.cat of the whole keyframe set per frame,
         data_util.py:84-102) by an `isdf_amd.frame_store.FrameData` holding the same keyframes (False: keep the caller's store).
    engine_factory: tests only (a stand-in engine for hosts without a GPU)."""
    if isinstance(trainer, HotPath) and getattr(trainer, "_hip", None) is not None:
        return trainer
    if engine_factory is None:
        engine_factory = ENGINE_FACTORY
    dev = torch.device(trainer.device)
    if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
        # "cuda" names the current device; tensors on it report "cuda:N", and `frames.frame_avg_losses.device == hip.device` decides
        # whether the step's closing launch writes the keyframe losses in place (else: a separate launch and a blocking index_put)
        dev = torch.device("cuda", torch.cuda.current_device())
    if engine_factory is None and dev.type != "cuda":
        raise _ffi.IsdfError(UNSUPPORTED_HINT % ("device %r is not a HIP device" % (trainer.device,)))
    why = unsupported_reason(trainer)
    if why is not None:                # (can_graft() is the non-raising form: the reference's path then stays in place)
        raise _ffi.IsdfError(UNSUPPORTED_HINT % why)
    if rng not in ("philox", "torch"):
        raise ValueError("rng must be 'philox' or 'torch'")

    old = getattr(trainer, "sdf_map", None)
    if old is not None and not isinstance(old, SDFMapHIP):
        pe_old = old.positional_encoding
        pe = PositionalEncodingHIP(min_deg=pe_old.min_deg, max_deg=pe_old.max_deg, scale=pe_old.scale,
                                   transform=pe_old.transform)
        n_block = len(old.mid1)
        hidden = old.out_alpha.in_features
        with torch.random.fork_rng(devices=[]):    # grafting mid-run must not advance the caller's generator: the initial
            new = SDFMapHIP(pe, hidden_size=hidden, hidden_layers_block=n_block, scale_output=old.scale_output,   # weights
                            device=dev, fwd_operand=fwd_operand, engine_factory=engine_factory,    # are overwritten below
                            bwd_operand=bwd_operand, spill_operand=spill_operand)
        new.load_state_dict({k: v.detach() for k, v in old.state_dict().items()})
        new.train(old.training)
        trainer.sdf_map = new
        og = trainer.optimiser.param_groups[0]
        old_state = trainer.optimiser.state_dict()
        trainer.optimiser = FlatAdamW(new, lr=og["lr"], weight_decay=og["weight_decay"], betas=tuple(og["betas"]),
                                      eps=og["eps"])
        if old_state.get("state"):     # a trainer that has already stepped (or loaded optimiser state): keep exp_avg /
            trainer.optimiser.load_state_dict(old_state)    # exp_avg_sq / step; parameter order = named_parameters() order
        if getattr(trainer, "frozen_sdf_map", None) is not None:
            fz = copy.deepcopy(new)
            fz.load_state_dict({k: v.detach() for k, v in trainer.frozen_sdf_map.state_dict().items()})
            trainer.frozen_sdf_map = fz
    elif old is None:
        raise ValueError("graft() needs a trainer whose load_networks() has run")

    hip = types.SimpleNamespace(rng=rng, seed=int(seed), dist_group=dist_group, fix_normal_window=bool(fix_normal_window),
                                fuse_optimiser=bool(fuse_optimiser), device=dev, draw_count=0, noise_count=0,
                                step_count=0, idx_cache=None, timing_events=None, prof_events=None,
                                inline_window=True,      # bench.py flips it for the A/B of the window's transport
                                overlap_allreduce=bool(overlap_allreduce) and dist_group is not None, split_event=None,
                                comm_stream=None,
                                virtual_step_ms=None if virtual_step_ms is None else float(virtual_step_ms),
                                loss_host=torch.zeros(8, dtype=torch.float32,
                                                      pin_memory=(dev.type == "cuda")))
    # the module whose two geometry functions get_data() redirects: `geometry.transform` as the reference's trainer module sees it
    geo = getattr(getattr(sys.modules.get(trainer.__class__.__module__), "geometry", None), "transform", None)
    hip.geometry_transform = geo if (geo is not None and hasattr(geo, "pointcloud_from_depth_torch")
                                     and hasattr(geo, "estimate_pointcloud_normals")) else None
    hip.ingest_launches = 0
    if migrate_frames and getattr(trainer, "frames", None) is not None and not isinstance(trainer.frames, frame_store.FrameData):
        trainer.frames = frame_store.FrameData.from_reference(trainer.frames)     # same fields, the existing keyframes carried over
    trainer._hip = hip
    base = trainer.__class__
    if not issubclass(base, HotPath):
        trainer.__class__ = type("Hip" + base.__name__, (HotPath, base), {"__module__": HotPath.__module__})
    hip.clock_slots, hip.prev_step_ms, hip.rank, hip.rccl, hip.collective = 0, 0.0, 0, None, None
    if dist_group is not None:                           # replicated weights / moments
        eng = trainer.sdf_map.engine
        for t in (eng.params, eng.exp_avg, eng.exp_avg_sq):
            torch.distributed.broadcast(t, dp.src_rank(dist_group), group=dist_group)
        eng.pack()
        hip.rank, hip.world = torch.distributed.get_rank(dist_group), torch.distributed.get_world_size(dist_group)
        hip.window_rng_state = np.random.RandomState(hip.seed + 104729).get_state()   # replicated select_keyframes stream
        if hip.virtual_step_ms is None:                  # per-rank step-time slots in the tail of the all-reduce message
            hip.clock_slots = eng.reduce_extra = hip.world
        if hasattr(eng, "allreduce_direct") and hip.device.type == "cuda":      # (the CPU tests' stand-in engine has no C library behind it)
            from .engine import _stream
            hip.rccl = dp.rccl_agree(eng.lib, dp.rccl_direct(dist_group, hip.device), dist_group, hip.device, _stream(hip.device))
        hip.collective = "rccl on the step's stream (isdf_allreduce_sum_f32)" if hip.rccl is not None else "torch.distributed.all_reduce"
    return trainer
