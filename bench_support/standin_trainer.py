"""TEST / BENCH INFRASTRUCTURE -- not part of the product package.

Stand-in for the DRIVER side of the reference `Trainer` (`isdf/modules/trainer.py`) on hosts where the reference itself is not
importable (the GPU box has no /root/reference; bench.py and the `-m gpu` tests run there).  It restates exactly the methods of
the reference class that the hot path does NOT replace but that the drivers' frame-scheduling loop calls (train.py:102-136):

    __init__ / set_params      trainer.py:35-96,157-333   (hot-path subset of the JSON schema)
    get_latest_frame_id        trainer.py:100-101
    add_data / add_frame       trainer.py:564-582
    check_keyframe_latest      trainer.py:622-650
    select_keyframes           trainer.py:652-674

with the reference's attribute names, so that `isdf_amd.hot_path.graft()` -- the product's only binding -- treats it exactly like
a reference `Trainer` instance.  Where the reference IS importable, graft the real `Trainer` instead (INTEGRATION.md,
tests/test_graft_reference.py); nothing here is needed then.  `step`, `sample_points`, `sdf_eval_and_loss`, `is_keyframe` are
deliberately absent: they exist only as HIP kernels (hot_path.HotPath).

    HipTrainer(device, config, ...)  ==  graft(StandinTrainer(device, config, ...), ...)

`add_data`, `add_frame`, `check_keyframe_latest` and `select_keyframes` below follow the reference's methods statement by statement
(they ARE the reference's driver-side logic, restated because the GPU box has no reference checkout).  The reference is
Copyright (c) Meta Platforms, Inc. and affiliates, released under the MIT license (iSDF, facebookresearch/iSDF, LICENSE):
permission is granted, free of charge, to deal in the software without restriction, provided the copyright notice and the
permission notice are included in all copies or substantial portions of it; the software is provided "as is", without warranty.
"""
import copy
import json

import numpy as np
import torch

from isdf_amd.frame_store import FrameData
from isdf_amd.hot_path import FlatAdamW, HotPath, StepLosses, graft      # noqa: F401
from isdf_amd.modules import PositionalEncodingHIP, SDFMapHIP


class StandinTrainer:
    def __init__(self, device, config_file, chkpt_load_file=None, incremental=True, grid_dim=200, *,
                 inv_bounds_transform=None, fwd_operand="fp16x2", engine_factory=None, bwd_operand=None, spill_operand=None):
        """Positional signature of the reference constructor (trainer.py:35-42).  config_file: path to / dict
        with the reference's JSON schema.  The reference derives `inv_bounds_transform` from the GT mesh
        (trainer.py:76-87, 102-123); with no mesh IO here it is an argument (None = live modes)."""
        self.device = device
        self.incremental = incremental
        self.tot_step_time = 0.
        self.last_is_keyframe = False
        self.steps_since_frame = 0
        self.optim_frames = 0
        self.grid_dim = grid_dim
        if isinstance(config_file, str):
            with open(config_file) as f:
                self.config = json.load(f)
        else:
            self.config = config_file
        self.frames = FrameData()  # keyframes
        self.set_params()
        self.inv_bounds_transform = inv_bounds_transform
        self.active_idxs = None
        self.active_pixels = None
        self._net_opts = dict(fwd_operand=fwd_operand, engine_factory=engine_factory, bwd_operand=bwd_operand, spill_operand=spill_operand)
        self.load_networks()
        if chkpt_load_file is not None:
            self.load_checkpoint(chkpt_load_file)
        self.sdf_map.train()

    def get_latest_frame_id(self):
        return int(self.tot_step_time * self.fps)

    # ---- trainer.py:157-333 (hot-path subset) ---------------------------------
    def set_params(self):
        c = self.config
        self.dataset_format = c["dataset"].get("format", "synthetic")
        self.live = self.dataset_format in ["arkit", "realsense", "realsense_franka"]
        cam = c["dataset"]["camera"]
        self.fx, self.fy, self.cx, self.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        self.H, self.W = cam["h"], cam["w"]
        self.fps = 30 if self.live else c["dataset"].get("fps", 30)
        self.n_steps = c.get("trainer", {}).get("steps", 0)
        m = c["model"]
        self.do_active = bool(m.get("do_active", 0))
        self.scale_output = m["scale_output"]
        self.noise_std, self.noise_kf, self.noise_frame = m["noise_std"], m["noise_kf"], m["noise_frame"]
        self.window_size = m["window_size"]
        self.hidden_layers_block, self.hidden_feature_size = m["hidden_layers_block"], m["hidden_feature_size"]
        self.frac_time_perception = m["frac_time_perception"]
        self.iters_per_kf, self.iters_per_frame = m["iters_per_kf"], m["iters_per_frame"]
        self.kf_dist_th, self.kf_pixel_ratio = m["kf_dist_th"], m["kf_pixel_ratio"]
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
        self.min_depth, self.max_depth = s["depth_range"][0], s["depth_range"][1]
        self.dist_behind_surf, self.n_rays = s["dist_behind_surf"], s["n_rays"]
        self.n_rays_is_kf = s["n_rays_is_kf"]
        self.n_strat_samples, self.n_surf_samples = s["n_strat_samples"], s["n_surf_samples"]
        self.loss_approx_factor = 8

    def load_networks(self):
        """trainer.py:419-439 -- the network IS the HIP module here (no torch-eager SDFMap is ever built)"""
        pe = PositionalEncodingHIP(min_deg=0, max_deg=self.n_embed_funcs, scale=self.scale_input,
                                   transform=self.inv_bounds_transform)
        self.sdf_map = SDFMapHIP(pe, hidden_size=self.hidden_feature_size,
                                 hidden_layers_block=self.hidden_layers_block, scale_output=self.scale_output,
                                 device=self.device, **self._net_opts)
        self.optimiser = FlatAdamW(self.sdf_map, lr=self.learning_rate, weight_decay=self.weight_decay)

    def load_checkpoint(self, checkpoint_load_file):
        checkpoint = torch.load(checkpoint_load_file)
        self.sdf_map.load_state_dict(checkpoint["model_state_dict"])

    # ---- frames (trainer.py:564-582) ----------------------------------------------
    def add_data(self, data, replace=False):
        replace = self.last_is_keyframe is False
        self.frames.add_frame_data(data, replace)
        if self.last_is_keyframe:
            print("New keyframe. KF ids:", self.frames.frame_id[:-1])

    def add_frame(self, frame_data):
        if self.last_is_keyframe:
            self.frozen_sdf_map = copy.deepcopy(self.sdf_map)
        self.add_data(frame_data)
        self.steps_since_frame = 0
        self.last_is_keyframe = False
        self.optim_frames = self.iters_per_frame
        self.noise_std = self.noise_frame

    def make_frame(self, frame_id, depth, T_WC, im=None):
        """`Trainer.get_data` (trainer.py:530-562) for one frame already in memory: device tensors + normals
        from the HIP stencil kernel (reference: pointcloud_from_depth_torch + estimate_pointcloud_normals,
        trainer.py:553-557).  depth [H,W] metres (0 = invalid), T_WC [4,4]."""
        depth = torch.as_tensor(depth, dtype=torch.float32).to(self.device)[None, ...]
        T = torch.as_tensor(T_WC, dtype=torch.float32).to(self.device)[None, ...]
        normals = None
        if self.do_normal:
            normals = self.sdf_map.engine.estimate_normals(depth[0], self._sample_cfg())[None, ...]
        return FrameData(frame_id=np.array([frame_id]), depth_batch=depth, T_WC_batch=T, normal_batch=normals,
                         im_batch=im)

    # ---- keyframe bookkeeping (trainer.py:622-674) -----------------------------------------
    def check_keyframe_latest(self):
        """returns whether or not to add a new frame."""
        add_new_frame = False
        if self.last_is_keyframe:
            add_new_frame = True
        else:
            T_WC = self.frames.T_WC_batch[-1].unsqueeze(0)
            depth_gt = self.frames.depth_batch[-1].unsqueeze(0)
            self.last_is_keyframe = self.is_keyframe(T_WC, depth_gt)
            time_since_kf = self.tot_step_time - self.frames.frame_id[-2] / 30.
            if time_since_kf > 5. and not self.live:
                print("More than 5 seconds since last kf, so add new")
                self.last_is_keyframe = True
            if self.last_is_keyframe:
                self.optim_frames = self.iters_per_kf
                self.noise_std = self.noise_kf
            else:
                add_new_frame = True
        return add_new_frame

    def select_keyframes(self):
        """last two keyframes + (window-2) drawn without replacement with p ~ frame_avg_losses (numpy global
        RNG, as the reference)."""
        n_frames = len(self.frames)
        limit = n_frames - 2
        denom = self.frames.frame_avg_losses[:-2].sum()
        loss_dist = (self.frames.frame_avg_losses[:-2] / denom).cpu().numpy()
        rand_ints = np.random.choice(np.arange(0, limit), size=self.window_size - 2, replace=False, p=loss_dist)
        last = n_frames - 1
        return [*rand_ints, last - 1, last]


class HipTrainer(HotPath, StandinTrainer):
    def __init__(self, device, config, incremental=True, inv_bounds_transform=None, rng="philox",
                 seed=1, dist_group=None, fix_normal_window=False, fwd_operand="fp16x2", virtual_step_ms=None,
                 engine_factory=None, overlap_allreduce=False, bwd_operand=None, spill_operand=None):
        """config: path to / dict with the reference's JSON schema (replicaCAD.json).
        rng: "philox" (in-kernel, no host sync) or "torch" (draw with torch in the
        reference's order and shapes -- parity mode, one host sync per step)."""
        self._hip = None
        StandinTrainer.__init__(self, device, config, None, incremental, inv_bounds_transform=inv_bounds_transform,
                                fwd_operand=fwd_operand, engine_factory=engine_factory, bwd_operand=bwd_operand, spill_operand=spill_operand)
        graft(self, rng=rng, seed=seed, dist_group=dist_group, fix_normal_window=fix_normal_window,
              fwd_operand=fwd_operand, virtual_step_ms=virtual_step_ms, engine_factory=engine_factory,
              overlap_allreduce=overlap_allreduce, bwd_operand=bwd_operand, spill_operand=spill_operand)

    # aliases kept for checkpoint files / callers of round 1
    def state_dict(self):
        return self.hip_state_dict()

    def load_state_dict(self, sd):
        return self.load_hip_state_dict(sd)
