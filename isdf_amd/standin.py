"""Stand-in for the DRIVER side of the reference `Trainer` (`isdf/modules/trainer.py`) on hosts where the
reference itself is not importable (the GPU box has no /root/reference; bench.py and the `-m gpu` tests run
there).  It holds exactly the methods of the reference class that the hot path does NOT replace but that the
drivers' frame-scheduling loop calls (train.py:102-136):

    __init__ / set_params      trainer.py:35-96,157-333   (hot-path subset of the JSON schema)
    get_latest_frame_id        trainer.py:100-101
    add_data / add_frame       trainer.py:564-582
    check_keyframe_latest      trainer.py:622-650
    select_keyframes           trainer.py:652-674
    FrameData                  isdf/datasets/data_util.py:11-102

with the reference's attribute names, so that `isdf_amd.hot_path.graft()` -- the product's only binding --
treats it exactly like a reference `Trainer` instance.  Where the reference IS importable, graft the real
`Trainer` instead; nothing here is needed then.  `step`, `sample_points`,
`sdf_eval_and_loss`, `is_keyframe` are deliberately absent: they exist only as HIP kernels (hot_path.HotPath).
"""
import copy
import json

import numpy as np
import torch

from .hot_path import FlatAdamW
from .modules import PositionalEncodingHIP, SDFMapHIP

_FIELDS = ("frame_id", "im_batch", "im_batch_np", "depth_batch", "depth_batch_np", "T_WC_batch", "T_WC_batch_np",
           "normal_batch", "frame_avg_losses", "T_WC_track", "T_WC_gt")


class FrameData:
    """Keyframe store (`data_util.FrameData`, isdf/datasets/data_util.py:11-81): same fields and the same
    `add_frame_data(data, replace)` contract (append, or overwrite the last slot when the previous frame was
    not promoted to a keyframe), but the device batches live in pre-allocated buffers that grow
    geometrically instead of being re-built with `torch.cat` on every frame (data_util.py:84-102 copies the
    whole keyframe set -- ~13 MB per keyframe at 680x1200 -- each time a frame arrives).
    The public attributes stay plain tensors: views of the first len(self) rows of the backing buffers.
    Accepts the reference's own FrameData objects as `data` (same attribute names)."""

    def __init__(self, frame_id=None, im_batch=None, im_batch_np=None, depth_batch=None, depth_batch_np=None,
                 T_WC_batch=None, T_WC_batch_np=None, normal_batch=None, frame_avg_losses=None, T_WC_track=None,
                 T_WC_gt=None):
        self.frame_id = frame_id
        self.im_batch, self.im_batch_np = im_batch, im_batch_np
        self.depth_batch, self.depth_batch_np = depth_batch, depth_batch_np
        self.T_WC_batch, self.T_WC_batch_np = T_WC_batch, T_WC_batch_np
        self.normal_batch = normal_batch
        self.frame_avg_losses = frame_avg_losses
        self.T_WC_track, self.T_WC_gt = T_WC_track, T_WC_gt
        self._back = {}          # field name -> backing tensor (capacity >= len)

    def __len__(self):
        return 0 if self.frame_id is None else len(self.frame_id)

    def __deepcopy__(self, memo):   # snapshots carry only the live rows
        out = FrameData()
        for k in _FIELDS:
            v = getattr(self, k, None)
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
        if isinstance(data, np.ndarray):     # host twins / frame ids
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
        for k in _FIELDS:
            if k == "frame_avg_losses":
                continue
            if k == "T_WC_gt" and getattr(data, k, None) is None:
                continue
            setattr(self, k, self._expand(k, getattr(self, k, None), getattr(data, k, None), replace))
        empty = torch.zeros([n_new], device=data.depth_batch.device)
        self.frame_avg_losses = self._expand("frame_avg_losses", self.frame_avg_losses, empty, replace)


class StandinTrainer:
    def __init__(self, device, config_file, chkpt_load_file=None, incremental=True, grid_dim=200, *,
                 inv_bounds_transform=None, fwd_operand="fp16x2", engine_factory=None):
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
        self._net_opts = dict(fwd_operand=fwd_operand, engine_factory=engine_factory)
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
