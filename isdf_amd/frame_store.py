"""Keyframe store of the hot path (`data_util.FrameData`, isdf/datasets/data_util.py:11-102): same fields, same
`add_frame_data(data, replace)` contract, device batches in pre-allocated buffers that grow geometrically instead of one
`torch.cat` of the whole keyframe set per frame.  graft() works with the reference's own FrameData as well;
this one is what bench.py / the GPU tests use where the reference is absent, and a drop-in for callers that want the ring."""
import numpy as np
import torch

_FIELDS = ("frame_id", "im_batch", "im_batch_np", "depth_batch", "depth_batch_np", "T_WC_batch", "T_WC_batch_np",
           "normal_batch", "frame_avg_losses", "T_WC_track", "T_WC_gt")


class FrameData:
    """Keyframe store (`data_util.FrameData`, isdf/datasets/data_util.py:11-81): same fields and the same
    `add_frame_data(data, replace)` contract (append, or overwrite the last slot when the previous frame was
    not promoted to a keyframe), but the device batches live in pre-allocated buffers that grow
    geometrically instead of being re-built with `torch.cat` on every frame (data_util.py:84-102 copies the
    whole keyframe set -- ~13 MB per keyframe at 680x1200 -- each time a frame arrives).
    The public attributes stay plain tensors: views of the first len(self) rows of the backing buffers.
    Accepts the reference's own FrameData objects as `data` (same attribute names).

    `frame_avg_losses` (K floats) can live in PINNED HOST memory (`host_losses`, `losses_to_host()`): the hot path moves it
    there once the keyframe set outgrows the window.  From then on the step's closing launch writes the window's averages
    straight into it (zero-copy, like the loss sums' mailbox), and the reference's `select_keyframes` (trainer.py:652-674) --
    `.sum()`, a division and `.cpu().numpy()` on that tensor, EVERY step in that regime -- runs entirely on the host instead of
    as two device ops and a synchronising device->host copy (-40 us of a 0.42 ms step).  Values are valid after the step's closing
    synchronisation, which is when the reference reads them.  While K <= window_size nothing reads the losses per step and they
    stay on the device (the zero-copy store costs the closing launch ~3 us)."""

    def __init__(self, frame_id=None, im_batch=None, im_batch_np=None, depth_batch=None, depth_batch_np=None,
                 T_WC_batch=None, T_WC_batch_np=None, normal_batch=None, frame_avg_losses=None, T_WC_track=None,
                 T_WC_gt=None, host_losses=True):
        self.frame_id = frame_id
        self.im_batch, self.im_batch_np = im_batch, im_batch_np
        self.depth_batch, self.depth_batch_np = depth_batch, depth_batch_np
        self.T_WC_batch, self.T_WC_batch_np = T_WC_batch, T_WC_batch_np
        self.normal_batch = normal_batch
        self.frame_avg_losses = frame_avg_losses
        self.T_WC_track, self.T_WC_gt = T_WC_track, T_WC_gt
        self._back = {}          # field name -> backing tensor (capacity >= len)
        self.host_losses = bool(host_losses)

    @classmethod
    def from_reference(cls, other, host_losses=True):
        """a store holding the keyframes of `other` (the reference's data_util.FrameData, or anything with its attributes): the
        tensors are shared, the growing buffers are allocated at the next append.  What graft() swaps in for `trainer.frames`."""
        out = cls(host_losses=host_losses)
        for k in _FIELDS + ("count",):
            if hasattr(other, k):
                setattr(out, k, getattr(other, k))
        return out

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
            buf = torch.empty((cap,) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device,
                              pin_memory=name == "frame_avg_losses" and data.device.type == "cpu" and torch.cuda.is_available())
            if n:
                buf[:n] = batch
            back[name] = buf
        buf[n:n + k] = data
        return buf[:n + k]

    def losses_to_host(self):
        """Move frame_avg_losses into pinned host memory (one synchronising copy; later growth stays there).  No-op when they
        already are, when `host_losses` is off, or on a host without a HIP device."""
        fal = self.frame_avg_losses
        if fal is None or not getattr(self, "host_losses", True) or fal.device.type != "cuda":
            return False
        n = fal.shape[0]
        buf = torch.empty(max(2 * n, 8), dtype=fal.dtype, pin_memory=True)
        buf[:n].copy_(fal)
        self._back["frame_avg_losses"] = buf
        self.frame_avg_losses = buf[:n]
        return True

    def add_frame_data(self, data, replace):
        """data_util.py:45-78"""
        n_new = len(data)
        for k in _FIELDS:
            if k == "frame_avg_losses":
                continue
            if k == "T_WC_gt" and getattr(data, k, None) is None:
                continue
            setattr(self, k, self._expand(k, getattr(self, k, None), getattr(data, k, None), replace))
        # the losses follow the tensor that is there: on the device until the hot path moves them (losses_to_host)
        fal = self.frame_avg_losses
        empty = torch.zeros([n_new], device=data.depth_batch.device if fal is None else fal.device)
        self.frame_avg_losses = self._expand("frame_avg_losses", self.frame_avg_losses, empty, replace)
