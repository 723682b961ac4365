"""Host logic: the pre-allocated keyframe store keeps data_util.FrameData's contract
(isdf/datasets/data_util.py:45-102: append, or overwrite the last slot) without re-concatenating."""
import copy

import numpy as np
import torch

from isdf_amd.frame_store import FrameData


def _frame(i, H=6, W=8):
    g = torch.Generator().manual_seed(i)
    return FrameData(frame_id=np.array([i]), depth_batch=torch.rand(1, H, W, generator=g),
                     T_WC_batch=torch.rand(1, 4, 4, generator=g), normal_batch=torch.rand(1, H, W, 3, generator=g))


def test_matches_concatenation_semantics_and_reuses_storage():
    store = FrameData()
    ref = dict(frame_id=[], depth_batch=[], T_WC_batch=[], normal_batch=[])
    # schedule: first frame is a keyframe; then alternate "promoted" / "not promoted" like train.py:86-136
    pattern = [False, False, True, False, True, True, False, False, False, True, False, False]
    ptrs = set()
    for i, replace in enumerate(pattern):
        f = _frame(i)
        store.add_frame_data(f, replace and i > 0)
        for k in ref:
            v = getattr(f, k)
            if replace and i > 0:
                ref[k][-1] = v
            else:
                ref[k].append(v)
        ptrs.add(store.depth_batch.data_ptr())
        assert len(store) == len(ref["frame_id"])
        np.testing.assert_array_equal(store.frame_id, np.concatenate(ref["frame_id"]))
        for k in ("depth_batch", "T_WC_batch", "normal_batch"):
            assert torch.equal(getattr(store, k), torch.cat(ref[k])), (i, k)
        assert store.frame_avg_losses.shape == (len(store),)
    # 9 appends with doubling capacity (8, 16): far fewer buffers than appends
    assert len(ptrs) <= 2, ptrs


def test_in_place_updates_survive_growth_and_deepcopy_is_detached():
    store = FrameData()
    for i in range(3):
        store.add_frame_data(_frame(i), False)
    store.frame_avg_losses[torch.tensor([0, 2])] = torch.tensor([0.5, 0.25])
    snap = copy.deepcopy(store)
    for i in range(3, 12):                      # forces one re-allocation
        store.add_frame_data(_frame(i), False)
    assert store.frame_avg_losses[:3].tolist() == [0.5, 0.0, 0.25]
    store.depth_batch[0].zero_()
    assert len(snap) == 3 and snap.depth_batch[0].abs().sum() > 0
    # a store assembled from full tensors (bench.py / checkpoint restore) keeps working
    ext = FrameData(frame_id=np.arange(2), depth_batch=torch.ones(2, 6, 8), T_WC_batch=torch.ones(2, 4, 4),
                    normal_batch=torch.ones(2, 6, 8, 3), frame_avg_losses=torch.zeros(2))
    ext.add_frame_data(_frame(9), False)
    assert len(ext) == 3 and torch.equal(ext.depth_batch[:2], torch.ones(2, 6, 8))
