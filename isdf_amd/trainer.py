"""`HipTrainer`: the reference's `Trainer` surface with the training hot path on HIP kernels.

    HipTrainer(device, config, ...)  ==  graft(StandinTrainer(device, config, ...), ...)

i.e. the product's one binding (`isdf_amd.hot_path.graft`, the code INTEGRATION.md shows applied to the
reference's own `Trainer`) applied to the in-repo stand-in for the reference's driver-side methods
(`isdf_amd.standin`), for hosts where /root/reference does not exist (the GPU box: bench.py, `-m gpu` tests).
Method names, argument meaning, return contracts and side effects are the reference's:

    step()               trainer.py:951-1016      hot_path.HotPath (HIP)
    sample_points()      trainer.py:683-766       hot_path.HotPath (HIP)
    sdf_eval_and_loss()  trainer.py:768-868       hot_path.HotPath (HIP)
    is_keyframe()        trainer.py:586-620       hot_path.HotPath (HIP)
    add_frame / check_keyframe_latest / select_keyframes / get_latest_frame_id    standin.StandinTrainer
    FrameData            isdf/datasets/data_util.py:11-102                        standin.FrameData
"""
from .hot_path import FlatAdamW, HotPath, StepLosses, graft      # noqa: F401  (public surface)
from .standin import FrameData, StandinTrainer                   # noqa: F401


class HipTrainer(HotPath, StandinTrainer):
    def __init__(self, device, config, incremental=True, inv_bounds_transform=None, rng="philox",
                 seed=1, dist_group=None, fix_normal_window=False, fwd_operand="fp16x2", virtual_step_ms=None,
                 engine_factory=None, overlap_allreduce=False):
        """config: path to / dict with the reference's JSON schema (replicaCAD.json).
        rng: "philox" (in-kernel, no host sync) or "torch" (draw with torch in the
        reference's order and shapes -- parity mode, one host sync per step)."""
        self._hip = None
        StandinTrainer.__init__(self, device, config, None, incremental, inv_bounds_transform=inv_bounds_transform,
                                fwd_operand=fwd_operand, engine_factory=engine_factory)
        graft(self, rng=rng, seed=seed, dist_group=dist_group, fix_normal_window=fix_normal_window,
              fwd_operand=fwd_operand, virtual_step_ms=virtual_step_ms, engine_factory=engine_factory,
              overlap_allreduce=overlap_allreduce)

    # aliases kept for checkpoint files / callers of round 1
    def state_dict(self):
        return self.hip_state_dict()

    def load_state_dict(self, sd):
        return self.load_hip_state_dict(sd)
