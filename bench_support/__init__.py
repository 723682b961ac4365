"""Driver-side support for bench.py, the GPU tests and the accuracy experiments -- NOT part of the product package.

`standin_trainer.HipTrainer` stands in for the reference's `Trainer` where /root/reference cannot be imported (the GPU box): the
handful of driver-side methods the hot path needs around it (`add_frame`, `check_keyframe_latest`, `select_keyframes`, restated
from trainer.py:574-582,622-674) with `isdf_amd.hot_path.graft` applied, exactly as it is applied to the real `Trainer` in
tests/test_graft_reference.py.  `driver_loop.run_train_loop` restates the reference's frame scheduling (train.py:86-136)."""
