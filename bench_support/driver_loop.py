"""The frame-scheduling loop of the reference's drivers (`isdf/train/train.py:86-136`; the same logic is in
`train_vis.py:20-62` and `batch_utils.py:64-197`), restated for tests and experiments (TEST INFRASTRUCTURE):
it only talks to the public `Trainer` surface

    steps_since_frame, optim_frames, check_keyframe_latest(), get_latest_frame_id(), get_data([id]) /
    add_frame(frame_data), last_is_keyframe, step()

so it runs unchanged against a grafted reference `Trainer`, a `HipTrainer`, or a CPU port."""


def run_train_loop(trainer, get_frame, size_dataset, n_steps, incremental=True, on_step=None, t0=0, first_frame_iters=200):
    """get_frame(frame_id) -> FrameData for `trainer.add_frame` (the drivers call trainer.get_data([id])).
    t0: step counter to continue from (t == 0 is the drivers' special first iteration).
    Returns (next step counter, list of (step, frame_id) ingests, last losses)."""
    ingests, losses = [], None
    t = t0
    for t in range(t0, t0 + n_steps):
        finish_optim = trainer.steps_since_frame == trainer.optim_frames            # train.py:102-103
        if incremental and (finish_optim or t == 0):
            if t == 0:
                add_new_frame = True
            else:
                add_new_frame = trainer.check_keyframe_latest()                      # train.py:109
            if add_new_frame:
                new_frame_id = trainer.get_latest_frame_id()                         # train.py:112
                if new_frame_id >= size_dataset:
                    return t, ingests, losses
                trainer.add_frame(get_frame(new_frame_id))                           # train.py:122-123
                ingests.append((t, new_frame_id))
                if t == 0:
                    trainer.last_is_keyframe = True                                  # train.py:125-127
                    trainer.optim_frames = first_frame_iters                         # 200 in the drivers; CPU tests shorten it
        losses, step_time = trainer.step()                                           # train.py:136
        if on_step is not None:
            on_step(t, losses, step_time)
    return t + 1, ingests, losses
