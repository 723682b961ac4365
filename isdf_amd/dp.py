"""Data-parallel protocol of the training step: rays shard over ranks,
every rank fills the same flat fp32 buffer with SUMS (never means)

    [ grad sums (n_params) | loss sums (8: sdf, grad, eik, total, count, -, -, -) |
      block_loss (F*64) | block_cnt (F*64) ]

and ONE all-reduce(sum) over it (RCCL on GPUs; gloo in the CPU tests) makes
gradients, logged losses and the per-frame block averages global.  AdamW then
divides the summed gradient by the reduced element count, which reproduces the
single-process `mean` exactly even when ranks drop different numbers of
invalid-depth rays (sample.py:39-55).  No other collective is on the step path:
keyframe selection stays identical across ranks because its inputs are
(frame_avg_losses from the reduced bins; a private numpy stream seeded
identically by graft(), hot_path.HotPath._select_window), and the shared
virtual clock rides in the same message -- `world` extra floats at its tail,
one slot per rank holding that rank's PREVIOUS step time (hot_path.HotPath._step).
Per-FRAME traffic (new keyframe broadcast, keyframe decision) is separate.
"""
import torch

LS_SDF, LS_GRAD, LS_EIK, LS_TOTAL, LS_COUNT = 0, 1, 2, 3, 4
N_LOSS = 8


def layout(n_params, n_frames):
    """slices of the flat reduction buffer"""
    o = n_params
    return dict(grad=slice(0, o), loss=slice(o, o + N_LOSS),
                block_loss=slice(o + N_LOSS, o + N_LOSS + 64 * n_frames),
                block_cnt=slice(o + N_LOSS + 64 * n_frames, o + N_LOSS + 128 * n_frames),
                total=o + N_LOSS + 128 * n_frames)


def allreduce_(buf, group=None):
    """in-place sum over ranks of the flat buffer (no-op without a process group)"""
    if group is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return buf
    torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM, group=group)
    return buf


def rccl_direct(group, device):
    """(address of ncclAllReduce, ncclComm_t) of `group`'s RCCL communicator on `device`, or None when the group is not an RCCL
    group on a HIP device (gloo in the CPU tests; `ISDF_DP_COLLECTIVE=torch` forces None).  With it the step's collective is
    enqueued on the step's own stream by the C library (isdf_allreduce_sum_f32) instead of going through
    torch.distributed.all_reduce, which runs it on ProcessGroupNCCL's side stream behind two cross-stream event waits.
    The communicator stays torch's (created by the group's first collective -- graft() has broadcast the weights by then);
    torch's RCCL build is the one it belongs to, so ncclAllReduce is taken from that library."""
    import os
    if os.environ.get("ISDF_DP_COLLECTIVE", "").lower() == "torch" or torch.device(device).type != "cuda":
        return None
    try:
        if torch.distributed.get_backend(group) != "nccl":
            return None
        import ctypes
        backend = group._get_backend(torch.device(device))
        comm = int(backend._comm_ptr())
        lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value
        return (fn, comm) if fn and comm else None
    except Exception as e:      # an older torch without _comm_ptr, a statically linked RCCL ...: the framework's collective still works
        import warnings
        warnings.warn("isdf_amd: direct RCCL all-reduce unavailable (%r); using torch.distributed.all_reduce" % (e,))
        return None


def rccl_agree(lib, cand, group, device, stream_ptr):
    """The group's decision on the direct path, identical on every rank (a split decision would deadlock the first step):
    (1) every rank found its communicator handle (`cand` = rccl_direct(...), agreed with the framework's own collective);
    (2) ONE small all-reduce through the direct path -- every rank contributes rank + 1 in 64 floats, the sum is known -- came out
    right on every rank.  Otherwise all ranks use torch.distributed.all_reduce together.  This is where the direct path meets a
    world size > 1 for the first time on a multi-GPU node (one-GPU boxes can only form an RCCL group of one rank), so it is checked
    where it runs.  Collective: every rank of an RCCL group on HIP devices must call it (graft() does)."""
    import os, warnings
    if torch.device(device).type != "cuda" or torch.distributed.get_backend(group) != "nccl":
        return None
    if os.environ.get("ISDF_DP_COLLECTIVE", "").lower() == "torch":      # (an environment switch is the same on every rank of a launch)
        return None
    rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
    flag = torch.tensor([1.0 if cand is not None else 0.0], device=device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
    if float(flag.item()) != 1.0:
        return None
    fn, comm = cand
    t = torch.full((64,), float(rank + 1), dtype=torch.float32, device=device)
    torch.cuda.synchronize(device)
    rc = lib.isdf_allreduce_sum_f32(fn, comm, t.data_ptr(), 64, stream_ptr)
    torch.cuda.synchronize(device)
    ok = rc == 0 and bool((t == world * (world + 1) / 2.0).all().item())
    flag.fill_(1.0 if ok else 0.0)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
    if float(flag.item()) != 1.0:
        warnings.warn("isdf_amd: the direct RCCL all-reduce failed its self-test (rank %d: rc %d, sums right: %s); "
                      "every rank uses torch.distributed.all_reduce" % (rank, rc, ok))
        return None
    return cand


def new_split_event(device):
    """an event whose native handle exists (torch creates it at the first record)"""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


def allreduce_split_(buf, cut, split_event, side_stream, group=None):
    """The step's collective in TWO parts ("overlap the all-reduce of early-finished layers' dW"): the suffix
    buf[cut:] -- final at `split_event`, which isdf_train_step recorded between its two closing launches -- is reduced on
    `side_stream` while the second launch still fills the prefix buf[:cut]; the prefix follows in stream order.  Element-wise
    sums over the same ranks: at world size 2 the result equals the one-message all-reduce bit for bit (a + b); with more
    ranks RCCL's ring order depends on the message layout, so the two forms may differ by fp32 re-association, each of them
    identical on every rank.  The caller's stream waits for both parts before this returns."""
    if group is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return buf
    side_stream.wait_event(split_event)
    with torch.cuda.stream(side_stream):
        w1 = torch.distributed.all_reduce(buf[cut:], op=torch.distributed.ReduceOp.SUM, group=group, async_op=True)
    w2 = torch.distributed.all_reduce(buf[:cut], op=torch.distributed.ReduceOp.SUM, group=group, async_op=True)
    w1.wait()
    w2.wait()
    torch.cuda.current_stream(buf.device).wait_stream(side_stream)
    return buf


def finish(buf, n_params, n_frames):
    """What the step does with the reduced buffer: mean gradient, mean losses,
    block averages (loss.py:208-240).  Used by the CPU protocol tests; on the GPU
    the same arithmetic is inside isdf_adamw / isdf_frame_avg."""
    L = layout(n_params, n_frames)
    cnt = buf[L["loss"]][LS_COUNT]
    grad = buf[L["grad"]] / cnt
    losses = {k: buf[L["loss"]][i] / cnt for k, i in
              (("sdf_loss", LS_SDF), ("grad_loss", LS_GRAD), ("eikonal_loss", LS_EIK), ("total_loss", LS_TOTAL))}
    bc = buf[L["block_cnt"]].clone()
    bc[bc == 0] = 1.0
    approx = (buf[L["block_loss"]] / bc).view(n_frames, 8, 8)
    return grad, losses, approx, approx.sum(dim=(1, 2)) / 64.0


def rank_seed(seed, rank):
    """rank-distinct Philox key for the sampler (each rank draws its own rays)"""
    return int(seed) + 7919 * int(rank)


FAR = 1.0e18   # coordinate of a "no surface point here" slot: never the nearest point, never overflows d^2 to inf*0


def gather_surface_points(pc, n_valid, group=None):
    """bounds_method "pc" under data parallelism: all-gather the surface sample (sample 0) of
    every ray slot of every rank -> [world * max_rays, 3].  Ray slots past a rank's n_valid (invalid-depth
    rays dropped by the sampler, sample.py:39-55) are replaced by FAR so that no device->host sync on the
    valid count is needed; 12 KB per 1000 rays and rank."""
    R0 = pc.shape[0]
    surf = pc[:, 0, :].contiguous()
    live = torch.arange(R0, device=pc.device) < n_valid.reshape(-1)[0]
    surf = torch.where(live[:, None], surf, torch.full_like(surf, FAR))
    if group is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return surf
    world = torch.distributed.get_world_size(group)
    if torch.distributed.get_backend(group) == "gloo" and surf.is_cuda:
        # gloo moves device tensors only for broadcast / all_reduce: stage the 12 KB through the host (functional
        # tests of the N>1 path on a box with fewer GPUs than ranks; RCCL takes the branch below)
        host = torch.empty(world * R0, 3, dtype=surf.dtype)
        torch.distributed.all_gather_into_tensor(host, surf.cpu(), group=group)
        return host.to(surf.device)
    out = torch.empty(world * R0, 3, dtype=surf.dtype, device=surf.device)
    torch.distributed.all_gather_into_tensor(out, surf, group=group)
    return out


def src_rank(group=None):
    """global rank of the group's rank 0 (the broadcast source)"""
    if group is None or group is torch.distributed.group.WORLD:
        return 0
    return torch.distributed.get_global_rank(group, 0)


def broadcast_floats(values, group, device):
    t = torch.as_tensor([float(v) for v in values], dtype=torch.float64, device=device)
    torch.distributed.broadcast(t, src_rank(group), group=group)
    return [float(v) for v in t.cpu()]
