"""The data-parallel step's collective as the product issues it on GPUs (`-m gpu`): RCCL's ncclAllReduce enqueued on the step's OWN
stream by the C library (isdf_allreduce_sum_f32; communicator and function address from dp.rccl_direct) instead of
torch.distributed.all_reduce on ProcessGroupNCCL's side stream.  A one-GPU box can only form an RCCL group of ONE rank (RCCL
refuses two ranks on a device; the two-rank protocol runs over gloo in test_dp_gpu.py), which exercises everything but the wire:
the communicator handle, the call through the C ABI, stream order against the kernels on either side.  Checked, in a child process
that owns the process group:

  * graft(dist_group=<nccl world of 1>) selects the direct form and step() never enters torch.distributed.all_reduce;
  * the same schedule with ISDF_DP_COLLECTIVE=torch (the framework's form) ends in the SAME parameters, moments, keyframe losses
    and loss sums bit for bit -- as it must: same kernels, a sum over one rank;
  * isdf_allreduce_sum_f32 rejects null arguments with ISDF_EINVAL and reports a refusing collective as ISDF_ECOLLECTIVE.
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import contextlib, io, os, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1)
from bench_support.standin_trainer import HipTrainer
from isdf_amd import synthetic
from tests.accuracy_experiment import config
CAM = dict(H=120, W=160, fx=150.0, fy=150.0, cx=79.5, cy=59.5)
cfg = config(CAM)
cfg["sample"].update(n_rays=64, n_rays_is_kf=128)
traj = synthetic.trajectory(20)
out = {{}}
for mode in ("direct", "torch"):
    os.environ["ISDF_DP_COLLECTIVE"] = "" if mode == "direct" else "torch"
    np.random.seed(5); torch.manual_seed(5)
    tr = HipTrainer("cuda:0", cfg, inv_bounds_transform=synthetic.bounds_transform(), rng="philox", seed=6,
                    dist_group=torch.distributed.group.WORLD, virtual_step_ms=10.0)
    out[mode + "_collective"] = np.array(tr._hip.collective)
    rng = np.random.RandomState(3)
    counts = {{"n": 0}}
    orig = torch.distributed.all_reduce
    def counted(*a, **k):
        counts["n"] += 1
        return orig(*a, **k)
    ls = []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(3):
            fr = tr.make_frame(k * 5, synthetic.render_depth(traj[k * 5], CAM, rng, noise_std=0.01), traj[k * 5])
            tr.last_is_keyframe = True
            tr.add_frame(fr)
            tr.noise_std = tr.noise_kf
            torch.distributed.all_reduce = counted
            for _ in range(3):
                losses, _ = tr.step()
                ls.append(float(losses["total_loss"]))
            torch.distributed.all_reduce = orig
    torch.cuda.synchronize()
    eng = tr.engine
    out.update({{mode + "_params": eng.params.cpu().numpy(), mode + "_m": eng.exp_avg.cpu().numpy(), mode + "_v": eng.exp_avg_sq.cpu().numpy(),
                mode + "_fal": tr.frames.frame_avg_losses.cpu().numpy(), mode + "_ls": np.array(ls), mode + "_torch_calls": np.array(counts["n"])}})
# the entry point's argument checks and error path
import ctypes as C
from isdf_amd import _ffi
lib = _ffi.lib()
buf = torch.zeros(16, device="cuda:0")
fn, comm = tr._hip.rccl if tr._hip.rccl is not None else (0, 0)
out["einval_fn"] = np.array(lib.isdf_allreduce_sum_f32(None, 1, buf.data_ptr(), 16, None))
out["einval_count"] = np.array(lib.isdf_allreduce_sum_f32(1, 1, buf.data_ptr(), 0, None))
REFUSE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)(lambda *a: 5)
out["ecollective"] = np.array(lib.isdf_allreduce_sum_f32(C.cast(REFUSE, C.c_void_p).value, 1, buf.data_ptr(), 16, None))
out["ecollective_text"] = np.array(lib.isdf_error_string(int(out["ecollective"])).decode())
np.savez({path!r}, **out)
torch.distributed.destroy_process_group()
"""


def test_step_collective_is_rccl_on_the_steps_stream():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    path = os.path.join(tempfile.mkdtemp(), "out.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, port=port, path=path)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    o = dict(np.load(path))
    assert str(o["direct_collective"]).startswith("rccl on the step's stream"), o["direct_collective"]
    assert str(o["torch_collective"]) == "torch.distributed.all_reduce"
    assert int(o["direct_torch_calls"]) == 0 and int(o["torch_torch_calls"]) == 9          # nine steps, one collective each
    for k in ("params", "m", "v", "fal", "ls"):
        assert np.isfinite(o["direct_" + k]).all(), k
        assert np.array_equal(o["direct_" + k], o["torch_" + k]), k
    from isdf_amd import _ffi
    assert int(o["einval_fn"]) == int(o["einval_count"]) == -1                              # ISDF_EINVAL
    assert int(o["ecollective"]) == -5 and "ncclResult_t 5" in str(o["ecollective_text"])   # ISDF_ECOLLECTIVE
    assert _ffi.ABI_VERSION >= 7
