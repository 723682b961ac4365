"""CPU-side checks of the C-ABI library: it loads, exports every symbol
include/isdf_hip.h declares, and its size queries (pure host code) are right.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np

import pytest

from isdf_amd import _ffi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _ffi.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "isdf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(isdf_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(_ffi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_abi_version_and_errors(lib):
    import re
    from isdf_amd import _ffi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "isdf_hip.h")).read()
    assert lib.isdf_abi_version() == _ffi.ABI_VERSION == int(re.search(r"#define ISDF_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.isdf_error_string(0) == b"ok"
    assert b"invalid" in lib.isdf_error_string(-1)


def test_size_queries_default_net(lib):
    from isdf_amd.engine import NetConfig
    c = NetConfig().to_c()
    assert lib.isdf_param_count(C.byref(c)) == 460033          # probe of the reference net
    assert lib.isdf_reduce_floats(C.byref(c), 5) == 460033 + 8 + 2 * 5 * 64
    sh = lib.isdf_shadow_bytes(C.byref(c))
    fwd = 256 * 256 + 4 * 256 * 256 + 256 * 512
    bwd = 5 * 256 * 256 + 256 * 512
    assert sh == 2 * (3 * fwd + 2 * bwd)      # default "fp16x2": + one forward set of fp16 weight residuals
    for op, sets in (("fp16", 2), ("bf16", 2)):
        assert lib.isdf_shadow_bytes(C.byref(NetConfig(fwd_operand=op).to_c())) == 2 * (sets * fwd + 2 * bwd)
    full = NetConfig(fwd_operand="fp16x2_full").to_c()       # exact-forward instrument: same residual set, used by every layer
    assert lib.isdf_shadow_bytes(C.byref(full)) == sh and lib.isdf_check_net(C.byref(full)) == 0
    for wide in (NetConfig(fwd_operand="fp16x2_full", n_freqs=9, blocks=3), NetConfig(fwd_operand="fp16x2_full", hidden=512, n_freqs=10)):
        assert lib.isdf_check_net(C.byref(wide.to_c())) == -2     # ISDF_EUNSUPPORTED: four operand regions do not fit those tiles
    # any hidden_feature_size up to 512 runs zero-padded on the 256 / 512 tiles; the parameter vector keeps the reference's shapes
    for hidden, blocks, nf in ((64, 1, 6), (64, 3, 11), (128, 2, 6), (300, 2, 6)):
        nc = NetConfig(hidden=hidden, blocks=blocks, n_freqs=nf)
        assert lib.isdf_check_net(C.byref(nc.to_c())) == 0
        assert lib.isdf_param_count(C.byref(nc.to_c())) == sum(int(np.prod(sh)) for _, sh in nc.param_shapes())
    assert lib.isdf_check_net(C.byref(NetConfig(hidden=513).to_c())) == -2
    bad = NetConfig().to_c()
    bad.fwd_operand = 4
    assert lib.isdf_shadow_bytes(C.byref(bad)) == -1
    assert lib.isdf_workspace_bytes(C.byref(c), 27000, 1) > lib.isdf_workspace_bytes(C.byref(c), 27000, 0) > 0


def test_invalid_arguments_are_reported_not_crashed(lib):
    assert lib.isdf_param_count(None) == -1
    from isdf_amd.engine import NetConfig
    bad = NetConfig(blocks=0).to_c()
    assert lib.isdf_param_count(C.byref(bad)) == -1
    assert lib.isdf_sample_rays(None, None, None, 0, None) == -1
    assert lib.isdf_adamw(C.byref(NetConfig().to_c()), None, None, None, None, None, 1.0, 1e-3, 0.9, 0.999,
                          1e-8, 0.0, 1, None, None) == -1


def test_param_layout_matches_reference_state_dict_order():
    from isdf_amd.engine import NetConfig
    import oracle.isdf_oracle as orc
    import numpy as np
    net = NetConfig()
    ref = orc.init_params(256, 2, 6, np.random.RandomState(0))
    assert [k for k, _ in net.param_shapes()] == list(ref.keys())
    assert all(tuple(ref[k].shape) == tuple(s) for k, s in net.param_shapes())


def test_engine_refuses_cpu():
    from isdf_amd.engine import Engine, NetConfig
    with pytest.raises(_ffi.IsdfError):
        Engine(NetConfig(), "cpu")


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """sizeof / offsetof of every struct that crosses the boundary, as gcc lays out include/isdf_hip.h, against the ctypes
    mirrors in isdf_amd/_ffi.py (a silent mismatch would hand the kernels shifted pointers)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"isdf_net_cfg": _ffi.NetCfg, "isdf_sample_args": _ffi.SampleArgs, "isdf_sample_out": _ffi.SampleOut,
               "isdf_loss_cfg": _ffi.LossCfg, "isdf_step_args": _ffi.StepArgs, "isdf_step_out": _ffi.StepOut,
               "isdf_optim_args": _ffi.OptimArgs}
    last = {"isdf_net_cfg": "spill_operand", "isdf_sample_args": "n_inline", "isdf_sample_out": "pc", "isdf_loss_cfg": "orien_loss",
            "isdf_step_args": "extra_value", "isdf_step_out": "split_event", "isdf_optim_args": "frame_avg_inline_n"}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "isdf_hip.h"', 'int main(void) {']
    for name in structs:
        src.append('  printf("%s %%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (name, name, name, last[name]))
    src += ['  return 0;', '}']
    c = tmp_path / "sizes.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(exe)])
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        name, size, off = line.split()
        ct = structs[name]
        assert C.sizeof(ct) == int(size), (name, C.sizeof(ct), size)
        assert getattr(ct, last[name]).offset == int(off), (name, last[name])


def test_allreduce_entry_point_checks_its_arguments_and_reports_a_refusing_collective():
    """isdf_allreduce_sum_f32 (the data-parallel step's collective on the step's own stream): no collective library is linked, the
    caller passes RCCL's ncclAllReduce -- here a stand-in that refuses, which never touches the buffer, so this runs without a GPU."""
    import ctypes as C
    lib = _ffi.lib()
    buf = (C.c_float * 16)()
    assert lib.isdf_allreduce_sum_f32(None, 1, C.addressof(buf), 16, None) == -1          # ISDF_EINVAL: no function
    assert lib.isdf_allreduce_sum_f32(1, None, C.addressof(buf), 16, None) == -1          # ... no communicator
    assert lib.isdf_allreduce_sum_f32(1, 1, None, 16, None) == -1
    assert lib.isdf_allreduce_sum_f32(1, 1, C.addressof(buf), 0, None) == -1
    seen = []

    def refuse(send, recv, count, dtype, op, comm, stream):
        seen.append((send, recv, count, dtype, op, comm))
        return 5
    fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)(refuse)
    rc = lib.isdf_allreduce_sum_f32(C.cast(fn, C.c_void_p).value, 0x1234, C.addressof(buf), 16, None)
    assert rc == -5 and b"ncclResult_t 5" in lib.isdf_error_string(rc)                   # ISDF_ECOLLECTIVE
    # in place, fp32 (ncclFloat32 = 7), sum (ncclSum = 0), the caller's communicator
    assert seen == [(C.addressof(buf), C.addressof(buf), 16, 7, 0, 0x1234)]


def test_rccl_direct_is_off_for_non_rccl_groups():
    from isdf_amd import dp
    assert dp.rccl_direct(None, "cpu") is None
