"""The built device code contains no instruction form this project has measured to misbehave on MI355X (isdf_amd/isa_lint.py:
round 4's `v_pk_fma_f32 ... op_sel:[0,1,0]` finding).  CPU-side: the library is disassembled, nothing is launched."""
import os
import shutil

import pytest

from isdf_amd import _ffi, build, isa_lint

BAD = [
    "\tv_pk_fma_f32 v[34:35], v[52:53], v[34:35], v[42:43] op_sel:[0,1,0]      // 000000012340: D3B04022 1C8A6934",
    "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,1]",
    "\tv_pk_add_f32 v[12:13], v[12:13], v[10:11] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "\tv_pk_fma_f32 v[2:3], v[4:5], a[6:7], v[8:9] op_sel:[0,1,0]",
]
GOOD = [
    "\tv_pk_fma_f32 v[34:35], v[34:35], v[52:53], v[42:43] op_sel:[1,0,0]",                    # the same product, selector on src0
    "\tv_pk_fma_f32 v[42:43], v[50:51], v[34:35], v[72:73] op_sel_hi:[1,0,1]",                 # high result <- low dword of src1: never failed
    "\tv_pk_fma_f32 v[6:7], v[26:27], v[18:19], v[34:35]",
    "\tv_pk_add_f32 v[12:13], v[12:13], s[10:11] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",   # SGPR pair (the sampler): bit-exact since round 1
    "\tv_pk_mul_f32 v[44:45], s[58:59], v[18:19]",
    "\tv_pk_fma_f16 v2, v3, v4, v5 op_sel:[0,1,0]",                                            # not a packed-fp32 op
]


def test_rule_flags_exactly_the_measured_form():
    text = "0000000000001b00 <kernel_a>:\n" + "\n".join(BAD) + "\n0000000000002b00 <kernel_b>:\n" + "\n".join(GOOD)
    hits = isa_lint.lint_text(text)
    assert [k for k, _ in hits] == ["kernel_a"] * len(BAD)
    assert [i.split()[0] for _, i in hits] == ["v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"]


def test_built_library_is_clean():
    if not os.path.exists(os.path.join(isa_lint.LLVM_BIN, "llvm-objdump")) and not shutil.which("llvm-objdump"):
        pytest.skip("no llvm-objdump on this host")
    build.build(verbose=False)
    n, bad = isa_lint.lint_library(_ffi.LIB_PATH)
    assert n >= 5                     # one code object per .hip source
    assert not bad, bad[:4]


def test_hot_kernels_keep_their_register_budgets():
    """What the kernels' design leans on: no scratch and no spilled VGPRs in any kernel of the step, and the <256, 256>
    chain kernels inside the 128-VGPR budget that lets two workgroups share a CU."""
    if not os.path.exists(os.path.join(isa_lint.LLVM_BIN, "llvm-readelf")):
        pytest.skip("no llvm-readelf on this host")
    build.build(verbose=False)
    res = isa_lint.kernel_resources(_ffi.LIB_PATH)
    hot = {k: v for k, v in res.items() if any(t in k for t in ("chain_kernel", "dw_kernel", "step_tail_kernel", "sample_rays_kernel"))}
    assert len(hot) >= 40, len(hot)
    # The dW kernel of the e4m3 spill formats (dw_kernel<256, true, 3>) holds three stage pipelines (PE-rebuilding, regular and
    # top-layer units) in one 512-register wave; the compiler parks a few thread constants in scratch AROUND the stage loops
    # (one store and one load per launch).  Everything else -- and every loop of every kernel -- is scratch-free.
    parked = "dw_kernelILi256ELb1ELi3EE"
    for k, v in hot.items():
        if parked in k:
            assert v["scratch"] <= 128 and v["vgpr_spill"] <= 24, (k, v)
        else:
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
    # round 6: no other kernel of the library uses scratch memory (the keyframe test's per-ray sort was the last one: ingest.hip)
    assert not {k: v for k, v in res.items() if (v["scratch"] or v["vgpr_spill"]) and parked not in k}
    assert not isa_lint.scratch_in_loops(_ffi.LIB_PATH)
    one_wave_per_simd = [v["vgpr"] for k, v in hot.items() if "dw_kernel" in k]
    assert len(one_wave_per_simd) == 6 and max(one_wave_per_simd) <= 512
    two_per_cu = [v["vgpr"] for k, v in hot.items() if "chain_kernelILi256ELi256ELi" in k and "ELi256ELi3E" not in k]    # (OPER 3: one per CU)
    assert len(two_per_cu) >= 9 and max(two_per_cu) <= 128, two_per_cu


def test_scratch_in_loops_is_cfg_based():
    """isa_lint.scratch_in_loops: a spill AROUND a loop is fine, one INSIDE a cycle of the control-flow graph is flagged -- and a
    backward branch that merely jumps to an earlier block (code layout) does not make everything in between a loop."""
    text = "\n".join([
        "0000000000001000 <k>:",
        "\tscratch_store_dword off, v1, off offset:4          // 000000001000: DC000000",      # before the loop: parked
        "\ts_nop 0                                            // 000000001008: BF800000",
        "\tscratch_load_dword v2, off, off                    // 00000000100C: DC000000",      # loop body (0x100c .. 0x1014): flagged
        "\ts_cbranch_scc1 65533                               // 000000001014: BF85FFFD <k+0xc>",
        "\ts_branch 3                                         // 000000001018: BF820003 <k+0x28>",   # forward jump over the next block
        "\tscratch_load_dword v1, off, off offset:4           // 00000000101C: DC000000",      # reached only by the backward jump below: no cycle
        "\ts_endpgm                                           // 000000001024: BF810000",
        "\ts_branch 65532                                     // 000000001028: BF82FFFC <k+0x1c>",
    ])
    bad = isa_lint.scratch_in_loops_text(text)
    assert bad == [("k", "scratch_load_dword v2, off, off")], bad
