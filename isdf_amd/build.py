"""Build libisdf_hip.so (the C-ABI library of include/isdf_hip.h) with hipcc for
gfx950, in-tree, so the built library travels with the repo snapshot.

    python -m isdf_amd.build          # build if sources are newer than the .so
    python -m isdf_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libisdf_hip.so")
SOURCES = ["chain.hip", "fwd_pair.hip", "dw.hip", "sampler.hip", "optim.hip", "ingest.hip", "capi.hip"]
HEADERS = ["isdf_common.h", "chain_params.h", "chain_dev.h", "chain_debug.h", os.path.join("..", "..", "include", "isdf_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-command-line-argument",
         "-fno-gpu-rdc"] + os.environ.get("ISDF_EXTRA_HIPCC_FLAGS", "").split()


# per-source extra flags.  fwd_pair.hip: its stages are hand-interleaved streams of [MFMA, one epilogue element, ...] groups; the SLP
# vectoriser pairs the elements of neighbouring groups into packed fp32 operations, which undoes the interleave (and packed fp32
# VALU next to MFMAs measures slower than two scalar operations on this chip)
# The stages are fully unrolled streams of up to 48 k-steps, each with its share of the epilogue: beyond the default size limit of
# `#pragma unroll` (a rolled stage would index the weight window dynamically, i.e. put it in scratch memory).
# dw.hip: the embedding-rebuilding units evaluate two adjacent columns per thread; SLP-vectorised, their shared point operand becomes the
# broadcast half of a v_pk_fma_f32 -- the op_sel form isa_lint.py refuses (and packed fp32 VALU beside MFMAs is slower anyway).
PER_FILE = {"fwd_pair.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
            "dw.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "isa_lint.py")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + PER_FILE.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("---- %s failed ----\n%s\n" % (s, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    staged = LIB + ".staged"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", staged] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # instruction forms measured to misbehave on MI355X (isa_lint.py): a library that contains one is not installed
    from . import isa_lint
    lint_ran = True
    try:
        n_obj, bad = isa_lint.lint_library(staged)
    except (OSError, subprocess.CalledProcessError) as e:   # no llvm-objdump on this host: the check cannot run (tests/test_isa_lint.py
        n_obj, bad, lint_ran = 0, [], False                 # then skips too) -- never a reason to ship nothing
        sys.stderr.write("isa_lint could not run (%s)\n" % e)
    except RuntimeError as e:                               # the tool ran and extracted NO gfx950 code object (a bundler / toolchain
        if not os.environ.get("ISDF_SKIP_ISA_LINT"):        # format change): the gate fails CLOSED -- an unchecked library is not installed
            os.remove(staged)
            raise RuntimeError("isa_lint found nothing to check in the freshly linked library (%s); set ISDF_SKIP_ISA_LINT=1 to install "
                               "it unchecked" % e)
        n_obj, bad, lint_ran = 0, [], False
        sys.stderr.write("isa_lint could not run (%s) -- overridden by ISDF_SKIP_ISA_LINT\n" % e)
    skip = bool(os.environ.get("ISDF_SKIP_ISA_LINT"))
    if bad and not skip:
        os.remove(staged)
        raise RuntimeError("isa_lint: %d instruction(s) of a form measured to misbehave on MI355X, e.g.\n  %s\n      %s\n"
                           "(see isdf_amd/isa_lint.py; rewrite the source so the compiler does not pick that form)"
                           % (len(bad), bad[0][0], bad[0][1]))
    os.replace(staged, LIB)
    if verbose:
        if not lint_ran:
            print("isa_lint: NOT RUN (see stderr)")
        elif bad:
            print("isa_lint: %d code objects, %d flagged instruction(s) -- overridden by ISDF_SKIP_ISA_LINT" % (n_obj, len(bad)))
        else:
            print("isa_lint: %d code objects, clean" % n_obj)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
