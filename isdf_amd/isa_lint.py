#!/usr/bin/env python3
"""Static check of the built device code (every gfx950 code object inside libisdf_hip.so) for instruction forms this project has
MEASURED to misbehave on MI355X.  isdf_amd/build.py runs it on the freshly linked library BEFORE installing it in-tree (a violating build is refused);
tests/test_isa_lint.py runs it on the library the tests load.

Rule 1 -- packed fp32 op whose LOW result reads the HIGH dword of a VGPR src1 pair (VOP3P `op_sel:[x,1,..]`, e.g.
    v_pk_fma_f32 v[34:35], v[52:53], v[34:35], v[42:43] op_sel:[0,1,0]
the form LLVM's SLP vectoriser picks for "pair (p0, p1) times a broadcast of the odd element of a register pair").  Round 4: with two
workgroups resident on a CU the product of the LOW half was intermittently dropped (low result = src2) in lanes 48-63, nowhere else:
  * found as sdf values off by 1e-3 .. 1e-2 in rows 16-31 of a tile, non-deterministically, only in builds whose output-layer
    reduction the compiler happened to vectorise that way (tests/fwd_race_probe.py: the errors decompose into single missing
    w_out[u] * a[u] terms, u in the lane half / odd elements those 7 instructions handle; the 8th odd element, which the compiler
    encoded with op_sel_hi instead, never failed);
  * ISA-level A/B on the failing build, nothing else changed (profiles/r04_pk_fma_opsel_erratum.txt): s_nop 3 before or after the
    instructions -- still failing; the same instruction with src0 and src1 swapped (op_sel:[1,0,0]) -- clean; two v_fma_f32 -- clean;
    one workgroup per CU -- clean.
  * reproduced in ISOLATION (tools/probes/pk_fma_opsel.hip, profiles/r04_pk_fma_opsel_probe.txt): a loop of the instruction against
    scalar FMAs while ANOTHER workgroup on the CU runs MFMAs -- 1.6e4 .. 3.4e4 wrong low results per 5e8, lanes 48-63 only, each equal
    to src2's low dword exactly (the product is dropped); v_pk_mul_f32 / v_pk_add_f32 op_sel:[0,1] also hit, rarely; 0 for the
    selector-on-src0 form, 0 without co-resident MFMA work (checkers alone, LDS-only, store-only, v_fma-only company).
An SGPR-pair src1 with the same selector (the sampler's `v_pk_add_f32 ..., s[10:11] op_sel:[0,1]`) has run bit-exact against the
reference at every size since round 1 and is not flagged.

usage: python -m isdf_amd.isa_lint [path/to/lib.so]   -> exit status 1 and the offending lines if a rule fires"""
import os, re, subprocess, sys, tempfile

LLVM_BIN = os.environ.get("ISDF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
# VOP3P packed fp32 ops: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (+ v_pk_mov_b32 moves data only).  operands: vdst, src0, src1[, src2]
_PK = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\s+(\S+),\s*(\S+),\s*(\S+?)(?:,\s*(\S+))?\s+(.*)$")
_OPSEL = re.compile(r"\bop_sel:\[([01]),([01])")


def code_objects(lib, workdir):
    """Extract the gfx950 code objects of a HIP shared library (llvm-objdump --offloading writes them next to the input)."""
    local = os.path.join(workdir, os.path.basename(lib))
    with open(lib, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], stderr=subprocess.STDOUT)
    return sorted(os.path.join(workdir, n) for n in os.listdir(workdir) if "amdgcn" in n)


def disassemble(obj):
    return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", obj],
                                   stderr=subprocess.DEVNULL).decode("utf-8", "replace")


def lint_text(text):
    """-> [(kernel, instruction)] violating rule 1."""
    out, kernel = [], "?"
    for line in text.split("\n"):
        if line.endswith(">:") and "<" in line:
            kernel = line[line.index("<") + 1:-2]
            continue
        if "v_pk_" not in line or "op_sel:" not in line:
            continue
        ins = line.split("//")[0].strip()
        m = _PK.search(ins)
        s = _OPSEL.search(ins)
        if not m or not s:
            continue
        src1 = m.group(4)
        if s.group(2) == "1" and src1[:1] in ("v", "a"):      # low result <- high dword of a VGPR (or AGPR) src1 pair
            out.append((kernel, ins))
    return out


def kernel_resources(lib):
    """{kernel name: dict(vgpr, sgpr, vgpr_spill, sgpr_spill, scratch)} from the code objects' metadata notes -- the facts the
    kernels' design leans on (no scratch anywhere; the <256, 256> chain kernels within the 128-VGPR budget of two workgroups per CU),
    checked by tests/test_isa_lint.py so that a compiler or source change cannot take them away silently."""
    keys = {".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
            ".private_segment_fixed_size": "scratch"}
    out = {}
    with tempfile.TemporaryDirectory(prefix="isdf_lint_") as wd:
        for o in code_objects(lib, wd):
            txt = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", o], stderr=subprocess.DEVNULL).decode("utf-8", "replace")
            cur = None
            for line in txt.split("\n"):
                t = line.strip()
                if t.startswith("- ."):              # first key of a new kernel record (or of an argument record: no .name follows)
                    cur = {}
                    t = t[2:]
                if cur is None or ":" not in t:
                    continue
                k, v = t.split(":", 1)
                if k == ".name" and "(" not in v and v.strip().startswith("_Z"):
                    cur["name"] = v.strip()
                elif k in keys:
                    cur[keys[k]] = int(v)
                if k == ".wavefront_size" and "name" in cur:      # last key of a kernel record
                    out[cur.pop("name")] = cur
                    cur = None
    return out


_ADDR = re.compile(r"//\s*([0-9A-Fa-f]{8,16}):")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")


def scratch_in_loops_text(text):
    """-> [(kernel, instruction)]: scratch (private-segment) accesses that sit INSIDE a loop (a cycle of the kernel's control-flow
    graph, rebuilt from the disassembly).  A register spilled around a loop costs a store and a load per launch; one reloaded inside
    the MFMA stream waits behind every global load in flight (scratch shares vmcnt, which retires in order)."""
    out = []

    def analyse(kernel, base, ins):          # ins: [(addr, text, branch target or None)]
        if not any(t.startswith("scratch_") for _, t, _ in ins):
            return
        addrs = [a for a, _, _ in ins]
        leaders = {addrs[0]} | {tg for _, _, tg in ins if tg is not None}
        for k, (a, t, tg) in enumerate(ins[:-1]):
            if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                leaders.add(addrs[k + 1])
        starts = sorted(x for x in leaders if x in set(addrs))
        import bisect
        block_of = lambda a: bisect.bisect_right(starts, a) - 1
        n = len(starts)
        succ = [set() for _ in range(n)]
        for k, (a, t, tg) in enumerate(ins):
            b = block_of(a)
            last = k + 1 == len(ins) or block_of(addrs[k + 1]) != b
            if tg is not None and tg in set(starts):
                succ[b].add(block_of(tg))
            if last and k + 1 < len(ins) and not t.startswith(("s_branch", "s_endpgm", "s_setpc")):
                succ[b].add(block_of(addrs[k + 1]))
        # Tarjan, iterative
        index, low, on, stack, comp, cnt = [None] * n, [0] * n, [False] * n, [], [None] * n, [0]
        for root in range(n):
            if index[root] is not None:
                continue
            work = [(root, iter(sorted(succ[root])))]
            index[root] = low[root] = cnt[0]; cnt[0] += 1; stack.append(root); on[root] = True
            while work:
                v, it = work[-1]
                adv = False
                for w in it:
                    if index[w] is None:
                        index[w] = low[w] = cnt[0]; cnt[0] += 1; stack.append(w); on[w] = True
                        work.append((w, iter(sorted(succ[w])))); adv = True
                        break
                    if on[w]:
                        low[v] = min(low[v], index[w])
                if adv:
                    continue
                work.pop()
                if work:
                    low[work[-1][0]] = min(low[work[-1][0]], low[v])
                if low[v] == index[v]:
                    members = []
                    while True:
                        w = stack.pop(); on[w] = False; members.append(w)
                        if w == v:
                            break
                    cyc = len(members) > 1 or v in succ[v]
                    for w in members:
                        comp[w] = cyc
        for a, t, _ in ins:
            if t.startswith("scratch_") and comp[block_of(a)]:
                out.append((kernel, t))

    kernel, base, ins = None, 0, []
    for line in text.split("\n"):
        if line.endswith(">:") and "<" in line:
            if kernel is not None and ins:
                analyse(kernel, base, ins)
            kernel, base, ins = line[line.index("<") + 1:-2], int(line.split()[0], 16), []
            continue
        m = _ADDR.search(line)
        if not m or kernel is None:
            continue
        t = line.split("//")[0].strip()
        tg = None
        if t.startswith(("s_cbranch", "s_branch")):
            mt = _TARGET.search(line)
            tg = base + int(mt.group(1), 16) if mt else None
        ins.append((int(m.group(1), 16), t, tg))
    if kernel is not None and ins:
        analyse(kernel, base, ins)
    return out


def scratch_in_loops(lib):
    with tempfile.TemporaryDirectory(prefix="isdf_lint_") as wd:
        bad = []
        for o in code_objects(lib, wd):
            bad += scratch_in_loops_text(disassemble(o))
        return bad


def lint_library(lib):
    with tempfile.TemporaryDirectory(prefix="isdf_lint_") as wd:
        objs = code_objects(lib, wd)
        if not objs:
            raise RuntimeError("no gfx950 code object found in %s" % lib)
        bad = []
        for o in objs:
            bad += lint_text(disassemble(o))
        return len(objs), bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libisdf_hip.so")
    n, bad = lint_library(lib)
    if bad:
        print("%s: %d instruction(s) of a form measured to misbehave on MI355X (isdf_amd/isa_lint.py, rule 1):" % (lib, len(bad)))
        for k, ins in bad[:40]:
            print("  %s\n      %s" % (k, ins))
        return 1
    print("%s: %d code objects, clean" % (lib, n))
    return 0


if __name__ == "__main__":
    sys.exit(main())
