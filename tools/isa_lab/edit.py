"""ISA edits of the v_pk_fma_f32 ... op_sel:[0,1,0] instructions of ONE kernel (chain_kernel<256, 256, fp16x2, MODE 1>) in a
save-temps assembly file: nop_after | nop_before (s_nop 3 around each), scalar (two v_fma_f32), swap01 (src0 <-> src1, the
selector moves to src0).  usage: edit.py <mode> in.s out.s"""
import re, sys
mode, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
K = "_ZN4isdf12chain_kernelILi256ELi256ELi2ELi1ELb0EEEvNS_11ChainParamsE"
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(K + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
pat = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]\s*$")
out, n = [], 0
for i, l in enumerate(lines):
    m = pat.match(l) if start <= i < end else None
    if not m:
        out.append(l); continue
    n += 1
    d0, d1, a0, a1, b0, b1, c0, c1 = map(int, m.groups())
    if mode == "nop_after":
        out += [l, "\ts_nop 3"]
    elif mode == "nop_before":
        out += ["\ts_nop 3", l]
    elif mode == "scalar":
        assert d0 != b1 and d0 != a1 and d0 != c1, l      # the first write must not clobber what the second reads
        out += ["\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, a0, b1, c0), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, a1, b1, c1)]
    elif mode == "swap01":     # src0 <-> src1 (commutative): op_sel moves to src0, as in the build that works
        out += ["\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0,0]" % (d0, d1, b0, b1, a0, a1, c0, c1)]
    else:
        raise SystemExit("mode?")
open(dst, "w").write("\n".join(out))
print(mode, "edited", n, "instructions")
