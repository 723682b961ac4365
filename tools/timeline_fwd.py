"""Dev tool: in-kernel stage timeline of one workgroup of the forward kernel (MODE 0).  Needs the instrumented build:
    python tools/build_variants.py dbg="-DISDF_DEBUG_HOOKS=1"   ->  variants/lib_dbg.so   (used automatically)
Prints the s_memtime stamps of wave 0 of workgroup 100 (fwd_pair.hip: stage ends before / after each barrier)."""
import os, sys
os.environ.setdefault("ISDF_DEBUG_TIMELINE", "1")   # "2": every wave of the workgroup (64 stamps each) instead of wave 0 + wall clocks
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dbg = os.path.join(_root, "variants", "lib_dbg.so")
if not os.environ.get("ISDF_HIP_LIB") and os.path.exists(_dbg):
    os.environ["ISDF_HIP_LIB"] = _dbg
sys.path.insert(0, _root)
import ctypes as C
import torch
from isdf_amd.engine import Engine, NetConfig, _stream
from isdf_amd import synthetic, _ffi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
eng = Engine(NetConfig(transform=synthetic.bounds_transform(), fwd_operand=os.environ.get("ISDF_FWD_OPERAND", "fp16")), "cuda")
torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
x = ((torch.rand(N, 3, device="cuda") - 0.5) * torch.tensor([6.0, 3.0, 5.0], device="cuda")).contiguous()
sdf = torch.empty(N, device="cuda")
ws = torch.zeros(8192, dtype=torch.uint8, device="cuda")
for _ in range(3):
    ws.zero_()
    _ffi.check(eng.lib.isdf_sdf_eval(C.byref(eng.cnet), _ffi.ptr(eng.params), _ffi.ptr(eng.shadow), _ffi.ptr(x), N, None, _ffi.ptr(sdf), None,
                                     _ffi.ptr(ws), ws.numel(), _stream(eng.device)), "isdf_sdf_eval")
    torch.cuda.synchronize()
raw = ws[-4096:].view(torch.int64).cpu().numpy()
if os.environ["ISDF_DEBUG_TIMELINE"] == "2":
    t = raw.reshape(8, 64)
    n = int((t > 0).all(axis=0).sum())
    t0 = t[:, 0].min()
    print("every wave of workgroup 100, %d stamps each (cycles since the first wave's first stamp), operand %s" % (n, eng.net.fwd_operand))
    print("  n " + "".join("   wave %d" % w for w in range(8)) + "   spread")
    for i in range(n):
        print("%3d " % i + "".join("%9d" % (t[w, i] - t0) for w in range(8)) + "%9d" % (t[:, i].max() - t[:, i].min()))
    sys.exit(0)
ts = raw[:128]; ts = ts[ts > 0]
print("n stamps", len(ts), "(s_memtime ticks = shader clock cycles), operand", eng.net.fwd_operand)
prev = ts[0]
for i, t in enumerate(ts):
    print("%3d  t=%8d  d=%7d" % (i, t - ts[0], t - prev)); prev = t
se = raw[128:128 + 2 * 190].reshape(-1, 2)
se = se[(se[:, 0] > 0) & (se[:, 1] > 0)]
if len(se):
    d = (se[:, 1] - se[:, 0]) / 100.0
    print("workgroups sampled %d: duration us min/med/max %.1f %.1f %.1f" % (len(se), d.min(), float(sorted(d)[len(d) // 2]), d.max()))
