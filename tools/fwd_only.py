"""Dev tool: run one chain-kernel mode a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isdf_amd.engine import Engine, NetConfig
from isdf_amd import synthetic
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
eng = Engine(NetConfig(transform=synthetic.bounds_transform(), fwd_operand=os.environ.get("ISDF_FWD_OPERAND", "fp16x2")), "cuda")
torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
x = (torch.rand(int(os.environ.get("ISDF_FWD_POINTS", "27000")), 3, device="cuda") * 4 - 2)
for _ in range(5):
    eng.sdf_eval(x, want_grad=(mode == "grad"))
torch.cuda.synchronize()
