"""Time the factorization kernels per kind with an ablated diagnostic library (tools/build_ablation.sh)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
variant = sys.argv[1]
if variant != "BASE":
    N._LIB_PATH = ROOT / "tools" / f"libparsy_abl_{variant}.bin"
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[2] if len(sys.argv) > 2 else "nd24k")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
plan.profile(2)
for _ in range(3):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    plan.profile_collect()
p = plan.profile_get()
# (bitwise checksum of the factor: variants that must not change a bit are compared through it)
chk = int(L.view(torch.int64).sum().item())
print(variant, {k: round(v / p["runs"], 3) for k, v in p["ms"].items() if v > 0}, "status", plan.status(), "checksum", chk)
