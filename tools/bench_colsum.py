import sys, os, torch
sys.path.insert(0, os.getcwd())
from spmm_amd import ops
for R, C in ((84256, 2304), (84256, 1536), (84256, 768), (28304, 2304), (13824, 2304)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16); out = torch.zeros(C, device="cuda")
    for _ in range(3): ops.colsum_bf16(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.colsum_bf16(x, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"colsum {R}x{C}: {us:7.1f} us  {R*C*2/us/1e6:6.2f} TB/s")
