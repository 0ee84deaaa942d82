import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
from spmm_amd._lib import lib
from bench_gemm import bench
shapes = [(93184, 3072, 768), (93184, 768, 3072), (93184, 768, 768), (93184, 2304, 768), (32768, 2304, 768), (13824, 768, 768), (8192, 8192, 8192)]
for rnd in range(2):
    lib().cdll.spmm_gemm_set_variant(101)
    for v in (600, 601, 602):
        lib().cdll.spmm_gemm_set_variant(v)
        out = []
        for (M, N, K) in shapes:
            ms, tf = bench(M, N, K, iters=20)
            out.append(f"{tf:7.1f}")
        print(f"round {rnd} variant {v}: " + " ".join(out), flush=True)
