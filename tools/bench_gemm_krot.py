"""NT GEMM throughput with / without per-tile k-loop rotation (spmm_gemm_set_variant 700/701)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd._lib import lib
from bench_gemm import bench
shapes = [(93184, 3072, 768), (93184, 768, 3072), (93184, 768, 768), (93184, 2304, 768), (32768, 2304, 768), (8192, 8192, 8192)]
for rnd in range(3):
    for v in (700, 701):
        lib().cdll.spmm_gemm_set_variant(v)
        out = []
        for (M, N, K) in shapes:
            ms, tf = bench(M, N, K, iters=20)
            out.append(f"{tf:7.1f}")
        print(f"round {rnd} krot {v - 700}: " + " ".join(out), flush=True)
