"""256x256 NT GEMM: 8 waves x (128x64) [v3] vs 4 waves x (128x128) [v7] (spmm_gemm_set_variant 1000/1001)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd._lib import lib
from bench_gemm import bench
shapes = [(84256, 768, 768), (84256, 3072, 768), (84256, 2304, 768), (84256, 768, 3072), (28304, 3072, 768), (8192, 8192, 8192)]
for rnd in range(2):
    for v in (1000, 1001):
        lib().cdll.spmm_gemm_set_variant(v)
        out = []
        for (M, N, K) in shapes:
            ms, tf = bench(M, N, K, iters=20)
            out.append(f"{tf:7.1f}")
        print(f"round {rnd} v{'7' if v == 1001 else '3'}: " + " ".join(out), flush=True)
lib().cdll.spmm_gemm_set_variant(1000)
