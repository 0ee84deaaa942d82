"""NT GEMM (256x256 kernel) with every other first-wave workgroup delayed by n x ~3.9 us (spmm_gemm_set_variant 900+n)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd._lib import lib
from bench_gemm import bench
shapes = [(84256, 768, 768), (84256, 3072, 768), (84256, 2304, 768), (84256, 768, 3072), (28304, 3072, 768), (8192, 8192, 8192)]
for rnd in range(2):
    for n in (0, 2, 4, 6, 9):
        lib().cdll.spmm_gemm_set_variant(900 + n)
        out = []
        for (M, N, K) in shapes:
            ms, tf = bench(M, N, K, iters=20)
            out.append(f"{tf:7.1f}")
        print(f"round {rnd} stagger {n}: " + " ".join(out), flush=True)
lib().cdll.spmm_gemm_set_variant(900)
