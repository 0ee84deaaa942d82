"""NT GEMM throughput by tile kernel (spmm_gemm_set_variant 801/802/803 = 128x128 / 256x128 / 256x256) on the small-M shapes of
the decoder (M = 5000, 2500) and of the unimodal / momentum encoders."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd._lib import lib
from bench_gemm import bench
shapes = [(M, N, K) for M in (2500, 5000, 6912, 11920, 13824, 23840, 28304) for (N, K) in ((768, 768), (2304, 768), (3072, 768), (768, 3072))]
print("shape".ljust(22), "128x128  256x128  256x256   (TF/s)")
for (M, N, K) in shapes:
    out = []
    for v in (801, 802, 803):
        lib().cdll.spmm_gemm_set_variant(v)
        ms, tf = bench(M, N, K, iters=15)
        out.append(f"{tf:7.1f}")
    print(f"{M}x{N}x{K}".ljust(22), "  ".join(out), flush=True)
lib().cdll.spmm_gemm_set_variant(800)
