import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import bench_tn
from spmm_amd._lib import lib
for var in (1, 512, 640, 768, 1000, 1024, 1536):
  lib().cdll.spmm_gemm_tn_set_variant(var)
  print('variant', var)
  for (M, N, K) in [(93184, 768, 768), (93184, 2304, 768), (93184, 3072, 768), (93184, 768, 3072), (32768, 2304, 768), (13824, 768, 768), (65536, 1536, 768)]:

      ms, tf = bench_tn(M, N, K)
      print(f"TN M={M} N={N} K={K}: {ms:.3f} ms {tf:.1f} TF", flush=True)
