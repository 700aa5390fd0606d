"""Our projection kernel on arbitrary shapes (GM GN GK env) for comparison with tools/blas_ref.py."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import engine
def t(M, N, K, it=20):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16; C = torch.empty(M, N, device="cuda")
    for _ in range(3): engine.gemm(A, K, W, None, C, N, M, N, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): engine.gemm(A, K, W, None, C, N, M, N, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"gn_gemm {M}x{N}x{K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
for s in ((54368, 1536, 256), (54368, 256, 1536), (54368, 256, 256), (21504, 256, 256), (2688, 1280, 256), (2688, 256, 1280), (8192, 8192, 8192), (4096, 4096, 4096)):
    t(*s, it=5 if s[0] in (8192, 4096) else 20)
