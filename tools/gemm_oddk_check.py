"""gn_gemm on shapes outside the path's usual multiples (K not a multiple of 32, tiny N): max error vs fp64, both modes."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import engine
torch.manual_seed(0)
for mode in ("f32", "split", "f16x2"):
    engine.GEMM_MODE = mode
    for M, N, K in [(23, 4, 68), (23, 8, 64), (69, 8, 64), (23, 64, 68), (23, 64, 36), (500, 4, 260), (23, 4, 64), (23, 64, 4), (23, 64, 12), (23, 64, 40)]:
        A = torch.randn(M, K, device="cuda") * (10.0 ** torch.randint(-6, 3, (M, 1), device="cuda").float()); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        engine.gemm(A, K, W, b, C, N, M, N, K)
        ref = (A.double() @ W.double().t() + b.double())
        print(mode, (M, N, K), "rel err %.2e" % float((C.double() - ref).abs().max() / ref.abs().max()))
