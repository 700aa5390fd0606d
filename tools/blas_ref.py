"""Vendor-library context for the projection kernel: torch.matmul (rocBLAS / hipBLASLt fp32) on the edge-sized shapes."""
import torch
def t(M, N, K, it=20):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    for _ in range(3): C = A @ W.T
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): C = A @ W.T
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"torch.matmul fp32 {M}x{N}x{K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
torch.backends.cuda.matmul.allow_tf32 = False
for s in ((54368, 1536, 256), (54368, 256, 1536), (54368, 256, 256), (21504, 256, 256), (2688, 1280, 256), (2688, 256, 1280), (8192, 8192, 8192)):
    t(*s, it=5 if s[0] == 8192 else 20)
