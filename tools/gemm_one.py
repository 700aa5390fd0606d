import os, sys, torch
sys.path.insert(0, os.getcwd())
from gotennet_amd import engine
M,N,K = int(os.environ.get("GM", 54368)), int(os.environ.get("GN", 1536)), int(os.environ.get("GK", 256))
A = torch.randn(M,K,device='cuda'); W = torch.randn(N,K,device='cuda')/16; b = torch.randn(N,device='cuda'); C = torch.empty(M,N,device='cuda')
for _ in range(5): engine.gemm(A,K,W,b,C,N,M,N,K)
torch.cuda.synchronize()
