import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gotennet_amd import engine
def t(M,N,K,reps=20):
    A = torch.randn(M,K,device='cuda'); W = torch.randn(N,K,device='cuda')/16; b = torch.randn(N,device='cuda'); C = torch.empty(M,N,device='cuda')
    for _ in range(3): engine.gemm(A,K,W,b,C,N,M,N,K)
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): engine.gemm(A,K,W,b,C,N,M,N,K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/reps*1e3
    tiles = ((M+127)//128)*((N+127)//128)
    print(f"{M:7d}x{N:5d}x{K:5d} tiles={tiles:6d} ({tiles/256:.2f}/CU) {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF  per-round(2/CU) {us/max(1,-(-tiles//512)):7.1f} us")
for K in (256, 1024):
    for mult in (1, 2, 3, 4, 8, 16):
        t(128*256*mult, 1024 if False else 128*4, K)   # 4 column tiles so it takes the big-tile path (>=384 tiles)
