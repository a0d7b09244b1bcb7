import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neat_amd import lib, ops
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
M,N,K=78848,2048,512
A=torch.randn(M,K,generator=g,device=dev).bfloat16()
W=ops.pack_weight(torch.randn(N,K,generator=g,device=dev)*K**-0.5)
bias=torch.randn(N,generator=g,device=dev)
o0=torch.empty(M,N,device=dev,dtype=torch.bfloat16); o1=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
half=int(os.environ.get("HALF","0"))
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
def run():
    if not half:
        ops.gemm(A,W,lib.EPI_BIAS_GELU,bias=bias,out0=o0,out1=o1)
    else:
        cur=torch.cuda.current_stream()
        h=M//2
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): ops.gemm(A[:h],W,lib.EPI_BIAS_GELU,bias=bias,out0=o0[:h],out1=o1[:h])
        with torch.cuda.stream(s2): ops.gemm(A[h:],W,lib.EPI_BIAS_GELU,bias=bias,out0=o0[h:],out1=o1[h:])
        cur.wait_stream(s1); cur.wait_stream(s2)
ts=[]
for r in range(14):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    if r>=3: ts.append(a.elapsed_time(b)*1e3)
ts.sort(); print("half" if half else "single", "grid", os.environ.get("NEAT_T256_GRID","256"), f"med {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f}")
