import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops
import numpy as np
epi, M, N, K = (int(x) for x in sys.argv[1:5])
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
A=torch.randn(M,K,generator=g,device=dev).bfloat16(); W=ops.pack_weight(torch.randn(N,K,generator=g,device=dev)*K**-0.5)
bias=torch.randn(N,generator=g,device=dev)
res=torch.randn(M,N,generator=g,device=dev) if epi==2 else None
aux=torch.randn(M,N,generator=g,device=dev).bfloat16() if epi==4 else None
for _ in range(3): ops.gemm(A,W,epi,bias=bias,res=res,aux=aux)
torch.cuda.synchronize()
buf=np.zeros(8*2*64*8,dtype=np.int64)
dll=lib.load(); dll.neat_lab_trace_read.argtypes=[ctypes.c_void_p]; 
rc=dll.neat_lab_trace_read(buf.ctypes.data); assert rc==0
t=buf.reshape(8,2,64,8)
nkh=K//128
for slot in range(8):
  for w in range(2):
    x=t[slot,w]; base=x[0,0]
    print(f"--- wg slot {slot} wave sel {w}")
    for j in range(min(64, 3*nkh+1)):
        if x[j,0]==0: break
        s=x[j]-base
        ep = f" epi_bar {s[5]-s[4]:6d} epi {s[6]-s[5]:6d}" if (j%nkh)==nkh-1 else ""
        print(f" j={j:2d} t={s[0]:8d} bar {s[1]-s[0]:6d} fill {s[2]-s[1]:5d} mma {s[3]-s[2]:6d} vmwait {s[4]-s[3]:6d}{ep}")
