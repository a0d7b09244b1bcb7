import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
for nseq,L,heads in ((250,257,16),(250,197,12),(250,50,12)):
    C=heads*64
    qkv=(torch.randn(nseq*L,3*C,generator=g,device=dev)*0.5).bfloat16()
    ops.attn_fwd(qkv,nseq,L,heads,False,save_lse=False,head_major=True)
    ts=[]
    for r in range(12):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); ops.attn_fwd(qkv,nseq,L,heads,False,save_lse=False,head_major=True); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b)*1e3)
    ts.sort(); by=nseq*L*C*2*4
    print(f"attn fwd nseq {nseq} L {L} heads {heads}: {ts[len(ts)//2]:8.1f} us  {by/ts[len(ts)//2]/1e6:5.2f} TB/s", flush=True)
