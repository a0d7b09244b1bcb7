"""Experiment: does running the text tower as two half-batches on two HIP streams absorb the kernels' tails
(GEMM round quantisation, launch boundaries)?  Forward + backward of the 12-layer stack, one stream at M rows vs two
streams at M/2 rows each.  Prints ms per pass."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, open_clip as oc, ops, synth
from neat_amd.engine import Stack

dev = "cuda"
name = "ViT-B-32"
model = oc.create_model(name, device=dev)
model.load_state_dict(synth.synth_state_dict(name, 0))
st = Stack(model.transformer, 8, True, True)
B, L, C = 256, 77, 512
nl = len(st.blocks)
def mk(nseq):
    return torch.randn(nseq * L, C, device=dev)
def grads():
    return [dict(ln1_w=torch.zeros(C, device=dev), ln1_b=torch.zeros(C, device=dev), ln2_w=torch.zeros(C, device=dev),
                 ln2_b=torch.zeros(C, device=dev)) for _ in range(nl)]

def one(x, nseq, nb):
    saved = []
    st.forward(x, nseq, L, saved, grad_rows=nb * L)
    M = nb * L
    dxf = torch.randn(M, C, device=dev) * 1e-3
    dxb = dxf.bfloat16()
    st.backward(dxf, dxb, nb, L, saved, grads(), False)

def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

xa = mk(4 * B)
print("one stream, 4B fwd / 3B bwd: %.2f ms" % timeit(lambda: one(xa, 4 * B, 3 * B)), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x1, x2 = mk(2 * B), mk(2 * B)
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        one(x1, 2 * B, 3 * B // 2)
    with torch.cuda.stream(s2):
        one(x2, 2 * B, 3 * B // 2)
    cur.wait_stream(s1); cur.wait_stream(s2)
print("two streams, 2 x (2B fwd / 1.5B bwd): %.2f ms" % timeit(two), flush=True)
def seq2():
    one(x1, 2 * B, 3 * B // 2); one(x2, 2 * B, 3 * B // 2)
print("one stream, the two halves back to back: %.2f ms" % timeit(seq2), flush=True)
