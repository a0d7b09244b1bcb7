"""Experiment (round 3): at the PACKED text batches' size (about 14 tokens per caption: M = 14 336 forward rows) every
kernel of the stack underfills the chip and is bound by one unit's latency.  Do two half-batches on two HIP streams
overlap those latencies?  Both forms are captured into HIP graphs (the host's launch path is out of the picture) and
replayed: one stream with all rows vs two streams with half of the sequences each.  Forward + backward of the 12 layers."""
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
B, L, C = 256, int(os.environ.get("L", 14)), 512
nl = len(st.blocks)


def grads():
    return [dict(ln1_w=torch.zeros(C, device=dev), ln1_b=torch.zeros(C, device=dev), ln2_w=torch.zeros(C, device=dev),
                 ln2_b=torch.zeros(C, device=dev)) for _ in range(nl)]


def one(x, nseq, nb, g):
    saved = []
    st.forward(x, nseq, L, saved, grad_rows=nb * L)
    M = nb * L
    dxf = x[:M] * 1e-3
    dxb = dxf.bfloat16()
    st.backward(dxf, dxb, nb, L, saved, g, False)


xa = torch.randn(4 * B * L, C, device=dev)
x1, x2 = xa[: 2 * B * L].clone(), xa[2 * B * L:].clone()
ga, g1, g2 = grads(), grads(), grads()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def single():
    one(xa, 4 * B, 3 * B, ga)


def double():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        one(x1, 2 * B, 3 * B // 2, g1)
    with torch.cuda.stream(s2):
        one(x2, 2 * B, 3 * B // 2, g2)
    cur.wait_stream(s1); cur.wait_stream(s2)


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def timeit(g, n=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gs = capture(single)
print(f"L={L}: one stream, {4 * B * L} fwd rows: {timeit(gs):.3f} ms per fwd+bwd pass (graph replay)", flush=True)
gd = capture(double)
print(f"L={L}: two streams, 2 x {2 * B * L} fwd rows: {timeit(gd):.3f} ms per fwd+bwd pass (graph replay)", flush=True)
print(f"L={L}: one stream again: {timeit(gs):.3f} ms", flush=True)
