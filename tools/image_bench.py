"""GPU micro-benchmark of neat_image_preprocess (COCO-sized images -> 224) next to Pillow + torch ops on the host."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from neat_amd import ops
from oracle import image_oracle as IO
from PIL import Image
dev = "cuda:0"
rng = np.random.default_rng(0)
for (h, w, s) in [(480, 640, 224), (427, 640, 224), (2000, 3000, 224)]:
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    d = torch.from_numpy(a).to(dev)
    out = torch.empty(3, s, s, device=dev)
    for _ in range(5): ops.image_preprocess(d, s, out=out)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): ops.image_preprocess(d, s, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.image_preprocess(d, s, out=out)
    e1.record(); torch.cuda.synchronize()
    dev_us = e0.elapsed_time(e1) * 1e3 / n
    alg = 3 * h * w + 12 * s * s
    t0 = time.perf_counter()
    m = 20
    for _ in range(m): IO.to_tensor_normalize(np.asarray(Image.fromarray(a).resize((s, s), Image.BICUBIC)))
    cpu = (time.perf_counter() - t0) / m
    print(f"{h}x{w}->{s}: GPU {dt*1e6:7.1f} us/image wall ({dev_us:7.1f} us device span), {1/dt:8.0f} img/s, "
          f"{alg/dev_us/1e3:6.2f} GB/s algorithmic; Pillow+numpy on 1 host core {cpu*1e3:6.2f} ms/image = {1/cpu:6.0f} img/s")

# batched form: a gallery batch of 256 COCO-sized images of mixed sizes in one launch set
sizes = [(480, 640), (427, 640), (640, 480), (500, 375), (375, 500), (333, 500), (480, 640), (640, 427)]
ims = [torch.from_numpy(rng.integers(0, 256, sizes[i % 8] + (3,), dtype=np.uint8)).to(dev) for i in range(256)]
out = torch.empty(256, 3, 224, 224, device=dev)
for _ in range(3): ops.image_preprocess_batch(ims, 224, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): ops.image_preprocess_batch(ims, 224, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
alg = sum(3 * h * w for h, w in [sizes[i % 8] for i in range(256)]) + 256 * 12 * 224 * 224
print(f"batch of 256 mixed COCO sizes -> 224: {us:8.1f} us per batch = {us/256:6.2f} us/image, {256/us*1e6:9.0f} img/s, "
      f"{alg/us/1e3:7.1f} GB/s algorithmic")

# ---- video frames (MSR-VTT shape): a loader batch of 64 clips x 4 frames, 240 x 320 -> 224, one call ----
fr = torch.randint(0, 256, (256, 240, 320, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    ops.frames_preprocess(fr, 224)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ops.frames_preprocess(fr, 224)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
byts = 256 * (240 * 320 * 3 + 3 * 224 * 224 * 4)
print(f"256 frames 240x320 -> 224 (float antialiased bicubic): {dt * 1e6:.1f} us per call = {dt / 256 * 1e6:.2f} us/frame, "
      f"{256 / dt:.0f} frames/s, {byts / dt / 1e9:.1f} GB/s algorithmic")
