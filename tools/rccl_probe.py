"""Cost of the data-parallel step's two collectives as torch.distributed (backend nccl = RCCL) issues them, measured on
whatever world size torchrun gives (1 on the one-GPU box): host time per call and device time per call."""
import os, time
import torch, torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
w = dist.get_world_size()
buf = torch.zeros(24578, device=dev)
x = torch.randn(256, 1024, device=dev)
out = torch.empty(256 * w, 1024, device=dev)
filler = torch.randn(4096, 4096, device=dev)
for name, fn in (("all_reduce 98 KB", lambda: dist.all_reduce(buf)),
                 ("all_gather_into_tensor 1 MB", lambda: dist.all_gather_into_tensor(out, x)),
                 ("matmul only", lambda: None)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        filler @ filler                      # some device work between collectives, as in the step
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name:30s} world {w}: host {t_host*10:.3f} ms per iteration, end to end {t_all*10:.3f} ms per iteration", flush=True)
dist.destroy_process_group()
