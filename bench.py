#!/usr/bin/env python3
"""Benchmark of the NEAT test-time-adaptation hot path on MI355X.

Metric (BASELINE.json): TTA image-text pairs/sec, ViT-B/32, batch 256, 3 steps.
A "step" of this harness = one adaptation batch: 4 text-tower forwards + coarse alignment + loss/grad +
3 text-tower backwards + AdamW, repeated tta_steps=3 times, and the batch's score rows written
(reference train_TTA.py:125-150).  value = caption rows fully adapted per second, whole job.

Workload at N=1: BASELINE config 2 -- CLIP ViT-B-32, gallery of 5 000 images, batches of 256 caption
rows (text/pos/neg/inv token sets), synthetic tokens/images of NegBench-COCO shape, random-init weights of
the ViT-B-32 architecture.  Inputs are resident in HBM when the timed region starts.  The gallery encode
(Phase A, once per epoch) is timed separately, after one untimed batch, and reported as ``gallery_images_per_s``.

N>1 (launched by torch.distributed.run): weak scaling -- every rank owns 256 rows of a global batch of
256*N (BASELINE config 3's sharding); features are all-gathered and LN gradients all-reduced over RCCL.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (c_fc bf16 MFMA GEMM + fused GELU, timed with HIP events
on its stream inside the timed region; FLOP/s against the dense bf16 MFMA peak, the HBM figure beside it) and ``cpu_baseline`` (the CPU oracle on a bounded sample of the same
workload, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # HBM3E peak, same guide
# HBM traffic per launch of the c_fc GEMM from the PMC counters (profiles/r03_bench_pmc_*.csv, unchanged from r02: separate
# `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over this same command; counters are in KiB;
# FETCH_SIZE doubled as the guide prescribes for gfx950 wide streaming reads -- the LN kernels in the same
# passes confirm the factor: ln_fwd reports 80.8 MB for a 161.5 MB read).  Keyed by (M, N, K); a launch mix
# that was not profiled reports traffic = null.
PMC_TRAFFIC_BYTES = {(78848, 2048, 512): int((2 * 119672.57 + 551936.00) * 1024)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6, help="timed adaptation batches")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="ViT-B-32")
    ap.add_argument("--batch", type=int, default=256, help="caption rows per GPU per batch")
    ap.add_argument("--gallery", type=int, default=5000)
    ap.add_argument("--tta-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=128, help="caption rows in the CPU-oracle sample")
    ap.add_argument("--pack-tokens", action="store_true",
                    help="profiling aid: run the MAIN measurement with model.pack_tokens = True (the workload string "
                         "says so; this is not the BASELINE configuration)")
    ap.add_argument("--no-packed", action="store_true",
                    help="skip the extra (non-headline) measurement with model.pack_tokens = True")
    ap.add_argument("--workload", default="tta", choices=["tta", "mcq", "gallery"],
                    help="tta: the headline (BASELINE config 2/3).  mcq: BASELINE config 5, the small-batch binary MCQ "
                         "scorer (default model ViT-B-16, 64 images x 2 captions): latency per batch.  gallery: Phase A "
                         "alone (vision tower forward over --gallery images), for kernel traces of the vision tower")
    ap.add_argument("--options", type=int, default=2, help="mcq: captions per image")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only form the process group, add up their ranks and print one JSON line: checks the "
                         "self-launch path of --gpus N without touching a GPU (tests/test_host_logic.py)")
    ap.add_argument("--skip-gallery-encode", action="store_true",
                    help="profiling aid: random unit-norm gallery features instead of Phase A, so that a "
                         "rocprofv3 summary holds text-tower launches only")
    return ap.parse_args()


def text_tower_flops_per_row_step(cfg, L):
    """Algorithmic FLOPs per caption row per TTA step: 4 fwd + 3 bwd (dgrad only)."""
    t = cfg["text_cfg"]
    C, layers = t["width"], t["layers"]
    lin = 12 * C * C            # MACs per token per layer: qkv 3C^2 + out C^2 + mlp 8C^2
    att = 2 * L * C             # QK^T and PV, dense
    fwd = 2.0 * L * layers * (lin + att)
    bwd = 2.0 * L * layers * (lin + 2.5 * att)
    return 4 * fwd + 3 * bwd


def _dbg(msg):
    if os.environ.get("NEAT_BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', 0)}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(a):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks ourselves, one per GPU, as
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` (the reference's
    intent: one process per GPU, rows = tta_total_bs // world, main_TTA.py:245-256).  This process is only the parent:
    it never touches a GPU (``torch.cuda.device_count()`` does not initialise one on this stack), relays the children's
    output -- rank 0's JSON line included -- and exits with the launcher's status."""
    import socket
    import subprocess
    backend = os.environ.get("NEAT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and not a.launcher_selftest and ndev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs (one RCCL rank per GPU), found {ndev}; "
              f"no line is printed for a run that did not happen", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL needs dmabuf IPC on this host driver
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(a, rank, world):
    import torch.distributed as dist
    dist.init_process_group("gloo")
    x = torch.tensor([float(rank)])
    dist.all_reduce(x)
    if rank == 0:
        print(json.dumps(dict(selftest=True, n_gpus=world, rank_sum=float(x.item()))))
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus and world > 1:
        a.gpus = world
    if a.launcher_selftest:
        return launcher_selftest(a, rank, world)
    if a.workload == "mcq":
        return mcq_workload(a)
    if a.workload == "gallery":
        return gallery_workload(a)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # rehearsal of N ranks on fewer GPUs (NEAT_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force = os.environ.get("NEAT_FORCE_COLLECTIVES", "") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force:
        import torch.distributed as dist
        backend = os.environ.get("NEAT_BENCH_BACKEND", "nccl")     # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from neat_amd import lib, open_clip as oc, ops, synth
    from neat_amd import train_TTA as T
    from neat_amd.optim import AdamW
    lib.load()

    _dbg('library loaded')
    cfg = synth.get_cfg(a.model)
    tcfg = cfg["text_cfg"]
    L = tcfg["context_length"]
    model = oc.create_model(a.model, device=dev)
    model.load_state_dict(synth.synth_state_dict(a.model, 0))
    oc.freeze_parameters(model, False)
    params, _ = oc.collect_params(model, False)
    opt = AdamW(params, lr=2e-4, weight_decay=1e-3)
    # the headline runs all 77 positions of every caption, as the reference does (FLOP accounting of SURVEY.md §8d);
    # the package's default (model.pack_tokens = True) is measured separately below as `packed_tokens`
    model.pack_tokens = False
    if a.pack_tokens:
        model.pack_tokens, a.no_packed = True, True

    _dbg('model ready')
    # ---- Phase A: gallery (timed separately; each rank encodes its shard, then all-gather) ----
    n_img = a.gallery
    lo, hi = T._shard(n_img, world, rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    G_local = torch.empty(hi - lo, cfg["embed_dim"], device=dev)
    isz = cfg["vision_cfg"]["image_size"]
    # Phase A as train_one_epoch runs it: loader batches of 256 images (the reference's tta_total_bs, main_TTA.py:256)
    # through train_TTA.GalleryEncoder, which groups them into ~50 000 token rows per encode
    gb = 256
    if not a.skip_gallery_encode:      # one untimed group: module load, workspaces and the vision tower's packed weights
        enc = T.GalleryEncoder(model, dev)
        for _ in range(4):
            enc.add(torch.randn(gb, 3, isz, isz, generator=g, device=dev), torch.arange(gb))
        enc.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        if a.skip_gallery_encode:
            G_local = torch.nn.functional.normalize(torch.randn(hi - lo, cfg["embed_dim"], generator=g, device=dev), dim=-1)
        else:
            enc = T.GalleryEncoder(model, dev)
            ready = []
            for s in range(0, hi - lo, gb):
                e = min(hi - lo, s + gb)
                img = torch.randn(e - s, 3, isz, isz, generator=g, device=dev)
                ready += enc.add(img, torch.arange(s, e))
            ready += enc.flush()
            for index, emb in ready:
                G_local[index] = emb
            del img, ready
    torch.cuda.synchronize()
    gallery_s = time.perf_counter() - t0
    G = T._all_gather_rows(G_local, n_img, world) if (world > 1 or force) else G_local

    _dbg('gallery done')
    # ---- synthetic caption batches, resident in HBM ----
    nb = a.steps + a.warmup
    B = a.batch
    batches = []
    for i in range(nb):
        sets = synth.synth_caption_sets(B, L, tcfg["vocab_size"], 1000 + 7919 * rank + i)
        batches.append([s.to(dev) for s in sets])
    n_text = B * world * nb
    score = torch.full((B * nb, n_img), -100.0, device=dev)      # this rank's score rows

    timer = ops.KernelTimer(lib.EPI_BIAS_GELU)       # c_fc GEMM: largest single GEMM of the step
    ops.KERNEL_TIMER = timer

    def run_batch(i):
        plan = T.batch_plan(model, batches[i])
        for _ in range(a.tta_steps):
            losses, idx, S = T.tta_step(model, opt, batches[i], G, 0.03, 0.07, world, rank, global_rows=B * world,
                                        plan=plan)
        score[i * B:(i + 1) * B] = S
        return losses

    def barrier():
        if world > 1 or force:
            dist.barrier()
        torch.cuda.synchronize()

    _dbg('batches ready')
    if os.environ.get('NEAT_BENCH_DEBUG'):
        import faulthandler
        faulthandler.dump_traceback_later(45, exit=True)
    # the warm-up batches run with the event timer ON, so that the first use of timing events (a one-off stall of
    # ~50 ms in the first process on a fresh box, tools/cold_start.py) is not inside the timed region; their events
    # are discarded
    timer.enabled = a.warmup > 0
    for i in range(a.warmup):
        run_batch(i)
        _dbg(f'warmup {i} done')
    barrier()
    timer.events.clear()
    timer.shapes.clear()
    timer.saved_rows.clear()
    _dbg('timed region starts')
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(a.warmup, nb):
        losses = run_batch(i)
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    _dbg('timed region done')
    if world > 1 or force:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    pairs = a.steps * B * world
    value = pairs / dt

    # ---- roofline of the dominant kernel (events recorded inside the timed region) ----
    # c_fc GEMM with fused bias + QuickGELU: reads A[M,C] bf16 and the packed weight, writes the saved
    # pre-activation h[M,4C] and u = gelu(h)[M,4C], both bf16.  Ceilings per launch: HBM bytes / 8 TB/s vs
    # FLOPs / 2.5 PFLOP/s -- for this shape the HBM ceiling is the longer one, so it is the bound.
    avg_ms, avg_flop, nl = timer.summary()
    # algorithmic bytes of a c_fc launch: A (bf16) + W (bf16) + bias + the activation for all M rows + the saved
    # pre-activation for the rows that will see a backward pass (3 of the 4 caption sets: 3/4 of M)
    avg_bytes = (sum(2.0 * m * k + 2.0 * m * n + 2.0 * r * n + 2.0 * n * k + 4.0 * n
                     for (m, n, k), r in zip(timer.shapes, timer.saved_rows)) / nl) if nl else 0.0
    sec = avg_ms * 1e-3
    gbs = avg_bytes / sec / 1e9 if sec > 0 else 0.0
    tfs = avg_flop / sec / 1e12 if sec > 0 else 0.0
    flops_row_step = text_tower_flops_per_row_step(cfg, L)
    shapes = set(timer.shapes)
    traffic = PMC_TRAFFIC_BYTES.get(next(iter(shapes))) if len(shapes) == 1 else None
    shape0 = next(iter(shapes)) if shapes else (0, 0, 0)
    t256 = bool(shapes) and lib.load().neat_gemm_prefers_t256(*shape0, lib.EPI_BIAS_GELU) == 1
    if not t256:
        traffic = None                                             # the PMC passes profiled the tile256 launch
    # SURVEY.md §8(d) classes the encoder GEMMs as MFMA rows: the headline fraction is FLOP/s against the dense bf16
    # MFMA peak.  The same launch against the HBM roof (its algorithmic bytes / 8 TB/s is the LONGER of its two
    # ceilings: 255 FLOP/B against a ridge of 312) is given beside it as hbm_*.
    roofline = dict(bound="mfma", kernel=("gemm_tile256_kernel" if t256 else "gemm_panel_kernel") +
                    "<NEAT_EPI_BIAS_GELU> (c_fc: M=4*B*77, N=4C, K=C)",
                    achieved=round(tfs, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(tfs / MFMA_BF16_PEAK_TFLOPS, 4),
                    traffic=traffic, launches=nl, avg_launch_ms=round(avg_ms, 4),
                    algorithmic_flop_per_launch=int(avg_flop), algorithmic_bytes_per_launch=int(avg_bytes),
                    hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4),
                    step_tflops=round(pairs / world * a.tta_steps * flops_row_step / dt / 1e12, 1),
                    step_mfma_frac=round(pairs / world * a.tta_steps * flops_row_step / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4))

    # ---- extra, NOT the headline: the same batches with the opt-in packed text batches (positions after EOT
    # dropped; identical outputs, tests/test_gpu_tta.py::test_packed_tokens_equal_full_length) ----
    packed = None
    if not a.no_packed:
        model.pack_tokens = True
        run_batch(0)
        barrier()
        ptimer = ops.KernelTimer(lib.EPI_BIAS_GELU)
        ops.KERNEL_TIMER = ptimer
        ptimer.enabled = True
        t0 = time.perf_counter()
        for i in range(a.warmup, nb):
            run_batch(i)
        barrier()
        dtp = time.perf_counter() - t0
        ptimer.enabled = False
        ops.KERNEL_TIMER = timer
        if world > 1:
            tmax = torch.tensor([dtp], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtp = float(tmax.item())
        model.pack_tokens = False
        # useful FLOPs of the packed run: the same per-token / per-pair counts as SURVEY.md §8(d), over the positions that
        # exist (0..EOT of every caption; attention dense over each caption's own length); 4 forwards, 3 backwards
        C, layers = tcfg["width"], tcfg["layers"]
        useful = 0.0
        ntok = 0.0
        for b in batches[a.warmup:]:
            lens = [(x.argmax(dim=-1) + 1).double() for x in b]              # text, pos, neg, inv
            for j, l in enumerate(lens):
                lin, att = float(l.sum()) * 12 * C * C, float((l * l).sum()) * 2 * C
                useful += 2.0 * layers * (lin + att)
                if j != 2:                                                   # neg has no backward
                    useful += 2.0 * layers * (lin + 2.5 * att)
            ntok += float(lens[0].mean())
        useful *= a.tta_steps
        ntok /= a.steps
        p_ms, p_flop, p_n = ptimer.summary()
        packed = dict(value=round(pairs / dtp, 2), unit="pairs/s", ms_per_step=round(dtp / a.steps * 1e3, 3),
                      mean_tokens_per_caption=round(ntok, 2), context_length=L,
                      note="the package default (model.pack_tokens): the text tower runs positions 0..EOT only; same "
                           "outputs (tests/test_gpu_tta.py::test_packed_tokens_*)",
                      roofline=dict(bound="mfma", kernel="whole packed step (useful FLOPs = positions that exist)",
                                    achieved=round(useful / dtp / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                                    frac=round(useful / dtp / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                                    useful_flop_per_pair=int(useful / (a.steps * B)),
                                    cfc_avg_launch_ms=round(p_ms, 4), cfc_launches=p_n,
                                    cfc_tflops=round(p_flop / (p_ms * 1e-3) / 1e12, 1) if p_ms > 0 else None))

    # recall@1 on the produced score rows (synthetic ground truth: caption r <-> image r % N)
    from neat_amd.evaluation import retrieval as R
    rows = torch.arange(B * nb, device=dev)
    pos = torch.zeros_like(score, dtype=torch.bool)
    pos[rows, rows % n_img] = True
    r1 = (R.recall_at_k(score, pos, 1) > 0).float().mean().item()

    out = dict(metric="TTA image-text pairs/sec (ViT-B/32, bs256, 3 steps)", value=round(value, 2), unit="pairs/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
               config=dict(workload=f"BASELINE config 2: CLIP {a.model} NEAT t2i TTA, gallery {n_img} images, "
                                    f"batches of {B} caption rows per GPU x {a.tta_steps} steps, L={L}"
                                    + (" [NOT BASELINE: packed text batches]" if a.pack_tokens else ""),
                           global_batch=B * world, tta_steps=a.tta_steps, gallery=n_img,
                           parallelism=f"dp{world}"),
               gallery_images_per_s=None if a.skip_gallery_encode else round(n_img / gallery_s, 1), recall_at_1_t2i=r1,
               last_losses=[round(x, 4) for x in losses.tolist()], roofline=roofline)
    if packed is not None:
        out["packed_tokens"] = packed

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, cfg)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force:
        dist.destroy_process_group()


def vision_flops_per_image(cfg):
    """Algorithmic forward FLOPs of the vision tower per image (8.82 / 35.1 / 162.0 GFLOP)."""
    v = cfg["vision_cfg"]
    C, layers, P = v["width"], v["layers"], v["patch_size"]
    ntok = (v["image_size"] // P) ** 2 + 1
    lin = 12 * C * C
    att = 2 * ntok * C
    return 2.0 * (ntok * layers * (lin + att) + (ntok - 1) * 3 * P * P * C + C * cfg["embed_dim"])


def text_fwd_flops_per_seq(cfg, L):
    t = cfg["text_cfg"]
    C, layers = t["width"], t["layers"]
    return 2.0 * L * layers * (12 * C * C + 2 * L * C)


def mcq_workload(a):
    """BASELINE config 5: the binary MCQ scorer of reference evaluation/mcq.py:149-217 (and its HF twin :220-294) at its
    own batch -- 64 images [64,3,224,224] with 2 captions each, ViT-B-16: encode_image + encode_text + normalise +
    einsum('bf,nbf->bn') + arg-max (:195-205).  Latency-bound: reported as the median wall time of one batch from
    device-resident inputs to predictions on the device, three ways -- eager (one library call per kernel, host
    synchronised per batch, as the reference's loop is by its `.item()` at :208), eager back-to-back (no per-batch
    synchronisation: what the GPU alone needs) and replayed from a HIP graph captured once per batch shape."""
    import statistics
    if a.model == "ViT-B-32" and "--model" not in " ".join(sys.argv):
        a.model = "ViT-B-16"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from neat_amd import lib, open_clip as oc, ops, synth
    lib.load()
    cfg = synth.get_cfg(a.model)
    t = cfg["text_cfg"]
    L, B, n_opt = t["context_length"], (a.batch if a.batch != 256 else 64), a.options
    model = oc.create_model(a.model, device=dev)
    model.load_state_dict(synth.synth_state_dict(a.model, 0))
    model.eval()
    isz = cfg["vision_cfg"]["image_size"]
    g = torch.Generator(device=dev).manual_seed(11)
    nb = a.steps + a.warmup
    images = [torch.randn(B, 3, isz, isz, generator=g, device=dev) for _ in range(nb)]
    toks = [torch.cat([synth.synth_tokens(B, L, t["vocab_size"], torch.Generator().manual_seed(50 + 7 * i + o))
                       for o in range(n_opt)]).to(dev) for i in range(nb)]
    calls = [0]
    real_call = lib.call

    def counting_call(name, *args):
        calls[0] += 1
        return real_call(name, *args)

    def score(img, tok):
        with torch.no_grad():
            fi = model.encode_image(img, normalize=True)
            ft = model.encode_text(tok, normalize=True).view(n_opt, B, -1)
            return ops.mcq_scores(fi, ft)

    for i in range(max(a.warmup, 1)):
        score(images[i % nb], toks[i % nb])
    torch.cuda.synchronize()
    lib.call = counting_call
    ops.lib.call = counting_call
    score(images[0], toks[0])
    lib.call = real_call
    ops.lib.call = real_call
    torch.cuda.synchronize()
    # (1) eager, synchronised per batch
    eager = []
    for i in range(a.warmup, nb):
        t0 = time.perf_counter()
        s_, pred = score(images[i], toks[i])
        torch.cuda.synchronize()
        eager.append((time.perf_counter() - t0) * 1e3)
    # (2) eager, back to back
    t0 = time.perf_counter()
    for i in range(a.warmup, nb):
        score(images[i], toks[i])
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) * 1e3 / a.steps
    # (3) HIP graph: static input buffers, one capture, replay per batch
    graph_ms, graph_err = None, None
    try:
        from neat_amd.evaluation.mcq import GraphedScorer
        gs = GraphedScorer(model, B, n_opt, L, isz, dev)
        ref_pred = score(images[a.warmup], toks[a.warmup])[1].clone()
        lat = []
        for i in range(a.warmup, nb):
            t0 = time.perf_counter()
            _, p = gs(images[i], toks[i])
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
            if i == a.warmup and not torch.equal(p, ref_pred):
                raise RuntimeError("graph replay disagrees with the eager scorer")
        graph_ms = statistics.median(lat)
    except Exception as e:                     # noqa: BLE001 -- reported in the line, the eager numbers stand
        graph_err = f"{type(e).__name__}: {e}"
    flops = B * vision_flops_per_image(cfg) + n_opt * B * text_fwd_flops_per_seq(cfg, L)
    best = min(x for x in (statistics.median(eager), graph_ms) if x is not None)
    out = dict(metric=f"binary MCQ scorer latency per batch ({a.model}, {B} images x {n_opt} captions)",
               value=round(best, 3), unit="ms", n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=round(best, 3),
               higher_is_better=False, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
               config=dict(workload=f"BASELINE config 5: {a.model} binary MCQ scorer, batch {B} x {n_opt} captions, "
                                    f"L={L} (reference evaluation/mcq.py:149-217)"),
               latency=dict(eager_sync_ms_median=round(statistics.median(eager), 3), eager_sync_ms_min=round(min(eager), 3),
                            eager_back_to_back_ms=round(b2b, 3), hip_graph_ms_median=None if graph_ms is None else round(graph_ms, 3),
                            hip_graph_error=graph_err, library_calls_per_batch=calls[0]),
               roofline=dict(bound="mfma", kernel="whole batch (launch-latency-bound path: see latency)",
                             achieved=round(flops / (best * 1e-3) / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(flops / (best * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                             algorithmic_flop_per_batch=int(flops)))
    print(json.dumps(out))


def gallery_workload(a):
    """Phase A alone (reference train_TTA.py:120-124): the vision tower forward over --gallery synthetic images in
    batches of 250, timed after one untimed batch.  For kernel traces of the vision tower (profiles/r03_gallery_*)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from neat_amd import lib, open_clip as oc, synth
    lib.load()
    cfg = synth.get_cfg(a.model)
    model = oc.create_model(a.model, device=dev)
    model.load_state_dict(synth.synth_state_dict(a.model, 0))
    isz = cfg["vision_cfg"]["image_size"]
    g = torch.Generator(device=dev).manual_seed(1234)
    from neat_amd import train_TTA as T
    bs = 256                                   # the reference's loader batch (tta_total_bs); grouped by GalleryEncoder
    img = torch.randn(bs, 3, isz, isz, generator=g, device=dev)
    idx = torch.arange(bs)
    with torch.no_grad():
        enc = T.GalleryEncoder(model, dev)
        for _ in range(4):
            enc.add(img, idx)
        enc.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        enc = T.GalleryEncoder(model, dev)
        while n < a.gallery:
            k = min(bs, a.gallery - n)
            enc.add(img[:k], idx[:k])
            n += k
        enc.flush()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = vision_flops_per_image(cfg)
    out = dict(metric=f"gallery images/sec ({a.model} vision tower forward)", value=round(a.gallery / dt, 1), unit="images/s",
               n_gpus=1, steps=1, warmup=1, ms_per_step=round(dt * 1e3, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="bf16", data="synthetic",
               config=dict(workload=f"Phase A of BASELINE config 2/3: {a.model} encode_image over {a.gallery} images, "
                                    f"loader batches of {bs} grouped to ~50 000 token rows per encode"),
               roofline=dict(bound="mfma", kernel="vision tower forward (all kernels)",
                             achieved=round(a.gallery * fl / dt / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(a.gallery * fl / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                             algorithmic_flop_per_image=int(fl)))
    print(json.dumps(out))


def cpu_baseline(a, cfg):
    """The CPU oracle (plain-PyTorch fp32 restatement of the reference loop, oracle/neat_oracle.py) on a
    bounded sample of the same workload: one batch of --cpu-sample caption rows x tta_steps, gallery of 64
    (gallery encode excluded, as in the GPU number)."""
    from neat_amd import synth
    from oracle import neat_oracle as O
    # every core this process may use: its affinity mask, bounded by OMP_NUM_THREADS where the box sets a CPU share
    # (os.cpu_count() reports the whole machine there, many times this job's share)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = min(avail, share) if share > 0 else avail
    torch.set_num_threads(cores)
    t = cfg["text_cfg"]
    Bc = a.cpu_sample
    # The sample's weights and data.  For ViT-B-32 (the headline model) it is the non-degenerate parity fixture of
    # tests/golden/make_spread_weights.py: "spread" weights (features far from collinear) and RELATED caption sets whose
    # row r describes image r, with the gallery = those images through the vision tower of either leg -- so recall@1 is
    # far from chance and the comparison below means something.  Other models have no fitted projections: seeded random
    # weights, independent caption sets and a random unit gallery (recall@1 at chance; the loss terms still compare).
    spread = a.model == "ViT-B-32" and Bc <= 256 and os.path.exists(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "spread_vitb32.npz"))
    if spread:
        sd = synth.spread_state_dict(a.model, 0)
        toks = tuple(x[:Bc] for x in synth.synth_related_caption_sets(256, t["context_length"], t["vocab_size"], 4321))
        images = synth.synth_images(256, cfg["vision_cfg"]["image_size"], 4322)[:Bc]
        G = torch.cat([O.encode_image_fwd(sd, images[i:i + 32], True) for i in range(0, Bc, 32)])      # untimed
        n_gal = Bc
    else:
        sd = synth.synth_state_dict(a.model, 0)
        toks = synth.synth_caption_sets(Bc, t["context_length"], t["vocab_size"], 1000)
        n_gal = 64
        G = torch.nn.functional.normalize(torch.randn(n_gal, cfg["embed_dim"], generator=torch.Generator().manual_seed(7)), dim=-1)
    sd0 = {k: v.clone() for k, v in sd.items()}                      # the oracle's step updates sd in place
    st = O.AdamWState({k: sd[k] for k in O.adapted_param_names(sd)})
    t0 = time.perf_counter()
    for _ in range(a.tta_steps):
        losses_c, idx_c, S_c = O.tta_step(sd, st, toks, G, 2e-4, 1e-3, 0.03, 0.07)[:3]
    dt = time.perf_counter() - t0
    out = dict(value=round(Bc / dt, 3), unit="pairs/s", cores=cores, cores_visible=avail, kind="port",
               sample=f"1 batch of {Bc} caption rows x {a.tta_steps} steps, gallery {n_gal}, {a.model}, fp32 torch CPU "
                      f"oracle, {dt:.1f} s" + ("; spread weights + related captions (tests/golden/make_spread_weights.py), "
                                               "gallery = the captions' images through each leg's vision tower" if spread else ""))
    # the same sample through the HIP path: recall@1 of both legs on the produced score rows (ground truth: caption r <->
    # gallery item r % n_gal) and how often their top-1 items agree
    from neat_amd import open_clip as oc, train_TTA as T
    from neat_amd.optim import AdamW
    dev = torch.device("cuda", torch.cuda.current_device())
    model = oc.create_model(a.model, device=dev)
    model.load_state_dict(sd0, strict=True)
    if spread:
        with torch.no_grad():
            G_d = model.encode_image(images.to(dev), normalize=True)
    else:
        G_d = G.to(dev)
    oc.freeze_parameters(model, False)
    opt = AdamW(oc.collect_params(model, False)[0], lr=2e-4, weight_decay=1e-3)
    toks_d = [x.to(dev) for x in toks]
    for _ in range(a.tta_steps):
        losses_g, idx_g, S_g = T.tta_step(model, opt, toks_d, G_d, 0.03, 0.07)
    truth = torch.arange(Bc) % n_gal
    S_g = S_g.float().cpu()
    out["recall_at_1"] = dict(gpu=round(float((S_g.argmax(1) == truth).float().mean()), 4),
                              cpu=round(float((S_c.argmax(1) == truth).float().mean()), 4),
                              top1_agreement=round(float((S_g.argmax(1) == S_c.argmax(1)).float().mean()), 4),
                              loss_rel_diff=round(float(abs(losses_g.sum().item() - float(sum(losses_c))) /
                                                        abs(float(sum(losses_c)))), 6),
                              # per term (entropy, semantics reversion, triplet): the hinge-max SR term moves when a
                              # coarse index flips under bf16 features; the other two stay within ~1e-4
                              loss_terms_gpu=[round(float(x), 6) for x in losses_g.tolist()],
                              loss_terms_cpu=[round(float(x), 6) for x in losses_c],
                              loss_term_rel_diff=[round(abs(float(g_) - float(c_)) / max(abs(float(c_)), 1e-12), 6)
                                                  for g_, c_ in zip(losses_g.tolist(), losses_c)],
                              coarse_idx_mismatch=int((idx_g.cpu().long() != idx_c).sum()),
                              weights="spread" if spread else "seeded random (collinear features: recall@1 at chance)")
    return out


if __name__ == "__main__":
    main()
