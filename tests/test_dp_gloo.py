"""Data-parallel host logic of train_TTA.tta_step on CPU: world_size 2, gloo.

The kernels need an MI355X, so here the three kernel-backed calls of tta_step (text tower forward/backward,
coarse alignment, row-sharded loss) are replaced -- in the TEST only -- by CPU stand-ins built from the oracle.
Everything else is the product code: row sharding (uneven shards), padded all-gather of text / coarse
features, all-reduce of the LayerNorm gradients and of the per-rank loss sums, optimizer step on every rank.
Pass criterion: after two steps every rank holds the LN parameters, losses and score rows that the
single-process oracle produces on the whole batch (reference train_TTA.py:133-147; main_TTA.py:221-226 for
the reference's own DistributedSampler sharding).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

T1, T2, LR, WD = 0.03, 0.07, 2e-4, 1e-3
MODEL, B, NIMG, STEPS = "neat-tiny", 7, 12, 2     # 7 rows -> shards of 4 and 3


def _sharded_loss_cpu(t, p, v, t_all, c_all, T1, T2, row0):
    """CPU stand-in for ops.tta_loss_fwd_bwd with the same contract as neat_tta_loss_fwd_bwd
    (include/neat_hip.h): local rows [row0, row0+Bl) of a global batch of B rows; losses[0] is the global
    entropy term, losses[1:] are this rank's contributions (sum over local rows / B)."""
    B, Bl = t_all.shape[0], t.shape[0]
    sl = slice(row0, row0 + Bl)
    Z = (c_all @ t_all.t()) / T1
    logp = torch.log_softmax(Z, dim=1)
    P = logp.exp()
    H = -(P * logp).sum(1)
    l_ent = H.mean()
    dZ = -P * (logp + H.unsqueeze(1)) / B
    dt = (dZ[:, sl].t() @ c_all) / T1
    U = v @ c_all.t()
    ar = torch.arange(Bl)
    diag = U[ar, ar + row0].unsqueeze(1)
    raw = diag - U
    cost = raw.clamp(min=0)
    cost[ar, ar + row0] = 0.0
    m, j = cost.max(dim=1)
    l_sr = m.sum() / (B * T2)
    act = (raw[ar, j] >= 0) & (j != ar + row0)
    gsr = act.to(t.dtype) / (B * T2)
    dU = torch.zeros_like(U)
    dU[ar, j] -= gsr
    dU[ar, ar + row0] += gsr
    dv = dU @ c_all
    e_tv, e_tp = t - v + 1e-6, t - p + 1e-6
    d_tv, d_tp = e_tv.norm(dim=1), e_tp.norm(dim=1)
    l_tri = ((2.0 - d_tv).sum() + 5.0 * d_tp.sum()) / B
    g_tv, g_tp = e_tv / d_tv.unsqueeze(1) / B, e_tp / d_tp.unsqueeze(1) / B
    return torch.stack([l_ent, l_sr, l_tri]), dt - g_tv + 5.0 * g_tp, -5.0 * g_tp, dv + g_tv


class _OracleTower:
    """Text tower with the engine.TextTower interface, computed by the CPU oracle on the model's live
    parameters."""

    def __init__(self, model):
        self.model = model

    def _sd(self):
        return {k: v.detach() for k, v in self.model.state_dict().items()}

    def forward(self, tokens, normalize, save, packed=False, n_grad=None):
        from oracle import neat_oracle as O
        feat, cache = O.encode_text_fwd(self._sd(), tokens, normalize, save=save)
        return feat, cache

    def backward(self, ctx, dfeat, grads, accumulate):
        from oracle import neat_oracle as O
        n_all = ctx["shape"][0]
        d = torch.zeros(n_all, dfeat.shape[1])
        d[: dfeat.shape[0]] = dfeat             # sequences beyond the first n_grad carry no gradient
        named = O.encode_text_bwd(self._sd(), ctx, d, {})
        for i, g in enumerate(grads):
            for key, nm in (("ln1_w", "ln_1.weight"), ("ln1_b", "ln_1.bias"), ("ln2_w", "ln_2.weight"),
                            ("ln2_b", "ln_2.bias")):
                src = named[f"transformer.resblocks.{i}.{nm}"]
                g[key].copy_(g[key] + src if accumulate else src)


class _FlatGradAdamW(torch.optim.AdamW):
    """torch's AdamW with the gradient layout of neat_amd.optim.AdamW (which needs a GPU): every .grad is a view of
    ``grad_buf[:n]``, two spare floats behind -- the layout tta_step's single all-gather relies on."""

    def __init__(self, params, **kw):
        params = list(params)
        super().__init__(params, **kw)
        n = sum(p.numel() for p in params)
        self.grad_buf = torch.zeros(n + 2)
        self.flat_grad = self.grad_buf[:n]
        off = 0
        for p in params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()


def _worker(rank, world, port, out, B=None, flat=False):
    B = B or globals()["B"]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neat_amd import open_clip as oc, ops, synth
        from neat_amd import train_TTA as T
        from oracle import neat_oracle as O

        model = oc.create_model(MODEL, device="cpu")
        model.load_state_dict(synth.synth_state_dict(MODEL, 0))
        oc.freeze_parameters(model, False)
        params, _ = oc.collect_params(model, False)
        opt = (_FlatGradAdamW if flat else torch.optim.AdamW)(params, lr=LR, weight_decay=WD)
        tower = _OracleTower(model)
        model._text_tower = lambda: tower
        # test-only stand-ins for the three kernel-backed calls
        ops.coarse_align = lambda t, p, n, G: O.coarse_align_t2i(t, p, n, G)
        ops.tta_loss_fwd_bwd = _sharded_loss_cpu

        cfg = synth.get_cfg(MODEL)["text_cfg"]
        toks = synth.synth_caption_sets(B, cfg["context_length"], cfg["vocab_size"], 77)
        gen = torch.Generator().manual_seed(5)
        G = torch.nn.functional.normalize(torch.randn(NIMG, 512, generator=gen), dim=-1)
        lo, hi = T._shard(B, world, rank)
        local = [x[lo:hi].contiguous() for x in toks]

        def _no_all_reduce(*a, **k):
            raise AssertionError("dist.all_reduce on the TTA step: both optimizer branches use one all-gather + ordered sum")
        dist.all_reduce = _no_all_reduce
        for _ in range(STEPS):
            losses, idx, S = T.tta_step(model, opt, local, G, T1, T2, world, rank, global_rows=B if flat else None)
        S_all = T._all_gather_rows(S, B, world)
        torch.save(dict(losses=losses, S=S_all, idx=idx, lo=lo,
                        params={k: v.detach().clone() for k, v in model.state_dict().items() if ".ln_" in k}),
                   os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_covers_rows_in_order():
    from neat_amd.train_TTA import _shard
    for n in (0, 1, 7, 256, 25014):
        for w in (1, 2, 3, 8):
            spans = [_shard(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,flat", [(7, False), (8, True)])
def test_tta_step_world2_matches_single_process_oracle(tmp_path, B, flat):
    """(7, False): uneven shards 4+3 (padded all_gather), plain torch.optim.AdamW -- the per-parameter gradients travel
    through one flat scratch (one all-gather + ordered sum, no all_reduce), row count by collective.
    (8, True): even shards (all_gather_into_tensor), ONE all-gather over grad_buf (gradients + two loss terms), the
    caller passes the global row count -- the path bench.py and train_one_epoch take with neat_amd.optim.AdamW.
    Either way the worker makes dist.all_reduce raise for the duration of the steps."""
    from neat_amd import synth
    from oracle import neat_oracle as O

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), B, flat), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)

    sd = synth.synth_state_dict(MODEL, 0)
    cfg = synth.get_cfg(MODEL)["text_cfg"]
    toks = synth.synth_caption_sets(B, cfg["context_length"], cfg["vocab_size"], 77)
    gen = torch.Generator().manual_seed(5)
    G = torch.nn.functional.normalize(torch.randn(NIMG, 512, generator=gen), dim=-1)
    st = O.AdamWState({k: sd[k] for k in O.adapted_param_names(sd)})
    for _ in range(STEPS):
        losses, idx, S, _, _ = O.tta_step(sd, st, toks, G, LR, WD, T1, T2)

    for r in (r0, r1):
        torch.testing.assert_close(r["losses"], losses, rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(r["S"], S, rtol=1e-5, atol=1e-6)
        for k, v in r["params"].items():
            if k.startswith("transformer."):
                torch.testing.assert_close(v, sd[k], rtol=1e-5, atol=1e-7, msg=lambda m, k=k: f"{k}: {m}")
    # ranks agree bit for bit (same reduced gradients, same optimizer arithmetic)
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert torch.equal(torch.cat([r0["idx"], r1["idx"]]), idx)
    assert (r0["lo"], r1["lo"]) == (0, 4)


# ------------------------------------------------------------------------------------------------------------------
# train_one_epoch under both loader wirings: DistributedSamplers as the reference's
# launcher hands them out (main_TTA.py:245-257, through the package's own create_sampler / create_loader), and replicated
# loaders that every rank iterates in full.  Same CPU stand-ins for the kernel-backed calls as above.
N_TEXT, N_IMG, BS_LOCAL, BS_IMG = 13, 9, 4, 4        # 13 -> padded to 14 rows (7 per rank: batches of 4 + 3); 9 -> 10 images


class _TextDS(torch.utils.data.Dataset):
    def __init__(self, sets):
        self.sets = sets
        self.text = list(range(sets[0].shape[0]))

    def __len__(self):
        return len(self.text)

    def __getitem__(self, i):
        return tuple(s[i] for s in self.sets) + (i,)


class _ImageDS(torch.utils.data.Dataset):
    def __init__(self, images):
        self.images = images
        self.image = list(range(images.shape[0]))

    def __len__(self):
        return len(self.image)

    def __getitem__(self, i):
        return self.images[i], i


class _ListLoader:
    def __init__(self, batches, n, attr):
        self.batches = batches
        self.dataset = type("DS", (), {})()
        setattr(self.dataset, attr, list(range(n)))

    def __iter__(self):
        return iter(self.batches)


def _epoch_inputs():
    from neat_amd import synth
    cfg = synth.get_cfg(MODEL)
    t = cfg["text_cfg"]
    sets = synth.synth_caption_sets(N_TEXT, t["context_length"], t["vocab_size"], 99)
    images = synth.synth_images(N_IMG, cfg["vision_cfg"]["image_size"], 98)
    return sets, images


def _sampler_order(n, world):
    """Index lists a DistributedSampler(shuffle=False, drop_last=False) hands to each rank."""
    total = (n + world - 1) // world * world
    order = list(range(n)) + list(range(total - n))
    return [order[r::world] for r in range(world)]


def _epoch_worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from neat_amd import open_clip as oc, ops
        from neat_amd import train_TTA as T
        from neat_amd.data import create_loader, create_sampler
        from neat_amd import synth
        from oracle import neat_oracle as O

        model = oc.create_model(MODEL, device="cpu")
        model.load_state_dict(synth.synth_state_dict(MODEL, 0))
        oc.freeze_parameters(model, False)
        params, _ = oc.collect_params(model, False)
        opt = _FlatGradAdamW(params, lr=LR, weight_decay=WD)
        tower = _OracleTower(model)
        model._text_tower = lambda: tower
        model.encode_image = lambda image, normalize=True: O.encode_image_fwd(tower._sd(), image, normalize)
        ops.coarse_align = lambda t, p, n, G: O.coarse_align_t2i(t, p, n, G)
        ops.tta_loss_fwd_bwd = _sharded_loss_cpu
        sets, images = _epoch_inputs()
        args = types.SimpleNamespace(distributed=True, tta_only_visual=False, name="neat", tta_steps=STEPS,
                                     tta_temperature=T1, tta_temperature2=T2)
        if mode == "sampler":
            ds = [_TextDS(sets), _ImageDS(images)]
            samplers = create_sampler(ds, [False, False], world, rank)
            lt, li = create_loader(ds, samplers, batch_size=[BS_LOCAL, BS_IMG], num_workers=[0, 0],
                                   is_trains=[False, False], collate_fns=[None, None])
        elif mode == "replicated":
            tb = [tuple(x[i:i + 2 * BS_LOCAL] for x in sets) + (torch.arange(N_TEXT)[i:i + 2 * BS_LOCAL],)
                  for i in range(0, N_TEXT, 2 * BS_LOCAL)]
            ib = [(images[i:i + BS_IMG], torch.arange(N_IMG)[i:i + BS_IMG]) for i in range(0, N_IMG, BS_IMG)]
            lt, li = _ListLoader(tb, N_TEXT, "text"), _ListLoader(ib, N_IMG, "image")
        elif mode == "mismatch":
            ds = [_TextDS(sets), _ImageDS(images)]
            samplers = create_sampler(ds, [False, False], world + 1, rank)      # sampler over 3 replicas, group of 2
            lt, li = create_loader(ds, samplers, batch_size=[BS_LOCAL, BS_IMG], num_workers=[0, 0],
                                   is_trains=[False, False], collate_fns=[None, None])
        else:                                                                    # "tiny_batch": 1 row for 2 ranks
            tb = [tuple(x[:1] for x in sets) + (torch.arange(1),)]
            ib = [(images, torch.arange(N_IMG))]
            lt, li = _ListLoader(tb, N_TEXT, "text"), _ListLoader(ib, N_IMG, "image")
        err = None
        try:
            _, score = T.train_one_epoch(model, lambda x: x, lt, li, opt, torch.device("cpu"), args)
        except RuntimeError as e:
            err, score = str(e), None
        torch.save(dict(score=score, err=err,
                        params={k: v.detach().clone() for k, v in model.state_dict().items() if ".ln_" in k}),
                   os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["sampler", "replicated"])
def test_train_one_epoch_world2_equals_single_process_oracle(tmp_path, mode):
    from neat_amd import synth
    from oracle import neat_oracle as O

    mp.spawn(_epoch_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert r0["err"] is None and r1["err"] is None, (r0["err"], r1["err"])

    sets, images = _epoch_inputs()
    if mode == "sampler":
        per_rank = _sampler_order(N_TEXT, 2)
        n_local = len(per_rank[0])
        batches_idx = [sum((pr[k:k + BS_LOCAL] for pr in per_rank), []) for k in range(0, n_local, BS_LOCAL)]
        assert [len(b) for b in batches_idx] == [8, 6] and batches_idx[1].count(0) == 1    # the padded duplicate of row 0
    else:
        batches_idx = [list(range(i, min(i + 2 * BS_LOCAL, N_TEXT))) for i in range(0, N_TEXT, 2 * BS_LOCAL)]
    tb = [tuple(x[torch.tensor(b)] for x in sets) + (torch.tensor(b),) for b in batches_idx]
    ib = [(images, torch.arange(N_IMG))]
    sd = synth.synth_state_dict(MODEL, 0)
    score, _ = O.train_one_epoch(sd, tb, ib, N_TEXT, N_IMG, lr=LR, wd=WD, T1=T1, T2=T2, steps=STEPS)
    for r in (r0, r1):
        torch.testing.assert_close(r["score"], score, rtol=1e-5, atol=2e-6)
        assert (r["score"] > -100).all()                                  # every caption row was written
        for k, v in r["params"].items():
            if k.startswith("transformer."):
                torch.testing.assert_close(v, sd[k], rtol=1e-5, atol=1e-7, msg=lambda m, k=k: f"{k}: {m}")
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert torch.equal(r0["score"], r1["score"])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,needle", [("mismatch", "DistributedSampler"), ("tiny_batch", "cannot be cut")])
def test_train_one_epoch_world2_rejects_bad_wiring_on_every_rank(tmp_path, mode, needle):
    """A sampler that does not match the process group, or a batch with fewer rows than ranks, must raise on EVERY rank
    before any collective (a rank failing alone would leave the others waiting: ADVICE r1)."""
    mp.spawn(_epoch_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    for r in (0, 1):
        res = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert res["err"] is not None and needle in res["err"], res["err"]


def test_sampler_without_process_group_is_rejected():
    """A DistributedSampler over several replicas with args.distributed off would silently adapt on 1/W of the rows."""
    import types
    from neat_amd import train_TTA as T
    from neat_amd.data import create_loader, create_sampler
    sets, images = _epoch_inputs()
    ds = [_TextDS(sets), _ImageDS(images)]
    lt, li = create_loader(ds, create_sampler(ds, [False, False], 2, 0), batch_size=[4, 4], num_workers=[0, 0],
                           is_trains=[False, False], collate_fns=[None, None])
    assert T._distributed_sampler(lt) is not None and T._distributed_sampler(li) is not None
    with pytest.raises(RuntimeError, match="args.distributed is off"):
        T._loader_is_sharded(lt, 1, 0, "text")
