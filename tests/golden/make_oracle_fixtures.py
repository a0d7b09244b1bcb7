"""Fixtures computed by the CPU ORACLE (oracle/neat_oracle.py), not by the reference: cases too slow to recompute inside
every GPU test run.  The oracle itself is pinned to the reference on the same model by tests/golden/vitl14_b128.npz
(tests/test_oracle_golden.py::test_vitl14_b128_matches_reference).

Run from the repo root:  python tests/golden/make_oracle_fixtures.py

vitl14_bl256_oracle_step.npz -- one adaptation step of ViT-L-14 at BASELINE config 3's per-GPU shape (256 caption rows x 4
variants, D = 768, gallery of 512 unit vectors): losses, coarse indices, the step-1 LayerNorm gradient, the first 32
feature rows of each caption set, the first 8 score rows.  Inputs are the seeded synthetic draws of neat_amd.synth
(bit-identical on any host), so only seeds are stored.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from neat_amd import synth  # noqa: E402
from oracle import neat_oracle as O  # noqa: E402

META = dict(model="ViT-L-14", wseed=0, B=256, tok_seed=4321, gallery=512, gallery_seed=8, lr=2e-4, wd=1e-3, T1=0.03, T2=0.07)


def inputs(meta):
    cfg = synth.get_cfg(meta["model"])
    t = cfg["text_cfg"]
    toks = synth.synth_caption_sets(meta["B"], t["context_length"], t["vocab_size"], meta["tok_seed"])
    # integer draws -> identical on every host (torch.randn is not guaranteed to be)
    g = torch.Generator().manual_seed(meta["gallery_seed"])
    G = torch.randint(-1000, 1001, (meta["gallery"], cfg["embed_dim"]), generator=g).float()
    G = torch.nn.functional.normalize(G, dim=-1)
    return toks, G


def main():
    m = META
    sd = synth.synth_state_dict(m["model"], m["wseed"])
    toks, G = inputs(m)
    names = O.adapted_param_names(sd)
    st = O.AdamWState({k: sd[k] for k in names})
    t0 = time.time()
    losses, idx, S, grads, feats = O.tta_step(sd, st, toks, G, m["lr"], m["wd"], m["T1"], m["T2"])
    m = dict(m, seconds=time.time() - t0, torch=torch.__version__)
    np.savez_compressed(os.path.join(HERE, "vitl14_bl256_oracle_step.npz"), meta=np.array(json.dumps(m)),
                        losses=losses.numpy(), idx=idx.numpy().astype(np.int32),
                        grad=torch.cat([grads[k].flatten() for k in names]).numpy(),
                        feats=torch.stack([f[:32] for f in feats]).numpy(), score_rows=S[:8].numpy(),
                        gallery_sum=np.array([G.double().sum().item(), G.double().abs().sum().item()]))
    print("done", losses.tolist(), m["seconds"])


if __name__ == "__main__":
    main()
