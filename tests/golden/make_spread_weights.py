"""Fit the two projection matrices of the "spread" synthetic CLIP (neat_amd.synth.spread_state_dict).

Run from the repo root:  python tests/golden/make_spread_weights.py

Why: with plain seeded random weights every caption feature shares one dominant direction (the EOT token's own embedding),
recall@1 sits at chance and every arg-max margin is tiny -- parity on such features says little.
Pretrained weights are not available offline.  This script needs NO reference code: it runs the CPU oracle's towers with
the residual branches scaled up (every layer contributes, as in a trained tower) and ridge-fits ``visual.proj`` so that
the gallery embeddings land on target codes with mutual cosine ~0.5, and ``text_projection`` so that caption r lands near
the code of image r % N (30 % of the captions are pointed at a WRONG image: recall@1 ~0.7, not saturated).  The fitted
matrices are rounded to bf16-representable values and stored as uint16 bit patterns: they are DATA that both the
reference (tests/golden/make_golden.py, case vitb32_spread_b256) and the HIP path load.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neat_amd import synth  # noqa: E402
from oracle import neat_oracle as O  # noqa: E402

MODEL, N_TEXT, N_IMAGE, DSEED = "ViT-B-32", 256, 256, 4321


def nz(x):
    return x / x.norm(dim=1, keepdim=True)


def ridge(Z, Y, lam):
    A = Z.t() @ Z
    A = A + lam * torch.trace(A) / A.shape[0] * torch.eye(A.shape[0], dtype=Z.dtype)
    return torch.linalg.solve(A, Z.t() @ Y)


def to_bf16_bits(w):
    """fp32 -> nearest-even bf16 bit pattern (uint16)."""
    b = w.float().contiguous().view(torch.int32)
    b = b + 0x7FFF + ((b >> 16) & 1)
    return ((b >> 16) & 0xFFFF).numpy().astype(np.uint16)


def main():
    torch.set_num_threads(8)
    cfg = synth.get_cfg(MODEL)
    t = cfg["text_cfg"]
    D = cfg["embed_dim"]
    sd = synth.synth_state_dict(MODEL, 0, branch_gain=synth.SPREAD_BRANCH_GAIN)
    sets = synth.synth_related_caption_sets(N_TEXT, t["context_length"], t["vocab_size"], DSEED)
    images = synth.synth_images(N_IMAGE, cfg["vision_cfg"]["image_size"], DSEED + 1)
    sdz = dict(sd)
    sdz["text_projection"] = torch.eye(t["width"])
    sdz["visual.proj"] = torch.eye(cfg["vision_cfg"]["width"])
    Zt = O.encode_text_fwd(sdz, sets[0], False)[0].double()          # ln_final output at EOT
    Zi = O.encode_image_fwd(sdz, images, False).double()              # ln_post output of the class token
    g = torch.Generator().manual_seed(5)
    u = nz(torch.randn(1, D, generator=g, dtype=torch.float64))
    Y = nz(u + nz(torch.randn(N_IMAGE, D, generator=g, dtype=torch.float64)))       # mutual cosine ~0.5
    Wv = ridge(Zi, Y, 1e-3)
    G = nz(Zi @ Wv)
    owner = torch.arange(N_TEXT) % N_IMAGE
    wrong = torch.rand(N_TEXT, generator=g) < 0.3
    tgt = torch.where(wrong, (owner + 1 + torch.randint(0, N_IMAGE - 1, (N_TEXT,), generator=g)) % N_IMAGE, owner)
    T = nz(G[tgt] + 1.0 * nz(torch.randn(N_TEXT, D, generator=g, dtype=torch.float64)))
    Wt = ridge(Zt, T, 1e-2)
    out = {"visual_proj": to_bf16_bits(Wv), "text_projection": to_bf16_bits(Wt)}
    np.savez_compressed(os.path.join(HERE, "spread_vitb32.npz"), **out)
    # report what the stored weights give
    sd2 = synth.spread_state_dict(MODEL, 0)
    F = [O.encode_text_fwd(sd2, x, True)[0] for x in sets]
    Gf = O.encode_image_fwd(sd2, images, True)
    off = lambda f: (f @ f.t())[~torch.eye(len(f), dtype=torch.bool)]
    S = F[0] @ Gf.t()
    top2 = S.topk(2, 1)[0]
    print("text-text cos mean %.3f  image-image cos mean %.3f  text-pos %.3f  text-inv %.3f  text-neg %.3f" % (
        off(F[0]).mean(), off(Gf).mean(), (F[0] * F[1]).sum(1).mean(), (F[0] * F[3]).sum(1).mean(), (F[0] * F[2]).sum(1).mean()))
    print("recall@1 %.3f  median top-1 margin %.4f  positive sim %.3f" % (
        (S.argmax(1) == owner).float().mean(), (top2[:, 0] - top2[:, 1]).median(), S[torch.arange(N_TEXT), owner].mean()))
    c, idx, _ = O.coarse_align_t2i(F[0], F[1], F[2], Gf)
    print("coarse idx == owner: %.3f" % (idx == owner).float().mean())


if __name__ == "__main__":
    main()
