"""AdamW over the adapted LayerNorm parameters as ONE kernel launch.

Same update rule and defaults as ``torch.optim.AdamW`` (reference main_TTA.py:271: lr = tta_init_lr,
weight_decay = tta_wd, betas (0.9, 0.999), eps 1e-8; state persists across batches).  The 48 (ViT-B-32)
parameter tensors are re-pointed at construction to views of one flat fp32 buffer, their ``.grad`` to
views of a flat gradient buffer, so ``step()`` is a single ``neat_adamw_flat`` launch instead of 48
small ones and a data-parallel gradient all-reduce is one collective over ``flat_grad``.
"""
from __future__ import annotations

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        if not params:
            raise ValueError("AdamW: empty parameter list")
        if any(isinstance(p, dict) for p in params):
            raise ValueError("neat_amd.optim.AdamW supports a single parameter group")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("neat_amd.optim.AdamW: parameters must be on the GPU")
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        # two spare floats behind the gradients: the data-parallel step sends its two row-summed loss terms with the
        # gradient in ONE all-reduce over grad_buf
        self.grad_buf = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self.flat_grad = self.grad_buf[:n]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        off = 0
        for p in params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[off:off + k].view_as(p)
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        # the parameters and their gradients must still be the views made in __init__: a later model.to()/.float()/
        # .cuda() re-creates p.data, and p.grad = None drops the gradient view -- the step would then update buffers
        # nobody reads.  Two pointer comparisons per step (first and last parameter).
        off_last = self.flat_param.numel() - g["params"][-1].numel()
        for p, off in ((g["params"][0], 0), (g["params"][-1], off_last)):
            if (p.data_ptr() != self.flat_param[off:].data_ptr() or p.grad is None
                    or p.grad.data_ptr() != self.flat_grad[off:].data_ptr()):
                raise RuntimeError("neat_amd.optim.AdamW: a parameter (or its .grad) no longer aliases the optimizer's flat "
                                   "buffers -- build the optimizer AFTER moving/casting the model, and clear gradients "
                                   "with optimizer.zero_grad() rather than p.grad = None")
        self.step_count += 1
        ops.adamw_flat(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, g["lr"],
                       g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])

    def zero_grad(self, set_to_none: bool = False):
        # gradients are views of flat_grad and must stay attached; the backward overwrites them
        self.flat_grad.zero_()
