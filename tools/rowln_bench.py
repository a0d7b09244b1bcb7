"""Round-3 prototype check + timing: residual GEMM with the following LayerNorm folded in (neat_gemm_bf16_res_ln) against
the two launches it replaces (neat_gemm_bf16 BIAS_RES_F32 + neat_ln_fwd), at the step's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for name, M, K in (("out  full", 78848, 512), ("proj full", 78848, 2048), ("out  packed", 14848, 512), ("proj packed", 14848, 2048),
                   ("ragged", 64 * 37 + 5, 256)):
    N = 512
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
    bias = torch.randn(N, generator=g, device=dev)
    res = torch.randn(M, N, generator=g, device=dev)
    gam = 1 + 0.2 * torch.randn(N, generator=g, device=dev)
    bet = 0.1 * torch.randn(N, generator=g, device=dev)
    x_ref, _ = ops.gemm(A, W, lib.EPI_BIAS_RES_F32, bias=bias, res=res, kernel="panel")
    y_ref, _, mu_ref, rs_ref = ops.ln_fwd(x_ref, gam, bet)
    x, y, mu, rs = ops.gemm_res_ln(A, W, bias, res, gam, bet)
    torch.cuda.synchronize()
    dx = (x - x_ref).abs().max().item()
    dy = (y.float() - y_ref.float()).abs().max().item()
    ny = (y != y_ref).float().mean().item()
    dm = (mu - mu_ref).abs().max().item()
    dr = ((rs - rs_ref).abs() / rs_ref).max().item()
    def t(fn, n=12):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return sorted(ts)[len(ts) // 2]
    t_gemm = t(lambda: ops.gemm(A, W, lib.EPI_BIAS_RES_F32, bias=bias, res=res))
    t_ln = t(lambda: ops.ln_fwd(x_ref, gam, bet))
    def both():
        xx, _ = ops.gemm(A, W, lib.EPI_BIAS_RES_F32, bias=bias, res=res)
        ops.ln_fwd(xx, gam, bet)
    t_both = t(both)
    t_fused = t(lambda: ops.gemm_res_ln(A, W, bias, res, gam, bet))
    print(f"{name:12s} M={M:6d} K={K:5d}: x max abs diff {dx:.2e}, y max abs diff {dy:.3f} ({ny * 100:.3f} % of bf16 values differ), "
          f"mean {dm:.1e}, rstd rel {dr:.1e} | gemm {t_gemm:7.1f} us + ln {t_ln:6.1f} us, back to back {t_both:7.1f} us, "
          f"fused {t_fused:7.1f} us", flush=True)
