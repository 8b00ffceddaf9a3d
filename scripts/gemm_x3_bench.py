#!/usr/bin/env python3
"""fp32 fma-chain GEMM (hpcs_gemm_tn) against the bf16x3 GEMM (hpcs_gemm_tn_x3) on the dense shapes of the step: time per
launch (HIP events) and error against the fp64 product.   usage: python scripts/gemm_x3_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpcs_amd import _lib  # noqa: E402

call = _lib.call
_lib.set_tuning("HPCS_GEMM_X3", 0)          # hpcs_gemm_tn below = the fma-chain kernel
dev = "cuda"
torch.manual_seed(0)
# (name, M, K, L, B, inner3): inner3 = walked as 3B matrices of L columns (the hoisted layers), else B matrices of 3L columns
shapes = [("vn1 fwd", 682, 341, 1024, 32, True), ("vn1 dgrad", 341, 682, 1024, 32, True), ("vn2 fwd", 340, 341, 1024, 32, False),
          ("vn2 dgrad", 341, 340, 1024, 32, False), ("conv6 fwd", 342, 63, 1024, 32, False), ("conv8 fwd", 256, 256, 1024, 32, None),
          ("conv9 fwd", 256, 256, 1024, 32, None), ("conv10 fwd", 128, 256, 1024, 32, None)]
for name, M, K, L, B, inner3 in shapes:
    W = torch.randn(M, K, device=dev) / K ** 0.5
    At = W.t().contiguous()
    lda = M
    if inner3 is None:                       # Conv1d head: [B, K, L]
        X = torch.randn(B, K, L, device=dev)
        args = lambda out: (At, lda, X, L, K * L, 0, out, L, M * L, 0, M, L, K, B, 1, None, 0)      # noqa: E731
        out_shape = (B, M, L)
        ref = lambda: torch.einsum("mk,bkl->bml", W.double(), X.double())                          # noqa: E731
    elif inner3:
        X = torch.randn(B, K, 3, L, device=dev)
        args = lambda out: (At, lda, X, 3 * L, K * 3 * L, L, out, 3 * L, M * 3 * L, L, M, L, K, B, 3, None, 0)      # noqa: E731
        out_shape = (B, M, 3, L)
        ref = lambda: torch.einsum("mk,bkal->bmal", W.double(), X.double())                        # noqa: E731
    else:
        X = torch.randn(B, K, 3, L, device=dev)
        args = lambda out: (At, lda, X, 3 * L, K * 3 * L, 0, out, 3 * L, M * 3 * L, 0, M, 3 * L, K, B, 1, None, 0)  # noqa: E731
        out_shape = (B, M, 3, L)
        ref = lambda: torch.einsum("mk,bkal->bmal", W.double(), X.double())                        # noqa: E731
    want = ref()
    flops = 2.0 * M * K * want.numel() / M
    res = {}
    for fn in ("hpcs_gemm_tn", "hpcs_gemm_tn_x3"):
        out = torch.empty(out_shape, device=dev)
        call(fn, *args(out))
        torch.cuda.synchronize()
        err = float((out.double() - want).norm() / want.norm())
        emax = float((out.double() - want).abs().max() / want.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call(fn, *args(out))
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        res[fn] = (t, err, emax)
    a, b = res["hpcs_gemm_tn"], res["hpcs_gemm_tn_x3"]
    print(f"{name:11s} M={M:4d} K={K:4d}  fma-chain {a[0] * 1e6:7.1f} us {flops / a[0] / 1e12:6.1f} TF  L2 {a[1]:.2e} max {a[2]:.2e} | "
          f"bf16x3 {b[0] * 1e6:7.1f} us {flops / b[0] / 1e12:6.1f} TF  L2 {b[1]:.2e} max {b[2]:.2e} | x{a[0] / b[0]:.2f}", flush=True)
