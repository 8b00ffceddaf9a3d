#!/usr/bin/env python3
"""One shape of the bf16x3 GEMM a few times (for rocprofv3 --pmc passes: scripts/gemm_x3_pmc.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpcs_amd import _lib  # noqa: E402

M, K, L, B = 682, 341, 1024, 32
W = torch.randn(M, K, device="cuda") / K ** 0.5
At = W.t().contiguous()
X = torch.randn(B, K, 3, L, device="cuda")
out = torch.empty(B, M, 3, L, device="cuda")
for _ in range(3):
    _lib.call("hpcs_gemm_tn_x3", At, M, X, 3 * L, K * 3 * L, L, out, 3 * L, M * 3 * L, L, M, L, K, B, 3, None, 0)
torch.cuda.synchronize()
