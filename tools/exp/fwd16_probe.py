#!/usr/bin/env python3
"""A few launches of cogdl_hip_linear_fwd_bf16 (232,965 x 602 fp32 -> 64) and, beside it, a plain device copy of the same bytes,
for rocprofv3 --pmc (tools/exp/fwd16_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import linear as cl  # noqa: E402

x = torch.randn(232965, 602, device="cuda:0")
w = torch.randn(602, 64, device="cuda:0") * 0.05
for _ in range(6):
    cl.tall_skinny_matmul_bf16(x, w, None, False)
torch.cuda.synchronize()
