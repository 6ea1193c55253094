#!/usr/bin/env python
"""What a batched isotropic-Gaussian callback costs in torch on the device tensor of proposals [n, d] (bench.py --callback): candidates
for the stand-in likelihood, timed with HIP events.  python tools/callback_cost.py [n] [d]"""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d = int(sys.argv[2]) if len(sys.argv) > 2 else 100
X = torch.randn(n, d, dtype=torch.float64, device="cuda")
ones = torch.ones(d, dtype=torch.float64, device="cuda")
cands = {
    "naive  -0.5 * (X * X).sum(-1)": lambda: -0.5 * (X * X).sum(-1),
    "norm   vector_norm(X, dim=-1).square_().mul_(-0.5)": lambda: torch.linalg.vector_norm(X, dim=-1).square_().mul_(-0.5),
    "vecdot linalg.vecdot(X, X).mul_(-0.5)": lambda: torch.linalg.vecdot(X, X).mul_(-0.5),
    "bmm    bmm(X[:, None, :], X[:, :, None]).view(-1).mul_(-0.5)": lambda: torch.bmm(X.unsqueeze(1), X.unsqueeze(2)).view(-1).mul_(-0.5),
    "einsum einsum('nd,nd->n', X, X).mul_(-0.5)": lambda: torch.einsum("nd,nd->n", X, X).mul_(-0.5),
    "sqsum  X.square().sum(-1).mul_(-0.5)": lambda: X.square().sum(-1).mul_(-0.5),
    "normT  vector_norm(X.view(-1, 2, d)...)": lambda: torch.linalg.vector_norm(X.view(-1, 2 * d)[:, :d], dim=-1),
    "copy   X.clone() (a plain read + write of the tensor, for scale)": lambda: X.clone(),
    "sum    X.sum() (a plain read of the tensor, for scale)": lambda: X.sum(),
}
for name, f in cands.items():
    try:
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-70s %8.1f us  %6.2f TB/s of the tensor's bytes" % (name, ms * 1e3, n * d * 8 / ms / 1e9))
    except Exception as e:      # noqa: BLE001
        print("%-70s failed: %r" % (name, e))
