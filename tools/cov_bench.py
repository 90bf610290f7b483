#!/usr/bin/env python3
"""ital_cov_block / ital_cov_abs_rowsum: check against a torch float64 restatement and time them (TFLOP/s on the FP64
matrix cores).  Usage: tools/cov_bench.py [n d m]...   (ITAL_HIP_LIB selects a library variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import _lib
from ital_amd.gp import _ptr, _stream

dev = "cuda:0"
lib = _lib.lib()
sizes = [(9273, 256, 9), (1000, 256, 9), (2048, 256, 9), (3000, 256, 9), (4096, 512, 64), (777, 48, 5), (20000, 512, 17), (20000, 2048, 17)]
if len(sys.argv) > 3:
    a = list(map(int, sys.argv[1:]))
    sizes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for n, d, m in sizes:
    g = torch.Generator(device="cpu").manual_seed(n)
    ldx = (d + 15) // 16 * 16
    X = torch.zeros(n, ldx, dtype=torch.float64)
    X[:, :d] = torch.rand(n, d, generator=g, dtype=torch.float64)
    X = X.to(dev)
    nb = n - 3
    Xa, Xb = X, X[3:].contiguous()
    an, bn = (Xa * Xa).sum(1), (Xb * Xb).sum(1)
    Va = (0.05 * torch.randn(m, n, generator=g, dtype=torch.float64)).to(dev)
    Vb = Va[:, 3:].contiguous()
    ls, var = float((d / 12.0) ** 0.5), 1.2
    ldo = nb + 5
    out = torch.full((n, ldo), -7.0, dtype=torch.float64, device=dev)
    call = lambda: _lib.check(lib.ital_cov_block(_ptr(Xa), _ptr(an), n, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), n, _ptr(Vb), nb, m,
                                                 var, ls, _ptr(out), ldo, _stream()))
    call()
    torch.cuda.synchronize()
    rows = slice(0, min(n, 2048))
    want = var * torch.exp(-(an[rows, None] + bn[None, :] - 2 * Xa[rows] @ Xb.T) / (2 * ls * ls)) - Va[:, rows].T @ Vb
    err = (out[rows, :nb] - want).abs().max().item()
    tail = slice(max(0, n - 300), n)
    want2 = var * torch.exp(-(an[tail, None] + bn[None, :] - 2 * Xa[tail] @ Xb.T) / (2 * ls * ls)) - Va[:, tail].T @ Vb
    err = max(err, (out[tail, :nb] - want2).abs().max().item())
    pad_ok = bool((out[:, nb:] == -7.0).all())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * n * nb * (ldx + m) / (ms * 1e-3) / 1e12
    print("n %d d %d m %d: cov_block %.3f ms %.1f TFLOP/s, max abs err %.2e, padding untouched %s" % (n, d, m, ms, tf, err, pad_ok), flush=True)
