#!/usr/bin/env python3
"""HBM-bound streaming kernels at scale: row norms, RBF columns, cross-covariance column, whiten-append.
Prints achieved GB/s against the algorithmic bytes of SURVEY.md 8(d):  read 8(d + m) per row, write 8c per row."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import _lib
from ital_amd.gp import _ptr, _stream

lib = _lib.lib()
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


for n, d, m in [(1_000_000, 256, 21), (1_000_000, 512, 81), (4_000_000, 256, 41), (9298, 256, 21)]:
    ldx = d
    ldv = (n + 15) // 16 * 16
    cap = (m + 16 + 15) // 16 * 16
    X = torch.rand((n, ldx), dtype=torch.float64, device=dev)
    xn = torch.empty(n, dtype=torch.float64, device=dev)
    V = torch.rand((cap, ldv), dtype=torch.float64, device=dev) * 0.01
    out = torch.empty((16, ldv), dtype=torch.float64, device=dev)
    Xs = X[:16].clone()
    sn = torch.empty(16, dtype=torch.float64, device=dev)
    W = torch.rand((16, cap), dtype=torch.float64, device=dev) * 0.01
    L22 = torch.eye(16, dtype=torch.float64, device=dev).repeat(1, cap // 16 + 1)[:, :cap].contiguous()
    alpha = torch.rand(16, dtype=torch.float64, device=dev)
    mu = torch.zeros(n, dtype=torch.float64, device=dev)
    s2 = torch.ones(n, dtype=torch.float64, device=dev)
    st = _stream()
    lib.ital_row_norms(_ptr(Xs), 16, ldx, _ptr(sn), st)
    t = timeit(lambda: lib.ital_row_norms(_ptr(X), n, ldx, _ptr(xn), st))
    print(f"n={n} d={d} m={m} row_norms      {t*1e3:8.3f} ms  {8*n*d/t/1e9:8.1f} GB/s")
    for c in (1, 4, 16):
        t = timeit(lambda: lib.ital_rbf_cols(_ptr(X), _ptr(xn), n, ldx, _ptr(Xs), _ptr(sn), c, 1.0, 3.0, _ptr(out), ldv, st))
        print(f"n={n} d={d} m={m} rbf_cols c={c:2d}  {t*1e3:8.3f} ms  {8*n*(d+1+c)/t/1e9:8.1f} GB/s")
    for c in (1, 8):
        t = timeit(lambda: lib.ital_cross_cov_cols(_ptr(X), _ptr(xn), n, ldx, _ptr(Xs), _ptr(sn), c, _ptr(W), cap, _ptr(V), ldv,
                                                   m, 1.0, 3.0, _ptr(out), ldv, st))
        print(f"n={n} d={d} m={m} cross_cov c={c:2d} {t*1e3:8.3f} ms  {8*n*(d+m+1+c)/t/1e9:8.1f} GB/s")
    for c in (4, 16):
        t = timeit(lambda: lib.ital_whiten_append(_ptr(X), _ptr(xn), n, ldx, _ptr(Xs), _ptr(sn), c, _ptr(W), cap,
                                                  W.data_ptr() + 8 * m, _ptr(alpha), _ptr(V), ldv, m, 1.0, 3.0, _ptr(mu), _ptr(s2), st))
        print(f"n={n} d={d} m={m} whiten c={c:2d}    {t*1e3:8.3f} ms  {8*n*(d+m+1+c+4)/t/1e9:8.1f} GB/s")
    del X, V, out, mu, s2, xn
    torch.cuda.empty_cache()
