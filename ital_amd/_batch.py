"""Device buffers of one greedy batch construction (the replicated batch state of include/ital_hip.h `ital_batch`,
the per-member covariance columns, the selection record and its gather target)."""
import torch

from ._lib import ItalBatch, lib as _lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def make_batch_buffers(device, kmax, ldx, cap, ldc, world):
    """kmax: batch capacity; ldx: padded feature dimension; cap: labelled-set capacity (leading dimension of the
    whitened columns); ldc: length of one covariance column; world: number of ranks exchanging records."""
    kmax = max(int(kmax), 4)
    f64, i64, i32 = torch.float64, torch.int64, torch.int32
    b = dict(kmax=kmax, ldw=cap, ldc=ldc, ldx=ldx)
    b["bidx"] = torch.zeros(kmax, dtype=i64, device=device)
    b["bgpos"] = torch.zeros(kmax, dtype=i64, device=device)
    b["bsort"] = torch.zeros(kmax, dtype=i32, device=device)
    b["bmu"] = torch.zeros(kmax, dtype=f64, device=device)
    b["sig"] = torch.zeros(kmax * kmax, dtype=f64, device=device)
    b["XB"] = torch.zeros((kmax, ldx), dtype=f64, device=device)
    b["XBn"] = torch.zeros(kmax, dtype=f64, device=device)
    b["VB"] = torch.zeros((kmax, cap), dtype=f64, device=device)
    b["C"] = torch.zeros((kmax, ldc), dtype=f64, device=device)
    b["ret"] = torch.zeros(kmax + 1, dtype=i64, device=device)   # last slot: copy of the status word (one download)
    b["work"] = torch.zeros(3 * 1024, dtype=f64, device=device)
    rec_len = int(_lib().ital_record_len(ldx, cap, kmax))      # = ITAL_REC_HEADER + ldx + cap + kmax
    b["rec_len"] = rec_len
    b["rec"] = torch.zeros(rec_len, dtype=f64, device=device)
    b["rec_all"] = torch.zeros((world, rec_len), dtype=f64, device=device)
    b["jump"] = {}
    b["jumppat"] = {}
    b["vk"] = {}
    b["batch"] = ItalBatch(kmax, ldx, cap, _ptr(b["bidx"]), _ptr(b["bgpos"]), _ptr(b["bsort"]), _ptr(b["bmu"]),
                           _ptr(b["sig"]), _ptr(b["XB"]), _ptr(b["XBn"]), _ptr(b["VB"]))
    return b
