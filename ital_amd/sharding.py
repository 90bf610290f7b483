"""Row / candidate sharding across ranks and the per-greedy-step record exchange.

Pure host logic (no GPU needed): used by ital_amd.gp / ital_amd.ital on the GPU box over RCCL ("nccl"
backend) and exercised on CPU with gloo in tests/test_dist_gloo.py.
"""
import numpy as np


def row_range(n_total, world, rank):
    """Contiguous row block [row0, row1) of `rank`; blocks differ by at most one row and tile [0, n_total)."""
    return n_total * rank // world, n_total * (rank + 1) // world


def shard_candidates(candidates, row0, row1):
    """Positions of the global candidate list that fall into this rank's rows.

    Returns (global data indices of the local positions, list position of the first one).  The local positions
    must form one contiguous run of the list (true for the ascending `get_unseen()` order) so that
    `global position = pos_offset + local position`, which is what the arg-max tie-break and the replay of the
    mvndst stream are keyed on."""
    cand = np.asarray(candidates, dtype=np.int64)
    mine = np.flatnonzero((cand >= row0) & (cand < row1))
    if len(mine) and not np.array_equal(mine, np.arange(mine[0], mine[0] + len(mine))):
        raise NotImplementedError("candidate order is not contiguous per rank (top_candidates across several ranks)")
    return cand[mine], (int(mine[0]) if len(mine) else 0)


def gather_records(record, out, group=None):
    """ONE collective per greedy step: every rank contributes its fixed-size record, all ranks receive all of them.
    `record` [R], `out` [world, R] (same dtype/device)."""
    import torch.distributed as dist
    world = out.shape[0]
    if world == 1:
        out[0].copy_(record)
        return out
    dist.all_gather(list(out.unbind(0)), record, group=group)
    return out


def winner(records, mode=0):
    """Host statement of the rule `ital_select_resolve` applies on the device (used by tests and documentation):
    records[:, 0] = score, records[:, 1] = global list position (< 0: rank had no live candidate).  mode 0: first
    maximum with NaN beating every number (np.argmax), mode 1: first minimum (np.argmin)."""
    best = -1
    for w in range(len(records)):
        v, p = records[w][0], records[w][1]
        if p < 0:
            continue
        if best < 0:
            best = w
            continue
        bv, bp = records[best][0], records[best][1]
        vn, bn = np.isnan(v), np.isnan(bv)
        if vn != bn:
            better = vn
        elif vn or v == bv:
            better = p < bp
        else:
            better = v > bv if mode == 0 else v < bv
        if better:
            best = w
    return best
