"""ital_amd -- MI355X-native hot path of ITAL (GP-based mutual-information candidate selection).

Drop-in for the reference's learner API on that path:
    from ital_amd import ITAL            # reference: from ital.ital import ITAL
    learner = ITAL(data, length_scale=3.0); learner.update({q: 1}); learner.fetch_unlabelled(4)
The arithmetic lives in hand-written HIP kernels (ital_amd/csrc, C ABI in include/ital_hip.h).
"""
from . import mvn_stream
from .mvn_stream import GLOBAL as mvn_global_stream

__all__ = ["ITAL", "MCMI_min", "GaussianProcess", "ActiveRetrievalBase", "mvn_stream", "mvn_global_stream", "serving_mode"]


def serving_mode():
    """For a long-running process that calls fetch_unlabelled() / update() in a loop (a retrieval server; bench.py): moves
    everything allocated so far -- the heap of an imported torch, ~10^5 objects -- out of the reach of CPython's cyclic
    garbage collector (gc.freeze()).  Without it the generation-2 collector walks that heap once every few rounds: a
    40 ms pause in a loop of 3 ms rounds.  Objects created afterwards are collected as usual."""
    import gc
    gc.collect()
    gc.freeze()


def __getattr__(name):
    # torch and the HIP library are only needed for the learners themselves
    if name == "ITAL":
        from .ital import ITAL
        return ITAL
    if name == "GaussianProcess":
        from .gp import GaussianProcess
        return GaussianProcess
    if name == "ActiveRetrievalBase":
        from .retrieval_base import ActiveRetrievalBase
        return ActiveRetrievalBase
    if name == "MCMI_min":
        from .mcmi import MCMI_min
        return MCMI_min
    raise AttributeError(name)
