"""ital_amd -- MI355X-native hot path of ITAL (GP-based mutual-information candidate selection).

Drop-in for the reference's learner API on that path:
    from ital_amd import ITAL            # reference: from ital.ital import ITAL
    learner = ITAL(data, length_scale=3.0); learner.update({q: 1}); learner.fetch_unlabelled(4)
The arithmetic lives in hand-written HIP kernels (ital_amd/csrc, C ABI in include/ital_hip.h).
"""
from . import mvn_stream
from .mvn_stream import GLOBAL as mvn_global_stream

__all__ = ["ITAL", "GaussianProcess", "ActiveRetrievalBase", "mvn_stream", "mvn_global_stream"]


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
