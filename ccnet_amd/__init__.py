"""ccnet_amd -- MI355X-native criss-cross attention (the hot path of speedinghzl/CCNet).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/ccnet_cca.h), the
ctypes binding (:mod:`ccnet_amd._lib`) and the host-side mirror of the reference's
``cc_attention/functions.py`` (:mod:`ccnet_amd.functions`).  Importing the package does not load the
device library; the first kernel call does, and fails loudly if it has not been built.
"""
from .functions import (CA_Map, CA_Weight, CrissCrossAttention, CrissCrossFunction, INF, ca_map, ca_softmax,
                        ca_weight, criss_cross_attention, graph_module)

__all__ = ["CrissCrossAttention", "CrissCrossFunction", "CA_Weight", "CA_Map", "ca_weight", "ca_map",
           "ca_softmax", "criss_cross_attention", "graph_module", "INF"]
__version__ = "0.1.0"
