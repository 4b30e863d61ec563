"""Drop-in for the reference's ``cc_attention`` package (/root/reference/cc_attention/__init__.py:1).

Put this repository ahead of the reference on ``sys.path`` and ``from cc_attention import
CrissCrossAttention`` (networks/ccnet.py:13) resolves to the MI355X-native module, unchanged call site.
"""
from ccnet_amd.functions import CrissCrossAttention, CA_Weight, CA_Map, ca_weight, ca_map, INF  # noqa: F401
