"""``cc_attention.functions`` alias (the reference keeps the module in cc_attention/functions.py)."""
from ccnet_amd.functions import *  # noqa: F401,F403
from ccnet_amd.functions import CrissCrossAttention, INF  # noqa: F401
