"""ctypes binding of the C ABI declared in include/ccnet_cca.h.

The product loads ``ccnet_amd/csrc/libccnet_cca.so`` (built for gfx950 by ``__graft_entry__.build()``).
There is deliberately NO fallback: if the library is missing, or a tensor is not on a HIP device, the
callers in :mod:`ccnet_amd.functions` raise.  ``import torch`` must precede the ``CDLL`` so that the HIP
runtime already mapped by PyTorch (same SONAME ``libamdhip64.so.7``) is the one the library binds to.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_int, c_long, c_size_t, c_void_p
from typing import List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libccnet_cca.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ccnet_cca.h")

CCNET_CCA_VERSION = 220        # include/ccnet_cca.h
CCNET_CA_ENERGY = 0
CCNET_CA_SOFTMAX = 1
CCNET_IMPL_AUTO = 0
CCNET_IMPL_DIRECT = 1
CCNET_IMPL_MFMA = 2
CCNET_PRECISION_F32 = 0
CCNET_PRECISION_BF16X3 = 1
CCNET_PRECISION_DEFAULT = 2
CCNET_WS_SOFTMAX_BACKWARD, CCNET_WS_FORWARD, CCNET_WS_BACKWARD = 0, 1, 2
CCNET_WS_PM_FORWARD, CCNET_WS_PM_BACKWARD, CCNET_WS_PLANES_FORWARD, CCNET_WS_PLANES_BACKWARD = 3, 4, 5, 6
CCNET_WS_SPLIT_COLSUM = 7
CCNET_WS_PLANES3_BACKWARD = 8

_P = c_void_p  # every tensor argument is a raw device pointer

# name -> (restype, argtypes); mirrors include/ccnet_cca.h one to one
_PROTOTYPES = {
    "ccnet_cca_version": (c_int, []),
    "ccnet_cca_arch": (c_char_p, []),
    "ccnet_cca_last_error_string": (c_char_p, []),
    "ccnet_cca_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ccnet_ca_forward_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ccnet_ca_backward_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ccnet_ca_softmax_forward_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ccnet_ca_softmax_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, _P]),
    "ccnet_ca_map_forward_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ccnet_ca_map_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ccnet_cca_forward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ccnet_cca_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t,
                                       c_int, c_int, c_int, c_int, c_int, _P]),
    "ccnet_cca_forward_ws_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                         c_long, c_long, c_long, _P, c_size_t, _P]),
    "ccnet_cca_attention_strided_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_long, c_long, _P]),
    "ccnet_cca_backward_strided_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t,
                                               c_int, c_int, c_int, c_int, c_int,
                                               c_long, c_long, c_long, c_long, c_long, c_long, _P]),
    "ccnet_cca_forward_pm_bf16": (c_int, [_P] * 7 + [c_int] * 5 + [c_long, c_int] * 5 + [_P, c_size_t, _P]),
    "ccnet_cca_backward_pm_bf16": (c_int, [_P] * 11 + [c_int] * 5 + [c_long, c_int] * 7 + [_P, c_size_t, _P]),
    "ccnet_cca_forward_pm_f32": (c_int, [_P] * 7 + [c_int] * 5 + [c_long, c_int] * 5 + [_P, c_size_t, _P]),
    "ccnet_cca_backward_pm_f32": (c_int, [_P] * 11 + [c_int] * 5 + [c_long, c_int] * 7 + [_P, c_size_t, _P]),
    "ccnet_cca_split_planes_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_int, _P, _P]),
    "ccnet_cca_split_planes_colsum_f32": (c_int, [_P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_int, _P]),
    "ccnet_cca_nchw_to_planes_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_long, c_long, c_int, c_int, _P]),
    "ccnet_cca_pack_projection_f32": (c_int, [_P] * 10 + [c_int, c_int, _P]),
    "ccnet_cca_projection_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_long, _P]),
    "ccnet_cca_projection_adjoint_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_long, c_long, c_long, _P]),
    "ccnet_cca_projection_wgrad_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_int, _P]),
    "ccnet_cca_forward_planes_f32": (c_int, [_P] * 9 + [c_int] * 5 + [c_long, c_int] * 4 + [_P, c_size_t, _P]),
    "ccnet_cca_backward_planes_f32": (c_int, [_P] * 12 + [c_int] * 5 + [c_long, c_int] * 7 + [_P, c_size_t, _P]),
    "ccnet_cca_backward_planes3_f32": (c_int, [_P] * 10 + [c_int] * 5 + [c_long, c_int] * 4 + [_P, c_size_t, _P]),
    "ccnet_cca_attention_pm": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_long, c_int, c_long, c_int, _P]),
    "ccnet_cca_shape_uses_mfma": (c_int, [c_int, c_int, c_int, c_int]),
    "ccnet_cca_mfma_selftest": (c_int, [_P, _P]),
    "ccnet_cca_set_option": (c_int, [c_char_p, c_int, ctypes.POINTER(c_int)]),
    "ccnet_cca_get_option": (c_int, [c_char_p, ctypes.POINTER(c_int)]),
    "ccnet_cca_probe_clock": (c_int, [_P, c_int, c_int, c_int, _P]),
    "ccnet_cca_probe_mfma": (c_int, [_P, _P, c_int, c_int, _P]),
    "ccnet_cca_probe_dma": (c_int, [_P, c_size_t, _P, c_int, c_int, c_int, _P]),
    "ccnet_cca_profile_begin": (c_int, [c_int]),
    "ccnet_cca_profile_end": (c_int, [_P, _P, c_int, c_int]),
}


def declared_symbols(header: str = HEADER_PATH) -> List[str]:
    """Every function name include/ccnet_cca.h declares (used by the symbol-export test)."""
    with open(header) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ccnet_\w+)\s*\(", text)))


class CcaError(RuntimeError):
    pass


class CcaLibrary:
    """A loaded libccnet_cca.so (or, in the CPU tests, the emulator build of the same sources)."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise CcaError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()').  "
                "ccnet_amd has no CPU or PyTorch fallback for the criss-cross attention kernels.")
        self.path = path
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(self.dll, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if self.ccnet_cca_version() != CCNET_CCA_VERSION:
            raise CcaError(f"{path} exports C ABI version {self.ccnet_cca_version()}, this binding is written against "
                           f"{CCNET_CCA_VERSION} (include/ccnet_cca.h): rebuild the extension")

    # ---- options by name: the C calls return a status and pass values through out-parameters ----
    def set_option(self, name, value: int) -> int:
        """Set option ``name``; returns its previous value.  Unknown names / out-of-range values raise CcaError."""
        prev = c_int(0)
        self.check(self.ccnet_cca_set_option(name.encode() if isinstance(name, str) else name, int(value), ctypes.byref(prev)),
                   f"set_option({name!r}, {value})")
        return prev.value

    def get_option(self, name) -> int:
        v = c_int(0)
        self.check(self.ccnet_cca_get_option(name.encode() if isinstance(name, str) else name, ctypes.byref(v)), f"get_option({name!r})")
        return v.value

    # ---- named forms of the two generic entry points (options by name, workspace sizes by entry code) ----
    def ccnet_cca_set_impl(self, impl: int) -> int:
        return self.set_option("impl", impl)

    def ccnet_cca_get_impl(self) -> int:
        return self.get_option("impl")

    def ccnet_cca_set_precision(self, precision: int) -> int:
        return self.set_option("precision", precision)

    def ccnet_cca_get_precision(self) -> int:
        return self.get_option("precision")

    def ccnet_cca_set_branch_mask(self, mask: int) -> int:
        return self.set_option("branch_mask", mask)

    def ccnet_ca_softmax_backward_workspace_bytes(self, B, H, W) -> int:
        return self.ccnet_cca_workspace_bytes(CCNET_WS_SOFTMAX_BACKWARD, B, 0, 0, H, W)

    def ccnet_cca_forward_workspace_bytes(self, B, C, Cq, H, W) -> int:
        return self.ccnet_cca_workspace_bytes(CCNET_WS_FORWARD, B, C, Cq, H, W)

    def ccnet_cca_backward_workspace_bytes(self, B, C, Cq, H, W) -> int:
        return self.ccnet_cca_workspace_bytes(CCNET_WS_BACKWARD, B, C, Cq, H, W)

    def ccnet_cca_pm_workspace_bytes(self, B, C, Cq, H, W, backward) -> int:
        return self.ccnet_cca_workspace_bytes(CCNET_WS_PM_BACKWARD if backward else CCNET_WS_PM_FORWARD, B, C, Cq, H, W)

    def ccnet_cca_planes_workspace_bytes(self, B, C, Cq, H, W, backward) -> int:
        return self.ccnet_cca_workspace_bytes(CCNET_WS_PLANES_BACKWARD if backward else CCNET_WS_PLANES_FORWARD, B, C, Cq, H, W)

    def last_error(self) -> str:
        return self.ccnet_cca_last_error_string().decode()

    def check(self, code: int, what: str = "") -> None:
        if code != 0:
            raise CcaError(f"{what or 'ccnet_cca'} failed with code {code}: {self.last_error()}")

    def profile_launches(self, fn, cap: int = 256):
        """Run ``fn()`` with the launch profiler armed: [(kernel name, ms)] of every launch the library issued, in
        issue order (HIP-event pairs on the launch stream; a measurement aid, see ccnet_cca_profile_begin)."""
        self.check(self.ccnet_cca_profile_begin(cap), "profile_begin")
        try:
            fn()
        finally:
            ms = (ctypes.c_float * cap)()
            names = ctypes.create_string_buffer(cap * 96)
            n = self.ccnet_cca_profile_end(ms, names, 96, cap)
        if n < 0:
            raise CcaError(f"profile_end failed with code {n}: {self.last_error()}")
        out = []
        for i in range(n):
            raw = names.raw[i * 96:(i + 1) * 96].split(b"\0", 1)[0].decode()
            out.append((raw, float(ms[i])))
        return out


_lib: Optional[CcaLibrary] = None


def get_lib() -> CcaLibrary:
    """The process-wide device library; raises CcaError when it has not been built."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (map PyTorch's HIP runtime first, see module docstring)
        _lib = CcaLibrary(LIB_PATH)
    return _lib


def kernel_source_sha16() -> str:
    """sha256 prefix over the kernel sources and the C ABI header (file names + contents, sorted).  hipcc output is not
    bit-reproducible, so measurements that must belong to "this build" (profiles/traffic_*.json) are keyed to the sources
    the library is compiled from; __graft_entry__.build() rebuilds the library whenever a source is newer than it."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    files = [os.path.join(root, "csrc", f) for f in sorted(os.listdir(os.path.join(root, "csrc")))
             if f.endswith((".hip", ".hpp"))] + [HEADER_PATH]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
