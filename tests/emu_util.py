"""Build and drive the CPU SIMT-emulator build of the kernel sources (tests/emu/).

The emulator library exports the same C ABI as libccnet_cca.so (it is the same cca_api.hip compiled
for the host against tests/emu/cca_platform.hpp), so the tests drive it through ccnet_amd._lib.CcaLibrary with numpy
buffers standing in for device memory.  Test infrastructure only.
"""
import os
import subprocess

import numpy as np

from ccnet_amd._lib import CCNET_CA_ENERGY, CCNET_CA_SOFTMAX, CcaLibrary

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "ccnet_amd", "csrc")
EMU_LIB = os.path.join(EMU_DIR, "libcca_emu.so")
HOST_CXX = "/opt/rocm/lib/llvm/bin/clang++"


def _sources():
    srcs = [os.path.join(EMU_DIR, f) for f in ("hip_emu.cpp", "hip_emu.hpp", "cca_platform.hpp")]
    srcs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(ROOT, "include", "ccnet_cca.h"))
    return srcs


def build_emu(force=False):
    cxx = HOST_CXX if os.path.exists(HOST_CXX) else "g++"
    if not force and os.path.exists(EMU_LIB):
        if os.path.getmtime(EMU_LIB) >= max(os.path.getmtime(s) for s in _sources()):
            return EMU_LIB
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wno-pass-failed",
           "-I" + EMU_DIR, "-I" + CSRC,          # tests/emu FIRST: <cca_platform.hpp> resolves to the emulator's
           os.path.join(CSRC, "cca_api.hip"),
           os.path.join(EMU_DIR, "hip_emu.cpp"), "-o", EMU_LIB]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return EMU_LIB


def _p(a):
    return None if a is None else a.ctypes.data


class EmuOps:
    """numpy front-end of the emulated C ABI; every method mirrors one entry point."""

    def __init__(self):
        self.lib = CcaLibrary(build_emu())

    def set_impl(self, impl):
        return self.lib.ccnet_cca_set_impl(impl)

    def ca_forward(self, q, k, softmax=False):
        B, Cq, H, W = q.shape
        out = np.full((B, H, W, H + W), np.nan, np.float32)
        self.lib.check(self.lib.ccnet_ca_forward_f32(_p(q), _p(k), _p(out), B, Cq, H, W,
                                                     CCNET_CA_SOFTMAX if softmax else CCNET_CA_ENERGY, None))
        return out

    def ca_backward(self, dE, q, k):
        B, Cq, H, W = q.shape
        dq = np.full_like(q, np.nan)
        dk = np.full_like(k, np.nan)
        self.lib.check(self.lib.ccnet_ca_backward_f32(_p(dE), _p(q), _p(k), _p(dq), _p(dk), B, Cq, H, W, None))
        return dq, dk

    def softmax_forward(self, e):
        B, H, W, S = e.shape
        out = np.full_like(e, np.nan)
        self.lib.check(self.lib.ccnet_ca_softmax_forward_f32(_p(e), _p(out), B, H, W, None))
        return out

    def softmax_backward(self, A, dA, gamma=None, want_dgamma=True):
        B, H, W, S = A.shape
        dE = np.full_like(A, np.nan)
        dgamma = np.full(1, np.nan, np.float32) if want_dgamma else None
        nbytes = self.lib.ccnet_ca_softmax_backward_workspace_bytes(B, H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        self.lib.check(self.lib.ccnet_ca_softmax_backward_f32(_p(A), _p(dA), _p(gamma), _p(dE), _p(dgamma),
                                                              _p(ws), nbytes, B, H, W, None))
        return dE, dgamma

    def ca_map_forward(self, A, v, x=None, gamma=None):
        B, C, H, W = v.shape
        out = np.full_like(v, np.nan)
        self.lib.check(self.lib.ccnet_ca_map_forward_f32(_p(A), _p(v), _p(x), _p(gamma), _p(out), B, C, H, W, None))
        return out

    def ca_map_backward(self, dout, A, v, gamma=None):
        B, C, H, W = v.shape
        dA = np.full_like(A, np.nan)
        dv = np.full_like(v, np.nan)
        self.lib.check(self.lib.ccnet_ca_map_backward_f32(_p(dout), _p(A), _p(v), _p(gamma), _p(dA), _p(dv),
                                                          B, C, H, W, None))
        return dA, dv

    def cca_forward(self, q, k, v, x, gamma):
        B, C, H, W = v.shape
        y = np.full_like(v, np.nan)
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_forward_f32(_p(q), _p(k), _p(v), _p(x), _p(gamma), _p(y), _p(A),
                                                      B, C, q.shape[1], H, W, None))
        return y, A

    def cca_forward_ws(self, q, k, v, x, gamma):
        """fused forward with the optional small-batch workspace (K-split slabs); returns (y, A, workspace bytes)"""
        B, C, H, W = v.shape
        cq = q.shape[1]
        y = np.full_like(v, np.nan)
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        n = self.lib.ccnet_cca_forward_workspace_bytes(B, C, cq, H, W)
        ws = np.full(n // 4 + 1, np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_forward_ws_f32(_p(q), _p(k), _p(v), _p(x), _p(gamma), _p(y), _p(A), B, C, cq, H, W,
                                                         cq * H * W, cq * H * W, C * H * W, _p(ws), n, None))
        return y, A, n

    def cca_backward_ws(self, dy, q, k, v, A, gamma):
        B, C, H, W = v.shape
        dq, dk, dv = np.full_like(q, np.nan), np.full_like(k, np.nan), np.full_like(v, np.nan)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_cca_backward_workspace_bytes(B, C, q.shape[1], H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_backward_f32(_p(dy), _p(q), _p(k), _p(v), _p(A), _p(gamma), _p(dq), _p(dk),
                                                       _p(dv), _p(dgamma), _p(scratch), _p(ws), nbytes,
                                                       B, C, q.shape[1], H, W, None))
        return dq, dk, dv, dgamma, nbytes

    def cca_attention_packed(self, qkv, cq, H, W):
        B = qkv.shape[0]
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        hw, bs = H * W * 4, qkv.shape[1] * H * W
        base = qkv.ctypes.data
        self.lib.check(self.lib.ccnet_cca_attention_strided_f32(base, base + cq * hw, _p(A), B, cq, H, W, bs, bs, None))
        return A

    def cca_backward(self, dy, q, k, v, A, gamma):
        B, C, H, W = v.shape
        dq, dk, dv = np.full_like(q, np.nan), np.full_like(k, np.nan), np.full_like(v, np.nan)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_ca_softmax_backward_workspace_bytes(B, H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_backward_f32(_p(dy), _p(q), _p(k), _p(v), _p(A), _p(gamma), _p(dq), _p(dk),
                                                       _p(dv), _p(dgamma), _p(scratch), _p(ws), nbytes,
                                                       B, C, q.shape[1], H, W, None))
        return dq, dk, dv, dgamma

    def cca_forward_packed(self, qkv, x, gamma, cq):
        """q, k, v are channel slices of one (B, 2*cq+C, H, W) array (the strided entry point)."""
        B, C, H, W = x.shape
        y = np.full_like(x, np.nan)
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        hw, bs = H * W * 4, (2 * cq + C) * H * W
        base = qkv.ctypes.data
        self.lib.check(self.lib.ccnet_cca_forward_ws_f32(base, base + cq * hw, base + 2 * cq * hw, _p(x), _p(gamma),
                                                         _p(y), _p(A), B, C, cq, H, W, bs, bs, bs, None, 0, None))
        return y, A

    def cca_backward_packed(self, dy, qkv, A, gamma, cq):
        B, C, H, W = dy.shape
        dqkv = np.full_like(qkv, np.nan)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_ca_softmax_backward_workspace_bytes(B, H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        hw, bs = H * W * 4, (2 * cq + C) * H * W
        p, g = qkv.ctypes.data, dqkv.ctypes.data
        self.lib.check(self.lib.ccnet_cca_backward_strided_f32(_p(dy), p, p + cq * hw, p + 2 * cq * hw, _p(A), _p(gamma),
                                                               g, g + cq * hw, g + 2 * cq * hw, _p(dgamma), _p(scratch),
                                                               _p(ws), nbytes, B, C, cq, H, W,
                                                               bs, bs, bs, bs, bs, bs, None))
        return dqkv, dgamma

    def cca_forward_pm_bf16(self, qkv, x, gamma, cq):
        """qkv: uint16 (B, H, W, 2*cq + C) packed pixel-major projection (q | k | v channel slices, bf16 bit patterns),
        x: uint16 (B, H, W, ps >= C); returns (y bits (B, H, W, C), A fp32).  float32 arrays take the fp32 entry points."""
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        y = np.zeros((B, H, W, C), qkv.dtype)
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        f32 = qkv.dtype == np.float32
        es = 4 if f32 else 2
        fwd = self.lib.ccnet_cca_forward_pm_f32 if f32 else self.lib.ccnet_cca_forward_pm_bf16
        nbytes = self.lib.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 0)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        base, bs = qkv.ctypes.data, H * W * ct
        self.lib.check(fwd(base, base + es * cq, base + 2 * es * cq, _p(x), _p(gamma), _p(y), _p(A),
                                                          B, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                                          H * W * x.shape[3], x.shape[3], H * W * C, C, _p(ws), nbytes, None))
        return y, A

    def cca_backward_pm_bf16(self, dy, qkv, A, gamma, cq):
        """dy: uint16 (B, H, W, C); returns (dqkv bits packed like qkv, dgamma)."""
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        dqkv = np.zeros_like(qkv)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 1)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        base, g, bs = qkv.ctypes.data, dqkv.ctypes.data, H * W * ct
        f32 = qkv.dtype == np.float32
        es = 4 if f32 else 2
        bwd = self.lib.ccnet_cca_backward_pm_f32 if f32 else self.lib.ccnet_cca_backward_pm_bf16
        self.lib.check(bwd(_p(dy), base, base + es * cq, base + 2 * es * cq, _p(A), _p(gamma),
                                                           g, g + es * cq, g + 2 * es * cq, _p(dgamma), _p(scratch),
                                                           B, C, cq, H, W, H * W * C, C, bs, ct, bs, ct, bs, ct,
                                                           bs, ct, bs, ct, bs, ct, _p(ws), nbytes, None))
        return dqkv, dgamma

    def split_planes(self, src_pm, C, c0=0, layout=2, bias=None):
        """src_pm: float32 (B, H, W, ps) pixel-major; channels [c0, c0 + C) -> planes uint16 (B, H, W, n, C)
        (layout = CCNET_PLANES_HL 2: hi | lo; HLH 3: hi | lo | hi; HHL 4: hi | hi | lo)."""
        B, H, W, ps = src_pm.shape
        n = 2 if layout == 2 else 3
        dst = np.full((B, H, W, n, C), 0xFFFF, np.uint16)
        self.lib.check(self.lib.ccnet_cca_split_planes_f32(src_pm.ctypes.data + 4 * c0, _p(dst), B, C, H, W, H * W * ps, ps,
                                                           H * W * n * C, n * C, layout, None if bias is None else _p(bias), None))
        return dst

    def split_planes_colsum(self, src_pm, layout=3):
        """ccnet_cca_split_planes_colsum_f32: (planes uint16 (B, H, W, n, C), column sums (C,)) of the whole (B, H, W, C) tensor"""
        B, H, W, C = src_pm.shape
        n = 2 if layout == 2 else 3
        dst = np.full((B, H, W, n, C), 0xFFFF, np.uint16)
        colsum = np.full(C, np.nan, np.float32)
        nbytes = self.lib.ccnet_cca_workspace_bytes(7, B, C, 0, H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_split_planes_colsum_f32(_p(src_pm), _p(dst), _p(colsum), _p(ws), nbytes, B, C, H, W,
                                                                  H * W * C, C, H * W * n * C, n * C, layout, None))
        return dst, colsum

    def nchw_to_planes(self, src, layout=2):
        B, C, H, W = src.shape
        n = 2 if layout == 2 else 3
        dst = np.full((B, H, W, n, C), 0xFFFF, np.uint16)
        self.lib.check(self.lib.ccnet_cca_nchw_to_planes_f32(_p(src), _p(dst), B, C, H, W, C * H * W, H * W * n * C, n * C, layout, None))
        return dst

    def cca_forward_planes(self, qkv, v_planes, x, gamma, cq, v_from_qkv=False, v_bias=None):
        """qkv: float32 (B, H, W, ct) packed pixel-major projection (q | k read from it); v_planes uint16 (B, H, W, 2, C): the
        pre-split value planes, or (``v_from_qkv``) an OUTPUT the entry point fills from the fp32 value slice of qkv (+ v_bias),
        or None: the PLANE-FREE form (v read as fp32 out of qkv, nothing written); x float32 NCHW; returns (y NCHW, A)."""
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        y = np.full((B, C, H, W), np.nan, np.float32)
        A = np.full((B, H, W, H + W), np.nan, np.float32)
        nbytes = self.lib.ccnet_cca_planes_workspace_bytes(B, C, cq, H, W, 0)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        base, bs = qkv.ctypes.data, H * W * ct
        self.lib.check(self.lib.ccnet_cca_forward_planes_f32(base, base + 4 * cq, base + 8 * cq if (v_from_qkv or v_planes is None) else None,
                                                             _p(v_bias), _p(v_planes), _p(x), _p(gamma), _p(y), _p(A),
                                                             B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * 2 * C, 2 * C,
                                                             _p(ws), nbytes, None))
        return y, A

    def cca_backward_planes(self, dy, qkv, v_planes, A, gamma, cq):
        """``v_planes`` None: the plane-free form (v read as fp32 out of qkv)"""
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        dqkv = np.full((B, H, W, 2 * cq + C), np.nan, np.float32)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_cca_planes_workspace_bytes(B, C, cq, H, W, 1)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        base, g, bs = qkv.ctypes.data, dqkv.ctypes.data, H * W * ct
        dct = 2 * cq + C
        dbs = H * W * dct
        self.lib.check(self.lib.ccnet_cca_backward_planes_f32(_p(dy), base, base + 4 * cq, base + 8 * cq if v_planes is None else None,
                                                              _p(v_planes), _p(A), _p(gamma),
                                                              g, g + 4 * cq, g + 8 * cq, _p(dgamma), _p(scratch),
                                                              B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * 2 * C, 2 * C,
                                                              dbs, dct, dbs, dct, dbs, dct, _p(ws), nbytes, None))
        return dqkv, dgamma

    def cca_backward_planes3(self, dy, qkv, A, gamma, cq):
        """ccnet_cca_backward_planes3_f32: (d3 (B, H, W, 3, ct) bf16 bits, dbias (ct), dgamma)"""
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        d3 = np.full((B, H, W, 3, ct), 0x7fc0, np.uint16)
        db = np.full(ct, np.nan, np.float32)
        dgamma = np.full(1, np.nan, np.float32)
        scratch = np.full_like(A, np.nan)
        nbytes = self.lib.ccnet_cca_workspace_bytes(8, B, C, cq, H, W)
        ws = np.full(nbytes // 4 + 1, np.nan, np.float32)
        base, bs = qkv.ctypes.data, H * W * ct
        self.lib.check(self.lib.ccnet_cca_backward_planes3_f32(_p(dy), base, base + 4 * cq, base + 8 * cq, _p(A), _p(gamma), _p(d3), _p(db),
                                                               _p(dgamma), _p(scratch), B, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                                               H * W * 3 * ct, 3 * ct, _p(ws), nbytes, None))
        return d3, db, dgamma

    def projection_bf16(self, a_bits, wt_bits, bias):
        """ccnet_cca_projection_bf16: a (M, K) / wt (N, K) uint16 bf16 bits, bias (N) fp32 or None -> out (M, N) fp32"""
        M, K = a_bits.shape
        N = wt_bits.shape[0]
        out = np.full((M, N), np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_projection_bf16(_p(a_bits), _p(wt_bits), _p(bias), _p(out), M, N, K, K, K, N, None))
        return out

    def projection_adjoint_bf16(self, w_bits, d_bits, add):
        """ccnet_cca_projection_adjoint_bf16: w (C, K) / d (B, P, K) uint16 bf16 bits, add (B, C, P) fp32 or None -> dx (B, C, P) fp32"""
        C, K = w_bits.shape
        B, P = d_bits.shape[0], d_bits.shape[1]
        dx = np.full((B, C, P), np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_projection_adjoint_bf16(_p(w_bits), _p(d_bits), _p(add), _p(dx), B, C, P, K, K, K, P * K, None))
        return dx

    def projection_wgrad_bf16(self, d_bits, x_bits, S):
        """ccnet_cca_projection_wgrad_bf16: d (R, N) / x (R, C) uint16 bf16 bits -> part (S, N, C) fp32"""
        R, N = d_bits.shape
        C = x_bits.shape[1]
        part = np.full((S, N, C), np.nan, np.float32)
        self.lib.check(self.lib.ccnet_cca_projection_wgrad_bf16(_p(d_bits), _p(x_bits), _p(part), R, N, C, N, C, S, None))
        return part

    def pack_projection(self, wq, bq, wk, bk, wv, bv, split=True):
        """ccnet_cca_pack_projection_f32: (w (N, C) fp32, b (N), w3 (N, 3C) bf16 bits, w3t (C, 3N) bf16 bits)"""
        cq, C = wq.shape[0], wq.shape[1]
        n = 2 * cq + C
        w, b = np.full((n, C), np.nan, np.float32), np.full(n, np.nan, np.float32)
        w3 = np.full((n, 3 * C), 0xFFFF, np.uint16) if split else None
        w3t = np.full((C, 3 * n), 0xFFFF, np.uint16) if split else None
        self.lib.check(self.lib.ccnet_cca_pack_projection_f32(_p(wq), _p(bq), _p(wk), _p(bk), _p(wv), _p(bv), _p(w), _p(b),
                                                              _p(w3), _p(w3t), C, cq, None))
        return w, b, w3, w3t

    def mfma_selftest(self):
        scratch = np.zeros(16, np.float32)
        return self.lib.ccnet_cca_mfma_selftest(_p(scratch), None)


def emu_stats(ops, reset=False):
    """(lds_read_instr, lds_read_cycles, lds_write_instr, lds_write_cycles, mfma lane-calls, launches)."""
    import ctypes
    buf = (ctypes.c_ulonglong * 6)()
    ops.lib.dll.cca_emu_stats(buf)
    if reset:
        ops.lib.dll.cca_emu_reset_stats()
    return tuple(buf)
