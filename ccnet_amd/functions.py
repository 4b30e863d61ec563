"""Host-side mirror of the reference's ``cc_attention/functions.py`` for MI355X.

Same public surface as /root/reference/cc_attention/functions.py:

    INF(B, H, W)                          functions.py:11-12  (kept for attribute parity; device-agnostic)
    CrissCrossAttention(in_dim)           functions.py:15-49  same ctor, same parameter / submodule names
                                          (query_conv, key_conv, value_conv, gamma, softmax, INF), same
                                          forward(x) -> tensor of x's shape

plus the autograd surface the reference's CUDA-extension branches expose and BASELINE.json's
north_star names (``ca_forward/ca_backward`` behind ``CA_Weight``, ``ca_map_forward/ca_map_backward``
behind ``CA_Map``) and the fused ``CrissCrossFunction`` the module uses.  The three 1x1 convolutions
(functions.py:29,32,35) stay torch ops (dense GEMMs -> MIOpen/hipBLASLt); everything between them and
the output -- the layout shuffles, both bmm pairs, INF, cat, softmax and the gamma/residual epilogue
(functions.py:30-49) -- runs in the hand-written HIP kernels of ``csrc/`` through the C ABI of
``include/ccnet_cca.h``.

There is no CPU / pure-PyTorch fallback: CPU tensors, non-fp32 tensors or a missing
``libccnet_cca.so`` raise.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable
from torch.nn import Softmax

from . import _lib
from ._lib import CCNET_CA_ENERGY, CCNET_CA_SOFTMAX

__all__ = ["INF", "CA_Weight", "CA_Map", "CrissCrossFunction", "ca_weight", "ca_map", "ca_softmax",
           "criss_cross_attention", "CrissCrossAttention"]


def INF(B, H, W, device=None):
    """Same values as the reference's ``INF`` (functions.py:11-12).  Like the reference it returns a tensor on the
    current HIP device (``.cuda()``) when one exists -- on a host without a GPU it stays on the CPU instead of
    raising; ``device=`` overrides.

    The kernels never materialise this (B*W, H, H) tensor -- the mask is the predicate ``j == h`` --
    it exists so that code poking at ``module.INF`` keeps working.
    """
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return -torch.diag(torch.tensor(float("inf"), device=device).repeat(H), 0).unsqueeze(0).repeat(B * W, 1, 1)


# ----------------------------------------------------------------------------------------------
# argument plumbing
# ----------------------------------------------------------------------------------------------
def _dev_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name}: tensor is on '{t.device}'. ccnet_amd's criss-cross attention runs only as HIP kernels on "
            "an AMD GPU (gfx950); there is no CPU fallback.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(nbytes, device):
    """(tensor or None, pointer or None, nbytes): scratch for the library's K-split slabs / reduction partials."""
    if nbytes <= 0:
        return None, None, 0
    t = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
    return t, t.data_ptr(), nbytes


def _check_qk(q, k):
    if q.dim() != 4 or q.shape != k.shape:
        raise RuntimeError(f"query/key must both be (B, C/8, H, W); got {tuple(q.shape)} and {tuple(k.shape)}")


def _check_attn(A, v):
    B, C, H, W = v.shape
    if A.dim() != 4 or tuple(A.shape) != (B, H, W, H + W):
        raise RuntimeError(f"attention must be (B, H, W, H+W) = {(B, H, W, H + W)}; got {tuple(A.shape)}")


# ----------------------------------------------------------------------------------------------
# CA_Weight: affinity  (ca_forward / ca_backward)
# ----------------------------------------------------------------------------------------------
class CA_Weight(torch.autograd.Function):
    """energy = CA_Weight.apply(query, key): (B,C/8,H,W) x2 -> (B,H,W,H+W), column self slot = -inf.

    Replaces functions.py:30-34,38-40 up to (not including) the softmax.
    """

    @staticmethod
    def forward(ctx, t, f):
        q, k = _dev_f32("query", t), _dev_f32("key", f)
        _check_qk(q, k)
        _same_device(q, k)
        B, Cq, H, W = q.shape
        lib = _lib.get_lib()
        out = torch.empty((B, H, W, H + W), device=q.device, dtype=torch.float32)
        with torch.cuda.device(q.device):
            lib.check(lib.ccnet_ca_forward_f32(q.data_ptr(), k.data_ptr(), out.data_ptr(), B, Cq, H, W,
                                               CCNET_CA_ENERGY, _stream()), "ca_forward")
        ctx.save_for_backward(q, k)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dw):
        q, k = ctx.saved_tensors
        dw = _dev_f32("grad_energy", dw)
        B, Cq, H, W = q.shape
        lib = _lib.get_lib()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        with torch.cuda.device(q.device):
            lib.check(lib.ccnet_ca_backward_f32(dw.data_ptr(), q.data_ptr(), k.data_ptr(), dq.data_ptr(),
                                                dk.data_ptr(), B, Cq, H, W, _stream()), "ca_backward")
        return dq, dk


# ----------------------------------------------------------------------------------------------
# CA_Map: aggregation  (ca_map_forward / ca_map_backward)
# ----------------------------------------------------------------------------------------------
class CA_Map(torch.autograd.Function):
    """out = CA_Map.apply(attention, value): (B,H,W,H+W), (B,C,H,W) -> (B,C,H,W) = out_H + out_W.

    Replaces functions.py:36-37,42,45-47.
    """

    @staticmethod
    def forward(ctx, weight, g):
        A, v = _dev_f32("attention", weight), _dev_f32("value", g)
        _check_attn(A, v)
        _same_device(A, v)
        B, C, H, W = v.shape
        lib = _lib.get_lib()
        out = torch.empty_like(v)
        with torch.cuda.device(v.device):
            lib.check(lib.ccnet_ca_map_forward_f32(A.data_ptr(), v.data_ptr(), None, None, out.data_ptr(),
                                                   B, C, H, W, _stream()), "ca_map_forward")
        ctx.save_for_backward(A, v)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        A, v = ctx.saved_tensors
        dout = _dev_f32("grad_out", dout)
        B, C, H, W = v.shape
        lib = _lib.get_lib()
        dA, dv = torch.empty_like(A), torch.empty_like(v)
        with torch.cuda.device(v.device):
            lib.check(lib.ccnet_ca_map_backward_f32(dout.data_ptr(), A.data_ptr(), v.data_ptr(), None,
                                                    dA.data_ptr(), dv.data_ptr(), B, C, H, W, _stream()),
                      "ca_map_backward")
        return dA, dv


class _CA_Softmax(torch.autograd.Function):
    """softmax over the H+W slots (functions.py:40) on the HIP kernels, differentiable."""

    @staticmethod
    def forward(ctx, energy):
        e = _dev_f32("energy", energy)
        B, H, W, S = e.shape
        if S != H + W:
            raise RuntimeError(f"energy must be (B, H, W, H+W); got {tuple(e.shape)}")
        lib = _lib.get_lib()
        A = torch.empty_like(e)
        with torch.cuda.device(e.device):
            lib.check(lib.ccnet_ca_softmax_forward_f32(e.data_ptr(), A.data_ptr(), B, H, W, _stream()), "ca_softmax")
        ctx.save_for_backward(A)
        return A

    @staticmethod
    @once_differentiable
    def backward(ctx, dA):
        (A,) = ctx.saved_tensors
        dA = _dev_f32("grad_attention", dA)
        B, H, W, _ = A.shape
        lib = _lib.get_lib()
        dE = torch.empty_like(A)
        with torch.cuda.device(A.device):
            lib.check(lib.ccnet_ca_softmax_backward_f32(A.data_ptr(), dA.data_ptr(), None, dE.data_ptr(), None,
                                                        None, 0, B, H, W, _stream()), "ca_softmax_backward")
        return dE


ca_weight = CA_Weight.apply
ca_map = CA_Map.apply
ca_softmax = _CA_Softmax.apply


# ----------------------------------------------------------------------------------------------
# fused core used by the module
# ----------------------------------------------------------------------------------------------
class CrissCrossFunction(torch.autograd.Function):
    """y = gamma * (out_H + out_W) + x from (q, k, v, x, gamma)  -- functions.py:38-49 in two launches
    families forward (affinity+softmax, aggregation+epilogue) and three backward."""

    @staticmethod
    def forward(ctx, q, k, v, x, gamma, recompute=False):
        q, k = _dev_f32("query", q), _dev_f32("key", k)
        v, x = _dev_f32("value", v), _dev_f32("x", x)
        gamma = _dev_f32("gamma", gamma)
        _check_qk(q, k)
        _same_device(q, k, v, x, gamma)
        B, C, H, W = v.shape
        if x.shape != v.shape or q.shape[0] != B or tuple(q.shape[2:]) != (H, W):
            raise RuntimeError(f"shape mismatch: q {tuple(q.shape)}, v {tuple(v.shape)}, x {tuple(x.shape)}")
        if gamma.numel() != 1:
            raise RuntimeError("gamma must hold exactly one element")
        lib = _lib.get_lib()
        y = torch.empty_like(x)
        A = torch.empty((B, H, W, H + W), device=x.device, dtype=torch.float32)
        cq = q.shape[1]
        with torch.cuda.device(x.device):
            _ws, wsp, wsn = _workspace(lib.ccnet_cca_forward_workspace_bytes(B, C, cq, H, W), x.device)
            lib.check(lib.ccnet_cca_forward_ws_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), x.data_ptr(),
                                                   gamma.data_ptr(), y.data_ptr(), A.data_ptr(),
                                                   B, C, cq, H, W, cq * H * W, cq * H * W, C * H * W, wsp, wsn,
                                                   _stream()), "cca_forward")
        ctx.recompute = bool(recompute)
        if ctx.recompute:
            ctx.save_for_backward(q, k, v, gamma)              # A is rebuilt from q, k in backward
        else:
            ctx.save_for_backward(q, k, v, A, gamma)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        if ctx.recompute:
            q, k, v, gamma = ctx.saved_tensors
            A = _recompute_attention(q, k, q.shape[1] * q.shape[2] * q.shape[3], q.shape[1] * q.shape[2] * q.shape[3])
        else:
            q, k, v, A, gamma = ctx.saved_tensors
        dy = _dev_f32("grad_output", dy)
        B, C, H, W = v.shape
        lib = _lib.get_lib()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dgamma = torch.empty_like(gamma)
        scratch = torch.empty_like(A)
        with torch.cuda.device(v.device):
            ws, wsp, nbytes = _workspace(lib.ccnet_cca_backward_workspace_bytes(B, C, q.shape[1], H, W), v.device)
            lib.check(lib.ccnet_cca_backward_f32(dy.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                 A.data_ptr(), gamma.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                                 dv.data_ptr(), dgamma.data_ptr(), scratch.data_ptr(),
                                                 wsp, nbytes, B, C, q.shape[1], H, W, _stream()),
                      "cca_backward")
        return dq, dk, dv, dy, dgamma.view_as(gamma), None


def _recompute_attention(q, k, q_bs, k_bs):
    """A = softmax(affinity(q, k)) again, for backward passes that did not keep it (q, k: (B,Cq,H,W) views whose
    batch strides are q_bs / k_bs elements)."""
    B, Cq, H, W = q.shape
    lib = _lib.get_lib()
    A = torch.empty((B, H, W, H + W), device=q.device, dtype=torch.float32)
    with torch.cuda.device(q.device):
        lib.check(lib.ccnet_cca_attention_strided_f32(q.data_ptr(), k.data_ptr(), A.data_ptr(), B, Cq, H, W,
                                                      q_bs, k_bs, _stream()), "cca_attention")
    return A


_PM_DTYPES = {torch.bfloat16: (2, 8, 132, "bf16"), torch.float32: (4, 4, 100, "f32")}   # bytes, alignment (elements), max strip


def _pm_view(name, t, dtype=None):
    """(B, H, W, C) bf16 / fp32 view whose channel axis is contiguous and whose rows follow each other at the pixel
    stride: returns (tensor, batch stride, pixel stride) in elements; copies only when the view does not qualify."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on an AMD GPU (there is no CPU fallback)")
    if t.dtype not in _PM_DTYPES or (dtype is not None and t.dtype != dtype) or t.dim() != 4:
        raise RuntimeError(f"{name}: expected a 4-D (B, H, W, C) {dtype or 'bfloat16 / float32'} tensor, got {t.dtype} "
                           f"{tuple(t.shape)}")
    B, H, W, c = t.shape
    es, al = _PM_DTYPES[t.dtype][:2]
    ok = (t.stride(3) == 1 and t.stride(1) == W * t.stride(2) and t.stride(2) % al == 0 and t.stride(0) % al == 0
          and t.stride(2) >= c and t.stride(0) >= H * W * t.stride(2) - (t.stride(2) - c) and t.data_ptr() % 16 == 0)
    if not ok:
        t = t.contiguous()
    return t, t.stride(0), t.stride(2)


def pm_covers(dtype, B, C, Cq, H, W):
    """geometry of the pixel-major kernels (csrc/cca_gmap.hpp) for this element type"""
    if dtype not in _PM_DTYPES:
        return False
    _, al, longest, _ = _PM_DTYPES[dtype]
    return max(H, W) <= longest and C % al == 0 and Cq % al == 0 and H * W * (C + 2 * Cq) < 2 ** 29


def pm_bf16_covers(B, C, Cq, H, W):
    return pm_covers(torch.bfloat16, B, C, Cq, H, W)


class CrissCrossPMFunction(torch.autograd.Function):
    """Fused core on PIXEL-MAJOR features (csrc/cca_gmap.hpp; bf16 = BASELINE configs[4], fp32 = the small-batch path):
    ``qkv`` is the packed (B, H, W, 2*Cq + C) projection (query | key | value channel slices, functions.py:29-35
    computed as one ``x^T W^T`` GEMM), ``x`` the (B, H, W, C) residual input of the same dtype; returns y (B, H, W, C).
    Attention, softmax, accumulation and gamma are fp32.  Backward returns the packed dqkv, so the projection's backward
    is again one GEMM."""

    @staticmethod
    def forward(ctx, qkv, x, gamma, cq, recompute=False):
        qkv, q_bs, q_ps = _pm_view("qkv", qkv)
        x, x_bs, x_ps = _pm_view("x", x, qkv.dtype)
        gamma = _dev_f32("gamma", gamma)
        _same_device(qkv, x, gamma)
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        es, _, longest, tag = _PM_DTYPES[qkv.dtype]
        if tuple(x.shape) != (B, H, W, C):
            raise RuntimeError(f"shape mismatch: qkv {tuple(qkv.shape)} (Cq = {cq}), x {tuple(x.shape)}")
        if not pm_covers(qkv.dtype, B, C, cq, H, W):
            raise RuntimeError(f"pixel-major {tag} kernels cover strips <= {longest} and channel counts divisible by "
                               f"{_PM_DTYPES[qkv.dtype][1]}; got C = {C}, Cq = {cq}, H = {H}, W = {W}")
        lib = _lib.get_lib()
        y = torch.empty((B, H, W, C), device=x.device, dtype=qkv.dtype)
        A = torch.empty((B, H, W, H + W), device=x.device, dtype=torch.float32)
        _ws, ws_ptr, nbytes = _workspace(lib.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 0), x.device)
        p = qkv.data_ptr()
        fwd = getattr(lib, "ccnet_cca_forward_pm_" + tag)
        with torch.cuda.device(x.device):
            lib.check(fwd(p, p + es * cq, p + 2 * es * cq, x.data_ptr(), gamma.data_ptr(), y.data_ptr(), A.data_ptr(),
                          B, C, cq, H, W, q_bs, q_ps, q_bs, q_ps, q_bs, q_ps, x_bs, x_ps, H * W * C, C,
                          ws_ptr, nbytes, _stream()), "cca_forward_pm_" + tag)
        ctx.recompute = bool(recompute)
        ctx.save_for_backward(*((qkv, gamma) if ctx.recompute else (qkv, A, gamma)))
        ctx.cq = cq
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        cq = ctx.cq
        lib = _lib.get_lib()
        if ctx.recompute:
            # the attention rebuilt by the forward's own affinity + softmax kernels: bit-identical to what a saving forward kept
            qkv, gamma = ctx.saved_tensors
            B, H, W, ct = qkv.shape
            es, tag = _PM_DTYPES[qkv.dtype][0], _PM_DTYPES[qkv.dtype][3]
            p = qkv.data_ptr()
            A = _attention_pm(lib, p, p + es * cq, tag == "bf16", B, cq, H, W, qkv.stride(0), qkv.stride(2), qkv.device)
        else:
            qkv, A, gamma = ctx.saved_tensors
        dy, dy_bs, dy_ps = _pm_view("grad_output", dy, qkv.dtype)
        B, H, W, ct = qkv.shape
        C = ct - 2 * cq
        es, _, _, tag = _PM_DTYPES[qkv.dtype]
        dqkv = torch.empty((B, H, W, ct), device=qkv.device, dtype=qkv.dtype)
        dgamma = torch.empty_like(gamma)
        scratch = torch.empty_like(A)
        _ws, ws_ptr, nbytes = _workspace(lib.ccnet_cca_pm_workspace_bytes(B, C, cq, H, W, 1), qkv.device)
        p, g = qkv.data_ptr(), dqkv.data_ptr()
        bs, ps = qkv.stride(0), qkv.stride(2)
        bwd = getattr(lib, "ccnet_cca_backward_pm_" + tag)
        with torch.cuda.device(qkv.device):
            lib.check(bwd(dy.data_ptr(), p, p + es * cq, p + 2 * es * cq, A.data_ptr(), gamma.data_ptr(),
                          g, g + es * cq, g + 2 * es * cq, dgamma.data_ptr(), scratch.data_ptr(), B, C, cq, H, W,
                          dy_bs, dy_ps, bs, ps, bs, ps, bs, ps, H * W * ct, ct, H * W * ct, ct, H * W * ct, ct,
                          ws_ptr, nbytes, _stream()), "cca_backward_pm_" + tag)
        return dqkv, dy, dgamma.view_as(gamma), None, None


CrissCrossPMBF16Function = CrissCrossPMFunction          # (the name round-2 code and tests imported first)


PLANES_HL, PLANES_HLH, PLANES_HHL = 2, 3, 4          # include/ccnet_cca.h CCNET_PLANES_*


def self_planes3_ok(lib):
    """the three-plane backward runs the library's default launch forms only (it refuses while an A/B option is set)"""
    return (lib.get_option("dqdk_wpc3") == 1 and lib.get_option("dqdk_exact") == 1 and lib.get_option("planes_ring") == 2
            and lib.get_option("planes_stream") >= 1)


def _attention_pm(lib, qptr, kptr, bf16, B, cq, H, W, bs, ps, device):
    """A rebuilt from pixel-major q, k views (recompute instead of save): ccnet_cca_attention_pm"""
    A = torch.empty((B, H, W, H + W), device=device, dtype=torch.float32)
    with torch.cuda.device(device):
        lib.check(lib.ccnet_cca_attention_pm(qptr, kptr, A.data_ptr(), int(bf16), B, cq, H, W, bs, ps, bs, ps, _stream()),
                  "cca_attention_pm")
    return A


def split_planes(t: torch.Tensor, c0: int, C: int, layout: int = PLANES_HL, dtype=torch.int16, bias=None) -> torch.Tensor:
    """Channels [c0, c0 + C) of the fp32 pixel-major tensor ``t`` (B, H, W, ps) as SPLIT PLANES (B, H, W, n, C):
    bf16 hi | lo halves of every value (include/ccnet_cca.h, "split-plane path"), produced once for all their consumers
    (``layout`` HLH / HHL: the three-plane rows of a K-concatenated split-bf16 GEMM; ``dtype`` bfloat16 for those)."""
    B, H, W, ps = t.shape
    n = 2 if layout == PLANES_HL else 3
    out = torch.empty((B, H, W, n, C), device=t.device, dtype=dtype)
    lib = _lib.get_lib()
    with torch.cuda.device(t.device):
        lib.check(lib.ccnet_cca_split_planes_f32(t.data_ptr() + 4 * c0, out.data_ptr(), B, C, H, W, t.stride(0), t.stride(2),
                                                 H * W * n * C, n * C, layout, None if bias is None else bias.data_ptr(),
                                                 _stream()), "split_planes")
    return out


def split_planes_colsum(t: torch.Tensor, layout: int = PLANES_HLH, dtype=torch.bfloat16):
    """``split_planes(t, 0, ps, layout)`` of the WHOLE fp32 pixel-major tensor ``t`` (B, H, W, ps) and, from the same pass, the sum of
    ``t`` over all images and pixels (ps floats, fixed summation order): the module's backward needs dqkv both ways (three-plane
    rows for its GEMMs, column sums as the bias gradients) -- ccnet_cca_split_planes_colsum_f32."""
    B, H, W, ps = t.shape
    n = 2 if layout == PLANES_HL else 3
    out = torch.empty((B, H, W, n, ps), device=t.device, dtype=dtype)
    colsum = torch.empty((ps,), device=t.device, dtype=torch.float32)
    lib = _lib.get_lib()
    with torch.cuda.device(t.device):
        _ws, wsp, wsn = _workspace(lib.ccnet_cca_workspace_bytes(_lib.CCNET_WS_SPLIT_COLSUM, B, ps, 0, H, W), t.device)
        lib.check(lib.ccnet_cca_split_planes_colsum_f32(t.data_ptr(), out.data_ptr(), colsum.data_ptr(), wsp, wsn, B, ps, H, W,
                                                        t.stride(0), t.stride(2), H * W * n * ps, n * ps, layout, _stream()),
                  "split_planes_colsum")
    return out, colsum


def nchw_to_planes(x: torch.Tensor, layout: int = PLANES_HL, dtype=torch.int16) -> torch.Tensor:
    """fp32 NCHW (B, C, H, W) -> planes (B, H, W, n, C) (transposed and split in one pass, csrc/cca_gmap.hpp)."""
    B, C, H, W = x.shape
    n = 2 if layout == PLANES_HL else 3
    out = torch.empty((B, H, W, n, C), device=x.device, dtype=dtype)
    lib = _lib.get_lib()
    with torch.cuda.device(x.device):
        lib.check(lib.ccnet_cca_nchw_to_planes_f32(x.data_ptr(), out.data_ptr(), B, C, H, W, C * H * W, H * W * n * C, n * C,
                                                   layout, _stream()), "nchw_to_planes")
    return out


def _split_weight(w: torch.Tensor):
    """fp32 (N, K) -> bf16 halves (hi, lo) with w = hi + lo + O(2^-17 |w|)"""
    hi = w.to(torch.bfloat16)
    return hi, (w - hi.float()).to(torch.bfloat16)


def planes_cover(B, C, Cq, H, W):
    """geometry of the split-plane kernels (csrc/cca_gmap.hpp, bf16p_t): strips up to 132 positions, and -- as long as
    C/8 <= 64 -- rows and / or columns of up to 4 x 132 positions, which run as blocks (the 129 x 257 feature map of the
    reference's whole-image evaluation, evaluate.py:102-143, 246; both sides beyond 132 at the larger scales of its multi-scale
    variant, evaluate.py:146-166)."""
    if C % 8 or Cq % 4 or H * W * (C + 2 * Cq) >= 2 ** 29 or H * W * (H + W) >= 2 ** 29:
        return False
    if max(H, W) <= 132:
        return True
    return max(H, W) <= 4 * 132 and Cq <= 64


def _pack_projection(wq, bq, wk, bk, wv, bv, split):
    """Stacked projection operands of one module application, packed by ONE launch (``ccnet_cca_pack_projection_f32``): the
    stacked fp32 weight / bias and, with ``split``, the K-concatenated bf16 hi | lo operands of the split-bf16 x3 GEMMs.

    Rounds 3-4 cached these per module, keyed on the parameters' ``data_ptr`` and ``_version``: ``p.data.add_()`` /
    ``p.data.copy_()`` (EMA swaps, weight clipping, fused multi-tensor optimizers) change the values without bumping
    ``_version``, and the node then ran forward AND backward on stale weights.  There is no cache any more: every
    forward packs the CURRENT parameter values (one 3 us launch over 1.3 MB -- less than the key comparison's six ``data_ptr``
    / ``_version`` reads cost in Python) and hands the result to its own backward through ``ctx``."""
    cq, C = wq.shape[0], wq.shape[1]
    n = 2 * cq + C
    dev = wq.device
    ps = [t.detach() if t.is_contiguous() else t.detach().contiguous() for t in (wq, bq, wk, bk, wv, bv)]
    if any(t.dtype != torch.float32 or t.device != dev for t in ps):
        raise RuntimeError("the fused projection needs fp32 parameters on one device")
    w = torch.empty((n, C), device=dev, dtype=torch.float32)
    b = torch.empty((n,), device=dev, dtype=torch.float32)
    w3 = torch.empty((n, 3 * C), device=dev, dtype=torch.bfloat16) if split else None
    w3t = torch.empty((C, 3 * n), device=dev, dtype=torch.bfloat16) if split else None
    lib = _lib.get_lib()
    with torch.cuda.device(dev):
        lib.check(lib.ccnet_cca_pack_projection_f32(*(t.data_ptr() for t in ps), w.data_ptr(), b.data_ptr(),
                                                    None if w3 is None else w3.data_ptr(), None if w3t is None else w3t.data_ptr(),
                                                    C, cq, _stream()), "pack_projection")
    val = {"w": w, "b": b, "bqk": b[:2 * cq], "bv": b[2 * cq:]}
    if split:
        val["w3"] = w3.t()                                                                      # (3C, 2Cq + C) view
        val["w3t"] = w3t                                                                        # (C, 3 (2Cq + C))
    return val


def _projection_gemm(lib, a, wt, bias):
    """``a @ wt.T + bias`` on bf16 operands, fp32 accumulation and output, by ``ccnet_cca_projection_bf16`` (functions.py:29,32,35 of
    the reference as one stacked GEMM).  ``a``: (M, K) and ``wt``: (N, K), both K-contiguous.  Row counts beyond the entry point's
    31-bit byte offsets run as several launches over row ranges.  None when the shape is outside the contract (K % 8, N % 4) --
    the caller then uses the stock GEMM."""
    M, K = a.shape
    N = wt.shape[0]
    if K % 8 or N % 4 or a.stride(1) != 1 or wt.stride(1) != 1 or a.stride(0) % 8 or wt.stride(0) % 8 or N * wt.stride(0) >= 1 << 30:
        return None
    rows = min(((1 << 30) - 1) // a.stride(0), ((1 << 29) - 1) // N) // 256 * 256          # per launch: whole 256-row tiles
    if rows <= 0:
        return None
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        for m0 in range(0, M, rows):
            m = min(rows, M - m0)
            lib.check(lib.ccnet_cca_projection_bf16(a.data_ptr() + 2 * m0 * a.stride(0), wt.data_ptr(), bias.data_ptr(), out.data_ptr() + 4 * m0 * N,
                                                    m, N, K, a.stride(0), wt.stride(0), N, _stream()), "projection_bf16")
    return out


def _projection_adjoint_gemm(lib, w, d, add):
    """``add + w @ d[b].T`` per image by ``ccnet_cca_projection_adjoint_bf16``: ``w`` (C, K) and ``d`` (B, P, K) bf16, K-contiguous,
    ``add`` (B, C, P) fp32 contiguous -> (B, C, P) fp32.  None when the shape is outside the entry point's contract."""
    C, K = w.shape
    B, P = d.shape[0], d.shape[1]
    if K % 8 or w.stride(1) != 1 or d.stride(2) != 1 or w.stride(0) % 8 or d.stride(1) % 8 or d.stride(0) % 8 or not add.is_contiguous():
        return None
    if C * w.stride(0) >= 1 << 30 or P * d.stride(1) >= 1 << 30 or C * P >= 1 << 29:
        return None
    out = torch.empty((B, C, P), device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        lib.check(lib.ccnet_cca_projection_adjoint_bf16(w.data_ptr(), d.data_ptr(), add.data_ptr(), out.data_ptr(), B, C, P, K,
                                                        w.stride(0), d.stride(1), d.stride(0), _stream()), "projection_adjoint_bf16")
    return out


def _projection_wgrad_gemm(lib, d, x):
    """``d.T @ x`` over all rows by ``ccnet_cca_projection_wgrad_bf16``: ``d`` (R, N) and ``x`` (R, C) bf16 with contiguous rows ->
    (N, C) fp32 (the partial sums of the launch -- of the launches, where the rows exceed the entry point's 31-bit byte offsets --
    added in a fixed order).  None when the shape is outside the entry point's contract."""
    R, N = d.shape
    C = x.shape[1]
    if N % 8 or C % 8 or d.stride(1) != 1 or x.stride(1) != 1 or d.stride(0) % 8 or x.stride(0) % 8 or x.shape[0] != R:
        return None
    rows = ((1 << 30) - 1) // max(d.stride(0), x.stride(0)) // 64 * 64                     # per launch: whole 64-row stages
    if rows <= 0:
        return None
    tiles = -(-N // 128) * -(-C // 256)
    cus = torch.cuda.get_device_properties(d.device).multi_processor_count
    chunks = [(r0, min(rows, R - r0)) for r0 in range(0, R, rows)]
    slabs = [max(1, min(cus // tiles if tiles <= cus else 1, -(-r // 64))) for _, r in chunks]
    part = torch.empty((sum(slabs), N, C), device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        s0 = 0
        for (r0, r), S in zip(chunks, slabs):
            lib.check(lib.ccnet_cca_projection_wgrad_bf16(d.data_ptr() + 2 * r0 * d.stride(0), x.data_ptr() + 2 * r0 * x.stride(0),
                                                          part.data_ptr() + 4 * s0 * N * C, r, N, C, d.stride(0), x.stride(0), S, _stream()),
                      "projection_wgrad_bf16")
            s0 += S
    return part.sum(0) if part.shape[0] > 1 else part[0]


class CrissCrossPlanesModuleFunction(torch.autograd.Function):
    """The whole module as ONE autograd node on the SPLIT-PLANE path (fp32, no autocast), the module's own tensors NCHW:
    the stacked projection is the GEMM ``x^T W^T`` whose (B, HW, 2Cq + C) output holds the pixel-major q | k | v; the core's
    forward entry point takes the fp32 value slice as it is: maps with strips <= 100 (the headline geometry) run the PLANE-FREE
    form -- the aggregation (functions.py:42-47) and the dA contraction of its adjoint read v as fp32 tiles and split every
    fragment into bf16 hi | lo in registers; larger maps have the entry point split v into planes first (its first launch; the
    projection's value bias is added there).  dy is transposed out of NCHW into planes once inside the backward; q, k stay fp32
    (exact energies); three exact bf16 products per term everywhere else.  ``dx = dy + W^T dqkv^T`` is one GEMM with beta = 1
    writing NCHW.

    ``split_gemm``: the three projection GEMMs (functions.py:29-35 and their adjoints) run split-bf16 x3 as well -- ONE stock
    bf16 -> fp32 GEMM each on K-concatenated three-plane operands (x.w ~ xh.wh + xh.wl + xl.wh: rows [xh | xh | xl] of x
    against [wh | wl | wh] of the stacked weight, K = 3C; the adjoints pair [dh | dl | dh] of dqkv with [wh | wh | wl] and,
    row by row over 3 B HW rows, with x's planes for the weight gradient).  The planes are written by the library's
    producers (one pass over x, one over dqkv); fp32 accumulation, relative error ~1e-5 (the lo x lo term is dropped).

    Kept for the backward: the packed projection (plane-free form), or a copy of its q | k fifth + v as planes (the projection
    is released); x (or, with ``split_gemm``, its three planes instead); and the attention tensor -- or,
    with ``recompute``, nothing of the attention: the backward rebuilds it from q | k with the forward's own kernels
    (bit-identical, one affinity + softmax launch pair)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma, split_gemm=False, recompute=False):
        x, gamma = _dev_f32("x", x), _dev_f32("gamma", gamma)
        B, C, H, W = x.shape
        cq, hw = wq.shape[0], H * W
        ct = 2 * cq + C
        pc = _pack_projection(wq, bq, wk, bk, wv, bv, split_gemm)     # (the current parameter values, every call: no cache)
        x3 = None
        direct = max(H, W) <= 100           # strips <= 100: the plane-free form of the core (v stays fp32, no split pass)
        lib = _lib.get_lib()
        y = torch.empty_like(x)
        A = torch.empty((B, H, W, H + W), device=x.device, dtype=torch.float32)
        vpl = None if direct else torch.empty((B, H, W, 2, C), device=x.device, dtype=torch.int16)
        bs, ps = hw * ct, ct
        if split_gemm:
            # (round 5 also built this forward as TWO GEMMs -- q | k, then v -- with the affinity + softmax launches on a second stream
            #  next to the v GEMM and the aggregation half of the entry point after both, VERDICT r4 item 8: module forward 0.607-0.623
            #  against 0.618-0.621 ms in the same run, inside the spread; kill criterion (-40 us) missed, removed.
            #  profiles/r05g_module_fwd_ab.txt, commits 743958e..ecb3563.)
            x3 = nchw_to_planes(x, PLANES_HHL, torch.bfloat16)                              # (B, H, W, 3, C): xh | xh | xl
            # the library's own GEMM (csrc/cca_gemm.hpp, round 6): bias in the accumulators, 199-206 us at (8,512,97,97) against
            # 260-264 us for the stock product + an in-place bias pass and 285-318 us for the stock bias epilogue (torch.addmm, rounds
            # 3-5) -- profiles/r06n_fwd_gemm_ab.txt, r06h_fwd_gemm_bias_ab.txt.  Shapes it does not take fall back to the stock pair.
            qkv = _projection_gemm(lib, x3.view(B * hw, 3 * C), pc["w3"].t(), pc["b"])
            if qkv is None:
                qkv = torch.mm(x3.view(B * hw, 3 * C), pc["w3"], out_dtype=torch.float32).add_(pc["b"])
            qkv = qkv.view(B, hw, ct)
            v_bias = None
        else:
            qkv = torch.baddbmm(pc["b"].view(1, 1, -1), x.view(B, C, hw).transpose(1, 2), pc["w"].t().unsqueeze(0).expand(B, -1, -1))
            v_bias = None
        p = qkv.data_ptr()
        with torch.cuda.device(x.device):
            _ws, wsp, wsn = _workspace(lib.ccnet_cca_planes_workspace_bytes(B, C, cq, H, W, 0), x.device)
            lib.check(lib.ccnet_cca_forward_planes_f32(p, p + 4 * cq, p + 8 * cq, None if v_bias is None else v_bias.data_ptr(),
                                                       None if direct else vpl.data_ptr(), x.data_ptr(), gamma.data_ptr(), y.data_ptr(),
                                                       A.data_ptr(), B, C, cq, H, W, bs, ps, bs, ps, bs, ps,
                                                       hw * 2 * C, 2 * C, wsp, wsn, _stream()), "cca_forward_planes")
        if not any(ctx.needs_input_grad):
            return y
        # plane-free: the packed projection itself is kept (its value slice is read again by the dA contraction); otherwise a copy
        # of the q | k fifth + the planes, and the projection is released
        qk = qkv if direct else qkv[..., :2 * cq].contiguous()
        ctx.recompute = bool(recompute)
        ctx.split_gemm = bool(split_gemm)
        ctx.direct = direct
        # (the packed weight the backward multiplies with -- the values this forward saw -- goes through save_for_backward like
        #  every other kept tensor: saved-tensor hooks / offload apply to it)
        keep = [x3 if split_gemm else x, qk, qk if direct else vpl, gamma, wq, bq, wk, bk, wv, bv, pc["w3t"] if split_gemm else pc["w"]]
        if not ctx.recompute:
            keep += [A]
        ctx.save_for_backward(*keep)
        ctx.cq, ctx.geom = cq, (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        cq = ctx.cq
        B, C, H, W = ctx.geom
        xs, qk, vpl, gamma, wq, bq, wk, bk, wv, bv, wpack = ctx.saved_tensors[:11]
        dy = _dev_f32("grad_output", dy)
        hw, ct = H * W, 2 * cq + C
        lib = _lib.get_lib()
        direct = ctx.direct
        p, bs, ps = (qk.data_ptr(), hw * ct, ct) if direct else (qk.data_ptr(), hw * 2 * cq, 2 * cq)
        if ctx.recompute:
            A = _attention_pm(lib, p, p + 4 * cq, False, B, cq, H, W, bs, ps, dy.device)
        else:
            A = ctx.saved_tensors[11]
        dgamma = torch.empty_like(gamma)
        scratch = torch.empty_like(A)
        if ctx.split_gemm and direct and cq <= 64 and self_planes3_ok(lib):
            # the core writes dq | dk | dv straight as the three-plane rows of the split-bf16 GEMMs below and the bias gradients as
            # their column sums (ccnet_cca_backward_planes3_f32): no fp32 dqkv, no pass that reads it back to split it (round 6)
            d3 = torch.empty((B, H, W, 3, ct), device=dy.device, dtype=torch.bfloat16)
            db = torch.empty((ct,), device=dy.device, dtype=torch.float32)
            with torch.cuda.device(dy.device):
                _ws, wsp, wsn = _workspace(lib.ccnet_cca_workspace_bytes(_lib.CCNET_WS_PLANES3_BACKWARD, B, C, cq, H, W), dy.device)
                lib.check(lib.ccnet_cca_backward_planes3_f32(dy.data_ptr(), p, p + 4 * cq, p + 8 * cq, A.data_ptr(), gamma.data_ptr(),
                                                             d3.data_ptr(), db.data_ptr(), dgamma.data_ptr(), scratch.data_ptr(),
                                                             B, C, cq, H, W, bs, ps, bs, ps, bs, ps, hw * 3 * ct, 3 * ct, wsp, wsn, _stream()),
                          "cca_backward_planes3")
            return CrissCrossPlanesModuleFunction._projection_adjoint(ctx, d3, db, dy, xs, wpack, gamma, dgamma, B, C, H, W, cq)
        dqkv = torch.empty((B, hw, ct), device=dy.device, dtype=torch.float32)
        g, gbs = dqkv.data_ptr(), hw * ct
        with torch.cuda.device(dy.device):
            _ws, wsp, wsn = _workspace(lib.ccnet_cca_planes_workspace_bytes(B, C, cq, H, W, 1), dy.device)
            lib.check(lib.ccnet_cca_backward_planes_f32(dy.data_ptr(), p, p + 4 * cq, p + 8 * cq if direct else None,
                                                        None if direct else vpl.data_ptr(), A.data_ptr(),
                                                        gamma.data_ptr(), g, g + 4 * cq, g + 8 * cq, dgamma.data_ptr(),
                                                        scratch.data_ptr(), B, C, cq, H, W, bs, ps, bs, ps, bs, ps, hw * 2 * C, 2 * C,
                                                        gbs, ct, gbs, ct, gbs, ct, wsp, wsn, _stream()), "cca_backward_planes")
        if ctx.split_gemm:
            # (B, H, W, 3, ct): dh | dl | dh for the two GEMMs, and the bias gradients (the sum of dqkv over all pixels) out of the same
            # pass over dqkv (round 5: split 72 us + torch sum 47 us -> one pass; maps the three-plane backward does not serve)
            d3, db = split_planes_colsum(dqkv.view(B, H, W, ct), PLANES_HLH, torch.bfloat16)
            return CrissCrossPlanesModuleFunction._projection_adjoint(ctx, d3, db, dy, xs, wpack, gamma, dgamma, B, C, H, W, cq)
        db = dqkv.sum(dim=(0, 1))
        xm = xs.view(B, C, hw)
        dqt = dqkv.transpose(1, 2)                                                        # (B, 2Cq + C, HW) view
        dx = torch.baddbmm(dy.view(B, C, hw), wpack.t().unsqueeze(0).expand(B, -1, -1), dqt)  # dy + W^T dqkv^T  (NCHW)
        dw = torch.bmm(dqt, xm.transpose(1, 2)).sum(0)                                    # (2Cq + C, C)
        dwq, dwk, dwv = dw[:cq], dw[cq:2 * cq], dw[2 * cq:]
        return (dx.view(B, C, H, W), dwq.reshape(cq, C, 1, 1), db[:cq], dwk.reshape(cq, C, 1, 1), db[cq:2 * cq],
                dwv.reshape(C, C, 1, 1), db[2 * cq:], dgamma.view_as(gamma), None, None)

    @staticmethod
    def _projection_adjoint(ctx, d3, db, dy, x3, wpack, gamma, dgamma, B, C, H, W, cq):
        """dx and the weight gradients from dqkv as three-plane rows ``d3`` (B, H, W, 3, ct) -- the two split-bf16 GEMMs."""
        hw, ct = H * W, 2 * cq + C
        # dy + W^T dqkv^T (NCHW): the library's GEMM, one launch over the batch, dy starting the accumulators (csrc/cca_gemm.hpp).
        # (The stock pair -- torch.bmm(..., out_dtype=fp32).add_(dy) -- ran at 265-293 us; with dy as the stock GEMM's C operand,
        # torch.baddbmm, beta = 1, VERDICT r5 item 5a, at 293-301 us: profiles/r06c_dx_gemm_dw_split_ab.txt.)
        dx = _projection_adjoint_gemm(_lib.get_lib(), wpack, d3.view(B, hw, 3 * ct), dy.view(B, C, hw))
        if dx is None:
            dx = torch.bmm(wpack.unsqueeze(0).expand(B, -1, -1), d3.view(B, hw, 3 * ct).transpose(1, 2),
                           out_dtype=torch.float32).add_(dy.view(B, C, hw))
        # rows (dh, xh), (dl, xh), (dh, xl) of every pixel: the three products, contracted over all 3 B HW rows by the library's
        # row-contraction GEMM (csrc/cca_gemm.hpp: S slabs of rows x ten output tiles ~ one workgroup per CU, S partials added here).
        # (The stock route: 3 B batch entries of K = HW rows summed by .sum(0), 264-268 us at (8,512,97,97); B entries of K = 3 HW:
        # 348-361 us -- profiles/r06c_dx_gemm_dw_split_ab.txt, r06o_dx_gemm_ab.txt.)
        dw = _projection_wgrad_gemm(_lib.get_lib(), d3.view(B * hw * 3, ct), x3.view(B * hw * 3, C))
        if dw is None:
            ks = 3 if B <= 12 else 1
            dw = torch.bmm(d3.view(B * ks, 3 * hw // ks, ct).transpose(1, 2), x3.view(B * ks, 3 * hw // ks, C),
                           out_dtype=torch.float32).sum(0)
        dwq, dwk, dwv = dw[:cq], dw[cq:2 * cq], dw[2 * cq:]
        return (dx.view(B, C, H, W), dwq.reshape(cq, C, 1, 1), db[:cq], dwk.reshape(cq, C, 1, 1), db[cq:2 * cq],
                dwv.reshape(C, C, 1, 1), db[2 * cq:], dgamma.view_as(gamma), None, None)


def criss_cross_attention(q, k, v, x, gamma, recompute_attention=False):
    """Functional form of the fused core (``recompute_attention``: rebuild A in backward instead of keeping it)."""
    return CrissCrossFunction.apply(q, k, v, x, gamma, recompute_attention)


# ----------------------------------------------------------------------------------------------
# the module (drop-in for networks/ccnet.py:13,105)
# ----------------------------------------------------------------------------------------------
class CrissCrossAttention(nn.Module):
    """Criss-Cross Attention Module -- same constructor, attributes, parameter names and forward
    contract as the reference (functions.py:15-49), so reference checkpoints load unchanged."""

    def __init__(self, in_dim):
        super(CrissCrossAttention, self).__init__()
        self.query_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.softmax = Softmax(dim=3)      # attribute parity only; the softmax runs inside the HIP path
        self.INF = INF
        self.gamma = nn.Parameter(torch.zeros(1))

    #: run query/key/value as ONE 1x1 convolution (stacked weights) feeding the kernels through channel-slice
    #: strides; set False (class or instance) for three separate convolutions exactly as functions.py:29-35.
    fuse_projections = True

    #: activation memory (networks/ccnet.py:118-119 applies the module R times): when True the
    #: (B,H,W,H+W) attention tensor is NOT kept for backward -- it is recomputed from q, k (one affinity + softmax
    #: launch pair, ~8 % of a fwd+bwd) -- so an application holds q, k, v only.  Every fp32 / bf16 node honours it (round 4: the
    #: split-plane and pixel-major nodes rebuild A with the forward's own affinity + softmax kernels: bit-identical gradients).
    #: Under torch.no_grad() / eval nothing is kept either way.
    recompute_attention = False

    #: fp32 NCHW inputs (no autocast; strips <= 528 positions -- see ``planes_cover``): the SPLIT-PLANE node
    #: (``CrissCrossPlanesModuleFunction``: q | k | v out of one GEMM pixel-major, v and dy pre-split into bf16 hi | lo planes,
    #: x / y / dy NCHW).  Measured on MI355X, core fwd+bwd at (8,512,97,97): 0.75-0.78 ms vs 0.88-0.90 ms on the NCHW strips of
    #: the same boxes (profiles/r03q_bench.json, r03z_bench.json); whole-image inference (1,512,129,257): 0.475 vs 1.18 ms.
    split_planes = True
    #: the split-plane node also runs its three projection GEMMs split-bf16 x3 (one bf16 -> fp32 GEMM each on three-plane
    #: operands, fp32 accumulate, ~1e-5 relative): fwd 496 -> 278, dx 488 -> 265, dW 508 -> 325 us at (8,512,97,97)
    #: (profiles/r03p_split_gemm_probe2.txt) for two extra producer passes.  False = fp32 GEMMs (torch's default fp32 path).
    split_bf16_projections = True
    #: ... from this many pixels per call on.  0 since the GEMMs are the library's own kernels (round 6: every shape measured gains --
    #: module fwd+bwd at (B,512,97,97), B = 1 / 2 / 3: 0.49-0.50 -> 0.44-0.45, 0.68 -> 0.47, 0.90 -> 0.62 ms; (1,512,129,129) 0.78 -> 0.56;
    #: maps of 1-8 k pixels, bound by host launches, 0.48-0.50 -> 0.43-0.44: fewer torch ops on the path --
    #: profiles/r06s_split_threshold_ab.txt).  Rounds 3-5, on stock GEMMs: 32768 (below ~30k pixels the extra producer passes cost
    #: more than the GEMMs saved).
    split_bf16_min_pixels = 0
    #: bf16 inputs: strips <= 132 (C, C/8 divisible by 8) run on the pixel-major bf16 MFMA kernels; strips of 133 .. 528 positions
    #: run the blocked fp32 plane kernels on fp32 copies (route ``f32-planes-cast``), anything else the strip family through fp32
    #: copies (route ``separate-strips``).
    native_bf16 = True

    #: route name -> what runs (``route(x)`` picks one; ``forward`` only dispatches on it)
    ROUTES = {
        "bf16-pixel-major": "one x^T W^T projection + pixel-major bf16 MFMA kernels (BASELINE configs[4])",
        "f32-planes": "one autograd node: projection GEMM, v / dy as bf16 hi | lo planes, NCHW x / y / dy (channels_last inputs: one copy)",
        "f32-planes-cast": "fp32 inputs under autocast, fp16 inputs, bf16 inputs beyond the bf16 kernels' 132 positions: the f32-planes node on fp32 copies, autocast off inside",
        "separate-strips": "three convolutions + NCHW strip / windowed / any-shape kernels (functions.py:29-35 as written): every other input",
    }
    # (Round 6 removed three routes nobody's default had used since round 3 -- the one-node form on the NCHW strip kernels, the
    #  stacked-conv2d form on the same kernels, the fp32 pixel-major family for channels_last inputs (1.71 ms on the split-plane
    #  node + one 70 us copy against 2.17 ms) -- and with them two autograd Functions and 40 GPU tests: VERDICT r5 item 7.)

    def route(self, x):
        """Which implementation ``forward`` runs for this input (a key of ``ROUTES``).  The pixel-major and split-plane
        routes always compute split-bf16 x3 (exact fp32 energies): they are skipped while the process-wide knobs pin exact
        fp32 arithmetic (option ``"precision"`` = CCNET_PRECISION_F32) or the any-shape kernels (``"impl"`` = CCNET_IMPL_DIRECT)
        and while ``fuse_projections`` is off.  ``recompute_attention`` is honoured by every route."""
        B, C, H, W = x.shape
        cq = self.query_conv.out_channels
        lib = _lib.get_lib()
        knobs_ok = (lib.ccnet_cca_get_precision() != _lib.CCNET_PRECISION_F32 and lib.ccnet_cca_get_impl() != _lib.CCNET_IMPL_DIRECT)
        fast_ok = knobs_ok and self.fuse_projections
        if x.dtype == torch.bfloat16 and self.native_bf16:
            if (fast_ok and (self._fusable(x) or (torch.is_autocast_enabled() and self._fusable()))   # (autocast casts W for linear)
                    and pm_bf16_covers(B, C, cq, H, W)):
                return "bf16-pixel-major"
        if x.dtype == torch.float32 and not torch.is_autocast_enabled() and self._fusable(x):
            # (strips beyond 132 positions -- rows, columns or both, up to 528 -- run in blocks.  Round 3 ran a TALL map whose width
            # fits 132 on its spatial transpose, two transposing copies each way; the blocked column passes are faster:
            # (1,512,257,129) inference 0.63 -> 0.47 ms, fwd+bwd 1.85 -> 1.52 ms, profiles/r04lc_tall_map_ab.txt -- that route is gone.)
            if fast_ok and self.split_planes and planes_cover(B, C, cq, H, W):
                return "f32-planes"
        # half-precision activations (or fp32 under autocast) on a map beyond the bf16 kernels' 132 positions -- mixed-precision
        # whole-image evaluation, evaluate.py:102-166 -- used to fall to the windowed / any-shape strip kernels through fp32 copies;
        # the blocked plane kernels take such maps (strips <= 528): the fp32 node on fp32 copies of x and of the parameters
        # ... and an fp32 input under autocast (what RCCAModule hands this module in an autocast training run: InPlaceABNSync returns
        # fp32) or an fp16 one (no native kernels) takes the same node at EVERY covered size: against stacked conv2d under autocast +
        # the strip kernels, module fwd+bwd at (B,512,97,97) 0.54 -> 0.41 ms (B = 1), 1.83 -> 1.75 ms (B = 8), with fp32-accurate
        # projections (profiles/r04lv_autocast_route_probe.txt)
        half = x.dtype in (torch.bfloat16, torch.float16) or (x.dtype == torch.float32 and torch.is_autocast_enabled())
        not_native = x.dtype != torch.bfloat16 or max(H, W) > 132 or not self.native_bf16
        if (half and not_native and fast_ok and self.split_planes and self._fusable()
                and self.query_conv.weight.device == x.device and x.shape[1] == self.query_conv.in_channels
                and planes_cover(B, C, cq, H, W)):
            return "f32-planes-cast"
        return "separate-strips"

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError(
                "CrissCrossAttention (ccnet_amd): input is on the CPU. This module runs its attention core as HIP "
                "kernels on an AMD GPU and has no CPU fallback; move the module and its input to the device.")
        r = self.route(x)
        cq = self.query_conv.out_channels
        params = (self.query_conv.weight, self.query_conv.bias, self.key_conv.weight, self.key_conv.bias,
                  self.value_conv.weight, self.value_conv.bias)
        if r == "bf16-pixel-major":
            # x as (B, H, W, C) is a free view of a channels_last tensor (one transposing copy otherwise); query | key | value
            # are ONE GEMM x^T W^T whose output the kernels read through channel-slice strides; y comes back in x's memory format
            xp = x.permute(0, 2, 3, 1)
            qkv = torch.nn.functional.linear(xp, self._stacked_weight().flatten(1), self._stacked_bias()).to(torch.bfloat16)
            y = CrissCrossPMBF16Function.apply(qkv, xp, self.gamma.float(), cq, self.recompute_attention).permute(0, 3, 1, 2)
            return y if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() else y.contiguous()
        if r == "f32-planes":
            split_gemm = self.split_bf16_projections and x.shape[0] * x.shape[2] * x.shape[3] >= self.split_bf16_min_pixels
            y = CrissCrossPlanesModuleFunction.apply(x.contiguous(), *params, self.gamma, split_gemm, self.recompute_attention)
            # (a channels_last input gets its output back in its own memory format, as torch's own operators do)
            cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
            return y.contiguous(memory_format=torch.channels_last) if cl else y
        if r == "f32-planes-cast":
            with torch.autocast(device_type="cuda", enabled=False):          # (the node's GEMMs are its own: fp32 / split-bf16 x3)
                split_gemm = self.split_bf16_projections and x.shape[0] * x.shape[2] * x.shape[3] >= self.split_bf16_min_pixels
                # (the node's projections are its own fp32 / split-bf16 x3 GEMMs on fp32 copies of the parameters -- a no-op for an
                #  fp32 module under autocast: autocast does NOT govern them, unlike the reference's autocast convolutions)
                y = CrissCrossPlanesModuleFunction.apply(x.float().contiguous(), *(p.float() for p in params), self.gamma.float(),
                                                         split_gemm, self.recompute_attention)
            return y.to(x.dtype)
        proj_query, proj_key, proj_value = self.query_conv(x), self.key_conv(x), self.value_conv(x)
        out = CrissCrossFunction.apply(proj_query.float(), proj_key.float(), proj_value.float(),
                                       x.float(), self.gamma.float(), self.recompute_attention)
        return out.to(x.dtype)

    def _stacked_weight(self):
        return torch.cat([self.query_conv.weight, self.key_conv.weight, self.value_conv.weight], 0)

    def _stacked_bias(self):
        return torch.cat([self.query_conv.bias, self.key_conv.bias, self.value_conv.bias], 0)

    def _fusable(self, x=None):
        """The packed path bypasses ``nn.Conv2d.forward``: it needs the three projections to still be the plain dense
        biased 1x1 convolutions the constructor made (a user may have swapped one out, changed its geometry, cast it
        or hooked it) and, when ``x`` is given, their weights to live where and as what ``x`` does."""
        convs = (self.query_conv, self.key_conv, self.value_conv)
        cin = convs[0].in_channels

        def plain(c):
            return (type(c) is nn.Conv2d and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0)
                    and c.dilation == (1, 1) and c.groups == 1 and c.bias is not None and c.in_channels == cin
                    and c.padding_mode == "zeros" and tuple(c.weight.shape) == (c.out_channels, cin, 1, 1)
                    and not c._forward_hooks and not c._forward_pre_hooks and not c._backward_hooks
                    and not getattr(c, "_backward_pre_hooks", None)
                    and c.weight.dtype == c.bias.dtype and c.weight.device == c.bias.device)

        if not all(plain(c) for c in convs):
            return False
        if self.value_conv.out_channels != cin or self.query_conv.out_channels != self.key_conv.out_channels:
            return False
        if x is not None:
            return all(c.weight.device == x.device and c.weight.dtype == x.dtype for c in convs) and x.shape[1] == cin
        return len({(c.weight.device, c.weight.dtype) for c in convs}) == 1


def graph_module(module: "CrissCrossAttention", sample_input: torch.Tensor, warmup_iters: int = 3):
    """Static-shape fast path for SMALL per-GPU batches (the reference trains at 1-2 images per GPU, engine.py:88): the module's
    forward and its backward each captured into ONE hipGraph (``torch.cuda.make_graphed_callables``).  At (1,512,97,97) a step
    is ~25 torch ops + 12 kernel launches for ~0.3 ms of GPU work -- eager, the host cannot issue them that fast (VERDICT r3
    item 7); replayed as two graphs the step is bound by its kernels.  Everything the node does is capturable: the C ABI launches
    on the capturing stream, its side stream forks and joins inside the capture, workspaces come from torch's allocator (the
    graph's private pool), and the stacked / split weights are packed INSIDE the graph (one launch per forward, no cache), so
    replays see in-place optimizer updates.  Returns a callable ``f(x) -> y`` bound to ``sample_input``'s shape,
    dtype and memory format; gradients flow to ``x`` and to the module's parameters as usual."""
    if not sample_input.is_cuda:
        raise RuntimeError("graph_module: the module runs on an AMD GPU only")
    return torch.cuda.make_graphed_callables(module, (sample_input,), num_warmup_iters=warmup_iters)

