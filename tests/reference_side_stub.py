"""The reference-side binding a CCNet maintainer would add (INTEGRATION.md section 2 shows this file verbatim).

It is what replaces lines 29-49 of ``CrissCrossAttention.forward`` in /root/reference/cc_attention/functions.py when the
reference keeps its own module and binds ``libccnet_cca.so`` with ctypes: the three 1x1 projections become ONE ``F.linear`` on
the channels-last view of x (its (B, H, W, 2 Cq + C) output holds q | k | v pixel-major -- the layout the fast kernels read in
place), and functions.py:38-49 + its autograd become one call each of ``ccnet_cca_forward_planes_f32`` /
``ccnet_cca_backward_planes_f32``.  Nothing of ``ccnet_amd`` is imported: the C ABI of include/ccnet_cca.h is the boundary.

Test infrastructure: tests/test_gpu_parity.py::test_reference_side_stub_binds_the_fast_kernels runs it against the oracle.
"""
import ctypes

import torch
import torch.nn.functional as F

_P, _I, _L, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
CCNET_WS_PLANES_FORWARD, CCNET_WS_PLANES_BACKWARD = 5, 6


def bind(path):
    lib = ctypes.CDLL(path)                                   # after ``import torch``: shares its HIP runtime
    lib.ccnet_cca_version.restype = _I
    assert lib.ccnet_cca_version() == 220, "written against C ABI 0.2.2 (include/ccnet_cca.h)"
    lib.ccnet_cca_last_error_string.restype = ctypes.c_char_p
    lib.ccnet_cca_workspace_bytes.restype = _Z
    lib.ccnet_cca_workspace_bytes.argtypes = [_I] * 6
    lib.ccnet_cca_forward_planes_f32.argtypes = [_P] * 9 + [_I] * 5 + [_L, _I] * 4 + [_P, _Z, _P]
    lib.ccnet_cca_backward_planes_f32.argtypes = [_P] * 12 + [_I] * 5 + [_L, _I] * 7 + [_P, _Z, _P]
    return lib


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.ccnet_cca_last_error_string().decode())


def make_function(lib):
    class CrissCrossCore(torch.autograd.Function):
        """functions.py:38-49 on the packed projection: (qkv (B, H, W, 2 Cq + C) fp32, x NCHW fp32, gamma) -> y NCHW."""

        @staticmethod
        def forward(ctx, qkv, x, gamma, cq):
            B, H, W, ct = qkv.shape
            C = ct - 2 * cq
            y, A = torch.empty_like(x), x.new_empty(B, H, W, H + W)
            # strips <= 100 positions: the plane-free form (v stays the fp32 slice it is); longer ones: the call writes v's planes
            vpl = None if max(H, W) <= 100 else torch.empty(B, H, W, 2, C, dtype=torch.int16, device=x.device)
            n = lib.ccnet_cca_workspace_bytes(CCNET_WS_PLANES_FORWARD, B, C, cq, H, W)
            ws = x.new_empty(n // 4 + 1)
            p, bs = qkv.data_ptr(), H * W * ct
            _check(lib, lib.ccnet_cca_forward_planes_f32(
                p, p + 4 * cq, p + 8 * cq, None, None if vpl is None else vpl.data_ptr(), x.data_ptr(), gamma.data_ptr(),
                y.data_ptr(), A.data_ptr(), B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * 2 * C, 2 * C,
                ws.data_ptr(), n, torch.cuda.current_stream().cuda_stream))
            ctx.save_for_backward(qkv, A, gamma, *([] if vpl is None else [vpl]))
            ctx.cq = cq
            return y

        @staticmethod
        def backward(ctx, dy):
            qkv, A, gamma = ctx.saved_tensors[:3]
            vpl = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
            cq = ctx.cq
            B, H, W, ct = qkv.shape
            C = ct - 2 * cq
            dy = dy.contiguous()
            dqkv, dg, scratch = torch.empty_like(qkv), torch.empty_like(gamma), torch.empty_like(A)
            n = lib.ccnet_cca_workspace_bytes(CCNET_WS_PLANES_BACKWARD, B, C, cq, H, W)
            ws = dy.new_empty(n // 4 + 1)
            p, g, bs = qkv.data_ptr(), dqkv.data_ptr(), H * W * ct
            _check(lib, lib.ccnet_cca_backward_planes_f32(
                dy.data_ptr(), p, p + 4 * cq, p + 8 * cq if vpl is None else None, None if vpl is None else vpl.data_ptr(),
                A.data_ptr(), gamma.data_ptr(), g, g + 4 * cq, g + 8 * cq, dg.data_ptr(), scratch.data_ptr(),
                B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * 2 * C, 2 * C, bs, ct, bs, ct, bs, ct,
                ws.data_ptr(), n, torch.cuda.current_stream().cuda_stream))
            return dqkv, dy, dg, None                      # (d/dx of the residual of functions.py:49 is dy itself)

    return CrissCrossCore


def forward(self, x, core):
    """Body of ``CrissCrossAttention.forward(self, x)`` (functions.py:27-49) on the binding above."""
    cq = self.query_conv.out_channels
    w = torch.cat([self.query_conv.weight, self.key_conv.weight, self.value_conv.weight]).flatten(1)
    b = torch.cat([self.query_conv.bias, self.key_conv.bias, self.value_conv.bias])
    qkv = F.linear(x.permute(0, 2, 3, 1), w, b)              # functions.py:29,32,35 as one GEMM: (B, H, W, 2 Cq + C), q | k | v
    return core.apply(qkv.contiguous(), x.contiguous(), self.gamma, cq)
