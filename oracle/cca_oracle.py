"""CPU oracle for CCNet's criss-cross attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ccnet_amd/`` or ``cc_attention/`` may import this
file; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
do, and there only as the checker / the timed CPU baseline -- never as the product path.

What it restates (reference = /root/reference, pure-python branch):

    cc_attention/functions.py:11-12   INF(B,H,W): -inf on the diagonal of the column branch
    cc_attention/functions.py:27-49   CrissCrossAttention.forward

The reference has no custom backward (autograd derives it) and no tests / golden vectors, so the
oracle is pinned against the *live* reference module run in the build container:
``tests/golden/make_golden.py`` imports /root/reference/cc_attention, overrides the per-instance
``INF`` attribute (functions.py:23) with a device-agnostic equivalent (functions.py:12 hard-codes
``.cuda()``), runs forward + autograd backward and stores the results under ``tests/golden/``.
``tests/test_oracle.py`` checks every function below against those fixtures.

Tensor conventions (all identical to the reference's intermediates):

    q, k          (B, Cq, H, W)      proj_query / proj_key         functions.py:29,32
    v, x, y       (B, C,  H, W)      proj_value / input / output   functions.py:35,49
    energy, A     (B, H, W, H+W)     ``concate`` before / after softmax, slot-fastest,
                                     slots [0,H) = column branch (key/value at (j, w), slot j==h
                                     masked to -inf), slots [H,H+W) = row branch (key/value at
                                     (h, j))                          functions.py:38-40

Two independent restatements are provided: an einsum form (fast, any float dtype, used as the
checker and as the timed CPU baseline) and a literal loop form (tiny shapes only) that follows
the index definitions one multiply-add at a time.  They are checked against each other and
against the live-reference fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

__all__ = [
    "ca_forward", "ca_softmax", "ca_map_forward", "ca_map_backward", "ca_softmax_backward",
    "ca_backward", "cca_core_forward", "cca_core_backward", "cca_module_forward",
    "cca_module_forward_backward", "ca_forward_loops", "ca_map_forward_loops",
    "algorithmic_bytes", "algorithmic_flops",
]


# ----------------------------------------------------------------------------------------------
# einsum restatement
# ----------------------------------------------------------------------------------------------
def ca_forward(q: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """Row/column restricted affinity with the column self-slot masked.

    functions.py:38  energy_H = bmm(Q_H, K_H) + INF  -> eH[b,h,w,j] = sum_c q[b,c,h,w] k[b,c,j,w],
                     j == h -> -inf
    functions.py:39  energy_W = bmm(Q_W, K_W)        -> eW[b,h,w,j] = sum_c q[b,c,h,w] k[b,c,h,j]
    functions.py:40  cat([energy_H, energy_W], 3)
    """
    B, _, H, W = q.shape
    eH = torch.einsum("bchw,bcjw->bhwj", q, k)
    eW = torch.einsum("bchw,bchj->bhwj", q, k)
    idx = torch.arange(H)
    eH = eH.clone()
    eH[:, idx, :, idx] = -math.inf          # functions.py:11-12 (diag of every (H,H) block)
    return torch.cat([eH, eW], dim=3)


def ca_softmax(energy: torch.Tensor) -> torch.Tensor:
    """functions.py:40  Softmax(dim=3) over the H+W slots (masked slot -> exactly 0)."""
    return torch.softmax(energy, dim=3)


def ca_map_forward(A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Aggregation out_H + out_W (without gamma / residual).

    functions.py:42,46  out_H[b,c,h,w] = sum_j A[b,h,w,j]   v[b,c,j,w]
    functions.py:45,47  out_W[b,c,h,w] = sum_j A[b,h,w,H+j] v[b,c,h,j]
    """
    H = v.shape[2]
    oH = torch.einsum("bhwj,bcjw->bchw", A[..., :H], v)
    oW = torch.einsum("bhwj,bchj->bchw", A[..., H:], v)
    return oH + oW


def ca_map_backward(do: torch.Tensor, A: torch.Tensor, v: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Adjoint of :func:`ca_map_forward` (autograd of functions.py:42-47).

    dA[b,h,w,j]   = sum_c do[b,c,h,w] v[b,c,j,w]      dA[b,h,w,H+j] = sum_c do[b,c,h,w] v[b,c,h,j]
    dv[b,c,j,w]   = sum_h A[b,h,w,j] do[b,c,h,w]  +  sum_w' A[b,j,w',H+w] do[b,c,j,w']
    """
    H = v.shape[2]
    dA = torch.cat([torch.einsum("bchw,bcjw->bhwj", do, v),
                    torch.einsum("bchw,bchj->bhwj", do, v)], dim=3)
    dv = (torch.einsum("bhwj,bchw->bcjw", A[..., :H], do)
          + torch.einsum("bhwj,bchw->bchj", A[..., H:], do))
    return dA, dv


def ca_softmax_backward(A: torch.Tensor, dA: torch.Tensor) -> torch.Tensor:
    """Adjoint of Softmax(dim=3) (functions.py:40): dE = A * (dA - sum_s A dA); A==0 -> dE==0."""
    r = (A * dA).sum(dim=3, keepdim=True)
    return A * (dA - r)


def ca_backward(dE: torch.Tensor, q: torch.Tensor, k: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Adjoint of :func:`ca_forward` (autograd of functions.py:38-39); dE at the masked slot is 0.

    dq[b,c,h,w] = sum_j dE[b,h,w,j] k[b,c,j,w] + sum_j dE[b,h,w,H+j] k[b,c,h,j]
    dk[b,c,j,w] = sum_h dE[b,h,w,j] q[b,c,h,w] + sum_w' dE[b,j,w',H+w] q[b,c,j,w']
    """
    H = q.shape[2]
    dEH, dEW = dE[..., :H], dE[..., H:]
    dq = torch.einsum("bhwj,bcjw->bchw", dEH, k) + torch.einsum("bhwj,bchj->bchw", dEW, k)
    dk = torch.einsum("bhwj,bchw->bcjw", dEH, q) + torch.einsum("bhwj,bchw->bchj", dEW, q)
    return dq, dk


def cca_core_forward(q, k, v, x, gamma) -> Tuple[torch.Tensor, torch.Tensor]:
    """Attention core of functions.py:38-49: (q,k,v,x,gamma) -> (y, A)."""
    A = ca_softmax(ca_forward(q, k))
    y = gamma * ca_map_forward(A, v) + x                      # functions.py:49
    return y, A


def cca_core_backward(dy, q, k, v, A, gamma) -> Dict[str, torch.Tensor]:
    """Adjoint of :func:`cca_core_forward` w.r.t. q, k, v, x, gamma."""
    t, dv_unscaled = ca_map_backward(dy, A, v)               # t = d(out_H+out_W)/dA un-scaled
    dgamma = (A * t).sum().reshape(1)                        # = sum dy * (out_H+out_W)
    dE = ca_softmax_backward(A, gamma * t)
    dq, dk = ca_backward(dE, q, k)
    return {"dq": dq, "dk": dk, "dv": gamma * dv_unscaled, "dx": dy.clone(), "dgamma": dgamma}


# ----------------------------------------------------------------------------------------------
# module level (adds the three 1x1 convolutions, functions.py:29,32,35)
# ----------------------------------------------------------------------------------------------
def _conv1x1(x, w, b):
    return torch.einsum("oc,bchw->bohw", w.reshape(w.shape[0], w.shape[1]), x) + b.view(1, -1, 1, 1)


def cca_module_forward(x, params: Dict[str, torch.Tensor]):
    """CrissCrossAttention.forward (functions.py:27-49) from a state_dict-shaped ``params``."""
    q = _conv1x1(x, params["query_conv.weight"], params["query_conv.bias"])
    k = _conv1x1(x, params["key_conv.weight"], params["key_conv.bias"])
    v = _conv1x1(x, params["value_conv.weight"], params["value_conv.bias"])
    y, A = cca_core_forward(q, k, v, x, params["gamma"])
    return y, (q, k, v, A)


def cca_module_forward_backward(x, params, dy):
    """Forward + closed-form backward of the whole module: returns y, dx and the 7 param grads."""
    y, (q, k, v, A) = cca_module_forward(x, params)
    g = cca_core_backward(dy, q, k, v, A, params["gamma"])
    grads = {"gamma": g["dgamma"]}
    dx = g["dx"]
    for name, dz in (("query_conv", g["dq"]), ("key_conv", g["dk"]), ("value_conv", g["dv"])):
        w = params[name + ".weight"]
        w2 = w.reshape(w.shape[0], w.shape[1])
        grads[name + ".weight"] = torch.einsum("bohw,bchw->oc", dz, x).reshape(w.shape)
        grads[name + ".bias"] = dz.sum(dim=(0, 2, 3))
        dx = dx + torch.einsum("oc,bohw->bchw", w2, dz)
    return y, dx, grads


# ----------------------------------------------------------------------------------------------
# literal loop restatement (tiny shapes only) -- pins the slot <-> source index map
# ----------------------------------------------------------------------------------------------
def ca_forward_loops(q: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    B, Cq, H, W = q.shape
    e = torch.empty(B, H, W, H + W, dtype=q.dtype)
    for b in range(B):
        for h in range(H):
            for w in range(W):
                for j in range(H):          # column branch: key at (j, w); self slot masked
                    e[b, h, w, j] = -math.inf if j == h else float((q[b, :, h, w] * k[b, :, j, w]).sum())
                for j in range(W):          # row branch: key at (h, j); self slot kept
                    e[b, h, w, H + j] = float((q[b, :, h, w] * k[b, :, h, j]).sum())
    return e


def ca_map_forward_loops(A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    B, C, H, W = v.shape
    o = torch.zeros(B, C, H, W, dtype=v.dtype)
    for b in range(B):
        for h in range(H):
            for w in range(W):
                for j in range(H):
                    o[b, :, h, w] += A[b, h, w, j] * v[b, :, j, w]
                for j in range(W):
                    o[b, :, h, w] += A[b, h, w, H + j] * v[b, :, h, j]
    return o


# ----------------------------------------------------------------------------------------------
# accounting used by bench.py / DESIGN.md
# ----------------------------------------------------------------------------------------------
def algorithmic_bytes(B: int, C: int, H: int, W: int, elt_size: int = 4) -> int:
    """Compulsory HBM bytes of the core fwd+bwd: six C-sized and six Cq-sized streams."""
    Cq = C // 8
    return elt_size * B * H * W * (6 * C + 6 * Cq)


def algorithmic_flops(B: int, C: int, H: int, W: int) -> int:
    """2*P*S*(C+Cq) forward, twice that backward."""
    Cq = C // 8
    return 3 * 2 * B * H * W * (H + W) * (C + Cq)
