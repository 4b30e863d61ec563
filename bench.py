#!/usr/bin/env python3
"""bench.py -- criss-cross attention fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8 --steps 50 --warmup 10          # self-spawns 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: forward (q,k,v,x,gamma -> y, A) and backward
(dy -> dq, dk, dv, dgamma) of the attention core at BASELINE.json configs[1], (8,512,97,97) fp32, R=1,
issued through the C ABI (include/ccnet_cca.h) with every buffer already resident in HBM.  The 1x1
projections around the core stay torch ops and are reported separately (``module_ms_per_step``), not in
``value``.

Multi-GPU (reference: run_local.sh:18, engine.py:52-57,85-88): the path shards along the batch with no exchange
inside the op, so every rank runs the same per-GPU batch on its own shard ("weak" scaling, no
data-path collective); the timed region is bracketed by barrier + synchronize and the maximum over ranks is used.
``--gpus N`` with no WORLD_SIZE in the environment re-executes this file under ``torch.distributed.run`` with N
ranks; a WORLD_SIZE that disagrees with ``--gpus`` is an error.  ``--allreduce-grads`` adds the all-reduce of the
seven CrissCrossAttention parameter gradients (328,321 floats, what DDP moves for this module: engine.py:75) to
every step.

Rank 0 prints ONE JSON line: the driver contract plus
  ``roofline``      op-level: SURVEY 8(d) algorithmic bytes of one step / the step time, against 8 TB/s; the
                    slowest single launch is described under ``roofline.dominant_kernel`` (live HIP events);
  ``cpu_baseline``  the CPU oracle timed on a bounded sample on this box's host cores;
  ``imgs_per_s``    the other half of BASELINE.json's metric: ResNet-101 + RCCA(R=2) synthetic 769x769 train
                    step (ccnet_amd.train_synthetic, DDP over RCCL when N > 1), whole-job images/s;
  ``step_ms_stats`` median / p10 / p90 of per-step HIP-event times, warm (back to back) and cold (a 512 MiB
                    buffer rewritten between steps, so nothing of the step's tensors survives in the 256 MiB
                    Infinity Cache).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4-copy ceiling, same table
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_16x16x4_f32 = fp32 vector rate
CCA_PARAM_FLOATS = lambda C: 2 * ((C // 8) * C + C // 8) + C * C + C + 1      # noqa: E731  (7 tensors)


# --------------------------------------------------------------------------------------------
# accounting (DESIGN.md "algorithmic bytes")
# --------------------------------------------------------------------------------------------
def core_bytes(B, C, H, W, elt=4):
    """Compulsory HBM bytes of one core fwd+bwd: six C-sized and six C/8-sized streams."""
    return elt * B * H * W * (6 * C + 6 * (C // 8))


def core_flops(B, C, H, W):
    return 3 * 2 * B * H * W * (H + W) * (C + C // 8)


def kernel_accounting(kind, B, K, H, W, row):
    """Algorithmic bytes / flops of ONE strip-kernel launch (one branch).

    weight kernel: reads X and Y (B,K,H,W) once, writes its half of the (B,H,W,H+W) attention tensor.
    map kernel   : reads its half of the attention tensor and F (B,K,H,W), writes out (B,K,H,W); the column
                   launch also reads the residual when there is one, the row launch re-reads the column partial.
    """
    L = W if row else H
    feat = 4 * B * K * H * W
    att = 4 * B * H * W * L
    flops = 2 * B * H * W * L * K
    if kind == "weight":
        return 2 * feat + att, flops
    nbytes = att + 2 * feat
    if row:
        nbytes += feat                      # column partial re-read
    elif kind == "map_resid":
        nbytes += feat                      # residual x (added by the column launch)
    return nbytes, flops


def metric_label(C, H, W):
    return f"CrissCrossAttention core fwd+bwd algorithmic GB/s at (B,{C},{H},{W})"


# --------------------------------------------------------------------------------------------
# distributed helpers (pure host logic; covered by tests/test_dist_gloo.py on the gloo backend)
# --------------------------------------------------------------------------------------------
def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def shard_seed(base_seed, rank):
    """Per-rank seed as in the reference's train.py:154-155 (seed = local rank offset)."""
    return base_seed + rank


def max_over_ranks(seconds, device, world):
    """MAX-reduce a local wall time over all ranks (identity when no process group is up)."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(obj, world):
    """Every rank's ``obj`` as a list indexed by rank (a one-element list when no process group is up)."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def rank_description(device, on_gpu):
    """What a rank runs on: the first thing to read when the per-rank times of an N-GPU run differ (engine.py:52-57 binds one
    process to one device by LOCAL_RANK; a rank that landed on the wrong device or a shared one shows up here)."""
    d = {"host": socket.gethostname(), "pid": os.getpid(), "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}
    if on_gpu:
        p = torch.cuda.get_device_properties(device)
        d.update({"device": str(device), "name": p.name, "cus": p.multi_processor_count,
                  "hbm_gib": round(p.total_memory / 2 ** 30, 1)})
        try:
            d["pci_bus_id"] = getattr(p, "pci_bus_id", None)
            d["gcn_arch"] = getattr(p, "gcnArchName", None)
        except Exception:
            pass
    else:
        d["device"] = "cpu"
    return d


def collective_library(on_gpu):
    """The collective library behind backend "nccl" (= RCCL on ROCm) -- the line NCCL_DEBUG=VERSION would print."""
    info = {"backend": "nccl (RCCL)" if on_gpu else "gloo", "torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    if on_gpu:
        try:
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as e:
            info["rccl_version"] = f"unavailable: {e}"
        for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME", "RCCL_MSCCL_ENABLE"):
            if k in os.environ:
                info.setdefault("env", {})[k] = os.environ[k]
    return info


def aggregate_value(bytes_per_step_per_rank, steps, world, seconds):
    """Whole-job GB/s: all ranks' bytes over the slowest rank's time."""
    return world * bytes_per_step_per_rank * steps / seconds / 1e9


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, backend="nccl"):
    """``--gpus N`` without a launcher: re-execute this file as N ranks of one node (the reference's
    ``python -m torch.distributed.launch --nproc_per_node=N`` of run_local.sh:18).  Returns (rc, parsed JSON line)."""
    if backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.exit(f"bench.py: --gpus {n} requested but this node exposes {have} HIP device(s); refusing to "
                     f"report a {n}-GPU number from fewer devices")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] spawning:", " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            line = ln
    if line is not None:
        print(line, flush=True)
    return p.returncode, (json.loads(line) if line else None)


# --------------------------------------------------------------------------------------------
class CoreWorkload:
    """Pre-allocated device buffers + the two C-ABI calls of one step."""

    def __init__(self, lib, B, C, H, W, device, seed):
        self.lib, self.shape, self.device = lib, (B, C, H, W), device
        g = torch.Generator(device="cpu").manual_seed(seed)
        Cq = C // 8

        def rnd(*s):
            return torch.randn(*s, generator=g).to(device)

        self.q, self.k = rnd(B, Cq, H, W), rnd(B, Cq, H, W)
        self.v, self.x, self.dy = rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, C, H, W)
        self.gamma = torch.full((1,), 0.5, device=device)
        self.y = torch.empty_like(self.x)
        self.A = torch.empty(B, H, W, H + W, device=device)
        self.scratch = torch.empty_like(self.A)
        self.dq, self.dk, self.dv = torch.empty_like(self.q), torch.empty_like(self.k), torch.empty_like(self.v)
        self.dgamma = torch.empty(1, device=device)
        self.ws_bytes = lib.ccnet_cca_backward_workspace_bytes(B, C, Cq, H, W)      # incl. small-batch K-split slabs
        self.ws = torch.empty(self.ws_bytes // 4 + 1, device=device)
        self.fws_bytes = lib.ccnet_cca_forward_workspace_bytes(B, C, Cq, H, W)
        self.fws = torch.empty(self.fws_bytes // 4 + 1, device=device)

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def forward(self):
        B, C, H, W = self.shape
        L = self.lib
        Cq, hw = C // 8, H * W
        L.check(L.ccnet_cca_forward_ws_f32(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), self.x.data_ptr(),
                                           self.gamma.data_ptr(), self.y.data_ptr(), self.A.data_ptr(),
                                           B, C, Cq, H, W, Cq * hw, Cq * hw, C * hw,
                                           self.fws.data_ptr() if self.fws_bytes else None, self.fws_bytes,
                                           self.stream()), "cca_forward")

    def backward(self):
        B, C, H, W = self.shape
        L = self.lib
        L.check(L.ccnet_cca_backward_f32(self.dy.data_ptr(), self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(),
                                         self.A.data_ptr(), self.gamma.data_ptr(), self.dq.data_ptr(),
                                         self.dk.data_ptr(), self.dv.data_ptr(), self.dgamma.data_ptr(),
                                         self.scratch.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                         B, C, C // 8, H, W, self.stream()), "cca_backward")

    def step(self):
        self.forward()
        self.backward()

    # ---- single-kernel launches for the roofline object (branch mask restricts to ONE kernel) ----
    def stage_table(self):
        B, C, H, W = self.shape
        L, s, Cq = self.lib, self.stream, C // 8
        P = lambda t: t.data_ptr()  # noqa: E731
        return {
            # name: (callable, kind, K)
            "ca_forward[q.k]": (lambda: L.ccnet_ca_forward_f32(P(self.q), P(self.k), P(self.scratch), B, Cq, H, W, 0, s()),
                                "weight", Cq, "weight_strip_kernel"),
            "ca_map_forward[A.v]": (lambda: L.ccnet_ca_map_forward_f32(P(self.A), P(self.v), P(self.x), P(self.gamma),
                                                                       P(self.y), B, C, H, W, s()),
                                    "map_resid", C, "map_strip_kernel"),
            "ca_map_backward.dA[dy.v]": (lambda: L.ccnet_ca_map_backward_f32(P(self.dy), P(self.A), P(self.v), P(self.gamma),
                                                                             P(self.scratch), None, B, C, H, W, s()),
                                         "weight", C, "weight_strip_kernel"),
            "ca_map_backward.dv[A^T.dy]": (lambda: L.ccnet_ca_map_backward_f32(P(self.dy), P(self.A), P(self.v),
                                                                               P(self.gamma), None, P(self.dv),
                                                                               B, C, H, W, s()),
                                           "map", C, "map_strip_kernel"),
        }

    def extra_stages(self):
        """Stages that are not single strip-kernel launches (timed as stages only)."""
        B, C, H, W = self.shape
        L, s, Cq = self.lib, self.stream, C // 8
        P = lambda t: t.data_ptr()  # noqa: E731
        return {
            "ca_backward[dq,dk]": lambda: L.ccnet_ca_backward_f32(P(self.scratch), P(self.q), P(self.k), P(self.dq),
                                                                  P(self.dk), B, Cq, H, W, s()),
            "softmax_forward": lambda: L.ccnet_ca_softmax_forward_f32(P(self.scratch), P(self.A), B, H, W, s()),
            "softmax_backward+dgamma": lambda: L.ccnet_ca_softmax_backward_f32(P(self.A), P(self.scratch), P(self.gamma),
                                                                               P(self.scratch), P(self.dgamma),
                                                                               P(self.ws), self.ws_bytes, B, H, W, s()),
        }


class PixelMajorBF16Workload:
    """BASELINE.json configs[4]: bf16 features as pixel-major views (q | k | v = channel slices of one packed projection),
    fp32 attention / softmax / accumulation -- ccnet_cca_forward_pm_bf16 + ccnet_cca_backward_pm_bf16 (csrc/cca_gmap.hpp)."""

    def __init__(self, lib, B, C, H, W, device, seed):
        self.lib, self.shape, self.device = lib, (B, C, H, W), device
        g = torch.Generator(device="cpu").manual_seed(seed)
        Cq = C // 8
        self.ct = ct = C + 2 * Cq

        def rnd(*s, scale=1.0):
            return (torch.randn(*s, generator=g) * scale).to(device).to(torch.bfloat16)

        self.qkv = rnd(B, H, W, ct, scale=0.5)
        self.x, self.dy = rnd(B, H, W, C), rnd(B, H, W, C)
        self.gamma = torch.full((1,), 0.5, device=device)
        self.y, self.dqkv = torch.empty_like(self.x), torch.empty_like(self.qkv)
        self.A = torch.empty(B, H, W, H + W, device=device)
        self.scratch = torch.empty_like(self.A)
        self.dgamma = torch.empty(1, device=device)
        self.fws_bytes = lib.ccnet_cca_pm_workspace_bytes(B, C, Cq, H, W, 0)
        self.ws_bytes = lib.ccnet_cca_pm_workspace_bytes(B, C, Cq, H, W, 1)
        self.ws = torch.empty(max(self.fws_bytes, self.ws_bytes) // 4 + 64, device=device)

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def forward(self):
        B, C, H, W = self.shape
        L, cq, ct, p = self.lib, C // 8, self.ct, self.qkv.data_ptr()
        bs = H * W * ct
        L.check(L.ccnet_cca_forward_pm_bf16(p, p + 2 * cq, p + 4 * cq, self.x.data_ptr(), self.gamma.data_ptr(),
                                            self.y.data_ptr(), self.A.data_ptr(), B, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                            H * W * C, C, H * W * C, C, self.ws.data_ptr(), self.fws_bytes, self.stream()),
                "cca_forward_pm_bf16")

    def backward(self):
        B, C, H, W = self.shape
        L, cq, ct, p, g = self.lib, C // 8, self.ct, self.qkv.data_ptr(), self.dqkv.data_ptr()
        bs = H * W * ct
        L.check(L.ccnet_cca_backward_pm_bf16(self.dy.data_ptr(), p, p + 2 * cq, p + 4 * cq, self.A.data_ptr(),
                                             self.gamma.data_ptr(), g, g + 2 * cq, g + 4 * cq, self.dgamma.data_ptr(),
                                             self.scratch.data_ptr(), B, C, cq, H, W, H * W * C, C, bs, ct, bs, ct, bs, ct,
                                             bs, ct, bs, ct, bs, ct, self.ws.data_ptr(), self.ws_bytes, self.stream()),
                "cca_backward_pm_bf16")

    def step(self):
        self.forward()
        self.backward()


class PlanesWorkload:
    """The fp32 core on the SPLIT-PLANE path (include/ccnet_cca.h): the step's inputs are what the reference's op gets --
    q | k | v fp32 (pixel-major channel slices of the packed projection, functions.py:29-37), x / dy NCHW fp32 -- and EVERY pass
    the op needs is inside the step (round 3 split v into planes outside it).  Strips <= 100 -- the headline -- run the
    PLANE-FREE form: v is never split into planes, its consumers read fp32 tiles and split per fragment; larger maps have the
    forward entry point write the planes (its first launch).  dy is transposed out of NCHW into planes inside the backward.
    Outputs: y NCHW, dq | dk | dv fp32 pixel-major, dgamma."""

    def __init__(self, lib, B, C, H, W, device, seed):
        self.lib, self.shape = lib, (B, C, H, W)
        g = torch.Generator(device="cpu").manual_seed(seed)
        Cq = C // 8
        self.ct = ct = C + 2 * Cq
        rnd = lambda *s: torch.randn(*s, generator=g).to(device)              # noqa: E731
        self.qkv = rnd(B, H, W, ct)
        self.x, self.dy = rnd(B, C, H, W), rnd(B, C, H, W)
        self.gamma = torch.full((1,), 0.5, device=device)
        self.y, self.dqkv = torch.empty_like(self.x), torch.empty_like(self.qkv)
        self.A = torch.empty(B, H, W, H + W, device=device)
        self.scratch = torch.empty_like(self.A)
        self.dgamma = torch.empty(1, device=device)
        # strips <= 100: the plane-free form (v is read as fp32 by every consumer); else the forward writes planes for the backward
        self.direct = max(H, W) <= 100
        self.vpl = None if self.direct else torch.empty(B, H, W, 2, C, dtype=torch.int16, device=device)
        self.fws_bytes = lib.ccnet_cca_planes_workspace_bytes(B, C, Cq, H, W, 0)
        self.ws_bytes = lib.ccnet_cca_planes_workspace_bytes(B, C, Cq, H, W, 1)
        self.ws = torch.empty(max(self.fws_bytes, self.ws_bytes) // 4 + 64, device=device)

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def step(self):
        self.forward()
        self.backward()

    def forward(self):
        B, C, H, W = self.shape
        L, cq, ct, p = self.lib, C // 8, self.ct, self.qkv.data_ptr()
        bs = H * W * ct
        L.check(L.ccnet_cca_forward_planes_f32(p, p + 4 * cq, p + 8 * cq, None, None if self.direct else self.vpl.data_ptr(), self.x.data_ptr(),
                                               self.gamma.data_ptr(), self.y.data_ptr(), self.A.data_ptr(),
                                               B, C, cq, H, W, bs, ct, bs, ct, bs, ct, H * W * 2 * C, 2 * C,
                                               self.ws.data_ptr(), self.fws_bytes, self.stream()),
                "cca_forward_planes")

    def backward(self):
        B, C, H, W = self.shape
        L, cq, ct, p, g = self.lib, C // 8, self.ct, self.qkv.data_ptr(), self.dqkv.data_ptr()
        bs = H * W * ct
        L.check(L.ccnet_cca_backward_planes_f32(self.dy.data_ptr(), p, p + 4 * cq, p + 8 * cq if self.direct else None,
                                                None if self.direct else self.vpl.data_ptr(), self.A.data_ptr(),
                                                self.gamma.data_ptr(), g, g + 4 * cq, g + 8 * cq,
                                                self.dgamma.data_ptr(), self.scratch.data_ptr(), B, C, cq, H, W, bs, ct, bs, ct, bs, ct,
                                                H * W * 2 * C, 2 * C, bs, ct, bs, ct, bs, ct, self.ws.data_ptr(), self.ws_bytes,
                                                self.stream()),
                "cca_backward_planes")


def bf16_config5(lib, device, shape=(16, 512, 129, 129), iters=20):
    """BASELINE.json configs[4] as an extra of the fp32 line: fwd+bwd of the pixel-major bf16 core, per-step statistics."""
    wl = PixelMajorBF16Workload(lib, *shape, device, 4321)
    st = per_step_stats(wl.step, iters)
    nbytes = core_bytes(*shape, elt=2)
    traffic = measured_traffic(lib, "traffic_bf16_latest.json")
    out = {"shape": list(shape), "dtype": "bf16 features (pixel-major), fp32 attention/softmax/accumulate",
           "ms_per_step": st, "fwd_ms": round(time_region(wl.forward, 10), 4), "bwd_ms": round(time_region(wl.backward, 10), 4),
           "algorithmic_bytes_per_step": nbytes, "GB_per_s": round(nbytes / (st["median"] * 1e-3) / 1e9, 1),
           "frac_of_hbm_roofline": round(nbytes / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "traffic_bytes_per_step": traffic.get("_step_total_bytes") if traffic else None}
    del wl
    torch.cuda.empty_cache()
    try:
        from ccnet_amd import CrissCrossAttention
        B, C, H, W = shape
        m = CrissCrossAttention(C).to(device).to(torch.bfloat16)
        with torch.no_grad():
            m.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        dy = torch.randn(B, C, H, W, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

        def one():
            x.grad = None
            m.zero_grad(set_to_none=True)
            m(x).backward(dy)

        one()
        out["module_ms_per_step"] = round(time_region(one, 5), 4)
        out["module_what"] = "CrissCrossAttention(512) bf16, channels_last in / out: one x^T W^T projection + the core + autograd"
        del m, x, dy
    except Exception as e:          # the metric does not depend on it
        out["module_ms_per_step"] = f"failed: {e}"
    torch.cuda.empty_cache()
    return out


def time_region(fn, iters):
    """HIP events on torch's current stream (the stream the C ABI launches on)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters        # ms


def quantiles(xs):
    s = sorted(xs)
    pick = lambda p: s[min(len(s) - 1, max(0, int(round(p * (len(s) - 1)))))]  # noqa: E731
    return {"median": round(pick(0.5), 4), "p10": round(pick(0.1), 4), "p90": round(pick(0.9), 4),
            "min": round(s[0], 4), "n": len(s)}


def per_step_stats(fn, iters, flush=None):
    """One HIP-event pair per step (median & p10/p90).  ``flush`` (a > 256 MiB tensor) is rewritten
    before every step, outside the event pair: the step then starts with none of its tensors in the Infinity Cache."""
    pairs = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return quantiles([a.elapsed_time(b) for a, b in pairs])


def capture_step_graph(step_fn):
    """One step captured into a hipGraph (torch.cuda.CUDAGraph): the C ABI launches on torch's current stream, which is
    the capture stream inside the context.  Two eager runs on the side stream first (lazy module loading, workspace
    touch), as torch's capture recipe asks."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step_fn()
        step_fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step_fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def launch_accounting(lib, wl, graph, steps=20):
    """Where a step's wall time goes: the step timed eagerly and as a hipGraph replay with the shipped
    options, and -- with every launch on ONE stream ("planes_overlap" 0: the backward's dv passes otherwise run on the library's
    side stream next to dA / softmax-backward / dq | dk, and concurrent launches stretch each other) -- the step again and the
    sum of its launches' OWN durations (HIP-event pair around every launch, ccnet_cca_profile_*).  single-stream step minus
    that sum = GPU idle time between dependent launches (dispatch, host); single-stream step minus the shipped step = what
    the overlap buys."""
    out = {}
    for _ in range(3):
        wl.step()
    out["eager_ms_per_step"] = round(time_region(wl.step, steps), 4)
    if graph is not None:
        graph.replay()
        out["graph_ms_per_step"] = round(time_region(graph.replay, steps), 4)
    torch.cuda.synchronize()
    nrep = 5
    prev = lib.set_option("planes_overlap", 0)
    try:
        for _ in range(3):
            wl.step()
        out["single_stream_eager_ms_per_step"] = round(time_region(wl.step, steps), 4)
        rec = lib.profile_launches(lambda: [wl.step() for _ in range(nrep)])
    finally:
        lib.set_option("planes_overlap", prev)
    n = len(rec) // nrep
    ksum = sum(ms for _, ms in rec) / nrep
    out["launches_per_step"] = n
    out["gpu_kernel_sum_ms"] = round(ksum, 4)
    out["idle_ms"] = {"single_stream_eager": round(out["single_stream_eager_ms_per_step"] - ksum, 4)}
    out["overlap_gain_ms"] = round(out["single_stream_eager_ms_per_step"] - out["eager_ms_per_step"], 4)
    # launch i of a step, averaged over the profiled steps, in issue order (single stream: each launch alone on the GPU)
    out["launch_ms"] = [[rec[i][0].replace("cca::", ""), round(sum(rec[r * n + i][1] for r in range(nrep)) / nrep, 4)]
                        for i in range(n)]
    return out


def gpu_state_under_load(step, device_index=0, replays=3000):
    """Clocks and power of the GPU WHILE the step is running (rocm-smi sampled against a queue of asynchronously enqueued
    steps), the power CAP of the socket, and up to four power samples spread over the queue: rocm-smi's sclk is the REQUESTED
    level -- what a wave really ran at is measured in-band by ``gpu_probe`` below.  None when rocm-smi is not there."""
    import re
    num = lambda v: (lambda m: float(m.group(1)) if m else None)(re.search(r"([0-9.]+)", str(v)))   # noqa: E731
    try:
        cap = None
        try:
            p = subprocess.run(["rocm-smi", "-d", str(device_index), "--showmaxpower", "--json"],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
            card = next(iter(json.loads(p.stdout).values()))
            cap = next((num(v) for k, v in card.items() if "max" in k.lower() and "power" in k.lower()), None)
        except Exception:
            pass
        for _ in range(replays):
            step()
        samples, card, busy = [], {}, False
        for _ in range(4):
            p = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--showperflevel", "--json"],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
            if torch.cuda.current_stream().query():              # (the queue must still be running when the sample returns)
                break
            busy = True
            card = next(iter(json.loads(p.stdout).values()))
            samples.append(num(card.get("Current Socket Graphics Package Power (W)", "")))
        torch.cuda.synchronize()
        return {"sclk_mhz": num(card.get("sclk clock speed:", "")), "mclk_mhz": num(card.get("mclk clock speed:", "")),
                "fclk_mhz": num(card.get("fclk clock speed:", "")),
                "socket_power_w": samples[-1] if samples else None, "socket_power_w_samples": samples, "power_cap_w": cap,
                "perf_level": card.get("Performance Level"), "sampled_while_busy": bool(busy)}
    except Exception as e:          # the metric does not depend on it
        torch.cuda.synchronize()
        return {"error": str(e)[:120]}


def gpu_probe(lib, step, device, step_ms):
    """In-band state of the device this line was measured on (csrc/cca_probe.hpp).  A wave reads shader
    cycles (s_memtime) and the constant 100 MHz reference clock (s_memrealtime): the ratio of two deltas is the clock it really
    ran at.  Reported: that clock with the GPU otherwise idle, under a matrix-pipe burn, and WHILE THE STEP RUNS (one sampling
    wave per XCD next to a queue of replayed steps); the bf16 matrix rate of the burn; the issue -> landed time of the 25 KB
    LDS-DMA tile fill every strip kernel waits for, alone and with 768 workgroups streaming such tiles (and the rate they
    stream at); a float4 copy.  A box whose step is slow shows here WHY: clock under load, matrix rate, fill latency, copy."""
    out = {}
    NWG = 8
    side = torch.cuda.Stream(device)
    main = torch.cuda.current_stream()

    def sample_clock(nsamples, interval_us, load):
        buf = torch.zeros(2 * NWG * nsamples + NWG, dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        load()                                                       # (asynchronous: enqueued on the current stream)
        lib.check(lib.ccnet_cca_probe_clock(buf.data_ptr(), NWG, nsamples, int(interval_us * 100), side.cuda_stream), "probe_clock")
        torch.cuda.synchronize()
        v = buf.cpu().numpy()
        sm = v[:2 * NWG * nsamples].reshape(NWG, nsamples, 2).astype("float64")
        d = sm[:, 1:, :] - sm[:, :-1, :]
        mhz = d[:, :, 0] / d[:, :, 1] * 100.0                        # shader cycles per 10 ns tick
        flat = sorted(mhz.reshape(-1).tolist())
        pick = lambda q: flat[min(len(flat) - 1, int(q * (len(flat) - 1)))]   # noqa: E731
        per_xcc = {}
        for wg in range(NWG):
            per_xcc.setdefault(int(v[2 * NWG * nsamples + wg]), []).append(float(sorted(mhz[wg].tolist())[mhz.shape[1] // 2]))
        return {"median": round(pick(0.5), 1), "p05": round(pick(0.05), 1), "min": round(flat[0], 1), "max": round(flat[-1], 1),
                "per_xcc_median": {str(k): round(sum(x) / len(x), 1) for k, x in sorted(per_xcc.items())},
                "window_ms": round(nsamples * interval_us * 1e-3, 2)}

    try:
        time.sleep(0.05)
        out["clock_mhz_idle"] = sample_clock(200, 10, lambda: None)
        # matrix-pipe burn: 512 workgroups x 4 waves (two waves per SIMD), ~10 ms
        nwg, iters = 512, 40000
        clk = torch.zeros(nwg * 4, dtype=torch.int64, device=device)
        sink = torch.zeros(nwg * 256, device=device)
        burn = lambda: lib.check(lib.ccnet_cca_probe_mfma(clk.data_ptr(), sink.data_ptr(), nwg, iters, main.cuda_stream), "probe_mfma")   # noqa: E731
        burn()
        torch.cuda.synchronize()
        out["clock_mhz_mfma_burn"] = sample_clock(300, 10, burn)
        c = clk.cpu().numpy().reshape(nwg, 4).astype("float64")
        secs = (c[:, 3].max() - c[:, 1].min()) * 1e-8
        out["mfma_bf16_tflops"] = round(nwg * 4 * iters * 131072.0 / secs / 1e12, 1)
        out["mfma_burn_ms"] = round(secs * 1e3, 2)
        # the step: a queue of replays next to one sampling wave per XCD
        nrep = max(20, int(40.0 / max(step_ms, 0.05)))
        out["clock_mhz_during_step"] = sample_clock(2000, 10, lambda: [step() for _ in range(nrep)])
        out["clock_mhz_during_step"]["replays_enqueued"] = nrep
        # tile fills out of a 512 MiB source (they miss the L2): "strided" = a COLUMN strip of the packed fp32 projection at W = 97
        # (rows of 256 B, 97 * 640 * 4 B apart: every row in another page), "contiguous" = 25 KiB in one piece (a ROW strip's
        # tile).  A box that is slow on the column passes only shows up in the strided numbers.
        src = torch.empty(128 * 1024 * 1024, device=device).normal_()
        out["tile_fill"] = {}
        for layout, stride in (("strided", 97 * 640 * 4), ("contiguous", 256)):
            for label, n, reps in (("idle", 1, 300), ("loaded", 768, 200)):
                ck = torch.zeros(n * 4, dtype=torch.int64, device=device)
                for _ in range(2):
                    lib.check(lib.ccnet_cca_probe_dma(src.data_ptr(), src.numel() * 4, ck.data_ptr(), n, reps, stride, main.cuda_stream), "probe_dma")
                torch.cuda.synchronize()
                k = ck.cpu().numpy().reshape(n, 4).astype("float64")
                span = (k[:, 3].max() - k[:, 2].min()) * 1e-8
                out["tile_fill"][f"{layout}_{label}"] = {
                    "us_per_fill": round(float(((k[:, 3] - k[:, 2]) / reps).mean()) * 1e-2, 2),
                    "shader_cycles_mean": round(float((k[:, 0] / reps).mean()), 0), "shader_cycles_max": int(k[:, 1].max()),
                    "stream_tb_s": round(n * reps * 25600.0 / span / 1e12, 3), "workgroups": n}
        dst = torch.empty_like(src)
        dst.copy_(src)
        out["copy_f32_gb_s"] = round(2 * src.numel() * 4 / (time_region(lambda: dst.copy_(src), 10) * 1e-3) / 1e9, 1)
        del src, dst
        out["how"] = ("shader cycles (s_memtime) / 100 MHz reference ticks (s_memrealtime) read by the waves themselves; "
                      "ccnet_cca_probe_* in include/ccnet_cca.h")
    except Exception as e:          # the metric does not depend on it
        torch.cuda.synchronize()
        out["error"] = str(e)[:200]
    return out


def planes_launch_bytes(B, C, H, W):
    """Algorithmic (compulsory) bytes of every launch of one split-plane step, in issue order: what the launch must read and
    write once, from the tensor sizes of SURVEY 8(d) (feature C-sized 4*P*C, Cq-sized 4*P*Cq, attention-shaped 4*P*S)."""
    P, S, Cq = B * H * W, H + W, C // 8
    fc, fq, att = 4 * P * C, 4 * P * Cq, 4 * P * S
    # (label, algorithmic bytes, regex of the launch's kernel name as rocprofv3 spells it: the key of the PMC traffic summary --
    #  the launch profiler only knows the SOURCE spelling of the launch, template parameters by name)
    split = [("v fp32 -> planes", 2 * fc, r"pm_split_kernel")] if max(H, W) > 100 else []
    return split + [
        ("energies q.k (both branches)", 2 * fq + att, r"gweight_kernel<\d+, true, float"),
        ("softmax", 2 * att, r"softmax_fwd_kernel"),
        ("aggregation, column pass (v, A/2 -> partial)", 2 * fc + att // 2, r"gmap3_kernel<\d+, false, false, false"),
        ("aggregation, row pass (v, A/2, partial, x -> y NCHW)", 4 * fc + att // 2,
         r"gmap_kernel<\d+, true, false, true, (cca::bf16p_t|float), float, true"),
        ("dy NCHW -> planes", 2 * fc, r"nchw_to_planes_kernel"),
        ("dA = dy.v (both branches)", 2 * fc + att, r"gweight_stream_kernel|gweight_kernel<\d+, false, cca::bf16p_t"),
        ("dv, column pass (dy, A/2 -> partial)", 2 * fc + att // 2, r"gmap3_kernel<\d+, false, true, false"),
        ("dv, row pass (dy, A/2, partial -> dv)", 3 * fc + att // 2, r"gmap_kernel<\d+, true, true, true, cca::bf16p_t"),
        ("softmax backward + dgamma partials", 3 * att, r"softmax_bwd_kernel"),
        ("dq | dk, column pass (+ dgamma reduction)", att // 2 + 4 * fq, r"gmap_kernel<\d+, false, false, false, float, float, false, true, \d, false, (true|false)"),
        ("dq | dk, row pass", att // 2 + 6 * fq, r"gmap_kernel<\d+, true, false, true, float, float, false, true, \d, false, (true|false)"),
    ]


def planes_roofline(lib, wl, step_ms, launch_ms):
    """Op-level roofline of the split-plane step + its dominant launch (timed INSIDE the step by the launch profiler)."""
    B, C, H, W = wl.shape
    nbytes = core_bytes(B, C, H, W)
    ach = nbytes / (step_ms * 1e-3) / 1e9
    traffic = measured_traffic(lib, "traffic_planes_latest.json")
    obj = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(ach / HBM_COPY_GBS, 4),
           "level": "op (one core fwd+bwd)", "algorithmic_bytes": nbytes, "step_ms": round(step_ms, 4),
           "traffic": (traffic or {}).get("_step_total_bytes"),
           "traffic_source": ("profiles/traffic_planes_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the builder "
                              "(tools/pmc.sh), accepted because its kernel-source hash equals this build's; NOT a counter of this run"
                              if traffic else "none: no PMC summary for this build's kernel sources under profiles/")}
    table = planes_launch_bytes(B, C, H, W)

    def pmc_bytes(pattern):
        import re
        hits = [v for k, v in (traffic or {}).items() if not k.startswith("_") and re.search(pattern, k)]
        return hits[0] if len(hits) == 1 else None

    if len(launch_ms) == len(table):
        rows = [{"launch": what, "kernel": name, "ms": ms, "bytes": nb, "traffic": pmc_bytes(pat)}
                for (what, nb, pat), (name, ms) in zip(table, launch_ms)]
        obj["launches"] = [{"launch": r["launch"], "ms": r["ms"], "algorithmic_bytes": r["bytes"], "traffic": r["traffic"],
                            "achieved_gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1) if r["ms"] > 0 and r["bytes"] else None} for r in rows]
        dom = max(rows, key=lambda r: r["ms"])
        obj["dominant_kernel"] = {"kernel": dom["kernel"], "launch": dom["launch"], "kernel_ms": round(dom["ms"], 4),
                                  "algorithmic_bytes": dom["bytes"],
                                  "achieved_gbs": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                                  "frac": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "how": "HIP-event pair around the launch inside eager steps (ccnet_cca_profile_*)",
                                  "traffic": dom["traffic"]}
    return obj


def strips_family_summary(lib, B, C, H, W, device):
    """The same core on the NCHW strip kernels (q, k, v NCHW), for the family choice: eager step + in-step launch durations."""
    wl = CoreWorkload(lib, B, C, H, W, device, 1234)
    for _ in range(3):
        wl.step()
    out = {"ms_per_step": round(time_region(wl.step, 20), 4)}
    rec = lib.profile_launches(lambda: [wl.step() for _ in range(3)])
    n = len(rec) // 3
    out["launch_ms"] = [[rec[i][0].replace("cca::", ""), round(sum(rec[r * n + i][1] for r in range(3)) / 3, 4)] for i in range(n)]
    del wl
    torch.cuda.empty_cache()
    return out


def lib_sha16(lib):
    """identity of the build being benched: hash of the kernel sources it is compiled from (hipcc output itself is not
    bit-reproducible; build() recompiles whenever a source is newer than the library)"""
    from ccnet_amd import _lib
    return _lib.kernel_source_sha16()


def measured_traffic(lib, name="traffic_latest.json"):
    """Per-launch HBM bytes from the PMC passes (tools/pmc.sh -> tools/traffic_from_pmc.py), accepted only when
    they were taken on a build of the kernel sources that are being benched (source hash recorded next to them); otherwise None."""
    tpath = os.path.join(ROOT, "profiles", name)
    try:
        with open(tpath) as f:
            t = json.load(f)
        if t.get("_src_sha16") == lib_sha16(lib):
            return t
    except Exception:
        pass
    return None


def roofline_object(wl, step_ms, iters=20):
    """Op-level roofline (algorithmic bytes of one fwd+bwd / step time, against 8 TB/s) plus the
    slowest single launch, timed alone with HIP events.

    A step issues the weight kernel ONCE for both branches (column and row workgroups of one launch) and the map
    kernel once per branch, so those are the units timed here (branch mask 3 for the weight stages, 1 / 2 for the
    map stages)."""
    B, C, H, W = wl.shape
    lib = wl.lib
    rows, stages = [], {}
    try:
        for name, (fn, kind, K, kname) in wl.stage_table().items():
            masks = ((3, None),) if kind == "weight" else ((1, False), (2, True))
            for mask, row in masks:
                lib.ccnet_cca_set_branch_mask(mask)
                for _ in range(3):
                    lib.check(fn(), name)
                ms = time_region(lambda: lib.check(fn(), name), iters)
                if row is None:                 # both branches: inputs once, the whole attention tensor out
                    feat, att = 4 * B * K * H * W, 4 * B * H * W * (H + W)
                    nbytes, flops = 2 * feat + att, 2 * B * H * W * (H + W) * K
                    label = f"{kname} {name}"
                else:
                    nbytes, flops = kernel_accounting(kind, B, K, H, W, row)
                    label = f"{kname}<{'row' if row else 'col'}> {name}"
                rows.append({"kernel": label, "ms": ms, "bytes": nbytes, "flops": flops})
        lib.ccnet_cca_set_branch_mask(3)
        for name, (fn, kind, K, kname) in wl.stage_table().items():      # whole stages, as a step issues them
            for _ in range(3):
                lib.check(fn(), name)
            stages[name] = time_region(lambda: lib.check(fn(), name), iters)
        for name, fn in wl.extra_stages().items():
            for _ in range(3):
                lib.check(fn(), name)
            stages[name] = time_region(lambda: lib.check(fn(), name), iters)
        wl.forward()                                  # leave A / scratch consistent again
    finally:
        lib.ccnet_cca_set_branch_mask(3)
    roofline_object.stages = stages
    nbytes = core_bytes(B, C, H, W)
    ach = nbytes / (step_ms * 1e-3) / 1e9
    traffic = measured_traffic(lib)
    obj = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(ach / HBM_COPY_GBS, 4),
           "level": "op (one core fwd+bwd)", "algorithmic_bytes": nbytes, "step_ms": round(step_ms, 4),
           "traffic": (traffic or {}).get("_step_total_bytes"),
           "traffic_source": "profiles/traffic_latest.json (builder-side PMC passes, source-hash keyed)" if traffic else "none"}
    if rows:
        dom = max(rows, key=lambda r: r["ms"])
        obj["dominant_kernel"] = {
            "kernel": dom["kernel"], "kernel_ms": round(dom["ms"], 4), "algorithmic_bytes": dom["bytes"],
            "algorithmic_flops": dom["flops"],
            "achieved_gbs": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
            "frac": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": (traffic or {}).get(dom["kernel"])}
    return obj, rows


def host_description():
    """CPU model / core counts of the box (for the cpu_baseline object)."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "Model name":
                info["model"] = val
            elif key == "Socket(s)":
                info["sockets"] = int(val)
            elif key == "Core(s) per socket":
                info["cores_per_socket"] = int(val)
    except Exception:
        pass
    if "sockets" in info and "cores_per_socket" in info:
        info["physical_cores"] = info["sockets"] * info["cores_per_socket"]
    return info


def _time_cpu(one, budget_s, max_iters=50):
    one()                                            # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_iters:
            return el / n, n


def cpu_baseline(C, H, W, budget_s=20.0):
    """The reference's CPU path beside the device number: the REFERENCE FORMULATION
    of functions.py:27-49 (``reference_formulation``: the module's own bmm / permute / cat / softmax sequence + autograd,
    without the 1x1 convolutions -- the same core the device step runs) on this box's host cores, one image of the bench
    shape, fp32, swept over thread counts {1, 8, 16, 32, physical cores} (small bmm batches oversubscribe badly on
    many-core hosts, so "all cores" is not the best setting); the best setting is the headline, the sweep is kept.
    Also BASELINE configs[0] (2,64,32,32), and -- as an extra -- the einsum oracle port that the parity tests use."""
    host = host_description()
    phys = host.get("physical_cores") or host.get("logical_cpus") or 1
    ncpu = host.get("logical_cpus") or phys
    counts = sorted({t for t in (1, 8, 16, 32, phys) if 1 <= t <= ncpu})
    prev = torch.get_num_threads()
    B = 1
    sweep, best = {}, None
    try:
        one = reference_formulation_step(B, C, H, W, torch.device("cpu"))
        slice_s = budget_s * 0.7 / len(counts)
        for t in counts:
            torch.set_num_threads(t)
            sec, n = _time_cpu(one, slice_s)
            sweep[str(t)] = {"ms_per_image": round(sec / B * 1e3, 1), "iters": n}
            if best is None or sec < best[0]:
                best = (sec, t, n)
        torch.set_num_threads(best[1])
        cfg0 = reference_formulation_step(2, 64, 32, 32, torch.device("cpu"))
        sec0, n0 = _time_cpu(cfg0, budget_s * 0.05, 200)
        from oracle import cca_oracle as O
        torch.manual_seed(0)
        q, k = torch.randn(B, C // 8, H, W), torch.randn(B, C // 8, H, W)
        v, x, dy = torch.randn(B, C, H, W), torch.randn(B, C, H, W), torch.randn(B, C, H, W)
        gamma = torch.full((1,), 0.5)

        def port():
            y, A = O.cca_core_forward(q, k, v, x, gamma)
            O.cca_core_backward(dy, q, k, v, A, gamma)

        secp, n_p = _time_cpu(port, budget_s * 0.25, 20)
        # the quoted configuration itself -- ONE full-batch iteration of BASELINE configs[1] -- at the best thread
        # count of the sweep and at all physical cores (a fresh closure: 8 x the tensors of the one-image sample)
        full = {}
        try:
            fb = reference_formulation_step(8, C, H, W, torch.device("cpu"))
            for t in sorted({best[1], phys}):
                torch.set_num_threads(t)
                fb()                                     # warm-up (allocations)
                t0 = time.perf_counter()
                fb()
                sec_f = time.perf_counter() - t0
                full[str(t)] = {"ms_per_step": round(sec_f * 1e3, 1), "GB/s": round(core_bytes(8, C, H, W) / sec_f / 1e9, 3)}
            del fb
        except Exception as e:                           # (memory on a small host: the sample above stands)
            full = {"failed": str(e)}
    finally:
        torch.set_num_threads(prev)
    sec, t, n = best
    return {"value": round(core_bytes(B, C, H, W) / sec / 1e9, 3), "unit": "GB/s", "cores": t, "threads": t,
            "cores_used": t, "cores_total": phys, "logical_cpus": ncpu,
            "kind": "port", "port_of": "reference formulation (bench.reference_formulation: the reference's own torch op sequence; "
                                       "the reference module hard-codes .cuda() and does not travel to the GPU box)",
            "full_batch_configs1": {"shape": [8, C, H, W], "iters": 1, "by_threads": full},
            "sample": f"reference formulation of functions.py:27-49 (bmm / cat / softmax + autograd, no convolutions), "
                      f"batch {B} of ({B},{C},{H},{W}) fp32, {n} iters at {t} torch threads (best of the sweep)",
            "ms_per_image": round(sec / B * 1e3, 1), "thread_sweep": sweep, "host": host,
            "configs0_2x64x32x32": {"ms_per_step": round(sec0 * 1e3, 2), "iters": n0, "threads": t},
            "oracle_port": {"ms_per_image": round(secp / B * 1e3, 1), "iters": n_p, "threads": t,
                            "what": "oracle/cca_oracle.py (einsum restatement, the parity checker) at the same thread count"}}


def reference_formulation(q, k, v, x, gamma):
    """The reference's op sequence for functions.py:30-49 restated with torch ops on whatever device the tensors live on:
    permuted contiguous copies, four torch.bmm, the -inf diagonal (INF, :11-12), cat, softmax, the gamma / residual
    epilogue.  (The reference module itself hard-codes ``.cuda()`` in INF and is not importable on the GPU box.)"""
    B, _, H, W = q.shape
    device = q.device

    def cols(t):       # (B, c, H, W) -> (B*W, c, H): one matrix per image column
        return t.permute(0, 3, 1, 2).contiguous().view(B * W, -1, H)

    def rows(t):       # (B, c, H, W) -> (B*H, c, W): one matrix per image row
        return t.permute(0, 2, 1, 3).contiguous().view(B * H, -1, W)

    ninf = -torch.diag(torch.full((H,), float("inf"), device=device)).unsqueeze(0).repeat(B * W, 1, 1)
    e_col = (torch.bmm(cols(q).permute(0, 2, 1), cols(k)) + ninf).view(B, W, H, H).permute(0, 2, 1, 3)
    e_row = torch.bmm(rows(q).permute(0, 2, 1), rows(k)).view(B, H, W, W)
    att = torch.softmax(torch.cat([e_col, e_row], 3), dim=3)
    a_col = att[:, :, :, 0:H].permute(0, 2, 1, 3).contiguous().view(B * W, H, H)
    a_row = att[:, :, :, H:H + W].contiguous().view(B * H, W, W)
    o_col = torch.bmm(cols(v), a_col.permute(0, 2, 1)).view(B, W, -1, H).permute(0, 2, 3, 1)
    o_row = torch.bmm(rows(v), a_row.permute(0, 2, 1)).view(B, H, -1, W).permute(0, 2, 1, 3)
    return gamma * (o_col + o_row) + x


def reference_formulation_step(B, C, H, W, device, seed=0):
    """Seeded inputs + a callable running one fwd+bwd (autograd) of ``reference_formulation``."""
    g = torch.Generator().manual_seed(seed)
    Cq = C // 8
    q, k = (torch.randn(B, Cq, H, W, generator=g).to(device).requires_grad_() for _ in range(2))
    v, x = (torch.randn(B, C, H, W, generator=g).to(device).requires_grad_() for _ in range(2))
    gamma = torch.full((1,), 0.5, device=device, requires_grad=True)
    dy = torch.randn(B, C, H, W, generator=g).to(device)

    def one():
        for t in (q, k, v, x, gamma):
            t.grad = None
        reference_formulation(q, k, v, x, gamma).backward(dy)

    return one


def stock_pytorch_core(B, C, H, W, device, iters=10):
    """What stock PyTorch gives on the same device for the same core (``reference_formulation`` -> rocBLAS + elementwise
    kernels, autograd backward).  Reported beside ``value`` with the same algorithmic-byte numerator; it is a baseline,
    never the product path."""
    one = reference_formulation_step(B, C, H, W, device)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    ms = time_region(one, iters)
    return {"ms_per_step": round(ms, 4), "GB/s": round(core_bytes(B, C, H, W) / (ms * 1e-3) / 1e9, 1),
            "what": "torch.bmm / cat / softmax formulation of functions.py:30-49 + autograd on this GPU"}


def module_level_ms(B, C, H, W, device, iters=10, fuse=True):
    """fwd+bwd of the whole CrissCrossAttention module (adds the 1x1 projections + autograd); ``fuse`` False
    runs the three projections as separate convolutions exactly as functions.py:29-35."""
    from ccnet_amd import CrissCrossAttention
    torch.manual_seed(0)
    m = CrissCrossAttention(C).to(device)
    m.fuse_projections = fuse
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device=device, requires_grad=True)
    dy = torch.randn(B, C, H, W, device=device)

    def one():
        m.zero_grad(set_to_none=True)
        x.grad = None
        m(x).backward(dy)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    return time_region(one, iters)


def projection_gemms(lib, B, C, H, W, device, iters=20):
    """The three GEMMs of the stacked 1x1 projections around the core (functions.py:29,32,35 and their adjoints) as the library's
    own MFMA kernels (csrc/cca_gemm.hpp), timed launch by launch on synthetic three-plane bf16 operands of the module's shapes;
    2 * M * N * K = the flops of ONE bf16 product (the split-bf16 x3 structure is inside K), priced against the dense bf16 MFMA peak."""
    from ccnet_amd import functions as F
    hw, ct = H * W, C + 2 * (C // 8)
    g = torch.Generator(device=device).manual_seed(3)
    r16 = lambda *shape: torch.randn(shape, device=device, generator=g).to(torch.bfloat16)
    x3, w3, w3t, d3 = r16(B * hw, 3 * C), r16(ct, 3 * C), r16(C, 3 * ct), r16(B, hw, 3 * ct)
    dy, bias = torch.randn((B, C, hw), device=device, generator=g), torch.randn((ct,), device=device, generator=g)
    legs = {"forward (bias in the accumulators)": (lambda: F._projection_gemm(lib, x3, w3, bias), 2.0 * B * hw * ct * 3 * C),
            "dx (NCHW, dy in the accumulators)": (lambda: F._projection_adjoint_gemm(lib, w3t, d3, dy), 2.0 * B * hw * C * 3 * ct),
            "dW (row contraction in slabs + sum of partials)": (lambda: F._projection_wgrad_gemm(lib, d3.view(B * hw * 3, ct), x3.view(B * hw * 3, C)),
                                                               2.0 * 3 * B * hw * ct * C)}
    out = {"what": "ccnet_cca_projection_bf16 / _adjoint_bf16 / _wgrad_bf16 at the module's shapes, one launch each", "peak_tflops": 2500.0}
    for name, (f, flops) in legs.items():
        if f() is None:
            out[name] = "outside the entry point's contract at this shape (the module uses the stock GEMM)"
            continue
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        ms = time_region(f, iters)
        out[name] = {"us": round(ms * 1e3, 1), "TFLOP/s": round(flops / (ms * 1e-3) / 1e12, 1), "frac_of_mfma_peak": round(flops / (ms * 1e-3) / 2.5e15, 3)}
    return out


def rcca_head_ms(device, batches=(1, 2), iters=5):
    """BASELINE.json configs[2] (networks/ccnet.py:116-123): RCCAModule(2048, 512, 19), recurrence 2, fwd+bwd on a
    (B,2048,97,97) input -- the layer4 output of a 769x769 crop -- B in {1, 2}."""
    from ccnet_amd.segmodel import RCCAModule
    out = {}
    torch.manual_seed(0)
    head = RCCAModule(2048, 512, 19).to(device).train()
    with torch.no_grad():
        head.cca.gamma.fill_(0.5)
    for b in batches:
        x = torch.randn(b, 2048, 97, 97, device=device, requires_grad=True)

        def one():
            head.zero_grad(set_to_none=True)
            x.grad = None
            y = head(x, 2)
            y.backward(torch.ones_like(y))

        for _ in range(2):
            one()
        torch.cuda.synchronize()
        out[f"B{b}_ms"] = round(time_region(one, iters), 3)
    out["what"] = "RCCAModule(2048,512,19) R=2 fwd+bwd, input (B,2048,97,97) fp32 (BASELINE configs[2])"
    return out


def small_batch_ms(lib, C, H, W, device, batches=(1, 2), iters=30):
    """CCNet's own recipe runs 1-2 images per GPU (engine.py:88): the core step at those batches on the NCHW strip kernels
    and on the pixel-major family (module boundary: q | k | v pixel-major, x / y / dy NCHW)."""
    out = {}
    for B in batches:
        wl = CoreWorkload(lib, B, C, H, W, device, 77 + B)
        wl.step()
        out[f"B{B}_ms"] = round(time_region(wl.step, iters), 4)
        del wl
    pl = {}
    for B in tuple(batches):
        wl = PlanesWorkload(lib, B, C, H, W, device, 277 + B)
        wl.step()
        pl[f"B{B}_ms"] = round(time_region(wl.step, iters), 4)
        del wl
    out["split_plane_family"] = pl
    out["what"] = ("fp32 core fwd+bwd, eager; B1_ms / B2_ms = NCHW strip kernels; split_plane_family = ccnet_cca_*_planes_f32 (one "
                   "workgroup per strip; q | k slices of the packed pixel-major projection, v / dy as bf16 hi|lo planes, x / y / dy "
                   "NCHW: the module's default route)")
    torch.cuda.empty_cache()
    return out


def train_imgs_per_s(steps, warmup, batch_per_gpu, bf16=False, size=769):
    """The imgs/s half of the metric (engine.py:85-88 global batch / world; train.py:160-183 step): every rank of the
    job runs ccnet_amd.train_synthetic.run (DDP over the already initialised process group when world > 1)."""
    from ccnet_amd import train_synthetic as TS
    argv = ["--steps", str(steps), "--warmup", str(warmup), "--batch-per-gpu", str(batch_per_gpu),
            "--size", str(size), "--no-destroy-group"] + (["--bf16"] if bf16 else [])
    return TS.run(TS.build_parser().parse_args(argv), quiet=True)


def main(argv=None, workload_factory=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 8 (f32) / 16 (bf16)")
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--height", type=int, default=None, help="default 97 (f32) / 129 (bf16)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"),
                    help="bf16 = BASELINE.json configs[4]: pixel-major bf16 core, default shape (16,512,129,129)")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="all-reduce the module's 7 parameter gradients after every step (engine.py:75)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline/module/cpu/train legs (timed region only)")
    ap.add_argument("--no-train", action="store_true", help="skip the synthetic train-step leg (imgs_per_s)")
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--train-batch", type=int, default=1, help="images per GPU in the train leg (engine.py:88: 8/world)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--prewarm-s", type=float, default=0.5, help="untimed pre-run (seconds) before the warm-up steps")
    ap.add_argument("--family", default="planes", choices=("planes", "strips"),
                    help="f32 core: planes (default) = split-plane path (q | k fp32 pixel-major, v as bf16 hi | lo planes, x / y / dy "
                         "NCHW; what the module runs); strips = NCHW strip kernels (q, k, v NCHW)")
    ap.add_argument("--launch", default="graph", choices=("graph", "eager"),
                    help="graph (default): the step's launches are captured once into a hipGraph and the timed region replays "
                         "it (the task's 'capture launch-bound inner loops in hipGraphs'); eager: two C-ABI calls per step")
    ap.add_argument("--overlap", default="auto", choices=("auto", "0", "1", "2"),
                    help="the library's 'planes_overlap' option: auto (default, what ships) = the dv passes of the backward run on "
                         "the library's side stream; 0 = every launch on one stream (per-launch rocprof durations then match "
                         "the bench line's launch_ms)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = host-logic tests on CPU with an injected workload (the product has no CPU path)")
    ap.add_argument("--workload-factory", default=None,
                    help="tests only: 'module:callable' returning an object with .step() (used with --backend gloo)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        rc, parsed = spawn_ranks(args.gpus, argv, args.backend)
        if rc != 0:
            sys.exit(rc)
        return parsed
    rank, world, local = dist_env()
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; refusing to mislabel the run")
    on_gpu = args.backend == "nccl"
    if on_gpu:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a HIP device (the product has no CPU path)")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"     # (the latter: exercise the RCCL path on 1 GPU)
    if use_dist:
        if on_gpu:
            dist.init_process_group("nccl", device_id=device)     # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    bf16 = args.dtype == "bf16"
    B = args.batch or (16 if bf16 else 8)
    H = args.height or (129 if bf16 else 97)
    W = args.width or H
    C = args.channels
    lib = None
    if workload_factory is None and args.workload_factory:
        mod, attr = args.workload_factory.split(":")
        workload_factory = getattr(__import__(mod, fromlist=[attr]), attr)
    if workload_factory is not None:
        wl = workload_factory(B, C, H, W, device, shard_seed(1234, rank))
    else:
        if not on_gpu:
            sys.exit("bench.py: --backend gloo needs --workload-factory (tests); the product path is HIP-only")
        from ccnet_amd import _lib
        lib = _lib.get_lib()
        lib.set_option("planes_overlap", -1 if args.overlap == "auto" else int(args.overlap))
        use_planes = (not bf16 and args.family == "planes" and max(H, W) <= 132 and C % 8 == 0)
        cls = PixelMajorBF16Workload if bf16 else PlanesWorkload if use_planes else CoreWorkload
        wl = cls(lib, B, C, H, W, device, shard_seed(1234, rank))

    grads = torch.zeros(CCA_PARAM_FLOATS(C), device=device) if args.allreduce_grads else None

    def eager_step():
        wl.step()
        if grads is not None and dist.is_initialized():
            dist.all_reduce(grads)

    # hipGraph capture of one step (same kernels, same arguments, same stream order): the replay removes the host side of
    # the launches from the timed region.  Only for the device library's own workloads, without the gradient all-reduce.
    graph, launch_mode = None, "eager"
    if on_gpu and lib is not None and args.launch == "graph" and grads is None:
        try:
            graph = capture_step_graph(wl.step)
            launch_mode = "hipGraph replay"
        except Exception as e:           # capture is an optimisation of the host side only: report and run eagerly
            launch_mode = f"eager (graph capture failed: {e})"
            graph = None
    step = graph.replay if graph is not None else eager_step

    # clocks ramp with load: a short untimed pre-run settles DVFS before the contract's own W warm-up steps
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_s:
        for _ in range(10):
            step()
        sync()
    for _ in range(args.warmup):
        step()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    local_s = time.perf_counter() - t0
    secs = max_over_ranks(local_s, device, world)
    per_rank = gather_over_ranks({"rank": rank, "ms_per_step": round(local_s / args.steps * 1e3, 4),
                                  **rank_description(device, on_gpu)}, world)

    nbytes = core_bytes(B, C, H, W, elt=2 if bf16 else 4)
    value = aggregate_value(nbytes, args.steps, world, secs)
    ms = secs / args.steps * 1e3
    impl = "injected" if lib is None else ("pixel-major bf16 mfma" if bf16 else
                                           "split-bf16 mfma (three terms; ca_backward six terms = fp32-equivalent): v read as fp32 tiles (no split pass), dy transposed into bf16 hi|lo planes inside the step" if isinstance(wl, PlanesWorkload) else
                                           "mfma-strip" if lib.ccnet_cca_shape_uses_mfma(B, C, H, W) else "direct")
    out = {
        "metric": metric_label(C, H, W),
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[4]: bf16 mixed-precision CrissCrossAttention core fwd+bwd (bf16 "
                                f"pixel-major features, fp32 attention / softmax / accumulate), ({B},{C},{H},{W}) per GPU"
                                if bf16 else
                                f"BASELINE.json configs[1]: single-op CrissCrossAttention core fwd+bwd, "
                                f"({B},{C},{H},{W}) fp32 per GPU, R=1"
                                + ("; inputs q | k | v fp32 (pixel-major channel slices of the packed projection), x / dy NCHW fp32; "
                                   "outputs y NCHW, dq | dk | dv fp32 pixel-major; every pass of the op is inside the timed step"
                                   if isinstance(wl, PlanesWorkload) else "")),
                   "per_gpu_batch": B, "global_batch": B * world, "shape": [B, C, H, W],
                   "parallelism": f"batch-sharded x{world} (no data-path collective"
                                  + (", + all-reduce of the 7 parameter gradients per step)" if grads is not None else ")"),
                   "impl": impl},
        "per_rank": per_rank,
        "collective_library": collective_library(on_gpu),
        "launch": launch_mode,
        "overlap": ("library default: the backward's dv passes on a side stream (event fork / join inside the C call, captured "
                    "into the same graph)" if args.overlap == "auto" else f"planes_overlap={args.overlap}"),
        "algorithmic_bytes_per_step_per_gpu": nbytes,
        "frac_of_hbm_roofline": round(value / world / HBM_PEAK_GBS, 4),
        "frac_of_hbm_copy_ceiling": round(value / world / HBM_COPY_GBS, 4),
        "tflops": round(core_flops(B, C, H, W) * world / (ms * 1e-3) / 1e12, 2),
    }

    extras = on_gpu and lib is not None and not args.no_extras
    if extras and bf16 and rank == 0:
        out["step_ms_stats"] = {"warm": per_step_stats(step, 50)}
        out.update(launch_accounting(lib, wl, graph))
        out["fwd_ms"], out["bwd_ms"] = round(time_region(wl.forward, 10), 4), round(time_region(wl.backward, 10), 4)
        traffic = measured_traffic(lib, "traffic_bf16_latest.json")
        out["roofline"] = {"bound": "hbm", "achieved": round(value / world, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(value / world / HBM_PEAK_GBS, 4),
                           "traffic": traffic.get("_step_total_bytes") if traffic else None,
                           "traffic_source": "profiles/traffic_bf16_latest.json (builder-side PMC passes, source-hash keyed)" if traffic else "none",
                           "note": "op level: algorithmic bytes of the fwd+bwd step / step time; traffic = PMC bytes of "
                                   "the whole step (tools/pmc.sh --script tools/pm_bf16_time.py), null unless taken on "
                                   "this build"}
    extras = extras and not bf16
    if extras and rank == 0:
        out["step_ms_stats"] = {"warm": per_step_stats(step, 100)}
        flush = torch.zeros(128 * 1024 * 1024, device=device)                 # 512 MiB > the 256 MiB Infinity Cache
        out["step_ms_stats"]["cold"] = per_step_stats(step, 30, flush)
        out["step_ms_stats"]["cold"]["flush"] = "512 MiB read-modify-write before every step, outside the event pair"
        del flush
        fwd_ms = time_region(wl.forward, 10)
        bwd_ms = time_region(wl.backward, 10)
        out["fwd_ms"], out["bwd_ms"] = round(fwd_ms, 4), round(bwd_ms, 4)
        out.update(launch_accounting(lib, wl, graph))
        if isinstance(wl, PlanesWorkload):
            # what the six-term (fp32-equivalent) ca_backward products -- option "dqdk_exact" 1, the shipped default -- cost in THIS
            # step: the same step with the three-term form
            prev = lib.set_option("dqdk_exact", 0)
            try:
                for _ in range(3):
                    wl.step()
                out["dqdk_six_terms"] = {"eager_ms_per_step_three_terms": round(time_region(wl.step, 20), 4)}
            finally:
                lib.set_option("dqdk_exact", prev)
            out["dqdk_six_terms"]["cost_ms"] = round(out["eager_ms_per_step"] - out["dqdk_six_terms"]["eager_ms_per_step_three_terms"], 4)
        out["gpu_state_under_load"] = gpu_state_under_load(step, local)
        out["gpu_probe"] = gpu_probe(lib, step, device, ms)
        if isinstance(wl, PlanesWorkload):
            out["roofline"] = planes_roofline(lib, wl, ms, out.get("launch_ms", []))
            out["strips_family"] = strips_family_summary(lib, B, C, H, W, device)
        else:
            roof, rows = roofline_object(wl, ms)
            out["roofline"] = roof
            out["kernels_ms"] = {r["kernel"]: round(r["ms"], 4) for r in rows}
            out["stages_ms"] = {k: round(v, 4) for k, v in roofline_object.stages.items()}
        for key, fn in (("bf16_config5", lambda: bf16_config5(lib, device)),
                        ("small_batch_core_ms", lambda: small_batch_ms(lib, C, H, W, device)),
                        ("rcca_head_R2_2048x97x97", lambda: rcca_head_ms(device)),
                        ("projection_gemms", lambda: projection_gemms(lib, B, C, H, W, device)),
                        ("stock_pytorch_core", lambda: stock_pytorch_core(B, C, H, W, device))):
            try:
                out[key] = fn()
            except Exception as e:          # the metric does not depend on it
                out[key] = f"failed: {e}"
        try:
            out["module_ms_per_step"] = round(module_level_ms(B, C, H, W, device), 4)
            out["module_ms_per_step_unfused_projections"] = round(module_level_ms(B, C, H, W, device, fuse=False), 4)
        except Exception as e:
            out["module_ms_per_step"] = f"failed: {e}"
        if isinstance(out.get("stock_pytorch_core"), dict):
            out["speedup_vs_stock_pytorch"] = round(out["stock_pytorch_core"]["ms_per_step"] / out["ms_per_step"], 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, H, W, args.cpu_budget)
    if extras and not args.no_train:
        # every rank takes part (DDP + SyncBN collectives); rank 0 keeps the result.  Ranks 1.. wait HERE (not inside the train
        # leg's first collective) while rank 0 finishes its rank-0-only extras, and a failure of rank 0 up to this point is
        # agreed on before anybody enters DDP
        if use_dist:
            ok = torch.ones(1, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok) < 1:
                sys.exit("bench.py: a rank failed before the train leg")
        del wl
        torch.cuda.empty_cache()
        try:
            r = train_imgs_per_s(args.train_steps, 2, args.train_batch)
            if rank == 0:
                out["imgs_per_s"] = r["value"]
                out["train_step"] = {k: r[k] for k in ("metric", "ms_per_step", "dtype", "config", "final_loss")}
        except Exception as e:
            if rank == 0:
                out["imgs_per_s"] = None
                out["train_step"] = f"failed: {e}"
    if use_dist:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
