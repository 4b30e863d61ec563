"""World-size-2 checks of the N>1 path on the gloo backend (CPU): the path shards along the batch with no
data-path collective; what has to hold across ranks is (a) batch sharding gives the same
result as the un-sharded op, (b) parameter gradients summed over shards equal the full-batch gradients
(what DDP's all-reduce does around the module), (c) bench.py's max-over-ranks timing and aggregation.
The device op has no CPU path, so the oracle stands in for it here -- this exercises the host logic only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    from oracle import cca_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert bench.dist_env() == (rank, world, rank)
        # identical full problem on every rank (same seed), each rank computes its own batch shard
        torch.manual_seed(0)
        B, C, H, W = 4, 16, 6, 5
        conv = {n: torch.nn.Conv2d(C, o, 1) for n, o in (("query_conv", C // 8), ("key_conv", C // 8), ("value_conv", C))}
        params = {"gamma": torch.full((1,), 0.5)}
        for n, m in conv.items():
            params[n + ".weight"], params[n + ".bias"] = m.weight.detach(), m.bias.detach()
        x, dy = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
        lo, hi = rank * B // world, (rank + 1) * B // world
        y_s, dx_s, g_s = O.cca_module_forward_backward(x[lo:hi], params, dy[lo:hi])
        # (b) sum the parameter gradients over ranks (DDP averages; the sum is the un-sharded gradient)
        for n in sorted(g_s):
            dist.all_reduce(g_s[n], op=dist.ReduceOp.SUM)
        # (a) gather the shards
        ys = [torch.empty_like(y_s) for _ in range(world)]
        dist.all_gather(ys, y_s)
        y_full, dx_full, g_full = O.cca_module_forward_backward(x, params, dy)
        assert torch.allclose(torch.cat(ys), y_full, atol=1e-5)
        assert torch.allclose(dx_s, dx_full[lo:hi], atol=1e-5)
        for n in g_full:
            assert torch.allclose(g_s[n], g_full[n], rtol=1e-4, atol=1e-4), n
        # (c) timing reduction + whole-job aggregation used by bench.py
        t = bench.max_over_ranks(0.5 + rank, torch.device("cpu"), world)
        assert t == pytest.approx(0.5 + (world - 1))
        v = bench.aggregate_value(1_000_000_000, 10, world, t)
        assert v == pytest.approx(world * 10 / t)
        assert bench.shard_seed(1234, rank) == 1234 + rank
        out.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_two_batch_sharding_on_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: "ok", 1: "ok"}, res
