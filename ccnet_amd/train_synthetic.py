"""Synthetic-data DDP training driver for the Cityscapes-shape half of BASELINE.json's metric
("imgs/sec 1/2/4/8 GPU").

It replaces the reference's ``train.py`` + ``engine.py`` (which need apex-era APIs and the Cityscapes files) with
the same training step on random data, one process per GPU:

    model      ccnet_amd.segmodel.Seg_Model(19, CriterionDSN(), recurrence=2)       train.py:160-164
    data       images randn(b,3,769,769), labels randint(0,19) with ~5 % set to 255 train.py:28-33 (crop 769)
    optimiser  SGD(lr 1e-2, momentum 0.9, weight decay 1e-4 ... 5e-4), poly LR      train.py:126-133,183
    parallel   DistributedDataParallel over RCCL (backend "nccl"), SyncBN statistics engine.py:52-57,75
               per-rank seed = rank                                                  train.py:154-155

Launch:   python -m ccnet_amd.train_synthetic --steps 10 --batch-per-gpu 1
          python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                 -m ccnet_amd.train_synthetic --steps 10 --batch-per-gpu 1
Rank 0 prints one JSON line with images/s over the timed steps (max over ranks, barrier + synchronize on both
sides) -- the optimiser step and the gradient all-reduce are inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import torch.distributed as dist


def lr_poly(base_lr, it, max_it, power=0.9):
    """train.py:126-127"""
    return base_lr * (1.0 - float(it) / max_it) ** power


def synthetic_batch(batch, size, num_classes, device, generator):
    images = torch.randn(batch, 3, size, size, device=device, generator=generator)
    labels = torch.randint(0, num_classes, (batch, size, size), device=device, generator=generator)
    ignore = torch.rand(batch, size, size, device=device, generator=generator) < 0.05
    return images, labels.masked_fill(ignore, 255)


def run(args, model_factory=None, quiet=False):
    """One training job on the calling rank; returns the result dict (rank 0) or None.  ``quiet`` suppresses the
    JSON line (bench.py embeds the result in its own line)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available() and not args.cpu
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        # train.py:152 sets cudnn.benchmark unconditionally.  On ROCm that makes MIOpen search its solvers per convolution
        # configuration at first use: for ResNet-101 at 769 x 769 the search alone ran past a 10-minute limit on an MI355X box with
        # an empty MIOpen cache (measured, round 4) -- opt-in only.  Tri-state: without the option the process-wide flag is left as
        # the embedding process set it; --cudnn-benchmark / --no-cudnn-benchmark set it.
        if getattr(args, "cudnn_benchmark", None) is not None:
            torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    ddp = world > 1 or getattr(args, "force_ddp", False)      # (--force-ddp: the RCCL / DDP path on a single GPU)
    if ddp and not dist.is_initialized():
        dist.init_process_group("nccl" if use_cuda else "gloo")
    torch.manual_seed(args.seed + rank)                       # per-rank seed (train.py:154-155)

    if model_factory is None:
        from .segmodel import CriterionDSN, Seg_Model
        model = Seg_Model(args.num_classes, criterion=CriterionDSN(), recurrence=args.recurrence)
    else:
        model = model_factory()
    model = model.to(device).train()
    net = model
    if ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local] if use_cuda else None,
                                                        broadcast_buffers=False)
    opt = torch.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=args.weight_decay)
    gen = torch.Generator(device=device)
    gen.manual_seed(args.seed + rank)
    total = args.warmup + args.steps

    def step(it):
        for g in opt.param_groups:
            g["lr"] = lr_poly(args.lr, it, max(total, 1))
        images, labels = synthetic_batch(args.batch_per_gpu, args.size, args.num_classes, device, gen)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16 and use_cuda):
            loss = net(images, labels)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if ddp:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    loss = None
    for it in range(args.warmup):
        loss = step(it)
    fence()
    t0 = time.perf_counter()
    for it in range(args.warmup, total):
        loss = step(it)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if ddp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    result = None
    if rank == 0:
        result = {
            "metric": "CCNet (ResNet-101 + RCCA R=%d) synthetic train step, images/s" % args.recurrence,
            "value": round(args.steps * args.batch_per_gpu * world / elapsed, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 2), "scaling": "weak",
            "dtype": "bf16-autocast" if args.bf16 else "f32", "data": "synthetic",
            "config": {"image": [3, args.size, args.size], "per_gpu_batch": args.batch_per_gpu,
                       "global_batch": args.batch_per_gpu * world, "parallelism": f"ddp{world}+syncbn",
                       "optimizer": "sgd+poly"},
            "final_loss": round(float(loss.detach().float().item()), 4) if loss is not None else None,
        }
        if not quiet:
            print(json.dumps(result), flush=True)
    if ddp and args.destroy_group:
        dist.destroy_process_group()
    return result


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--size", type=int, default=769, help="square crop (train.py: 769)")
    ap.add_argument("--num-classes", type=int, default=19)
    ap.add_argument("--recurrence", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--weight-decay", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bf16", action="store_true", help="autocast to bf16: convolutions in bf16, the attention core on the pixel-major bf16 kernels (fp32 attention / softmax / accumulate)")
    ap.add_argument("--cudnn-benchmark", dest="cudnn_benchmark", action="store_true", default=None,
                    help="set torch.backends.cudnn.benchmark as train.py:152 does (MIOpen then searches its solvers per convolution "
                         "configuration at first use: minutes with an empty cache); default: leave the process-wide flag untouched")
    ap.add_argument("--no-cudnn-benchmark", dest="cudnn_benchmark", action="store_false")
    ap.add_argument("--cpu", action="store_true", help="tests only: gloo on CPU with an injected model")
    ap.add_argument("--no-destroy-group", dest="destroy_group", action="store_false")
    ap.add_argument("--force-ddp", action="store_true", help="wrap the model in DistributedDataParallel (RCCL process group) even at "
                                                             "WORLD_SIZE = 1: the multi-GPU code path on the one GPU a test box has")
    return ap


if __name__ == "__main__":
    run(build_parser().parse_args())
