"""The RCCL leg on ONE GPU: the code paths an 8-GPU run takes -- process group over backend "nccl"
(= RCCL on ROCm), the hipGraph-captured step next to a live process group and the library's side stream, the all-reduce of
the module's parameter gradients, DistributedDataParallel around the train step -- exercised with WORLD_SIZE = 1, so that the
driver's first multi-GPU run is not also the first RCCL run.  One subprocess per case (a process group is process state)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, **env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--allreduce-grads"], ["--launch", "eager", "--allreduce-grads"]])
def test_bench_core_step_under_a_live_rccl_process_group(extra):
    """bench.py's timed region with BENCH_FORCE_DIST=1: init_process_group("nccl"), barriers, the max-over-ranks all-reduce of
    the timing, the step as a captured hipGraph (fork / join of the library's side stream inside it) or eager + the gradient
    all-reduce -- the headline metric must come out the same as without a process group."""
    res = _run(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras"] + extra, BENCH_FORCE_DIST="1")
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["unit"] == "GB/s" and 0.2 < res["ms_per_step"] < 5.0, res
    assert ("all-reduce" in res["config"]["parallelism"]) == ("--allreduce-grads" in extra)


@pytest.mark.gpu
def test_train_step_through_ddp_on_one_gpu():
    """ccnet_amd.train_synthetic --force-ddp: ResNet-101 + RCCA (HIP attention) inside DistributedDataParallel over RCCL with
    one rank -- gradient buckets, SyncBN-free forward, optimiser steps (engine.py:52-57,75)."""
    res = _run(["-m", "ccnet_amd.train_synthetic", "--steps", "2", "--warmup", "1", "--size", "257", "--force-ddp"])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["final_loss"] == res["final_loss"], res
