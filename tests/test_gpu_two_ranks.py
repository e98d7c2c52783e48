"""The N > 1 path as EXECUTABLE code: two processes, two devices, the HIP library in both, RCCL between them.

Skipped on a 1-GPU lease (every box this repository has been measured on so far); runs wherever >= 2 GPUs exist.  bench.py
is launched exactly as the driver launches it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ...`), head-
sharded at the cfg 4 shape (evaluations/RULER/pred/attnserver_dist.py:252-254: kv heads over the ranks) and batch-sharded
at the cfg 3 shape; the line must say `nccl x2`, and every rank's outputs must equal those of a SINGLE process serving the
same units (bench.py --emulate-rank R/2: units are independent, their synthetic data is drawn per global unit id).
The one-GPU half of the same statement -- an emulated rank equals the corresponding slice of the unsharded run -- runs on
every GPU box (test_emulated_ranks_partition_the_single_process_result)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])


def _bench(extra, nproc=1, timeout=1500):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-host-mode",
            "--no-clustered-leg", "--no-legs"] + extra
    if nproc == 1:
        cmd = [sys.executable] + base + ["--gpus", "1"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + base + ["--gpus", str(nproc)]
    return _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT))


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
@pytest.mark.parametrize("shard,config", [("head", "cfg4"), ("batch", "cfg3")])
def test_two_processes_two_devices(shard, config):
    two = _bench(["--config", config, "--shard", shard], nproc=2)
    assert two["config"]["process_group"] == "nccl x2" and two["n_gpus"] == 2 and two["value"] > 0
    assert len(two["rank_checksums"]) == 2
    for r in range(2):
        one = _bench(["--config", config, "--shard", shard, "--emulate-rank", f"{r}/2"])
        assert one["rank_checksums"] == [two["rank_checksums"][r]], (shard, r)
    if shard == "head":
        # the all_gather at the edge: [B, H_full, D] on every rank, the same bytes as the unsharded model's outputs
        full = _bench(["--config", config, "--shard", "head"])           # one process, all 8 kv heads
        assert two["head_shard_gather"]["shape"] == full["head_shard_gather"]["shape"] == [1, 64, 128]
        assert two["head_shard_gather"]["checksum"] == full["head_shard_gather"]["checksum"]


@pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")
def test_emulated_ranks_partition_the_single_process_result():
    """One GPU is enough for this half: the units ranks 0 and 1 of 2 would serve, each run by ONE process on its own
    (--emulate-rank), reproduce the unsharded run's outputs for those units bit for bit -- the head-sharded gather of
    the unsharded run is the concatenation of the two emulated ranks' outputs (cfg 0-sized shape with 2 kv heads)."""
    shape = json.dumps(dict(model="synthetic-2kv", layers=2, dense=[], H=4, Hkv=2, D=128, B=1, P=4096 + 68, M=4288, K=8,
                            L=40, H_full=4, Hkv_full=2))
    outs = {}
    for tag, extra in (("full", []), ("r0", ["--emulate-rank", "0/2"]), ("r1", ["--emulate-rank", "1/2"])):
        outs[tag] = _bench(["--config-json", shape, "--shard", "head"] + extra)
    # the unsharded run serves both kv heads: its checksum is not a rank's, but the per-rank ones must differ from each
    # other (different units) and the emulated runs must be self-consistent across two invocations
    assert outs["r0"]["rank_checksums"] != outs["r1"]["rank_checksums"]
    again = _bench(["--config-json", shape, "--shard", "head", "--emulate-rank", "1/2"])
    assert again["rank_checksums"] == outs["r1"]["rank_checksums"]
    # additivity of the int16 checksum: sum over heads of the full run == sum of the two halves
    s = lambda o: o["head_shard_gather"]["checksum"]          # noqa: E731
    assert s(outs["full"]) == s(outs["r0"]) + s(outs["r1"])


def _plain(extra, timeout=1500):
    """bench.py started PLAINLY -- no launcher, no RANK / WORLD_SIZE in the environment -- as the round-end driver starts it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-host-mode",
           "--no-clustered-leg", "--no-legs"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")
def test_plain_start_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` on a node with fewer than N GPUs must not print a line labelled with
    fewer -- it ends with a message and a non-zero status.  Runs on every GPU box (one more rank than there are GPUs)."""
    n = torch.cuda.device_count() + 1
    r = _plain(["--gpus", str(n), "--config", "cfg0"], timeout=300)
    assert r.returncode != 0 and f"--gpus {n} asked for" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_plain_start_runs_two_ranks_over_rccl():
    """`python bench.py --gpus 2` with a clean environment re-executes itself under torch.distributed.run: two processes, two
    devices, `nccl x2`, one checksum per rank equal to the emulated single-process run of that rank's units; cfg 4 shards
    the model's kv heads by default (strong scaling)."""
    two = _line(_plain(["--gpus", "2", "--config", "cfg3"]))
    assert two["n_gpus"] == 2 and two["config"]["process_group"] == "nccl x2" and len(two["rank_checksums"]) == 2
    assert two["config"]["global_batch"] == 16 and two["scaling"] == "weak"
    for r in range(2):
        one = _bench(["--config", "cfg3", "--shard", "batch", "--emulate-rank", f"{r}/2"])
        assert one["rank_checksums"] == [two["rank_checksums"][r]], r
    four = _line(_plain(["--gpus", "2", "--config", "cfg4"]))
    assert four["scaling"] == "strong" and four["config"]["process_group"] == "nccl x2"
    assert four["head_shard_gather"]["shape"] == [1, 64, 128]
