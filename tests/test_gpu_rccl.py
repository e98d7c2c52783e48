"""RCCL on the GPU box with the one GPU a lease has: torch.distributed backend "nccl" (= RCCL on ROCm), world size 1.
Proves what the 2-rank gloo tests cannot: the library loads, the collectives of magicpig_amd/sharding.py run on DEVICE
tensors with the dtypes they use (uint8 views of bf16, f64 MAX), in stream order with the kernels around them, and
bench.py takes its distributed path under the launch contract.  No N > 1 curve exists for this repository: an 8-GPU node
was never available to a round.  Reference precedent: evaluations/RULER/pred/attnserver_dist.py:252-254, 279."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from magicpig_amd import sharding
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
out = {}
# one-time hyperplane broadcast (attnserver_dist.py:279): bf16 as raw bytes, on the device, behind a kernel on the stream
W = torch.randn((128, 1500), device=dev).to(torch.bfloat16)
W2 = (W.float() * 2).to(torch.bfloat16)                 # produced by a kernel just before the collective
got = sharding.sync_hash_func(W2, src=0)
assert got.dtype == torch.bfloat16 and got.is_cuda and torch.equal(got, (W.float() * 2).to(torch.bfloat16))
out["broadcast_bytes"] = got.numel() * 2
# outputs gathered at the edge, both layouts (uint8 all_gather of bf16 rows)
for mode, (B, H, Hkv) in (("batch", (3, 8, 2)), ("head", (1, 64, 8))):
    shard = sharding.partition(B, H, Hkv, 1, 0, mode)
    local = torch.randn((shard.local_batch, shard.local_heads, 128), device=dev).to(torch.bfloat16)
    full = sharding.gather_outputs(local.clone(), shard, B, H)
    assert tuple(full.shape) == (B, H, 128) and torch.equal(full, local)
    out["gather_" + mode] = list(full.shape)
# the bench's timing reduction: f64 MAX on the device
t = sharding.max_over_ranks(0.123456789012345, device=dev)
assert t == 0.123456789012345
dist.barrier()
torch.cuda.synchronize()
out["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else "?"
dist.destroy_process_group()
print("RCCL_WORKER_JSON " + json.dumps(out))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
                WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_rccl_world_size_one_drives_the_sharding_collectives():
    r = subprocess.run([sys.executable, "-c", _WORKER, ROOT], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RCCL_WORKER_JSON ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[0][len("RCCL_WORKER_JSON "):])
    assert out["gather_batch"] == [3, 8, 128] and out["gather_head"] == [1, 64, 128]


@pytest.mark.parametrize("shard", ["batch", "head"])
def test_bench_takes_the_distributed_path_with_one_rank(shard):
    """bench.py under the launch contract (RANK / WORLD_SIZE / MASTER_* in the environment) with one rank: process group
    on RCCL, hyperplane broadcast, barrier-bracketed timing, max over ranks, (head shard) all_gather at the edge."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "cfg0", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline", "--shard", shard]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["config"]["process_group"] == "nccl x1" and line["n_gpus"] == 1 and line["value"] > 0
    if shard == "head":
        assert line["head_shard_gather"]["shape"] == [1, 1, 128]


def test_default_line_carries_a_second_configuration_and_the_footprint():
    """VERDICT r04 item 2 on the GPU box: the default single-GPU line (cfg 1) holds another CONFIGURATION as a leg (here
    cfg 4's per-GPU share only, short CPU samples: ~40 s) with its roofline, its HBM footprint and the full-size first
    layer matched against the CPU path; the headline's footprint per layer is what the shapes say; nothing failed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-clustered-leg", "--no-host-mode",
           "--cpu-steps", "64", "--leg-cpu-steps", "32", "--legs", "cfg4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "cfg1" in d["config"]["workload"] and "failed_legs" not in d
    f = d["observed"]["hbm_bytes_per_layer"]
    assert f["kv"] == 8 * 98304 * 2 * 128 * 2 and f["table"] == 8 * 150 * 98304 * 4 and f["total"] == sum(f[k] for k in ("kv", "key_norms", "bounds", "table", "slots"))
    leg = d["legs"]["cfg4_share"]
    assert leg["ranges_per_head"] == 16 and leg["tokens_per_s"] > 0 and 0 < leg["roofline"]["frac"] < 1
    assert leg["cpu_baseline"]["gpu_matches"]["nnz_equal"] is True and leg["cpu_baseline"]["gpu_matches"]["max_abs_out_diff"] <= 1e-2
    assert d["cpu_baseline"]["gpu_matches"]["nnz_equal"] is True
