"""N > 1 path on CPU: two gloo processes shard the units (batch- and head-sharded), each computes
its shard with the ORACLE standing in for the HIP library (tests only), and the gathered result
must equal the unsharded oracle result on every rank.  Also covers the partition arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp_

import cases
import oracle
import synth
from magicpig_amd import sharding


def test_partition_arithmetic():
    s = sharding.partition(64, 32, 8, 8, 3, "batch")          # BASELINE cfg 3
    assert list(s.requests) == list(range(24, 32)) and s.local_kv_heads == 8 and s.local_heads == 32
    s = sharding.partition(1, 64, 8, 8, 5, "head")            # BASELINE cfg 4: 70B TP=8
    assert list(s.kv_heads) == [5] and list(s.heads) == list(range(40, 48)) and s.local_batch == 1
    # ragged batch: 10 requests over 4 ranks -> 3, 3, 2, 2 and every request owned exactly once
    owned = [r for k in range(4) for r in sharding.partition(10, 8, 2, 4, k).requests]
    assert owned == list(range(10))
    assert [sharding.partition(10, 8, 2, 4, k).local_batch for k in range(4)] == [3, 3, 2, 2]
    with pytest.raises(ValueError):
        sharding.partition(1, 32, 8, 3, 0, "head")            # 8 kv heads do not split over 3 ranks
    with pytest.raises(ValueError):
        sharding.partition(1, 32, 8, 2, 2, "batch")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


B, H, HKV, N, M, D, K, L = 3, 8, 2, 384, 448, 128, 8, 40


def _oracle_layer(keys, kns, vals, W, qb, b_ids, kv_ids):
    """Oracle decode of one layer restricted to requests b_ids x kv heads kv_ids."""
    G = H // HKV
    Bl, Hkvl = len(b_ids), len(kv_ids)
    Hl = Hkvl * G
    heads = [g * G + i for g in kv_ids for i in range(G)]
    q = np.stack([qb.reshape(B, H, D)[b][heads] for b in b_ids]).reshape(Bl * Hl, D)
    qcodes, qn = oracle.simhash_query(q, W, K, L)
    lsh = oracle.LSH()
    lsh.alloc(K, L, 1, Hl, Hkvl, Bl, M)
    srv = oracle.SparseAttentionServer(exp_mode=2, clamp_cos=1)
    srv.alloc(1, Hl, Hkvl, D, Bl, M)
    for i, b in enumerate(b_ids):
        kk = np.ascontiguousarray(keys[b][list(kv_ids)])
        sc, si = cases.stable_sort_codes(oracle.simhash_keys(kk, W, K, L))
        lsh.fill(0, i, sc, si)
        srv.fill(0, i, kk, np.ascontiguousarray(vals[b][list(kv_ids)]), np.ascontiguousarray(kns[b][list(kv_ids)]))
    res = np.zeros((Bl * Hl, M), np.int32)
    nnz = np.zeros((Bl * Hl,), np.int32)
    lsh.batch_retrieve(0, qcodes, res, nnz)
    ind = np.zeros_like(res)
    for h in range(Bl * Hl):
        ind[h, :nnz[h]] = np.sort(res[h, :nnz[h]])
    out = np.zeros((Bl * Hl, D), np.uint16)
    mve = np.zeros((2, Bl * Hl), np.float32)
    srv.attention_wrapper(0, K, L, out, mve, q, qn, ind, nnz)
    return out.reshape(Bl, Hl, D)


def _worker(rank, world, port, mode, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        keys, kns, vals, W, qb = cases.case_inputs(77, B, H, HKV, N, D, K, L)
        # rank 0 owns the hyperplanes; the others start from garbage and must receive them
        Wt = synth.to_torch_bf16(W if rank == 0 else np.zeros_like(W))
        Wt = sharding.sync_hash_func(Wt, src=0)
        Wl = Wt.view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(Wl, W)
        shard = sharding.partition(B, H, HKV, world, rank, mode)
        local = _oracle_layer(keys, kns, vals, Wl, qb, list(shard.requests), list(shard.kv_heads))
        full = sharding.gather_outputs(synth.to_torch_bf16(local), shard, B, H)
        t = sharding.max_over_ranks(1.0 + rank)
        ret[rank] = (full.view(torch.int16).numpy().copy(), t)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["batch", "head"])
def test_two_rank_sharded_decode_equals_unsharded(mode):
    world = 2
    port = _free_port()
    with mp_.Manager() as mgr:
        ret = mgr.dict()
        mp_.spawn(_worker, args=(world, port, mode, ret), nprocs=world, join=True)
        keys, kns, vals, W, qb = cases.case_inputs(77, B, H, HKV, N, D, K, L)
        ref = _oracle_layer(keys, kns, vals, W, qb, list(range(B)), list(range(HKV)))
        for r in range(world):
            got, t = ret[r]
            assert np.array_equal(got.view(np.uint16), ref), (mode, r)     # bit-identical: units are independent
            assert t == 2.0                                                 # max over ranks


def test_bench_launch_contract_two_ranks_dry_run():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2` (the driver's N > 1 launch)
    with --dry-run: process group from the environment, hyperplanes broadcast from rank 0, barrier +
    max-over-ranks timing, exactly ONE JSON line (rank 0), n_gpus = 2 and the slowest rank's time."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "2", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 2
    assert d["ms_per_step"] >= 1.9            # rank 1 sleeps 2 ms per step: max over ranks
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1",
                             "--dry-run"], capture_output=True, text=True, timeout=300, cwd=root)
    assert single.returncode == 0, single.stderr[-2000:]
    d1 = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["planes_checksum"] == d["planes_checksum"]   # rank 0's planes everywhere


def test_bench_head_shard_two_ranks_dry_run():
    """bench.py --shard head under torch.distributed.run with 2 ranks (gloo, --dry-run): the kv heads of the whole
    model (cfg 4: H = 64, Hkv = 8) are partitioned as evaluations/RULER/pred/attnserver_dist.py:252-254 does, each
    rank's outputs are all_gathered at the edge, and the gathered tensor holds every head exactly once."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--config", "cfg4", "--shard", "head"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["heads_per_rank"] == 32 and d["gathered_shape"] == [1, 64, 128]
    assert d["gathered_checksum"] == float(sum(range(64)) * 128)          # every global head once, in place


def _clean_env():
    import os

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_starts_its_own_ranks_from_a_plain_start():
    """`python bench.py --gpus 2` started PLAINLY (no launcher, no RANK / WORLD_SIZE in the
    environment) runs two ranks -- it re-executes itself under torch.distributed.run on 127.0.0.1 -- and rank 0's line
    says so: n_gpus 2, a process group of two, one checksum per rank.  cfg 4 on more than one rank shards the model's kv
    heads by default (evaluations/RULER/pred/attnserver_dist.py:252-254)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1",
                        "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["process_group"] == "gloo x2" and d["rank_checksums"] == [1000, 1001]
    assert d["config"]["shard"] == "batch" and d["ms_per_step"] >= 1.9
    r4 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--dry-run", "--config", "cfg4"], env=_clean_env(), capture_output=True, text=True, timeout=300,
                        cwd=root)
    assert r4.returncode == 0, r4.stderr[-2000:]
    d4 = json.loads([ln for ln in r4.stdout.splitlines() if ln.startswith("{")][0])
    assert d4["config"]["shard"] == "head" and d4["scaling"] == "strong" and d4["heads_per_rank"] == 32


def test_bench_refuses_a_world_that_is_not_what_was_asked_for():
    """No silent 1-GPU run under another label: more ranks asked for than GPUs visible (this container has none) ends
    with a message and a non-zero status before anything is launched; so does a WORLD_SIZE that disagrees with --gpus."""
    import os
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                           env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE = 1" in r.stderr and not r.stdout.strip()
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE = 2" in r.stderr
