"""bench.py -- decode throughput of the LSH-sampled sparse attention path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1]

A "step" is one decode token of the hot path: for every sparse layer of the model shape
(30 of Llama-3.1-8B's 32), q SimHash -> L table probes + collision-count dedupe -> gathered
sparse KV attention with importance-sampling correction (models/attnserver.py:264-300), on
synthetic Q/K/V already resident in HBM.  Loop shape as examples/bench.py:47-56 (32 warm-up +
128 timed steps by default).

One process per GPU.  `--gpus N` with N > 1 runs N ranks whichever way it is started: under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE / MASTER_* in the
environment) it IS one of the ranks; started plainly (`python bench.py --gpus N`, no WORLD_SIZE) it re-executes
itself under torch.distributed.run on 127.0.0.1 (evaluations/RULER/run_tensor_parallel.sh:84 starts the reference's
TP variant the same way).  A world size that is not --gpus, or fewer visible GPUs than --gpus, is an error, never a
silent 1-GPU run.  Ranks own disjoint units -- requests (cfg 1-3: --shard batch, weak scaling) or the model's kv
heads (cfg 4: --shard head, attnserver_dist.py:252-254, strong scaling); no collective inside the path, RCCL only for
the hyperplane broadcast, the barrier, the max-over-ranks time and the checksums.

Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline      the dominant kernel (lsh_decode_kernel, one launch per sparse layer): algorithmic
                bytes per launch / average launch duration from HIP events on the launch stream;
  cpu_baseline  the reference's own AVX512 path (oracle/_ref, kind "reference") or the oracle
                port (kind "port") timed on this box's host cores on a bounded sample;
  legs          (default single-GPU run of cfg 1) the same measurement, shorter, at cfg 2 and at cfg 4's per-GPU
                share -- tokens/s, us per layer, roofline, the full-size first layer against the CPU path -- so that
                those configurations are observed by whoever runs this file, not only claimed in profiles/.
A leg that fails is recorded in the line AND makes the exit status non-zero (unless --allow-missing-legs).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs: the model shape fixes layers/H/Hkv/D; P is the prompt
# length, n = P - 68 offloaded tokens, M the allocated rows.
CONFIGS = {
    # plumbing config: library/lsh + library/sparse_attention, 1 head, seq 4096
    "cfg0": dict(model="synthetic-1head", layers=1, dense=(), H=1, Hkv=1, D=128, B=1, P=4096 + 68,
                 M=4288, K=10, L=150),
    "cfg1": dict(model="Llama-3.1-8B", layers=32, dense=(0, 16), H=32, Hkv=8, D=128, B=1, P=98000,
                 M=98304, K=10, L=150),
    "cfg2": dict(model="Llama-3.1-8B", layers=32, dense=(0, 16), H=32, Hkv=8, D=128, B=8, P=32768,
                 M=32960, K=10, L=170),
    # per-GPU share of cfg 3 (B=64 over 8 GPUs) -- the N-GPU weak-scaling unit
    "cfg3": dict(model="Llama-3.1-8B", layers=32, dense=(0, 16), H=32, Hkv=8, D=128, B=8, P=32768,
                 M=32960, K=10, L=150),
    # cfg 4 (Llama-3.1-70B, H = 64, Hkv = 8, TP = 8).  Default (--shard batch): the per-GPU share of the TP = 8
    # layout, 1 kv head + 8 query heads, as an independent replica per rank.  --shard head: the WHOLE model's heads
    # are partitioned over the ranks (sharding.partition(mode="head"): Hkv_full // world kv heads each, exactly
    # attnserver_dist.py:252-254) -- total work fixed, strong scaling, outputs all_gathered for a checksum.
    "cfg4": dict(model="Llama-3.1-70B/TP8-shard", layers=80, dense=(0, 16, 32, 48, 64), H=8, Hkv=1,
                 D=128, B=1, P=131072, M=131264, K=11, L=300, H_full=64, Hkv_full=8),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

# --data: what the synthetic keys look like.  randn = isotropic Gaussian keys, Gaussian queries: the
# most uniform buckets SimHash can see, 1.56 % selected at cfg 1.  clustered / skewed = keys as a KV cache holds them
# after RoPE -- per-dimension offset (removed by the centring), anisotropic spectrum s_d ~ (1 + d)^-alpha, a mixture
# of cluster centres with unequal populations, a low-rank component shared by all tokens -- with heavy-hitter queries
# 0.5 q + 3 k_j (tests/synth.py: clustered_raw_bits is the same generator in numpy; scripts/tune_clustered.py tuned
# `clustered` to the README's ~2 % sampling rate at K10 L150, /root/reference/README.md:43; `skewed` is a stress
# case: ~8 % selected, 3.7 % of the probed (table, bucket, range) pieces longer than 126 ids).
DATA = {
    "randn": None,
    "clustered": dict(alpha=0.2, a=0.5, b=0.2, clusters=64, rank=4),
    "skewed": dict(alpha=0.5, a=1.0, b=0.5, clusters=64, rank=4),
}


def synth_kv(data, P, Hkv, D, dev, gen):
    """One request's token-major KV cache, bf16 [P, Hkv, D] x 2, drawn on the device."""
    vc = None
    if DATA[data] is None:
        kc = torch.randn((P, Hkv, D), device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        vc = torch.randn((P, Hkv, D), device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        return kc, vc
    p = DATA[data]
    f32 = dict(device=dev, dtype=torch.float32, generator=gen)
    sd = (1.0 + torch.arange(D, device=dev, dtype=torch.float32)) ** (-p["alpha"])
    sd = sd / sd.square().mean().sqrt()
    raw = torch.randn((P, Hkv, D), **f32) * sd
    C = torch.randn((Hkv, p["clusters"], D), **f32) * sd
    cid = torch.minimum(torch.randint(0, p["clusters"], (P, Hkv), device=dev, generator=gen),
                        torch.randint(0, p["clusters"], (P, Hkv), device=dev, generator=gen))
    raw += p["a"] * C[torch.arange(Hkv, device=dev)[None, :], cid]
    U = torch.randn((Hkv, p["rank"], D), **f32) * sd
    w = torch.randn((P, Hkv, p["rank"]), **f32)
    raw += p["b"] * torch.einsum("phr,hrd->phd", w, U)
    raw *= 1.0 / (1.0 + p["a"] ** 2 + p["rank"] * p["b"] ** 2) ** 0.5
    raw += 0.75 * torch.randn((1, Hkv, D), **f32)
    vc = torch.randn((P, Hkv, D), **f32).to(torch.bfloat16)
    return raw.to(torch.bfloat16), vc


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="cfg1", choices=sorted(CONFIGS) + ["custom"])
    ap.add_argument("--data", default="randn", choices=sorted(DATA),
                    help="synthetic key distribution: randn (isotropic), clustered (anisotropic clusters + low-rank "
                         "component, ~2 %% selected at cfg 1), skewed (stress)")
    ap.add_argument("--queries", default="auto", choices=["auto", "randn", "heavy"],
                    help="randn queries or heavy hitters 0.5 q + 3 k_j (SURVEY.md 8(d)); auto: randn with --data randn, "
                         "heavy otherwise")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-mode", action="store_true", help="skip the untimed host-buffer (unchanged caller) leg")
    ap.add_argument("--no-clustered-leg", action="store_true",
                    help="skip the second workload of the default run (clustered keys + heavy-hitter queries, the README's "
                         "~2 %% sampling rate), reported as value_clustered next to the headline value")
    ap.add_argument("--cpu-steps", type=int, default=8192,
                    help="decode steps of ONE sparse layer timed on the host cores per thread placement (medians): "
                         "~5 s of CPU work at cfg 1, ~20 s at cfg 2")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)   # tests: control flow on CPU / gloo
    ap.add_argument("--config-json", default=None,
                    help="tests: a workload shape as JSON (the keys of a CONFIGS entry), used as --config custom")
    ap.add_argument("--emulate-rank", default=None,
                    help="R/W: ONE process that serves the units rank R of W would own (no process group): what the "
                         "multi-GPU tests compare a real rank's outputs with")
    ap.add_argument("--ab-option", default="", help="NAME=v0,v1[,..]: A/B of a library debug option on one workload (alternating "
                    "timed regions of --steps graph replays each); prints its own JSON line instead of the bench line")
    ap.add_argument("--ab-reps", type=int, default=5)
    ap.add_argument("--ab-worker", action="store_true", help="one side of scripts/ab_libs.py (two builds of the library, one "
                    "process each, timed in alternating regions on one GPU): waits for go / quit lines on stdin")
    ap.add_argument("--distinct-layers", type=int, default=0,
                    help="DIAGNOSTIC (not a valid bench line): the step's launches cycle over only this many of the "
                         "model's sparse layers -- the HBM footprint a step touches shrinks (address translation, "
                         "MALL) while launch count and shapes stay")
    ap.add_argument("--table-build", default="counting", choices=["sort", "counting"])
    ap.add_argument("--shard", default=None, choices=["batch", "head"],
                    help="batch: every rank serves its own B requests (weak scaling); head: the kv heads of the "
                         "whole model are partitioned over the ranks like the reference's TP variant (strong scaling).  "
                         "Default: head for cfg4 on more than one GPU (the configuration IS the TP = 8 layout), batch otherwise")
    ap.add_argument("--legs", default="cfg2,cfg3,cfg4",
                    help="extra configurations measured after the headline in the default single-GPU cfg1 run and reported "
                         "under `legs` (cfg3 / cfg4 = their per-GPU shares, as --config cfg3 / cfg4 on one GPU; `e2e` = the "
                         "end-to-end decode step with its split, see --no-e2e-leg); '' or --no-legs: none")
    ap.add_argument("--no-e2e-leg", action="store_true",
                    help="skip legs.e2e: the whole decode step (examples/bench.py:43-59) with random Llama-3.1-8B weights at "
                         "cfg 1 (B = 1) and cfg 2 (B = 8), each with its split by kind of work")
    ap.add_argument("--e2e-steps", type=int, default=24, help="timed steps of an e2e leg (+ 4 warm-up)")
    ap.add_argument("--no-legs", action="store_true")
    ap.add_argument("--leg-cpu-steps", type=int, default=1024,
                    help="decode steps of one sparse layer the CPU path is timed on in a leg (one thread placement)")
    ap.add_argument("--allow-missing-legs", action="store_true",
                    help="a failing host-mode / CPU-baseline / clustered / config leg is recorded in the line but does "
                         "not turn the exit status non-zero")
    ap.add_argument("--lib", default=None,
                    help="A/B: path of an alternative build of libmagicpig_hip.so (the product reads no environment)")
    ap.add_argument("--cluster", type=int, default=0,
                    help="A/B: workgroups per query head of the decode kernel (1, 2, 4, 8, 16, 32; default: by B*H, CUs and "
                         "the tokens per member)")
    ap.add_argument("--no-direct-slots", action="store_true", help="A/B: sub-bounds + ids instead of direct piece slots")
    ap.add_argument("--split-hash", type=int, default=-1, choices=[-1, 0, 1],
                    help="A/B: hyperplanes split over the workgroups of a head's cluster (0 never, 1 always; default: auto)")
    ap.add_argument("--direct-slots", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="A/B: direct piece slots: 0 never, 1 always (where R > 1); default: auto")
    ap.add_argument("--slot-log2", type=int, default=0, choices=[0, 3, 4, 5],
                    help="A/B: width of the direct slots forced to 32 / 64 / 128 bytes (default: by the mean piece length)")
    ap.add_argument("--kn-payload", type=int, default=-1, choices=[-1, 0, 1],
                    help="A/B: key norms as a payload of the table entries (default: the library's choice)")
    ap.add_argument("--mfma-hash", action="store_true",
                    help="A/B: query SimHash by the MFMA kernel as its own launch, then the decode kernel")
    ap.add_argument("--by-products", type=int, default=0, choices=[0, 1],
                    help="0 (default): the timed decode is mp_decode_sparse_layer_ex with MP_DECODE_NO_BYPRODUCTS -- output, LSE "
                         "and counts only, what a serving loop reads; 1: mp_decode_sparse_layer as in rounds 1-5 (query codes, "
                         "result rows and logits written for get_mask / get_score).  The line says which form it timed")
    ap.add_argument("--accel-budget", type=float, default=None,
                    help="GB of HBM the LSH handle may spend on accelerator structures over all layers (direct piece slots, the "
                         "host-buffer mode's row copy): mp_lsh_alloc_ex.  Default: the library's rule (a third of what is free)")
    ap.add_argument("--ranges", type=int, default=0, help="token ranges per table row = workgroups per head (mp_lsh_alloc_ex; 0 = auto)")
    ap.add_argument("--quad-hash", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="A/B: the quad MFMA query hash at one workgroup per head (cfg 2 / 3): 0 never, 1 on, 2 on but nobody publishes")
    ap.add_argument("--two-launch", action="store_true",
                    help="A/B: the decode entry as (hash + retrieve) then attention instead of one launch")
    ap.add_argument("--end-to-end", action="store_true",
                    help="examples/bench.py-style full decode step with synthetic weights (SURVEY 8f-3) "
                         "instead of the hot path alone")
    args = ap.parse_args()
    if args.no_legs:
        args.legs = ""
    if args.config_json:
        CONFIGS["custom"] = dict(json.loads(args.config_json))
        CONFIGS["custom"]["dense"] = tuple(CONFIGS["custom"].get("dense", ()))
        args.config = "custom"
    return args


# ---------------------------------------------------------------------------- CPU baseline worker

def cpu_worker(path: str) -> None:
    """Runs in a subprocess with OMP_* set by the parent: times batch_retrieve +
    attention_wrapper of ONE sparse layer (same data as the GPU's first sparse layer)."""
    z = np.load(path)
    meta = {k: int(v) for k, v in zip(z["meta_keys"], z["meta_vals"])}
    B, H, Hkv, D, M, K, L, n, steps = (meta[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "n", "steps"))
    cores = meta["cores"]          # host cores seen by the parent (OMP_PROC_BIND pins this process's main thread)
    BH = B * H
    sorted_codes = torch.from_numpy(z["sorted_codes"])        # [B, Hkv, L, n] int16
    sorted_ids = torch.from_numpy(z["sorted_ids"])            # int32
    k = torch.from_numpy(z["k"]).view(torch.bfloat16)         # [B, Hkv, n, D]
    v = torch.from_numpy(z["v"]).view(torch.bfloat16)
    kn = torch.from_numpy(z["kn"])
    qs = torch.from_numpy(z["q"]).view(torch.bfloat16)        # [NQ, BH, D]
    codes = torch.from_numpy(z["qcodes"])                     # [NQ, BH, L] int32
    kind = "port"
    sys.path.insert(0, ROOT)
    from oracle import build_ref

    if build_ref.ref_available():
        ref_lsh, ref_attn = build_ref.load_ref()
        lsh, srv = ref_lsh.LSH(), ref_attn.SparseAttentionServer()
        kind = "reference"
        cores = min(cores, 64)     # LSH_THREADS / ATTENTION_THREADS are compile-time 64 (lsh.h:12, sparse_attention.h:10)
    else:
        import oracle

        lsh, srv = oracle.LSH(nthreads=cores), oracle.SparseAttentionServer(nthreads=cores)
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        lsh.fill(0, b, sorted_codes[b].contiguous(), sorted_ids[b].contiguous())
        srv.fill(0, b, k[b].contiguous(), v[b].contiguous(), kn[b].contiguous())
    results = torch.zeros((BH, M), dtype=torch.int32)
    nnz = torch.zeros((BH,), dtype=torch.int32)
    out = torch.zeros((BH, D), dtype=torch.bfloat16)
    mve = torch.zeros((2, BH), dtype=torch.float32)
    NQ = qs.shape[0]
    t_ret, t_att = [], []
    warm = max(2, steps // 8)
    outs = []
    for i in range(warm + steps):
        q = qs[i % NQ].contiguous()
        qn = q.float().norm(p=2, dim=-1)
        c = codes[i % NQ].contiguous()
        t0 = time.perf_counter()
        lsh.batch_retrieve(0, c, results, nnz)
        t1 = time.perf_counter()
        srv.attention_wrapper(0, K, L, out, mve, q, qn, results, nnz)
        t2 = time.perf_counter()
        if i >= warm:
            t_ret.append(t1 - t0)
            t_att.append(t2 - t1)
        if i < NQ:
            outs.append((nnz.clone(), out.clone(), mve.clone()))
    # medians: the host is shared, a handful of descheduled steps would dominate a mean
    res = dict(kind=kind, cores=cores, t_retrieve_us=float(np.median(t_ret)) * 1e6,
               t_attention_us=float(np.median(t_att)) * 1e6,
               steps=steps, nnz0=outs[0][0].tolist(),
               out0=outs[0][1].float().flatten().tolist(), lse0=outs[0][2][1].tolist())
    print("CPU_BASELINE_JSON " + json.dumps(res))


def run_cpu_baseline(cfg, server, qs, steps, H, Hkv, placements=(None, "cores")):
    """Dump the first sparse layer (tables, KV, queries) and time the CPU path on it.  H, Hkv: the heads this
    process actually serves (cfg's, or the head shard's).  placements: OMP_PLACES settings tried (best one reported)."""
    B, D, M, K, Lt = (cfg[k] for k in ("B", "D", "M", "K", "L"))
    n = cfg["P"] - 68
    layer = 0
    bounds, table = server.lsh_retriever.get_tables(layer)
    ids = table[:, :, :n].reshape(B, Hkv, Lt, n).contiguous()
    # recover the sorted codes from the CSR bounds: code of position p = #buckets with start <= p ...
    # simpler and exact: re-hash the stored (centred) keys and gather by the stored ids
    kc = server.attn_server.get_key_cache(layer)[:, :, :n].contiguous()           # [B,Hkv,n,D]
    vc = server.attn_server.get_value_cache(layer)[:, :, :n].contiguous()
    kn = server.attn_server.get_key_norm(layer)[:, :, :n].contiguous()
    codes = torch.stack([server.hasher.keys(kc[b]) for b in range(B)])            # [B,Hkv,L,n] int16
    sorted_codes = torch.gather(codes, -1, ids.long())
    NQ = qs.shape[0]
    BH = B * H
    qcodes = torch.stack([server.hasher.query(qs[i, layer].reshape(BH, D))[0] for i in range(NQ)])
    path = os.path.join(tempfile.gettempdir(), f"mp_cpu_baseline_{os.getpid()}.npz")
    cores = len(os.sched_getaffinity(0))
    keys = ["B", "H", "Hkv", "D", "M", "K", "L", "n", "steps", "cores"]
    np.savez(path, meta_keys=np.array(keys), meta_vals=np.array([B, H, Hkv, D, M, K, Lt, n, steps, cores]),
             sorted_codes=sorted_codes.cpu().numpy(), sorted_ids=ids.cpu().numpy(),
             k=kc.cpu().view(torch.int16).numpy(), v=vc.cpu().view(torch.int16).numpy(),
             kn=kn.cpu().numpy(), q=qs[:, layer].reshape(NQ, BH, D).cpu().view(torch.int16).numpy(),
             qcodes=qcodes.cpu().numpy())
    # two thread placements (the reference pins with numactl, examples/bench.sh:1); the faster one
    # is reported: (a) libgomp default places, (b) one thread per physical core, packed
    best = None
    try:
        for places in placements:
            env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_THREAD_LIMIT=str(cores),
                       OMP_PROC_BIND="close", MKL_NUM_THREADS=str(cores))
            if places:
                env["OMP_PLACES"] = places
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path],
                               env=env, capture_output=True, text=True, timeout=900)
            for line in r.stdout.splitlines():
                if line.startswith("CPU_BASELINE_JSON "):
                    cb = json.loads(line[len("CPU_BASELINE_JSON "):])
                    cb["omp_places"] = places or "default"
                    if best is None or cb["t_retrieve_us"] + cb["t_attention_us"] < best["t_retrieve_us"] + best["t_attention_us"]:
                        best = cb
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    if best is not None:
        return best
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            return json.loads(line[len("CPU_BASELINE_JSON "):])
    raise RuntimeError("cpu baseline worker failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


# ---------------------------------------------------------------------------- host-buffer mode leg

def host_mode_leg(server, cfg, qs, H, reps=40, pin_results=False):
    """Per-layer cost of the UNCHANGED caller: the decode lines of models/attnserver.py:264-303 -- q hash on the GPU,
    codes + query copied to pinned CPU tensors, batch_retrieve and attention_wrapper on CPU tensors (`results` and
    `nnz` pageable, the rest pinned, exactly as :59-66), output + LSE copied back -- eager, synchronised per layer.
    Untimed extra leg: `value` is measured with everything resident in HBM."""
    B, D, M, K, Lt = (cfg[k] for k in ("B", "D", "M", "K", "L"))
    BH = B * H
    dev = qs.device
    pin = lambda *shape, dtype: torch.zeros(shape, dtype=dtype).pin_memory()       # noqa: E731
    pinned_hashcode, pinned_query = pin(BH, Lt, dtype=torch.int32), pin(BH, D, dtype=torch.bfloat16)
    results, nnz = torch.zeros((BH, M), dtype=torch.int32), torch.zeros((BH,), dtype=torch.int32)      # :59-60: pageable
    if pin_results:      # the ONE caller-side change INTEGRATION.md 1 recommends: pin_memory=True on these two allocations
        results, nnz = results.pin_memory(), nnz.pin_memory()
    output, mve = pin(BH, D, dtype=torch.bfloat16), pin(2, BH, dtype=torch.float32)
    out_cuda = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev)
    lse_cuda = torch.zeros((BH,), dtype=torch.float32, device=dev)
    lsh, srv, hasher = server.lsh_retriever, server.attn_server, server.hasher

    def host_layer(q):
        codes, _ = hasher.query(q.reshape(BH, D))                       # :264-270 on the GPU
        pinned_hashcode.copy_(codes)                                    # :272
        pinned_query.copy_(q.reshape(BH, D))                            # :273
        lsh.batch_retrieve(0, pinned_hashcode, results, nnz)            # :299
        srv.attention_wrapper(0, K, Lt, output, mve, pinned_query, pinned_query.float().norm(p=2, dim=-1), results, nnz)
        lse_cuda.copy_(mve[1], non_blocking=True)                       # :302-303
        out_cuda.copy_(output, non_blocking=True)
        torch.cuda.synchronize()

    import magicpig_amd._lib as L

    NQ = qs.shape[0]

    per_rep = []

    def timed():
        for i in range(4):
            host_layer(qs[i % NQ, 0])
        t0 = time.perf_counter()
        for i in range(reps):
            t1 = time.perf_counter()
            host_layer(qs[i % NQ, 0])
            per_rep.append((time.perf_counter() - t1) * 1e6)
        return (time.perf_counter() - t0) / reps * 1e6

    for name in ("host_fast_hits", "host_fast_edited", "host_fast_unpaired"):
        L.set_option(name, 0)
    us = timed()
    served = {name[len("host_fast_"):]: L.get_option(name) for name in ("host_fast_hits", "host_fast_edited", "host_fast_unpaired")}
    host_layer(qs[(reps - 1) % NQ, 0])
    server.collect_nnz = True
    server.decode(qs[(reps - 1) % NQ, 0], 0)
    torch.cuda.synchronize()
    same = bool(torch.equal(server.nnz.cpu(), nnz)) and float((server.output.float().cpu() - output.float()).abs().max()) < 2e-2
    # the MEDIAN over the repetitions is the leg's figure, as in the CPU baseline (the host is shared: one descheduled
    # repetition -- 18 ms once in 40 on a box of round 5 -- would otherwise be the whole mean); the mean is kept next to it
    return {"us_per_layer": float(np.median(per_rep)), "mean_us": us, "max_us": float(np.max(per_rep)),
            "matches_device_entry": same, "reps": reps,
            "attention_calls_served": served,      # hits = the rows just handed out were recognised (no index upload)
            "what": "models/attnserver.py:264-303 unchanged: GPU q-hash, pinned codes/query/output, "
                    + ("results/nnz allocated with pin_memory=True (the one flag INTEGRATION.md 1 recommends), "
                       if pin_results else "pageable results/nnz, ")
                    + "batch_retrieve + attention_wrapper on CPU tensors, eager + synchronised per layer"}


# ---------------------------------------------------------------------------- end-to-end variant

def end_to_end(args, cfg, rank, world, dev, dist):
    """examples/bench.py:43-59 with synthetic weights: full decode step (embedding, 32 layers of
    projections + RoPE + attention server + MLP, lm_head).  Not the headline metric: the model GEMMs are
    torch-ROCm plumbing outside the north-star path."""
    from magicpig_amd import decode_harness as dh
    from magicpig_amd import sharding

    steps, warmup = args.steps, args.warmup
    big = cfg["model"].startswith("Llama-3.1-70B")
    shape = dh.LLAMA_3_1_70B if big else dh.LLAMA_3_1_8B
    # --shard head: the TENSOR-PARALLEL harness (llama_dist.py:195-220): every rank holds H / world heads and 1 / world of
    # the MLP, o_proj / down_proj partials are all-reduced over RCCL; the ranks decode the SAME B requests (strong
    # scaling).  --emulate-rank R/W (one process): rank R's share of a TP = W step with the all-reduce left out -- the
    # per-GPU compute of the step, for context.  --shard batch: an independent replica per rank (weak scaling).
    tp_rank, tp_world, reduce_fn = 0, 1, None
    if args.shard == "head":
        tp_rank, tp_world = rank, world
        if args.emulate_rank:
            tp_rank, tp_world = (int(x) for x in args.emulate_rank.split("/"))
            reduce_fn = lambda t: None                                            # noqa: E731
    assert big or cfg["model"].startswith("Llama-3.1-8B"), "--end-to-end is wired for the 8B and 70B shapes"
    assert not big or tp_world > 1, "the 70B shape needs --shard head over several ranks (or --emulate-rank R/W)"
    dec = dh.SyntheticLlamaDecoder(shape, K=cfg["K"], L=cfg["L"], batch_size=cfg["B"],
                                   max_length=cfg["M"], generation_buffer=max(256, steps + warmup + 8),
                                   dense_layers=cfg["dense"], device=str(dev), seed=0 if tp_world > 1 else rank,
                                   tp_rank=tp_rank, tp_world=tp_world, all_reduce=reduce_fn)
    ms, tps = dh.run_decode_benchmark(dec, cfg["P"], warmup=warmup, steps=steps, use_graph=not args.no_graph)
    ms = sharding.max_over_ranks(ms, device=dev)
    tokens = cfg["B"] if tp_world > 1 else world * cfg["B"]
    if rank == 0:
        what = (f"tp{tp_world} (heads and MLP sharded, 2 all-reduces per layer over RCCL"
                + (", ALL-REDUCE LEFT OUT: one emulated rank" if reduce_fn is not None else "") + ")") if tp_world > 1 \
            else f"dp{world} (independent replicas)"
        print(json.dumps({
            "metric": f"end-to-end decode tokens/sec, synthetic-weight {shape_name(big)} P={cfg['P']} K{cfg['K']}L{cfg['L']}",
            "value": tokens * 1e3 / ms, "unit": "tokens/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if tp_world > 1 else "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config} end-to-end: {shape_name(big)} B={cfg['B']} P={cfg['P']}, random weights, "
                                   f"{len(dec.sparse_layers)} LSH-sparse layers + {len(dec.dense_layers)} dense layers + "
                                   "projections/MLP/lm_head (torch-ROCm)",
                       "parallelism": what, "launch": "eager" if args.no_graph else "hipGraph"}}))
    if dist is not None:
        dist.destroy_process_group()


def shape_name(big: bool) -> str:
    return "Llama-3.1-70B" if big else "Llama-3.1-8B"


def dry_run(args, rank, world):
    """The launch contract without a GPU (tests/test_sharding_gloo.py): RANK / WORLD_SIZE / MASTER_*
    from the environment, process group, hyperplane broadcast, barrier-bracketed timed region,
    max over ranks, ONE JSON line from rank 0 -- with a sleep standing in for the kernels."""
    from magicpig_amd import sharding

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    cfg = CONFIGS[args.config]
    gen0 = torch.Generator(device="cpu").manual_seed(7 + rank)
    hash_func = torch.randn((cfg["D"], cfg["K"] * cfg["L"]), generator=gen0).to(torch.bfloat16)
    hash_func = sharding.sync_hash_func(hash_func, src=0)

    def sync_all():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))          # ranks differ: the slowest one sets the time
    sync_all()
    dt = sharding.max_over_ranks(time.perf_counter() - t0)
    line = {"metric": "dry-run", "value": world * cfg["B"] * args.steps / dt, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "config": {"workload": args.config + " (dry run: a sleep stands in for the kernels)", "shard": args.shard,
                       "process_group": None if dist is None else f"{dist.get_backend()} x{dist.get_world_size()}"},
            "rank_checksums": sharding.gather_scalars(1000 + rank),
            "planes_checksum": int(hash_func.view(torch.int16).to(torch.int64).sum())}
    if args.shard == "head":
        # the head-sharded layout of main(): partition -> (stand-in outputs: element = global head index) ->
        # all_gather at the edge -> the same checksum on every rank
        H_full, Hkv_full, B, D = cfg.get("H_full", cfg["H"]), cfg.get("Hkv_full", cfg["Hkv"]), cfg["B"], cfg["D"]
        shard = sharding.partition(B, H_full, Hkv_full, world, rank, mode="head")
        local = torch.tensor(list(shard.heads), dtype=torch.float32).view(1, -1, 1).expand(B, -1, D).to(torch.bfloat16)
        full = sharding.gather_outputs(local.contiguous(), shard, B, H_full)
        line.update({"value": B * args.steps / dt, "scaling": "strong", "heads_per_rank": shard.local_heads,
                     "gathered_shape": list(full.shape), "gathered_checksum": float(full.float().sum())})
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------- N ranks from a plain start

def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: re-execute this file under
    torch.distributed.run with N local ranks on 127.0.0.1 (evaluations/RULER/run_tensor_parallel.sh:84 starts the
    reference's tensor-parallel variant with `torchrun --nproc_per_node=8`).  Rank 0's JSON line is the child's stdout,
    the exit status the child's.  Fewer visible GPUs than ranks is an ERROR -- the measurement the caller asked for
    does not exist on this node -- never a smaller run under the same label."""
    import socket

    n = args.gpus
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.exit(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible on this node: refusing to run fewer ranks "
                     f"under the label n_gpus = {n}")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes needs it on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


# ---------------------------------------------------------------------------- one configuration on this rank

class Workload:
    """One configuration's server (all sparse layers of the model shape on synthetic keys), NQ steps' worth of
    queries, and the captured step.  Units -- (request, kv head) -- are drawn from their GLOBAL ids, so a rank's data
    does not depend on how many ranks share the work."""
    NQ = 16

    def __init__(self, mp, sharding, args, name, cfg, dev, rank, urank, uworld, shard_mode, data, queries):
        self.mp, self.args, self.name, self.cfg, self.dev, self.data = mp, args, name, cfg, dev, data
        B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
        self.shard = None
        if shard_mode == "head":
            # the model's heads over the ranks (evaluations/RULER/pred/attnserver_dist.py:252-254): rank r owns kv heads
            # [r * Hkv_full / world, ...) and their query heads; nothing is exchanged inside the path
            H_full, Hkv_full = cfg.get("H_full", H), cfg.get("Hkv_full", Hkv)
            self.shard = sharding.partition(B, H_full, Hkv_full, uworld, urank, mode="head")
            H, Hkv = self.shard.local_heads, self.shard.local_kv_heads
        self.B, self.H, self.Hkv, self.D, self.M, self.K, self.Lt, self.P = B, H, Hkv, D, M, K, Lt, P
        self.G = H // Hkv
        self.g_requests = list(range(B)) if self.shard is not None else list(range(urank * B, (urank + 1) * B))
        self.g_kv_heads = list(self.shard.kv_heads) if self.shard is not None else list(range(Hkv))
        self.NL = len([i for i in range(cfg["layers"]) if i not in cfg["dense"]])
        self.BH, self.n = B * H, P - 68
        # identical hyperplanes on every rank: rank 0's are broadcast once (attnserver_dist.py:279)
        gen0 = torch.Generator(device="cpu").manual_seed(7 + rank)
        hash_func = torch.randn((D, K * Lt), generator=gen0, dtype=torch.float32).to(torch.bfloat16).to(dev)
        self.hash_func = sharding.sync_hash_func(hash_func, src=0)
        self.graph = None
        self._build(queries)
        self.q_static = self.qs[0].clone()
        self.ND = args.distinct_layers if args.distinct_layers > 0 else self.NL

    def _build(self, queries):
        mp, dev, data, NQ = self.mp, self.dev, self.data, self.NQ
        B, H, Hkv, D, P, NL, n, G = self.B, self.H, self.Hkv, self.D, self.P, self.NL, self.n, self.G
        srv = mp.LSHSparseAttnServer(NL, H, Hkv, D, K=self.K, L=self.Lt, batch_size=B, max_length=self.M,
                                     dense_layers=(), device=str(dev), hash_func=self.hash_func,
                                     table_build=self.args.table_build,
                                     accel_budget_bytes=None if self.args.accel_budget is None else int(self.args.accel_budget * 1e9),
                                     ranges=self.args.ranges)
        t_s = time.time()
        for li in range(NL):
            for b in range(B):
                ks, vs = [], []
                for gkv in self.g_kv_heads:      # one (request, kv head) unit at a time, seeded by its GLOBAL id
                    gen_kv = torch.Generator(device=dev).manual_seed(1000 + 1_000_003 * li + 10_007 * self.g_requests[b] + gkv)
                    k1, v1 = synth_kv(data, P, 1, D, dev, gen_kv)
                    ks.append(k1)
                    vs.append(v1)
                kc, vc = torch.cat(ks, dim=1), torch.cat(vs, dim=1)
                srv.fill(li, b, kc, vc, P)
                srv.build_table(li, b, P)
                del kc, vc, ks, vs
        torch.cuda.synchronize()
        self.t_setup = time.time() - t_s
        import magicpig_amd._lib as _L
        self.build_rank_fallbacks = _L.get_option("build_rank_fallbacks")    # table builds redone with the exact ranking (expected 0)
        q = torch.empty((NQ, NL, B, H, 1, D), device=dev, dtype=torch.float32)
        hv = queries == "heavy" or (queries == "auto" and data != "randn")
        jj = torch.zeros((NQ, NL, B, H), device=dev, dtype=torch.long)
        for b in range(B):
            for hl in range(H):                 # one query head of one request at a time, seeded by its GLOBAL id
                gh = self.g_kv_heads[hl // G] * G + hl % G
                gen_q = torch.Generator(device=dev).manual_seed(2000 + 100_003 * self.g_requests[b] + gh)
                q[:, :, b, hl, 0] = torch.randn((NQ, NL, D), device=dev, dtype=torch.float32, generator=gen_q)
                if hv:
                    jj[:, :, b, hl] = torch.randint(0, n, (NQ, NL), device=dev, generator=gen_q)
        if hv:   # every query is pulled toward one (centred) key of its kv group: q <- 0.5 q + 3 k_j (one gather per layer)
            bi = torch.arange(B, device=dev)[None, :, None].expand(NQ, B, H)
            gi = (torch.arange(H, device=dev) // G)[None, None, :].expand(NQ, B, H)
            for li in range(NL):
                kc = srv.attn_server.get_key_cache(li)                             # [B, Hkv, M, D] centred keys
                q[:, li, :, :, 0] = 0.5 * q[:, li, :, :, 0] + 3.0 * kc[bi, gi, jj[:, li]].float()
        srv.by_products = bool(self.args.by_products)
        self.server, self.qs, self.heavy = srv, q.to(torch.bfloat16), hv

    # -- one decode token: every sparse layer once
    def step(self):
        for li in range(self.NL):
            self.server.decode(self.q_static[li], li % self.ND)

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.step()
        return g

    def run_steps(self, k0, count, graph="own"):
        g = self.graph if graph == "own" else graph
        for i in range(count):
            self.q_static.copy_(self.qs[(k0 + i) % self.NQ])    # this step's queries (produced by the QKV GEMM in a model)
            if g is not None:
                g.replay()
            else:
                self.step()

    def observe(self):
        """Untimed stats pass on step 0's queries: selected tokens per head, candidates, probed piece lengths."""
        srv, dev, BH, H, Hkv, Lt, NL = self.server, self.dev, self.BH, self.H, self.Hkv, self.Lt, self.NL
        srv.collect_nnz = True
        nnz_obs, cand_obs, piece_obs = [], [], []
        for li in range(NL):
            srv.decode(self.q_static[li], li)
            nnz_obs.append(srv.nnz.clone())
            if li < 4:
                codes, _ = srv.hasher.query(self.q_static[li].reshape(BH, self.D))
                bounds, _ = srv.lsh_retriever.get_tables(li)
                g = torch.arange(BH, device=dev) // (H // Hkv)
                be = bounds[g[:, None], torch.arange(Lt, device=dev)[None, :], codes.long()]   # [BH, L, R + 1]
                cand_obs.append((be[..., -1] - be[..., 0]).sum(-1))
                piece_obs.append((be[..., 1:] - be[..., :-1]).flatten())   # (table, bucket, token range) pieces probed
        torch.cuda.synchronize()
        srv.collect_nnz = False      # no statistics copies inside the timed region
        self.nnz_all = torch.stack(nnz_obs).float()
        self.nnz_mean = float(self.nnz_all.mean())
        self.cand_mean = float(torch.stack(cand_obs).float().mean())
        pieces = torch.cat(piece_obs).float()
        sample = pieces if pieces.numel() < (1 << 24) else pieces[:1 << 24]
        self.piece_stats = {"ranges_per_head": int(srv.lsh_retriever.R), "mean": float(pieces.mean()),
                            "p50": float(sample.quantile(0.5)), "p99": float(sample.quantile(0.99)),
                            "max": float(pieces.max()), "share_gt_30": float((pieces > 30).float().mean()),
                            "share_gt_126": float((pieces > 126).float().mean())}

    def bytes_per_launch(self):
        """Whole-layer algorithmic bytes, SURVEY.md 8(d): per head L bucket probes (8 B), the candidate ids (4 B), the
        selected ids (4 B), per selected token K row + V row + key norm + id, q and out; plus the hyperplanes once."""
        BH, Lt, D, K = self.BH, self.Lt, self.D, self.K
        return BH * (8 * Lt + 4 * self.cand_mean + 4 * self.nnz_mean + self.nnz_mean * (4 * D + 4) + 4 * self.nnz_mean
                     + 2 * D + 8) + 2 * D * K * Lt + BH * (2 * D + 4 * Lt)

    def launch_us(self, reps=8):
        """Average duration of one launch of the dominant kernel (a sparse layer IS one launch of lsh_decode_kernel):
        HIP events recorded on the launch stream around back-to-back launches (graph replays when captured)."""
        self.q_static.copy_(self.qs[0])

        def layer_pass():
            if self.graph is not None:
                self.graph.replay()
            else:
                for li in range(self.NL):
                    self.server.decode(self.q_static[li], li)
        layer_pass()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            layer_pass()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * self.NL), reps * self.NL

    def footprint(self):
        """HBM this configuration holds per sparse layer on this GPU: what the reference keeps in
        host memory (KV, norms, table) and what this implementation adds to it (sub-bounds, direct slots)."""
        f = dict(self.server.attn_server.footprint())
        f.update(self.server.lsh_retriever.footprint())
        total = f["kv"] + f["key_norms"] + f["bounds"] + f["table"] + f["slots"]
        f["total"] = total
        f["all_sparse_layers_GB"] = round(total * self.NL / 1e9, 2)
        f["index_over_kv"] = round((f["bounds"] + f["table"] + f["slots"]) / f["kv"], 2)
        return f

    def roofline(self, fused=True):
        k_us, n_timed = self.launch_us()
        bl = self.bytes_per_launch()
        achieved = bl / (k_us * 1e-6) / 1e9
        return {"bound": "hbm", "kernel": "lsh_decode_kernel" if fused else "lsh_retrieve_kernel + attn_sparse_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": None, "bytes_per_launch": bl, "avg_launch_us": k_us,
                "launches_timed": n_timed}

    def cpu_check(self, cpu_steps, placements):
        """The CPU path on this box's host cores on the first sparse layer, and the GPU's full-size first layer against it."""
        cb = run_cpu_baseline(self.cfg, self.server, self.qs, cpu_steps, self.H, self.Hkv, placements=placements)
        t_layer = cb["t_retrieve_us"] + cb["t_attention_us"]
        res = {"value": self.B / (self.NL * t_layer * 1e-6), "unit": "tokens/s", "cores": cb["cores"], "kind": cb["kind"],
               "sample": f"1 of {self.NL} sparse layers x {cb['steps']} decode steps, same tables/KV/queries as "
                         f"the GPU's first sparse layer; tokens/s = B / ({self.NL} x t_layer); "
                         f"OMP_PLACES={cb.get('omp_places')}, best of {len(placements)} placement(s)",
               "t_retrieve_us": cb["t_retrieve_us"], "t_attention_us": cb["t_attention_us"]}
        self.q_static.copy_(self.qs[0])
        self.server.collect_nnz = True
        o, _ = self.server.decode(self.q_static[0], 0)
        torch.cuda.synchronize()
        same_nnz = self.server.nnz.cpu().tolist() == cb["nnz0"]
        dmax = float((o.float().flatten().cpu() - torch.tensor(cb["out0"])).abs().max())
        self.server.collect_nnz = False
        res["gpu_matches"] = {"nnz_equal": same_nnz, "max_abs_out_diff": dmax}
        return res

    def release(self):
        self.graph = None
        self.server = None
        self.qs = self.q_static = None
        torch.cuda.empty_cache()


def config_leg(mp, sharding, args, name, dev):
    """A second / third CONFIGURATION in the same line: the headline's measurement at cfg 2 or at
    cfg 4's per-GPU share, shorter legs around it -- same loop, same graph capture, HIP-event launch average, SURVEY
    8(d) bytes, the full-size first layer against the CPU path of this box."""
    cfg = CONFIGS[name]
    w = Workload(mp, sharding, args, name, cfg, dev, 0, 0, 1, "batch", "randn", "auto")
    w.observe()
    if not args.no_graph:
        w.graph = w.capture()
    w.run_steps(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = w.roofline()
    w.server.attn_server.check()
    share = {"cfg4": " (per-GPU share of TP=8)", "cfg3": " (per-GPU share: B = 8 of 64)"}.get(name, "")
    leg = {"workload": f"{name}{share}: {cfg['model']} B={w.B} P={w.P} "
                       f"K={w.K} L={w.Lt}, {w.NL} sparse layers/step, H={w.H} Hkv={w.Hkv} D={w.D}",
           "tokens_per_s": w.B * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
           "us_per_layer": dt / args.steps * 1e6 / w.NL, "steps": args.steps, "warmup": args.warmup,
           "nnz_per_head": w.nnz_mean, "candidates_per_head": w.cand_mean, "selected_fraction": w.nnz_mean / w.n,
           "ranges_per_head": w.piece_stats["ranges_per_head"],
           "decode_form": "by-products on" if args.by_products else "MP_DECODE_NO_BYPRODUCTS",
           "roofline": {k: roof[k] for k in ("bytes_per_launch", "avg_launch_us", "achieved", "frac", "launches_timed")},
           "hbm_bytes_per_layer": w.footprint()}
    if not args.no_cpu_baseline:
        leg["cpu_baseline"] = w.cpu_check(args.leg_cpu_steps, ("cores",))
        leg["speedup_vs_cpu"] = leg["tokens_per_s"] / leg["cpu_baseline"]["value"]
    w.release()
    return leg


def e2e_leg(args, name, dev):
    """legs.e2e: the WHOLE decode step of examples/bench.py:43-59 -- embedding, 32 layers of projections +
    RoPE + attention (30 LSH-sparse layers through the hot path with the static window folded in, 2 dense layers) + MLP,
    lm_head -- on random Llama-3.1-8B weights at one BASELINE configuration, captured in a hipGraph; tokens/s of the step
    and its split by kind of work (each kind re-captured as a graph of its own over all layers, HIP events around
    back-to-back replays: decode_harness.split_decode_step).  The model GEMMs are torch-ROCm plumbing outside the
    north-star path: context for what "decode tokens/sec" means end to end, not the headline."""
    from magicpig_amd import decode_harness as dh

    cfg = CONFIGS[name]
    steps, warmup = args.e2e_steps, 4
    t_s = time.time()
    dec = dh.SyntheticLlamaDecoder(dh.LLAMA_3_1_8B, K=cfg["K"], L=cfg["L"], batch_size=cfg["B"], max_length=cfg["M"],
                                   generation_buffer=max(64, steps + warmup + 16), dense_layers=cfg["dense"], device=str(dev), seed=0)
    dec.attention_server.by_products = bool(args.by_products)
    ms, tps = dh.run_decode_benchmark(dec, cfg["P"], warmup=warmup, steps=steps, use_graph=not args.no_graph)
    split = dh.split_decode_step(dec)
    parts = sum(split.values())
    leg = {"workload": f"{name} end to end: Llama-3.1-8B (random weights) B={cfg['B']} P={cfg['P']} K={cfg['K']} L={cfg['L']}, "
                       f"{len(dec.sparse_layers)} LSH-sparse + {len(dec.dense_layers)} dense layers, projections / MLP / lm_head in torch-ROCm",
           "tokens_per_s": tps, "ms_per_step": ms, "steps": steps, "warmup": warmup,
           "launch": "eager" if args.no_graph else "hipGraph",
           "split_ms": {k: round(v, 4) for k, v in split.items()}, "split_sum_ms": round(parts, 4),
           "split_note": "each kind of work re-captured as its own hipGraph over all layers on static inputs, HIP events around "
                         "8 replays; sparse_attention = append + q-hash + retrieve + sampled attention + static window (one "
                         "launch + the append per sparse layer)",
           "hot_path_share_of_step": round(split["sparse_attention"] / ms, 4), "leg_wall_s": round(time.time() - t_s, 1)}
    del dec
    torch.cuda.empty_cache()
    return leg


# ---------------------------------------------------------------------------- main

def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_worker(args.cpu_baseline_worker)
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_rank:
        launch_ranks(args)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: start it as `python bench.py --gpus N` or under "
                 "`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU)")
    if args.shard is None:     # cfg 4 IS the TP = 8 layout of the 70B model: on several GPUs its kv heads are what is sharded
        emu_world = int(args.emulate_rank.split("/")[1]) if args.emulate_rank else 1
        args.shard = "head" if (args.config == "cfg4" and max(world, emu_world) > 1) else "batch"
    if args.dry_run:
        return dry_run(args, rank, world)
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have <= local_rank:
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU: {have} visible on this node")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:     # launched under torch.distributed.run (also with ONE rank: the
        import torch.distributed as dist             # RCCL path -- broadcast, barrier, f64 MAX, all_gather -- runs as with N)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    import magicpig_amd._lib as L
    if args.lib:
        L.LIB_PATH = os.path.abspath(args.lib)
    import magicpig_amd as mp
    from magicpig_amd import sharding

    cfg = CONFIGS[args.config]
    if args.two_launch:
        L.set_option("decode_two_launch", 1)
    if args.kn_payload >= 0:
        L.set_option("decode_kn_payload", args.kn_payload)
    if args.mfma_hash:
        L.set_option("decode_mfma_hash", 1)
    if args.cluster:
        L.set_option("decode_cluster", args.cluster)
    if args.split_hash >= 0:
        L.set_option("decode_split_hash", args.split_hash)
    if args.quad_hash >= 0:
        L.set_option("decode_quad_hash", args.quad_hash)
    if args.slot_log2:
        L.set_option("decode_slot_log2", args.slot_log2)
    if args.no_direct_slots:
        L.set_option("decode_direct", 0)
    elif args.direct_slots >= 0:
        L.set_option("decode_direct", args.direct_slots)
    if args.end_to_end:
        return end_to_end(args, cfg, rank, world, dev, dist)
    # which units (request, kv head) this process serves: by its rank in the process group, or -- --emulate-rank R/W,
    # one process, no group -- those rank R of W would serve
    urank, uworld = rank, world
    if args.emulate_rank:
        urank, uworld = (int(x) for x in args.emulate_rank.split("/"))
        if world != 1 or not 0 <= urank < uworld:
            sys.exit("bench.py: --emulate-rank R/W needs one process and 0 <= R < W")
    w = Workload(mp, sharding, args, args.config, cfg, dev, rank, urank, uworld, args.shard, args.data, args.queries)
    shard, server, qs, q_static = w.shard, w.server, w.qs, w.q_static
    B, H, Hkv, D, M, K, Lt, P, NL, NQ, n = w.B, w.H, w.Hkv, w.D, w.M, w.K, w.Lt, w.P, w.NL, w.NQ, w.n

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    w.observe()
    if not args.no_graph:
        w.graph = w.capture()

    if args.ab_worker:
        # One side of scripts/ab_libs.py: this process holds ONE build of the library (--lib), the workload and a captured
        # step; the parent alternates timed regions between two such processes on the same GPU.  Protocol on stdin / stdout:
        # "ready" once the step is captured; per "go" line one JSON line {"us_per_layer", "checksum"}; "quit" ends.
        g = w.capture()
        q_static.copy_(qs[0])
        g.replay()
        torch.cuda.synchronize()
        chk = int(server.output.view(torch.int16).to(torch.int64).sum().item()) * 1_000_003 + \
            int((server.max_value_expsum[1] * 1024).to(torch.int64).sum().item())
        print("ready", flush=True)
        for line in sys.stdin:
            cmd = line.strip()
            if cmd == "quit":
                break
            if cmd != "go":
                continue
            w.run_steps(0, args.warmup, g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w.run_steps(0, args.steps, g)
            torch.cuda.synchronize()
            print(json.dumps({"us_per_layer": round((time.perf_counter() - t0) / args.steps / NL * 1e6, 3),
                              "checksum": chk}), flush=True)
        server.attn_server.check()
        return

    if args.ab_option:
        # A/B of a library option that is read at call time: one captured step per value over the SAME workload, timed
        # alternately (profiles/*_ab_*.txt).  Not a bench line.
        name, vals = args.ab_option.split("=")
        vals = [int(v) for v in vals.split(",")]
        graphs = []
        for v in vals:
            L.set_option(name, v)
            graphs.append(w.capture())
        L.set_option(name, vals[0])
        # same queries through every captured variant: the last layer's outputs must be the same bits
        outs = []
        for g in graphs:
            q_static.copy_(qs[0])
            g.replay()
            torch.cuda.synchronize()
            outs.append((server.output.clone(), server.max_value_expsum.clone()))
        same = all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
        res = {str(v): [] for v in vals}
        for rep in range(args.ab_reps):
            for v, g in zip(vals, graphs):
                w.run_steps(0, args.warmup, g)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                w.run_steps(0, args.steps, g)
                torch.cuda.synchronize()
                res[str(v)].append(round((time.perf_counter() - t0) / args.steps / NL * 1e6, 3))
        server.attn_server.check()
        print(json.dumps({"ab_option": name, "config": args.config, "data": args.data, "steps": args.steps,
                          "outputs_bit_identical": same, "us_per_layer": res}))
        return

    w.run_steps(0, args.warmup)
    sync_all()
    t0 = time.perf_counter()
    w.run_steps(args.warmup, args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, device=dev)
    ms_per_step = dt / args.steps * 1e3
    # batch sharding: every rank decodes its own B requests; head sharding: all ranks share the same B requests
    tokens_per_s = (B if shard is not None else world * B) * args.steps / dt
    us_per_layer = ms_per_step * 1e3 / NL
    gathered = None
    if shard is not None:
        # the edge of the path (outside the timed region): every rank's [B, H_loc, D] outputs of the last layer
        # all_gathered through RCCL into [B, H_full, D]; the checksum is the same on every rank
        q_static.copy_(qs[0])
        o_loc, _ = server.decode(q_static[NL - 1], NL - 1)
        torch.cuda.synchronize()
        t_g = time.perf_counter()
        full = sharding.gather_outputs(o_loc.clone(), shard, B, cfg.get("H_full", H))
        torch.cuda.synchronize()
        gathered = {"allgather_us": (time.perf_counter() - t_g) * 1e6, "shape": list(full.shape),
                    "checksum": int(full.view(torch.int16).to(torch.int64).sum().item())}

    # per-rank checksum of the last layer's outputs + counts on step 0's queries, gathered from every rank: what the
    # multi-GPU tests compare with single-process runs of the same units (--emulate-rank)
    q_static.copy_(qs[0])
    server.collect_nnz = True
    o_chk, _ = server.decode(q_static[NL - 1], NL - 1)
    torch.cuda.synchronize()
    my_sum = int(o_chk.reshape(-1).view(torch.int16).to(torch.int64).sum().item()) * 1_000_003 + int(server.nnz.sum().item())
    server.collect_nnz = False
    rank_checksums = sharding.gather_scalars(my_sum, device=dev)

    # ---- roofline leg.  A sparse layer is ONE launch (lsh_decode_kernel: hash -> retrieve -> attention),
    # so the dominant kernel is the step itself: its average duration is taken from HIP events recorded
    # on the launch stream around back-to-back launches (graph replays when the step is captured), its
    # algorithmic bytes are the whole-layer figure of SURVEY.md 8(d).
    fused = not args.two_launch
    roof = w.roofline(fused)
    # the device-side validations of all the launches above (append overflow, a cluster seen on two XCDs): raises
    server.attn_server.check()
    # HBM traffic per launch cannot be counted from inside this process (the PMC counters need rocprofv3 around
    # it): the line carries the figure of the last committed PMC passes over this same command and says so
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    if os.path.exists(tpath) and shard is None:
        try:
            with open(tpath) as f:
                tj = json.load(f)
            key = "lsh_decode_bytes_per_launch" if fused else "two_launch_bytes_per_layer"
            if args.data != "randn":
                key += "_" + args.data              # PMC passes exist per key distribution (or not at all)
            elif fused and args.by_products:
                key += "_byproducts"                # ... and per form of the launch
            roof["traffic"] = (tj.get(key) or {}).get(args.config)
            if roof["traffic"] is not None:
                roof["traffic_source"] = "not measured in this run: rocprofv3 PMC passes of " + str(tj.get("source", "profiles/"))
        except Exception:
            roof["traffic"] = None

    out = {
        "metric": "decode tokens/sec (LSH sparse-attention path), " + cfg["model"] +
                  f" P={P} K{K}L{Lt}",
        "value": tokens_per_s, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if shard is not None else "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic" if args.data == "randn" else f"synthetic ({args.data} keys, heavy-hitter queries)" if w.heavy
                else f"synthetic ({args.data} keys)",
        "config": {"workload": f"{args.config}: {cfg['model']} B={B} P={P} K={K} L={Lt}, "
                               f"{NL} sparse layers/step, H={H} Hkv={Hkv} D={D}, KV+tables resident in HBM",
                   "batch_per_gpu": B, "global_batch": B if shard is not None else B * world,
                   "parallelism": (f"tp{world} (kv heads sharded: {Hkv} of {cfg.get('Hkv_full', Hkv)} kv heads, "
                                   f"{H} of {cfg.get('H_full', H)} query heads per GPU)") if shard is not None
                                  else f"dp{world} (requests sharded)",
                   "launch": "eager" if w.graph is None else "hipGraph",
                   "decode_form": ("mp_decode_sparse_layer (by-products on: codes, result rows, logits written)"
                                   if args.by_products or args.two_launch or args.mfma_hash else
                                   "mp_decode_sparse_layer_ex(MP_DECODE_NO_BYPRODUCTS): output + LSE + counts only, "
                                   "selected ids handed to the gather unordered"),
                   "process_group": None if dist is None else f"{dist.get_backend()} x{dist.get_world_size()}"},
        "sparse_attn_us_per_layer": us_per_layer,
        "rank_checksums": rank_checksums,
        **({"emulated_rank": args.emulate_rank} if args.emulate_rank else {}),
        **({"DIAGNOSTIC_distinct_layers": w.ND} if w.ND != NL else {}),
        "observed": {"nnz_per_head": w.nnz_mean, "candidates_per_head": w.cand_mean,
                     "selected_fraction": w.nnz_mean / n, "nnz_max_head": float(w.nnz_all.max()),
                     "probed_pieces": w.piece_stats, "key_distribution": args.data,
                     "queries": "heavy" if w.heavy else "randn", "setup_s": w.t_setup,
                     "build_rank_fallbacks": w.build_rank_fallbacks,
                     "hbm_bytes_per_layer": w.footprint()},
        "roofline": roof,
    }
    if gathered is not None:
        out["head_shard_gather"] = gathered

    # ---- the extra legs of a single-GPU run.  A leg that fails is recorded in the line and (unless
    # --allow-missing-legs) turns the exit status non-zero: a line without its CPU baseline must not look like a success
    failures = []
    solo = rank == 0 and world == 1 and dist is None and not args.emulate_rank

    def attempt(name, fn, on_fail):
        try:
            return fn()
        except Exception as e:
            failures.append(f"{name}: {e!r}"[:400])
            return on_fail(f"failed: {e!r}"[:300])

    if solo and shard is None and not args.no_host_mode:
        out["host_mode"] = attempt("host_mode", lambda: host_mode_leg(server, cfg, qs, H),
                                   lambda msg: {"us_per_layer": None, "what": msg})
        # the same caller with ONE allocation flag changed: results_lsh_cpu / nnz pinned
        out["host_mode_pinned_results"] = attempt("host_mode_pinned_results",
                                                  lambda: host_mode_leg(server, cfg, qs, H, pin_results=True),
                                                  lambda msg: {"us_per_layer": None, "what": msg})
    # (rank 0 of an N-rank run times it too -- on ITS share of the units: the line of a multi-GPU run is self-contained;
    # the other ranks wait at the barrier below)
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = attempt("cpu_baseline", lambda: w.cpu_check(args.cpu_steps, (None, "cores")),
                                      lambda msg: {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": msg})
        if out["cpu_baseline"].get("value"):
            if world > 1:      # the CPU figure is one rank's share on one host: compare it with that rank's share of the GPU rate
                out["cpu_baseline"]["sample"] = "rank 0's units only; " + out["cpu_baseline"]["sample"]
                out["speedup_vs_cpu"] = (tokens_per_s / world if shard is None else tokens_per_s) / out["cpu_baseline"]["value"]
            else:
                out["speedup_vs_cpu"] = tokens_per_s / out["cpu_baseline"]["value"]
            if out.get("host_mode", {}).get("us_per_layer"):      # the unchanged caller against the CPU path, same layer
                t_cpu = out["cpu_baseline"]["t_retrieve_us"] + out["cpu_baseline"]["t_attention_us"]
                out["host_mode"]["speedup_vs_cpu_layer"] = t_cpu / out["host_mode"]["us_per_layer"]
                if out["host_mode_pinned_results"].get("us_per_layer"):
                    out["host_mode_pinned_results"]["speedup_vs_cpu_layer"] = t_cpu / out["host_mode_pinned_results"]["us_per_layer"]
    # ---- second workload in the same line: clustered keys + heavy-hitter queries, the README's
    # ~2 % sampling rate -- randn keys are the easiest case SimHash can see.  Same model shape, same loop, after the
    # headline region and its legs; the first workload's HBM is released first.
    w.release()
    del server, qs, q_static
    if solo and shard is None and args.data == "randn" and not args.no_clustered_leg:
        def clustered():
            w2 = Workload(mp, sharding, args, args.config, cfg, dev, rank, urank, uworld, args.shard, "clustered", "auto")
            w2.observe()
            if not args.no_graph:
                w2.graph = w2.capture()
            w2.run_steps(0, args.warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w2.run_steps(args.warmup, args.steps)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            w2.server.attn_server.check()
            res = {"value_clustered": B * args.steps / dt2,
                   "sparse_attn_us_per_layer_clustered": dt2 / args.steps * 1e6 / NL,
                   "observed_clustered": {"nnz_per_head": w2.nnz_mean, "selected_fraction": w2.nnz_mean / n,
                                          "nnz_max_head": float(w2.nnz_all.max()), "key_distribution": "clustered",
                                          "queries": "heavy", "steps": args.steps, "warmup": args.warmup}}
            w2.release()
            return res
        out.update(attempt("clustered", clustered,
                           lambda msg: {"value_clustered": None, "observed_clustered": {"failed": msg}}))
    # ---- other CONFIGURATIONS in the same line
    legs = [x for x in args.legs.split(",") if x]
    if solo and shard is None and args.config == "cfg1" and args.data == "randn" and legs:
        out["legs"] = {}
        for name in legs:
            if name not in CONFIGS or name == args.config:
                continue
            t_leg = time.time()
            key = name + "_share" if name in ("cfg3", "cfg4") else name
            out["legs"][key] = attempt("leg " + name, lambda: config_leg(mp, sharding, args, name, dev),
                                       lambda msg: {"tokens_per_s": None, "failed": msg})
            out["legs"][key]["leg_wall_s"] = round(time.time() - t_leg, 1)
    if solo and shard is None and args.config == "cfg1" and args.data == "randn" and legs and not args.no_e2e_leg:
        out["legs"]["e2e"] = {}
        for name in ("cfg1", "cfg2"):
            out["legs"]["e2e"][name] = attempt("leg e2e " + name, lambda: e2e_leg(args, name, dev),
                                               lambda msg: {"tokens_per_s": None, "failed": msg})
    if failures:
        out["failed_legs"] = failures
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()              # (the other ranks wait here while rank 0 times the CPU path)
        dist.destroy_process_group()
    if failures and not args.allow_missing_legs:
        sys.exit("bench.py: " + "; ".join(failures))


if __name__ == "__main__":
    main()
