"""RECORD of an experiment (EXPERIMENTS.md R4-6) -- it drives the `host_register` option, which round 4 REMOVED from the
library after this hunt reproduced neither abort; to re-run it, check out commit "Fast-path query normalisation ..." of round 4
(the last one that still has the option).

Root-cause hunt for the two process aborts of round 3 (EXPERIMENTS.md R3-9): the opt-in `host_register` mode of the
MP_MEM_HOST calls (hipHostRegister of a caller's pageable `results` buffer, used in place by the kernels).

    python scripts/stress_host_register.py <scenario> [iterations]     (one scenario per process: an abort ends it)
    python scripts/stress_host_register.py all                         (every scenario in its own subprocess, under
                                                                        rocgdb when MP_STRESS_GDB=1: backtrace on abort)

Scenarios (every one allocates, uses, frees and re-allocates pageable buffers of ONE size, so that the allocator hands
the same addresses out again -- glibc serves >= 128 KiB by mmap / munmap):
  live        register, use, destroy the handle (it unregisters), THEN free the buffer: the documented contract
  free_first  register, use, FREE THE BUFFER, then destroy the handle (hipHostUnregister of an unmapped range)
  stale       handle A registers buffer X, the caller frees X; a new buffer lands on X's address and is handed to handle B
              in the DEFAULT mode while A (and its registration) is still alive: hipPointerGetAttributes then reports the
              stale registration as mapped host memory and the kernel writes through it
  resize      the same pointer seen again with a larger size (a registration that covers only a prefix)
"""
from __future__ import annotations

import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


# MP_STRESS_BIG=1 (environment): 64 heads x 300 000 rows, so that a `results` buffer is 77 MB -- above glibc's dynamic mmap threshold
# (at most 32 MiB): freeing it really unmaps it.  With the small shape a freed buffer stays in the heap, its pages never
# leave the process and a stale registration keeps pointing at live memory (round 4, first hunt: nothing failed).
BIG = bool(os.environ.get("MP_STRESS_BIG"))
M_ROWS = 300000 if BIG else 20480          # (the collision bitmaps of max_length tokens must fit the LDS: <= ~600 K)
HEADS = 64 if BIG else 8                    # BIG: results = 64 x 300 000 x 4 = 76.8 MB


def _setup(mp, torch, K=6, L=20, H=HEADS, Hkv=2, B=1, n=20000, M=M_ROWS, seed=1):
    gen = torch.Generator().manual_seed(seed)
    lsh = mp.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    codes = torch.randint(0, 1 << K, (Hkv, L, n), dtype=torch.int16, generator=gen)
    lsh.fastfill(0, 0, codes.cuda())
    q = torch.randint(0, 1 << K, (B * H, L), dtype=torch.int32, generator=gen)
    ref_res = torch.zeros((B * H, M), dtype=torch.int32, device="cuda")
    ref_nnz = torch.zeros((B * H,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, q.cuda(), ref_res, ref_nnz)
    torch.cuda.synchronize()
    return lsh, q, ref_res.cpu(), ref_nnz.cpu()


def _check(res, nnz, ref_res, ref_nnz, what):
    import torch
    assert torch.equal(nnz, ref_nnz), f"{what}: nnz differs"
    for h in range(nnz.numel()):
        z = int(nnz[h])
        assert torch.equal(res[h, :z], ref_res[h, :z]), f"{what}: ids of head {h} differ"


def run(scenario: str, iters: int) -> None:
    import torch
    import magicpig_amd as mp
    import magicpig_amd._lib as L_

    faulthandler.enable()
    BH, M = HEADS, M_ROWS                 # results: 8 x 20480 x 4 = 640 KiB (> the 256 KiB registration threshold); BIG: 77 MB
    addrs = set()
    reused = 0
    for it in range(iters):
        lsh, q, ref_res, ref_nnz = _setup(mp, torch, seed=it)
        L_.set_option("host_register", 1)
        res = torch.full((BH, M), -7, dtype=torch.int32)
        nnz = torch.zeros((BH,), dtype=torch.int32)
        addrs.add(res.data_ptr())
        lsh.batch_retrieve(0, q, res, nnz)
        _check(res, nnz, ref_res, ref_nnz, f"{scenario} it {it} registered")
        if scenario == "live":
            del lsh
            del res
        elif scenario == "free_first":
            del res                       # the registration now covers an unmapped range
            res2 = torch.full((BH, M), -7, dtype=torch.int32)       # ... or somebody else's pages
            res2.fill_(3)
            del lsh                       # hipHostUnregister of the stale range
            del res2
        elif scenario == "stale":
            del res
            L_.set_option("host_register", 0)
            lshB, qB, refB, refzB = _setup(mp, torch, seed=1000 + it)
            resB = torch.full((BH, M), -7, dtype=torch.int32)       # very likely X's address again
            nnzB = torch.zeros((BH,), dtype=torch.int32)
            same = resB.data_ptr() in addrs
            reused += int(same)
            lshB.batch_retrieve(0, qB, resB, nnzB)
            try:
                _check(resB, nnzB, refB, refzB, f"stale it {it} (address reused: {same})")
            except AssertionError as e:
                print(f"[stale] WRONG RESULT through a stale registration: {e}", flush=True)
                raise
            del lshB, resB
            del lsh
        elif scenario == "resize":
            big = torch.full((2 * BH, M), -7, dtype=torch.int32)
            lsh2, q2, r2, z2 = _setup(mp, torch, H=2 * HEADS, Hkv=4, seed=500 + it)
            nn2 = torch.zeros((2 * BH,), dtype=torch.int32)
            lsh2.batch_retrieve(0, q2, big, nn2)
            _check(big, nn2, r2, z2, f"resize it {it}")
            del lsh2, big, lsh, res
        else:
            raise SystemExit(f"unknown scenario {scenario}")
        L_.set_option("host_register", 0)
        if it % 20 == 0:
            print(f"[{scenario}] iteration {it} ok ({len(addrs)} distinct result addresses so far"
                  + (f", stale address handed out again {reused} times" if scenario == "stale" else "") + ")",
                  flush=True)
    print(f"[{scenario}] {iters} iterations ok, {len(addrs)} distinct result addresses", flush=True)


def main() -> None:
    scenario = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    if scenario != "all":
        run(scenario, iters)
        return
    env = dict(os.environ, AMD_LOG_LEVEL=os.environ.get("AMD_LOG_LEVEL", "1"))
    for sc in (os.environ.get("MP_STRESS_SCENARIOS", "live,resize,free_first,stale")).split(","):
        cmd = [sys.executable, os.path.abspath(__file__), sc, str(iters)]
        if os.environ.get("MP_STRESS_GDB") and os.path.exists("/opt/rocm/bin/rocgdb"):
            cmd = ["/opt/rocm/bin/rocgdb", "-batch", "-ex", "run", "-ex", "bt", "-ex", "info threads", "--args"] + cmd
        print(f"==== scenario {sc}: {' '.join(cmd)}", flush=True)
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
            keep = [ln for ln in r.stdout.splitlines() if not ln.startswith(("[Thread ", "[New Thread", "[New process"))]
            err = [ln for ln in r.stderr.splitlines() if "Cannot get amd_mem_obj" not in ln]
            tail = "\n".join(keep[-40:]) + "\n---- stderr (without the pageable-pointer lookups) ----\n" + "\n".join(err[-40:])
            print(f"==== scenario {sc}: exit code {r.returncode}\n{tail}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"==== scenario {sc}: TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
