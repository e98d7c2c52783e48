"""A/B of two (or more) BUILDS of the library on one GPU with the precision of a one-process A/B: every build gets a process of
its own (`bench.py --ab-worker --lib <build>`: the workload, one captured step), and this parent hands out timed regions in
turn -- A, B, A, B, ... -- so that both see the same minute of the same box.  The workloads are seeded: the same data in every
process; the last layer's output checksum is compared across builds.

usage: python scripts/ab_libs.py <cfg> <lib or 'product'> <lib or 'product'> [...] [--reps N] [--data D] [--steps K] [--warmup W]
prints one JSON line: {"config", "data", "libs", "us_per_layer": {lib: [...]}, "outputs_equal"}"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_path(name):
    if name == "product":
        return None
    if os.path.exists(name):
        return os.path.abspath(name)
    return os.path.join(ROOT, "magicpig_amd", "lib", "variants", name, "libmagicpig_hip.so")


def worker_cmd(cfg, lib, data, steps, warmup):
    """`lib` = a build (product | variant name | path), optionally followed by `@` and comma-separated extra bench.py
    arguments for that side (round 5: settings read at alloc time, e.g. product@--direct-slots=1 against product)."""
    lib, _, extra = lib.partition("@")
    cfg_args = ["--config-json", cfg] if cfg.lstrip().startswith("{") else ["--config", cfg]      # a shape as JSON works too
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *cfg_args, "--data", data, "--steps", str(steps),
           "--warmup", str(warmup), "--ab-worker", "--no-cpu-baseline", "--no-host-mode", "--no-clustered-leg"]
    p = lib_path(lib)
    return cmd + (["--lib", p] if p else []) + [a for a in extra.split(",") if a]


def read_until(proc, pred):
    """Next stdout line of `proc` that satisfies pred (other lines -- warnings of the runtime -- are skipped)."""
    while True:
        line = proc.stdout.readline()
        if not line:
            raise RuntimeError(f"worker exited (rc {proc.poll()})")
        line = line.strip()
        if pred(line):
            return line


def run(cfg, libs, reps, data, steps, warmup, make_cmd=worker_cmd):
    procs = [subprocess.Popen(make_cmd(cfg, lib, data, steps, warmup), stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True, bufsize=1) for lib in libs]
    res = {lib: [] for lib in libs}
    sums = {}
    try:
        for p in procs:
            read_until(p, lambda s: s == "ready")
        for _ in range(reps):
            for lib, p in zip(libs, procs):
                p.stdin.write("go\n")
                p.stdin.flush()
                d = json.loads(read_until(p, lambda s: s.startswith("{")))
                res[lib].append(d["us_per_layer"])
                sums[lib] = d["checksum"]
    finally:
        for p in procs:
            try:
                p.stdin.write("quit\n")
                p.stdin.flush()
            except (BrokenPipeError, OSError):
                pass
        for p in procs:
            try:
                p.wait(timeout=60)
            except subprocess.TimeoutExpired:
                p.kill()
    return {"config": cfg, "data": data, "libs": libs, "steps": steps, "us_per_layer": res,
            "outputs_equal": len(set(sums.values())) == 1}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg")
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--data", default="randn")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    a = ap.parse_args()
    print(json.dumps(run(a.cfg, a.libs, a.reps, a.data, a.steps, a.warmup)))
