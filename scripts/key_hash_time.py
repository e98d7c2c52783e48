"""Key SimHash (and, with `build`, the table build) at cfg-1 / cfg-4 size, timed with device events per library build:
python scripts/key_hash_time.py [build] <lib or variant name> ...  ("product" = the in-tree library).  One process per side."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch
    import magicpig_amd._lib as L0
    if sys.argv[2] != "product":
        L0.LIB_PATH = os.path.join(ROOT, "magicpig_amd", "lib", "variants", sys.argv[2], "libmagicpig_hip.so")
    import magicpig_amd as mp
    do_build = sys.argv[3] == "1"
    out = []
    for name, (n, M, D, K, Lt, Hkv, H) in {"cfg1": (97932, 98304, 128, 10, 150, 8, 32), "cfg4": (131004, 131072, 128, 11, 300, 1, 8)}.items():
        torch.manual_seed(0)
        W = torch.randn((D, K * Lt), device="cuda").to(torch.bfloat16)
        sh = mp.SimHash(W, K, Lt)
        keys = torch.randn((Hkv, n, D), device="cuda").to(torch.bfloat16)
        def timed(fn, reps=20):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        codes = sh.keys(keys)
        import hashlib
        digest = hashlib.sha256(codes.cpu().numpy().tobytes()).hexdigest()[:12]
        line = f"{name}: key hash {timed(lambda: sh.keys(keys)):7.1f} us  codes sha {digest}"
        if do_build:
            lsh = mp.LSH(); lsh.alloc(K, Lt, 1, H, Hkv, 1, M)
            line += f"  table build {timed(lambda: lsh.fastfill(0, 0, codes), 10):7.1f} us"
            tb = lsh.get_tables(0)
            line += "  tables sha " + hashlib.sha256(tb[0].cpu().numpy().tobytes() + tb[1].cpu().numpy().tobytes()).hexdigest()[:12]
        out.append(line)
    print(f"{sys.argv[2]:>10}: " + " | ".join(out), flush=True)
    sys.exit(0)
args = sys.argv[1:]
do_build = "0"
if args and args[0] == "build":
    do_build, args = "1", args[1:]
for rnd in range(2):                     # two rounds: box drift shows as a difference between a side's two lines
    for side in args or ["product"]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", side, do_build], check=False)
