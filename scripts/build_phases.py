"""Phase stamps of workgroup 0 of lsh_build_kernel (the -DMP_STAMPS=1 build) at cfg-1 / cfg-4 size, all rows of a layer in flight:
kernel start, row histogram done, the row's third tile (top, validated + zeroed, counted, scanned, ranked, written out), kernel end."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd._lib as L0
L0.LIB_PATH = os.path.join(ROOT, "magicpig_amd", "lib", "variants", "stamps", "libmagicpig_hip.so")
import magicpig_amd as mp, magicpig_amd._lib as L
for name, (n, M, D, K, Lt, Hkv, H) in {"cfg1": (97932, 98304, 128, 10, 150, 8, 32), "cfg4": (131004, 131072, 128, 11, 300, 1, 8)}.items():
    torch.manual_seed(0)
    W = torch.randn((D, K * Lt), device="cuda").to(torch.bfloat16)
    sh = mp.SimHash(W, K, Lt)
    keys = torch.randn((Hkv, n, D), device="cuda").to(torch.bfloat16)
    codes = sh.keys(keys)
    lsh = mp.LSH(); lsh.alloc(K, Lt, 1, H, Hkv, 1, M)
    lsh.fastfill(0, 0, codes); torch.cuda.synchronize()
    stamp = torch.zeros(64, dtype=torch.int64, device="cuda")
    L.check(L.lib().mp_debug_set_stamp_buffer(L.ptr(stamp)))
    acc = []
    for r in range(5):
        stamp.zero_(); lsh.fastfill(0, 0, codes); torch.cuda.synchronize(); acc.append(stamp.cpu().numpy().copy())
    L.check(L.lib().mp_debug_set_stamp_buffer(None))
    a = np.array(acc).astype(float)[:, 50:59] * 0.01                      # 100 MHz -> us
    d = np.median(np.diff(a, axis=1), axis=0)
    print(f"{name}: workgroup 0 of lsh_build_kernel, us: row histogram {d[0]:.1f} | ... to the third tile's top {d[1]:.1f} | validate + zero {d[2]:.1f} | "
          f"count {d[3]:.1f} | scan {d[4]:.1f} | rank {d[5]:.1f} | write-out {d[6]:.1f} | ... to the kernel's end {d[7]:.1f}; "
          f"whole kernel {np.median(a[:, 8] - a[:, 0]):.1f}", flush=True)
