"""Phase timestamps of workgroup 0 of the hot kernels at a BASELINE config shape (one layer).
usage: python scripts/phase_times.py [cfg1|cfg2|cfg3|cfg4] [reps]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd._lib as L
# the product's kernels carry no stamp code: load the -DMP_STAMPS=1 build (built on demand; MP_LIB= overrides)
STAMP_LIB = os.path.join(ROOT, "magicpig_amd", "lib", "variants", "stamps", "libmagicpig_hip.so")
if not os.environ.get("MP_LIB"):
    if not os.path.exists(STAMP_LIB):
        import subprocess
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "build_variant.py"), "stamps", "-DMP_STAMPS=1"], check=True)
    L.LIB_PATH = STAMP_LIB
if os.environ.get("MP_LIB"):            # A/B builds of the library (this script only; the product reads no environment)
    L.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
import magicpig_amd as mp
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
dev = torch.device("cuda:0")
NLAYER = 3
server = mp.LSHSparseAttnServer(NLAYER, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for li in range(NLAYER):
    for b in range(B):
        gen = torch.Generator(device=dev).manual_seed(100 * li + b)
        kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
        vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
        server.fill(li, b, kc, vc, P); server.build_table(li, b, P)
stamp = torch.zeros(64, dtype=torch.int64, device=dev)
qs = torch.randn((reps, NLAYER, B, H, 1, D), device=dev).to(torch.bfloat16)
server.collect_nnz = False
for r in range(3):
    for li in range(NLAYER): server.decode(qs[r, li], li)
torch.cuda.synchronize()
L.check(L.lib().mp_debug_set_stamp_buffer(L.ptr(stamp)))
acc = []
for r in range(reps):
    for li in range(NLAYER):
        stamp.zero_()
        server.decode(qs[r, li], li)
        torch.cuda.synchronize()
        acc.append(stamp.cpu().numpy().copy())
L.check(L.lib().mp_debug_set_stamp_buffer(None))
a = np.array(acc).astype(np.float64) * 0.01   # 100 MHz ticks -> us
def seg(name, slots):
    t = a[:, slots]
    d = np.diff(t, axis=1)
    print(f"{name}: total {np.median(t[:, -1] - t[:, 0]):.2f} us; phases " + ", ".join(f"{x:.2f}" for x in np.median(d, axis=0)))
seg("simhash  [start|planes+rows staged|mfma+guard|pack]", [0, 1, 2, 3])
seg("retrieve [start|probe+zero|chunk table|ids+atomics|scan|emit]", [16, 17, 18, 19, 20, 21])
seg("  fused hash prologue [start|q arrived|row normalised|barrier|pass0|pass1|barrier|pieces in]", [16, 28, 29, 22, 23, 24, 27, 17])
seg("partial  [start|prefix|gathers issued|qk+reduce|transform|pv|combine]", [32, 33, 34, 35, 36, 37, 38])
print("kernel-to-kernel (WG0 start to WG0 start): simhash->retrieve %.2f, retrieve->partial %.2f us" % (
    np.median(a[:, 16] - a[:, 0]), np.median(a[:, 32] - a[:, 16])))
print("lib", L.LIB_PATH)
seg("fused decode kernel [start|hash+probe|table streamed|scan|emit|ids staged|K gathers issued|qk|transform|pv|ticket|end]",
    [16, 17, 19, 20, 21, 33, 34, 35, 36, 37, 38, 39])
seg("hand-off [pv|states merged in LDS|partial stored + acked|ticket drawn]", [37, 40, 41, 38])
