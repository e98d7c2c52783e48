"""Phase stamps of EVERY workgroup of the decode kernel at a BASELINE config shape: per phase boundary the
median / max over the workgroups of (stamp - earliest kernel start), i.e. where the critical path runs.
usage: python scripts/phase_spread.py [cfg0|cfg1|cfg2|cfg3|cfg4] [reps] [randn|clustered|skewed]"""
import os, sys
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
if os.environ.get("MP_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
import magicpig_amd as mp
from bench import CONFIGS, synth_kv

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
data = sys.argv[3] if len(sys.argv) > 3 else "randn"
# "graph" [layers]: the stamps of the LAST launch of a hipGraph of `layers` back-to-back launches over as many DISTINCT
# layers (bench.py's timed region: 30 at the 8B shape) instead of eager launches over 2 layers with a synchronisation
# in between -- variants that win under idle-machine stamps have lost in the replayed graph three times (R3-6, R3-12, R4-1)
use_graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
glayers = int(sys.argv[5]) if len(sys.argv) > 5 else 30
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
dev = torch.device("cuda:0")
NLAYER = glayers if use_graph else 2
server = mp.LSHSparseAttnServer(NLAYER, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for li in range(NLAYER):
    for b in range(B):
        gen = torch.Generator(device=dev).manual_seed(100 * li + b)
        kc, vc = synth_kv(data, P, Hkv, D, dev, gen)
        server.fill(li, b, kc, vc, P); server.build_table(li, b, P)
R = server.lsh_retriever.R
BHp = (B * H + 7) // 8 * 8 if R > 1 else B * H     # the launch pads the heads per rank to a multiple of 8
grid = BHp * R
STRIDE = 64
stamp = torch.zeros(grid * STRIDE, dtype=torch.int64, device=dev)
qs = torch.randn((reps, NLAYER, B, H, 1, D), device=dev)
if data != "randn":     # heavy hitters (bench.py --queries heavy)
    G = H // Hkv
    bi = torch.arange(B, device=dev)[None, :, None].expand(reps, B, H)
    gi = (torch.arange(H, device=dev) // G)[None, None, :].expand(reps, B, H)
    for li in range(NLAYER):
        kcen = server.attn_server.get_key_cache(li)
        j = torch.randint(0, P - 68, (reps, B, H), device=dev)
        qs[:, li, :, :, 0] = 0.5 * qs[:, li, :, :, 0] + 3.0 * kcen[bi, gi, j].float()
qs = qs.to(torch.bfloat16)
server.collect_nnz = False
server.by_products = os.environ.get("MP_LEAN", "0") != "1"      # MP_LEAN=1: the MP_DECODE_NO_BYPRODUCTS launch
for r in range(3):
    for li in range(NLAYER): server.decode(qs[r % reps, li], li)
torch.cuda.synchronize()
for kv in os.environ.get("MP_OPTIONS", "").split(","):      # e.g. MP_OPTIONS=decode_split_hash=0,decode_kn_payload=0
    if "=" in kv:
        L.set_option(kv.split("=")[0], int(kv.split("=")[1]))
L.set_option("stamp_stride", STRIDE)
L.check(L.lib().mp_debug_set_stamp_buffer(L.ptr(stamp)))
slots = [16, 28, 22, 23, 27, 42, 43, 44, 45, 17, 19, 20, 33, 34, 35, 36, 37, 40, 41, 38, 39]
names = ["start", "q row in", "normalised", "own unit", "hashed", "slots issued", "first slot in", "last slot in",
         "wave counted", "pieces in", "stream done", "own words scanned", "ids staged", "gathers issued",
         "qk", "transform", "pv", "states met", "stores acked", "ticket", "end (merger)"]
acc = []
perwave = []
if use_graph:
    q_static = qs[0].clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for li in range(NLAYER): server.decode(q_static[li], li)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for li in range(NLAYER): server.decode(q_static[li], li)
    for r in range(reps):
        q_static.copy_(qs[r])
        graph.replay(); torch.cuda.synchronize()          # warm replay
        stamp.zero_()
        q_static.copy_(qs[(r + 1) % reps])
        graph.replay(); torch.cuda.synchronize()
        raw = stamp.cpu().numpy().reshape(grid, STRIDE)
        a = raw[:, slots].astype(np.float64) * 0.01
        a[a == 0] = np.nan
        if os.environ.get("MP_LEAN", "0") == "1":      # per-wave: (own fold done << 10 | list length) in slots 48 .. 63
            pw = raw[:, 48:64]
            t = (pw >> 10).astype(np.float64) * 0.01
            t[pw == 0] = np.nan
            perwave.append((t - np.nanmin(a[:, 0]), (pw & 1023).astype(np.int64)))
        # every launch of the replay overwrote the slots it passed: keep the workgroups whose stamps are those of ONE
        # launch (monotone), relative to the last launch's first start
        acc.append(a - np.nanmin(a[:, 0]))
else:
  for r in range(reps):
    for li in range(NLAYER):
        stamp.zero_()
        server.decode(qs[r, li], li)
        torch.cuda.synchronize()
        a = stamp.cpu().numpy().reshape(grid, STRIDE)[:, slots].astype(np.float64) * 0.01
        a[a == 0] = np.nan
        acc.append(a - np.nanmin(a[:, 0]))
L.check(L.lib().mp_debug_set_stamp_buffer(None))
L.set_option("stamp_stride", 0)
a = np.array(acc)                    # [runs, grid, phases]
print(f"{name} ({data}{', last launch of a replayed graph of %d layers' % NLAYER if use_graph else ''}): grid {grid} workgroups (R = {R}); us after the first workgroup's start, median over runs of the per-launch")
print(f"{'phase':>16} {'min':>7} {'median':>7} {'p90':>7} {'max':>7}   rank-0 median / other ranks median")
BH = B * H
for i, nm in enumerate(names):
    x = a[:, :, i]
    st = [np.nanmedian(f(x, axis=1)) for f in (np.nanmin, np.nanmedian, lambda v, axis: np.nanpercentile(v, 90, axis=axis), np.nanmax)]
    r0 = np.nanmedian(x[:, :BHp]); ro = np.nanmedian(x[:, BHp:]) if R > 1 else float("nan")
    print(f"{nm:>16} {st[0]:7.2f} {st[1]:7.2f} {st[2]:7.2f} {st[3]:7.2f}   {r0:7.2f} / {ro:7.2f}")

if perwave:
    t = np.array([p[0] for p in perwave]); n = np.array([p[1] for p in perwave])      # [runs, grid, 16]
    print("per-wave own fold done (us): median %.2f  p90 %.2f  p99 %.2f  max(median over runs of per-launch max) %.2f" % (
        np.nanmedian(t), np.nanpercentile(t, 90), np.nanpercentile(t, 99), np.nanmedian(np.nanmax(t, axis=(1, 2)))))
    for lo, hi in ((0, 8), (9, 16), (17, 24), (25, 32), (33, 1023)):
        sel = (n >= lo) & (n <= hi)
        if sel.any():
            print("  list length %3d..%3d: %5.1f %% of waves, fold done median %.2f  p90 %.2f  max %.2f" % (
                lo, hi, 100.0 * sel.mean(), np.nanmedian(t[sel]), np.nanpercentile(t[sel], 90), np.nanmax(t[sel])))
    wg_last = np.nanmax(t, axis=2)                      # [runs, grid]: the workgroup's slowest wave
    wg_n = n.sum(axis=2)
    print("workgroup's slowest wave: median %.2f  p90 %.2f  max %.2f;  tokens per workgroup: median %d  max %d" % (
        np.nanmedian(wg_last), np.nanpercentile(wg_last, 90), np.nanmedian(np.nanmax(wg_last, axis=1)), np.median(wg_n), wg_n.max()))
    big = wg_n >= np.percentile(wg_n, 90)
    print("  workgroups in the top decile by tokens: slowest wave median %.2f;  the rest: %.2f" % (
        np.nanmedian(wg_last[big]), np.nanmedian(wg_last[~big])))
    late = wg_last >= np.nanpercentile(wg_last, 95)
    print("  the 5 %% slowest workgroups: tokens median %d, their slowest wave's list length median %d" % (
        np.median(wg_n[late]), np.median(np.take_along_axis(n, np.nanargmax(np.where(np.isnan(t), -1, t), axis=2)[..., None], axis=2)[..., 0][late])))
