"""Cost of the host-buffer compatibility mode at a BASELINE config shape: the decode lines of the UNCHANGED
models/attnserver.py:264-308 (q hash on the GPU, codes + query copied to pinned CPU tensors, batch_retrieve and
attention_wrapper on CPU tensors, output + LSE copied back) against the device-resident entries, per layer.
usage: python scripts/host_mode_times.py [cfg1|cfg2|...] [reps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd._lib as L
if os.environ.get("MP_LIB"):            # A/B builds of the library (this script only; the product reads no environment)
    L.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
import magicpig_amd as mp
from bench import CONFIGS
for kv in os.environ.get("MP_OPTIONS", "").split(","):      # e.g. MP_OPTIONS=host_spin_wait=1 (this script only)
    if "=" in kv:
        L.set_option(kv.split("=")[0], int(kv.split("=")[1]))

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
dev = torch.device("cuda:0")
BH = B * H
server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for b in range(B):
    gen = torch.Generator(device=dev).manual_seed(b)
    kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    server.fill(0, b, kc, vc, P); server.build_table(0, b, P)
qs = torch.randn((reps + 5, B, H, 1, D), device=dev).to(torch.bfloat16)
# pinned CPU tensors of the reference (attnserver.py:59-66)
pin = lambda *shape, dtype: torch.zeros(shape, dtype=dtype).pin_memory()
pinned_hashcode, pinned_query = pin(BH, Lt, dtype=torch.int32), pin(BH, D, dtype=torch.bfloat16)
# (results_lsh_cpu and nnz are PAGEABLE in the reference, :59-60; argv[3] = "pinned" pins them for comparison)
if len(sys.argv) > 3 and sys.argv[3] == "pinned":
    results, nnz = pin(BH, M, dtype=torch.int32), pin(BH, dtype=torch.int32)
else:
    results, nnz = torch.zeros((BH, M), dtype=torch.int32), torch.zeros((BH,), dtype=torch.int32)
output, mve = pin(BH, D, dtype=torch.bfloat16), pin(2, BH, dtype=torch.float32)
out_cuda, lse_cuda = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev), torch.zeros((BH,), device=dev)
lsh, srv, hasher = server.lsh_retriever, server.attn_server, server.hasher

parts = {}
def lap(name, t):
    now = time.perf_counter()
    parts[name] = parts.get(name, 0.0) + (now - t)
    return now

def host_layer(q):
    t = time.perf_counter()
    codes, _ = hasher.query(q.reshape(BH, D))                       # :264-270 on the GPU
    pinned_hashcode.copy_(codes)                                    # :272
    t = lap("q hash + codes -> pinned (blocking D2H)", t)
    pinned_query.copy_(q.reshape(BH, D))                            # :273
    t = lap("query -> pinned (blocking D2H)", t)
    lsh.batch_retrieve(0, pinned_hashcode, results, nnz)            # :299
    t = lap("batch_retrieve (host buffers)", t)
    qn = pinned_query.float().norm(p=2, dim=-1)
    t = lap("||q|| on the CPU (torch)", t)
    srv.attention_wrapper(0, K, Lt, output, mve, pinned_query, qn, results, nnz)   # :300
    t = lap("attention_wrapper (host buffers)", t)
    lse_cuda.copy_(mve[1], non_blocking=True)                       # :302-303
    out_cuda.copy_(output, non_blocking=True)
    torch.cuda.synchronize()
    t = lap("out + LSE -> device, synchronize", t)

d_res = torch.zeros((BH, M), dtype=torch.int32, device=dev)
d_nnz = torch.zeros((BH,), dtype=torch.int32, device=dev)
d_out = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev)
d_mve = torch.zeros((2, BH), dtype=torch.float32, device=dev)

def device_three_call(q):
    codes, qn = hasher.query(q.reshape(BH, D))
    lsh.batch_retrieve(0, codes, d_res, d_nnz)
    srv.attention_wrapper(0, K, Lt, d_out, d_mve, q.reshape(BH, D), qn, d_res, d_nnz)
    torch.cuda.synchronize()

def device_one_launch(q):
    server.decode(q, 0)
    torch.cuda.synchronize()

for fn, label in ((host_layer, "host buffers (unchanged attnserver.py decode lines)"),
                  (device_three_call, "device buffers, three calls"),
                  (device_one_launch, "device buffers, mp_decode_sparse_layer")):
    for i in range(5):
        fn(qs[i])
    parts.clear()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(qs[5 + i])
    dt = (time.perf_counter() - t0) / reps * 1e6
    print(f"{label:58s} {dt:9.1f} us per layer (eager, synchronised after every layer)")
    if fn is host_layer:
        for k, v in parts.items():
            print(f"    {k:54s} {v / reps * 1e6:9.1f} us")
host_layer(qs[0]); a = out_cuda.clone(); device_three_call(qs[0])
# (the host path takes ||q|| from torch's CPU norm, attnserver.py:300: f32 sums in another order than the kernel's)
print("max |host - device| output:", float((a.float() - d_out.float()).abs().max()), " nnz equal:", torch.equal(nnz.cuda(), d_nnz))
import magicpig_amd._lib as _L
print("attention calls served:", {n: _L.get_option("host_fast_" + n) for n in ("hits", "edited", "unpaired")},
      "by the launch behind the retrieve:", {n: _L.get_option("host_" + n) for n in ("spec_hits", "spec_misses", "flag_timeouts")})
calls = max(1, _L.get_option("host_ret_calls"))
print("inside batch_retrieve (host buffers), us per call:", {n: round(_L.get_option("host_ret_ns_" + n) / calls / 1e3, 1) for n in ("enqueue", "wait", "copy")}, f"over {calls} calls")
