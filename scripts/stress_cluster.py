"""Stress of the cross-workgroup exchanges of the decode kernel (split hash, cluster hand-off) at a BASELINE config
shape: a hipGraph of 30 back-to-back launches of ONE layer is replayed `reps` times on a rotating set of queries;
every replay's output, LSE and nnz must equal the eager result of that query bit for bit, and the device flags
(mp_attn_check) must stay quiet.  With `contend` a second stream keeps the CUs busy with large GEMMs meanwhile, so
that members of a cluster are not resident together (the bounded waits time out and the fallbacks run).
`lean` (round 6): the replayed launches are mp_decode_sparse_layer_ex(MP_DECODE_NO_BYPRODUCTS) -- the claim list's spins and
the last wave's hand-off under replay and contention; counts bit for bit, outputs / LSE within the parity bar (the order in
which tokens enter the softmax is not reproducible there).
usage: python scripts/stress_cluster.py [cfg1] [reps] [contend] [lean]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd as mp
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
contend = "contend" in sys.argv[3:]
lean = "lean" in sys.argv[3:]
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
dev = torch.device("cuda:0")
server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for b in range(B):
    gen = torch.Generator(device=dev).manual_seed(b)
    kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    server.fill(0, b, kc, vc, P); server.build_table(0, b, P)
NQ = 8
qs = torch.randn((NQ, B, H, 1, D), device=dev).to(torch.bfloat16)
ref = []
for i in range(NQ):
    o, l = server.decode(qs[i], 0)
    ref.append((o.clone(), l.clone(), server.nnz.clone()))
server.by_products = not lean
q_static = qs[0].clone()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    server.decode(q_static, 0)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(30):
        o_g, l_g = server.decode(q_static, 0)
bad = 0
noise = torch.cuda.Stream()
xa = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
for r in range(reps):
    i = r % NQ
    q_static.copy_(qs[i])
    if contend:
        with torch.cuda.stream(noise):
            for _ in range(2):
                xb = xa @ xa
    graph.replay()
    torch.cuda.synchronize()
    if lean:
        ok = torch.equal(server.nnz, ref[i][2]) and \
            torch.allclose(o_g.float(), ref[i][0].float(), rtol=2 ** -7, atol=2e-4) and torch.allclose(l_g, ref[i][1], atol=1e-3)
    else:
        ok = torch.equal(o_g, ref[i][0]) and torch.equal(l_g, ref[i][1]) and torch.equal(server.nnz, ref[i][2])
    if not ok:
        bad += 1
server.attn_server.check()
print(f"{name}{' (lean launches)' if lean else ''}: R = {server.lsh_retriever.R}; {reps} replays x 30 launches{' against concurrent GEMMs' if contend else ''}, "
      f"{bad} replays differed from the eager result; device flags quiet")
sys.exit(1 if bad else 0)
