import sys, time, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import magicpig_amd as mp, magicpig_amd._lib as L_
from test_gpu_parity import _fused_server
server, _ = _fused_server(mp, 8, 32, 8, 3000, 3072, 128, 10, 150, 5)
q = torch.randn((8, 32, 1, 128), device="cuda").to(torch.bfloat16)
server.collect_nnz = False
for mode in (0, 1, 2, 1, 0):
    L_.set_option("decode_quad_hash", mode)
    for _ in range(20): server.decode(q, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): server.decode(q, 0)
    torch.cuda.synchronize(); print("quad", mode, round((time.perf_counter() - t0) / 200 * 1e6, 2), "us per launch (eager)")
