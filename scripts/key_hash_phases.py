import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd._lib as _L0
_L0.LIB_PATH = os.path.join(ROOT, 'magicpig_amd', 'lib', 'variants', 'stamps', 'libmagicpig_hip.so')   # -DMP_STAMPS=1 build (scripts/build_variant.py stamps -DMP_STAMPS=1)
import magicpig_amd as mp, magicpig_amd._lib as L
n, D, K, Lt, Hkv = 97932, 128, 10, 150, 8
W = torch.randn((D, K*Lt), device="cuda").to(torch.bfloat16)
sh = mp.SimHash(W, K, Lt)
keys = torch.randn((Hkv, n, D), device="cuda").to(torch.bfloat16)
sh.keys(keys); torch.cuda.synchronize()
stamp = torch.zeros(64, dtype=torch.int64, device="cuda")
L.check(L.lib().mp_debug_set_stamp_buffer(L.ptr(stamp)))
acc=[]
for r in range(5):
    stamp.zero_(); sh.keys(keys[:1].contiguous()); torch.cuda.synchronize(); acc.append(stamp.cpu().numpy().copy())
L.check(L.lib().mp_debug_set_stamp_buffer(None))
a = np.array(acc).astype(float)*0.01
d = np.median(np.diff(a[:, [40,41,45,42,43,44]], axis=1), axis=0)
print("WG0 phases us [prologue | first 4 tiles | remaining tiles of the chunk | exact pass | store]:", np.round(d,2))
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); sh.keys(keys); e1.record(); torch.cuda.synchronize(); print("8 heads ms", e0.elapsed_time(e1))
