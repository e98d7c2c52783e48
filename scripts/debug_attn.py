import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, synth, oracle, magicpig_amd as mp
B,H,Hkv,n,M,D,K,L = 1,4,2,256,320,128,4,8
keys,kns,vals,W,qb = cases.case_inputs(41,B,H,Hkv,n,D,K,L)
t = lambda b: synth.to_torch_bf16(np.ascontiguousarray(b)).cuda()
srv = mp.SparseAttentionServer(); srv.alloc(1,H,Hkv,D,B,M)
srv.fill(0,0,t(keys[0]),t(vals[0]),torch.from_numpy(kns[0]).cuda())
nnz = np.array([70, 64, 5, 130], np.int32)
ind = np.zeros((B*H, M), np.int32)
for h in range(B*H): ind[h,:nnz[h]] = np.arange(nnz[h])
_, qn = oracle.simhash_query(qb, W, K, L)
osrv = oracle.SparseAttentionServer(clamp_cos=1); osrv.alloc(1,H,Hkv,D,B,M); osrv.fill(0,0,keys[0],vals[0],kns[0])
qf = synth.bf16_bits_to_f32(qb)
for dense in (False, True):
  for qdt in ("bf16","f32"):
    out = torch.zeros((B*H,D),dtype=torch.bfloat16,device="cuda"); mve = torch.zeros((2,B*H),device="cuda")
    q = t(qb) if qdt=="bf16" else torch.from_numpy(qf).cuda()
    oo = np.zeros((B*H,D),np.uint16); om = np.zeros((2,B*H),np.float32)
    if dense:
        srv.full_attention(0,out,mve,q,torch.from_numpy(nnz).cuda())
        osrv.full_attention(0,oo,om,qf,nnz)
    else:
        srv.attention_wrapper(0,K,L,out,mve,q,torch.from_numpy(qn).cuda(),torch.from_numpy(ind).cuda(),torch.from_numpy(nnz).cuda())
        osrv.attention_wrapper(0,K,L,oo,om,qb,qn,ind,nnz)
    pg = srv.get_score().reshape(B*H,M).cpu().numpy(); po = osrv.get_score().reshape(B*H,M)
    errs = [float(np.abs(pg[h,:nnz[h]]-po[h,:nnz[h]]).max()) for h in range(B*H)]
    print("dense",dense,"q",qdt,"max prob err per head",errs, "lse gpu",mve[1].tolist(),"orc",om[1].tolist())
    if not dense and qdt=="bf16":
        h=1; z=nnz[h]
        print(" gpu logp", np.round(np.log(pg[h,:16]),2)); print(" orc logp", np.round(np.log(po[h,:16]),2))
