"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference compiled into oracle/_ref by
oracle/build_ref.py, and torch CPU):

    OMP_THREAD_LIMIT=8 python tests/golden/make_golden.py

Sources of truth:
  * query / key SimHash: the literal torch-CPU restatement of models/attnserver.py:264-270
    and :159-168 (the reference computes these with torch ops, not in its C++ libraries);
  * tables, retrieve, sparse attention: the reference's compiled C++ (library/lsh/lsh.cc,
    library/sparse_attention/sparse_attention.cc) driven through its own pybind11 API.
Inputs are regenerated from seeds by tests/synth.py (integer-only, platform independent);
only OUTPUTS are stored (np.savez_compressed).  Sorting uses torch.sort(stable=True) so the
bucket-internal order -- unspecified in the reference (models/attnserver.py:187 uses an
unstable sort) -- is reproducible with numpy's stable argsort.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from cases import (DATA_KINDS, FILL_CENTRE, FULL_DENSE_CASES, LLAMA_OPS, WINDOW_MERGE, case_inputs,  # noqa: E402
                   fill_centre_inputs, full_dense_inputs, full_dense_seed, llama_ops_inputs, window_merge_inputs)
from oracle.build_ref import load_ref, ref_uses_bf16_family  # noqa: E402


# ----------------------------------------------------------------- torch restatements

def torch_qhash(q: torch.Tensor, hash_func: torch.Tensor, K: int, L: int) -> torch.Tensor:
    """models/attnserver.py:264-270, verbatim on CPU tensors."""
    binary_pack = torch.Tensor([int(2 ** i) for i in range(K)]).to(dtype=torch.float16)
    norm_q = q.reshape(-1, q.shape[-1])
    norm_q = norm_q / norm_q.norm(p=2, dim=-1, keepdim=True)
    q_hashcode = torch.matmul(norm_q, hash_func).gt(0)
    q_hashcode = q_hashcode.reshape(-1, K).to(torch.float16)
    q_hashcode = torch.mv(q_hashcode, binary_pack).int()
    return q_hashcode.reshape(-1, L)


def torch_khash(offload_key: torch.Tensor, hash_func: torch.Tensor, K: int, L: int) -> torch.Tensor:
    """models/attnserver.py:159-168 (one chunk), offload_key bf16 [Hkv, n, D] -> int16 [Hkv, L, n]."""
    Hkv = offload_key.shape[0]
    binary_pack = torch.Tensor([int(2 ** i) for i in range(K)]).to(dtype=torch.float16)
    hash_code = torch.matmul(offload_key, hash_func)
    hash_code = hash_code > 0
    hash_code = hash_code.reshape(-1, K).to(torch.float16)
    hash_code = torch.mv(hash_code, binary_pack)
    hash_code = hash_code.reshape(Hkv, -1, L)
    return hash_code.transpose(1, 2).contiguous().to(torch.int16)


# ----------------------------------------------------------------- case builders

def exact_sign_ties(kcodes: torch.Tensor, keys: np.ndarray, W: np.ndarray, K: int, L: int):
    """Where does torch's bf16 GEMM (f32 accumulation in the library's own order, attnserver.py:159-162)
    disagree with the sign of the EXACT dot product?  Only where the exact value is zero or within f32
    rounding of it -- there the reference's code depends on the GEMM library's summation order (the
    reference runs this matmul through cuBLAS), so the hash is defined by the exact sign.
    Products of two bf16 numbers and their 128-term sums are exact in f64.  Returns (patched codes,
    ties int64 [T, 6] = (b, g, l, t, bit, torch_bit), exact dot products f64 [T])."""
    B, Hkv, _, n = kcodes.shape
    Wf = synth.bf16_bits_to_f32(W).astype(np.float64)                       # [D, K*L]
    patched = kcodes.clone()
    ties, dots = [], []
    for b in range(B):
        for g in range(Hkv):
            dot = synth.bf16_bits_to_f32(keys[b][g]).astype(np.float64) @ Wf          # [n, K*L] exact
            bits = (dot > 0).reshape(n, L, K)
            exact = (bits * (1 << np.arange(K))).sum(-1).T.astype(np.int16)           # [L, n]
            tc = kcodes[b, g].numpy()
            for l, t in np.argwhere(exact != tc):
                x = int(exact[l, t]) ^ int(tc[l, t])
                for bit in range(K):
                    if (x >> bit) & 1:
                        ties.append((b, g, l, t, bit, (int(tc[l, t]) >> bit) & 1))
                        dots.append(dot[t, l * K + bit])
            patched[b, g] = torch.from_numpy(exact)
    return patched, np.array(ties, np.int64).reshape(-1, 6), np.array(dots, np.float64)


def run_pipeline(name, seed, B, H, Hkv, n, M, D, K, L, ref_lsh, ref_attn, out, data="randn"):
    keys, kns, vals, W, qb = case_inputs(seed, B, H, Hkv, n, D, K, L, data)
    Wt = synth.to_torch_bf16(W)
    # --- SimHash (torch restatement)
    qcodes = torch_qhash(synth.to_torch_bf16(qb), Wt, K, L).contiguous()
    kcodes = torch.stack([torch_khash(synth.to_torch_bf16(keys[b]), Wt, K, L) for b in range(B)])
    kcodes, ties, tie_dots = exact_sign_ties(kcodes, keys, W, K, L)
    assert len(ties) <= 1e-7 * kcodes.numel() * K + 1 and (len(ties) == 0 or np.abs(tie_dots).max() < 1e-5)
    # --- tables + retrieve (compiled reference)
    lsh = ref_lsh.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        sc, si = kcodes[b].sort(stable=True)
        lsh.fill(0, b, sc.contiguous(), si.int().contiguous())
    results = torch.zeros((B * H, M), dtype=torch.int32)
    nnz = torch.zeros((B * H,), dtype=torch.int32)
    lsh.batch_retrieve(0, qcodes, results, nnz)
    mask = lsh.get_mask().clone().reshape(B * H, M)
    # --- sparse attention (compiled reference, bf16 query as models/attnserver.py:300)
    srv = ref_attn.SparseAttentionServer()
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, synth.to_torch_bf16(keys[b]), synth.to_torch_bf16(vals[b]),
                 torch.from_numpy(kns[b]))
    q_t = synth.to_torch_bf16(qb)
    qn = q_t.float().norm(p=2, dim=-1)
    output = torch.zeros((B * H, D), dtype=torch.bfloat16)
    mve = torch.zeros((2, B * H), dtype=torch.float32)
    srv.attention_wrapper(0, K, L, output, mve, q_t, qn, results, nnz)
    probs = srv.get_score().reshape(B * H, M)
    nz = nnz.numpy()
    out[name] = dict(
        meta=np.array([seed, B, H, Hkv, n, M, D, K, L], np.int64),
        qcodes=qcodes.numpy().astype(np.int32),
        kcodes_sha=np.frombuffer(hashlib.sha256(kcodes.numpy().tobytes()).digest(), np.uint8),
        kcodes_head0_table0=kcodes[0, 0, 0].numpy().copy(),
        kcodes_ties=ties, kcodes_tie_dots=tie_dots,
        nnz=nz.copy(),
        results_ref_order=np.concatenate([results[h, :nz[h]].numpy() for h in range(B * H)]),
        mask_hist=np.stack([np.bincount(mask[h].numpy().astype(np.int64), minlength=3)
                            for h in range(B * H)]),
        qnorm=qn.numpy().copy(),
        out_bits=output.view(torch.int16).numpy().view(np.uint16).copy(),
        mve=mve.numpy().copy(),
        probs=np.concatenate([probs[h, :nz[h]].numpy() for h in range(B * H)]),
    )
    if data != "randn":
        out[name]["data"] = np.array(DATA_KINDS.index(data), np.int64)
    print(f"{name}: nnz mean {nz.mean():.1f} min {nz.min()} max {nz.max()}; "
          f"{len(ties)} of {kcodes.numel() * K} key sign bits are summation-order ties")


def run_qhash_only(name, seed, R, D, K, L, out):
    qb = synth.normal_bf16_bits(seed, (R, D))
    W = synth.normal_bf16_bits(seed + 7, (D, K * L))
    codes = torch_qhash(synth.to_torch_bf16(qb), synth.to_torch_bf16(W), K, L)
    out[name] = dict(meta=np.array([seed, R, D, K, L], np.int64), qcodes=codes.numpy().astype(np.int32))
    print(f"{name}: done")


def run_lsh_edge(name, seed, ref_lsh, out):
    """Hand-built queries on random tables: an empty-result head, a head that selects a known
    token, heads sharing a kv group, re-query on the same tables (mask reset, lsh/test.py:59-76)."""
    K, L, H, Hkv, B, n, M = 8, 24, 4, 2, 2, 200, 264
    NB = 1 << K
    codes = synth.randint(seed, 0, NB, (B, Hkv, L, n)).astype(np.int16)
    lsh = ref_lsh.LSH()
    lsh.alloc(K, L, 2, H, Hkv, B, M)
    kc = torch.from_numpy(codes)
    for b in range(B):
        sc, si = kc[b].sort(stable=True)
        lsh.fill(1, b, sc.contiguous(), si.int().contiguous())
    G = H // Hkv
    q = synth.randint(seed + 1, 0, NB, (B * H, L)).astype(np.int32)
    q[1] = codes[0, 1 // G, :, 5]          # head 1 copies token 5's codes: count == L
    q[2, :] = codes[0, 2 // G, :, 17]      # head 2: token 17 ...
    q[2, 1:] = (q[2, 1:] + 1) % NB         # ... but only table 0 is guaranteed to collide
    for l in range(L):                     # head 3 probes an EMPTY bucket in every table: nnz == 0
        present = np.zeros(NB, bool)
        present[codes[0, 3 // G, l]] = True
        q[3, l] = int(np.flatnonzero(~present)[0])
    runs = []
    for rep in range(2):
        qq = torch.from_numpy(q if rep == 0 else ((q + 3) % NB).astype(np.int32)).contiguous()
        results = torch.zeros((B * H, M), dtype=torch.int32)
        nnz = torch.zeros((B * H,), dtype=torch.int32)
        lsh.batch_retrieve(1, qq, results, nnz)
        nz = nnz.numpy()
        runs.append((nz.copy(),
                     np.concatenate([np.sort(results[h, :nz[h]].numpy()) for h in range(B * H)])))
    out[name] = dict(meta=np.array([seed, K, L, H, Hkv, B, n, M], np.int64), q=q,
                     nnz0=runs[0][0], sorted0=runs[0][1], nnz1=runs[1][0], sorted1=runs[1][1])
    print(f"{name}: nnz {runs[0][0].tolist()} / {runs[1][0].tolist()}")


def run_attn_edge(name, seed, ref_attn, out):
    """attention_wrapper on explicit index lists: nnz = 0, 1, 15, 16, 17, 33 and a long head;
    random `ind` = prefix of a permutation like library/sparse_attention/test_sparse.py:52-55."""
    K, L, H, Hkv, B, n, M, D = 10, 150, 8, 2, 1, 512, 640, 128
    keys, kns, vals, W, qb = case_inputs(seed, B, H, Hkv, n, D, K, L)
    nnz_list = [0, 1, 15, 16, 17, 33, 200, 511]
    ind = np.zeros((B * H, M), np.int32)
    for h, z in enumerate(nnz_list):
        perm = np.argsort(synth.u64(seed + 50 + h, n), kind="stable").astype(np.int32)
        ind[h, :z] = perm[:z]
    srv = ref_attn.SparseAttentionServer()
    srv.alloc(1, H, Hkv, D, B, M)
    srv.fill(0, 0, synth.to_torch_bf16(keys[0]), synth.to_torch_bf16(vals[0]),
             torch.from_numpy(kns[0]))
    q_t = synth.to_torch_bf16(qb)
    qn = q_t.float().norm(p=2, dim=-1)
    output = torch.zeros((B * H, D), dtype=torch.bfloat16)
    mve = torch.zeros((2, B * H), dtype=torch.float32)
    nnz = torch.tensor(nnz_list, dtype=torch.int32)
    srv.attention_wrapper(0, K, L, output, mve, q_t, qn, torch.from_numpy(ind), nnz)
    probs = srv.get_score().reshape(B * H, M)
    out[name] = dict(meta=np.array([seed, K, L, H, Hkv, B, n, M, D], np.int64),
                     nnz=np.array(nnz_list, np.int32), qnorm=qn.numpy().copy(),
                     out_bits=output.view(torch.int16).numpy().view(np.uint16).copy(),
                     mve=mve.numpy().copy(),
                     probs=np.concatenate([probs[h, :z].numpy() for h, z in enumerate(nnz_list)]))
    print(f"{name}: lse {mve[1].tolist()}")


def cfg1_codes(seed, Hkv, L, n, NB):
    return synth.randint(seed, 0, NB, (Hkv, L, n)).astype(np.int16)


def run_cfg1_retrieve_sha(name, seed, ref_lsh, out):
    """BASELINE cfg-1-shaped single layer (B=1, H=32, Hkv=8, n=97932, M=98304, K10 L150) with
    uniformly random codes (as library/lsh/test.py:30,36); only SHA-256 of the sorted selected
    ids + nnz is stored so the big case is pinned without a big file."""
    K, L, H, Hkv, B, n, M = 10, 150, 32, 8, 1, 97932, 98304
    NB = 1 << K
    codes = cfg1_codes(seed, Hkv, L, n, NB)
    order = np.argsort(codes, axis=-1, kind="stable").astype(np.int32)
    sc = np.take_along_axis(codes, order, axis=-1)
    lsh = ref_lsh.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    lsh.fill(0, 0, torch.from_numpy(sc), torch.from_numpy(order))
    q = synth.randint(seed + 1, 0, NB, (B * H, L)).astype(np.int32)
    results = torch.zeros((B * H, M), dtype=torch.int32)
    nnz = torch.zeros((B * H,), dtype=torch.int32)
    lsh.batch_retrieve(0, torch.from_numpy(q), results, nnz)
    nz = nnz.numpy()
    hsh = hashlib.sha256()
    hsh.update(nz.tobytes())
    for h in range(B * H):
        hsh.update(np.sort(results[h, :nz[h]].numpy()).tobytes())
    out[name] = dict(meta=np.array([seed, K, L, H, Hkv, B, n, M], np.int64), nnz=nz.copy(),
                     sha256=np.frombuffer(hsh.digest(), np.uint8))
    print(f"{name}: nnz mean {nz.mean():.1f}")


def run_full_dense(name, seed, ref_attn, out):
    """SparseAttentionServer::full_attention (sparse_attention.cc:988-1037) of the compiled reference,
    f32 query as library/sparse_attention/test_dense.py:40, one FRESH server per (case, nnz): the
    softmax of the reference covers round_up(nnz, 16) score slots (softmax_kernel_optimized, :249-283:
    the tail mask is never taken because every block start is < nnz), so for nnz % 16 != 0 it also counts
    whatever the score buffer holds behind the list -- zeros on a fresh server (:579-580).  The oracle
    emulates that (`quirks` bit 1) to be pinned on ragged lengths too; the definition the HIP path is
    held to is the softmax over exactly nnz rows."""
    D = 128
    d = dict(meta=np.array([seed, D], np.int64))
    for tag, B, H, Hkv, n, M, nnz_list in FULL_DENSE_CASES:
        keys, vals, q = full_dense_inputs(full_dense_seed(seed, tag, H), B, H, Hkv, n, D)
        kn = np.zeros((Hkv, n), np.float32)           # key norms play no role in the dense path
        for z in nnz_list:
            srv = ref_attn.SparseAttentionServer()
            srv.alloc(1, H, Hkv, D, B, M)
            for b in range(B):
                srv.fill(0, b, synth.to_torch_bf16(keys[b]), synth.to_torch_bf16(vals[b]), torch.from_numpy(kn))
            output = torch.zeros((B * H, D), dtype=torch.bfloat16)
            mve = torch.zeros((2, B * H), dtype=torch.float32)
            nnz = torch.full((B * H,), z, dtype=torch.int32)
            srv.full_attention(0, output, mve, torch.from_numpy(q).reshape(B, H, 1, D), nnz)
            probs = srv.get_score().reshape(B * H, M)
            z16 = (z + 15) & ~15
            d[f"{tag}_z{z}_out"] = output.view(torch.int16).numpy().view(np.uint16).copy()
            d[f"{tag}_z{z}_mve"] = mve.numpy().copy()
            d[f"{tag}_z{z}_probs"] = probs[:, :min(z16, M)].numpy().copy()
        print(f"{name}/{tag}: nnz {nnz_list}")
    out[name] = d


def run_window_merge(name, out):
    """The sparse-layer decode of the reference stated with torch ops on CPU: the in-tree torch statement
    of the LSH-sampled half (evaluations/RULER/pred/attnserver_dist.py:813-851: collision mask, importance
    weight, masked softmax, base-2 LSE = logsumexp / ln 2), exact attention with a base-2 LSE over the
    static window (what BatchDecodeWithPagedKVCacheWrapper.run_return_lse returns,
    models/attnserver.py:293-296) and flashinfer.merge_state (:305-308, attnserver_dist.py:882).
    FlashInfer itself is not in /root/reference (install.sh:4, un-vendored, unpinned); its published
    definition of merge_state on base-2 LSEs is  s = log2(2^sa + 2^sb), v = (2^sa va + 2^sb vb) / 2^s.
    Two deviations from the literal lines, both toward the C++ hot path: q.K is accumulated in f32
    (attnserver_dist.py:843 rounds the bf16 matmul result to bf16; qk_kernel keeps f32,
    sparse_attention.cc:38-103) and softmax(z) is not rounded to bf16 before P.V (:851; wv_kernel keeps
    f32, sparse_attention.cc:321-384).  Also stored: the same attention as ONE softmax over the union of
    the two parts in f64 -- the defining property of merge_state, independent of its formula."""
    import math

    c = WINDOW_MERGE
    seed, B, H, Hkv, D, K, L, n, M = (c[k] for k in ("seed", "B", "H", "Hkv", "D", "K", "L", "n", "M"))
    G = H // Hkv
    keys, kns, vals, W, qb, wk, wv = window_merge_inputs(c)
    Wt = synth.to_torch_bf16(W)
    q = synth.to_torch_bf16(qb)                                            # bf16 [BH, D]
    qcodes = torch_qhash(q, Wt, K, L)                                      # [BH, L]
    BH = B * H
    sp_out = torch.zeros((BH, D), dtype=torch.bfloat16)
    sp_lse = torch.zeros((BH,))
    w_out = torch.zeros((BH, D), dtype=torch.bfloat16)
    w_lse = torch.zeros((BH,))
    mg_out = torch.zeros((BH, D), dtype=torch.bfloat16)
    mg_lse = torch.zeros((BH,))
    joint_out = torch.zeros((BH, D), dtype=torch.float64)
    joint_lse = torch.zeros((BH,), dtype=torch.float64)
    nnz = np.zeros((BH,), np.int32)
    for b in range(B):
        kcodes = torch_khash(synth.to_torch_bf16(keys[b]), Wt, K, L)       # int16 [Hkv, L, n]
        for hh in range(H):
            h, g = b * H + hh, hh // G
            qh = q[h].float()
            # --- attnserver_dist.py:813-851 for one head
            mask = (kcodes[g].int() == qcodes[h][:, None]).int().sum(dim=0) > 1          # :820-821
            nnz[h] = int(mask.sum())
            kf = synth.to_torch_bf16(keys[b][g]).float()
            vf = synth.to_torch_bf16(vals[b][g]).float()
            score = kf @ qh                                                              # :843 (f32)
            cos = score / (torch.from_numpy(kns[b][g]) * qh.norm(p=2))                   # :846
            theta = torch.arccos(cos)
            weight = 1 - theta / torch.pi
            weight = 1 - (1 - weight ** K) ** L - L * ((1 - weight ** K) ** (L - 1)) * (weight ** K)   # :851-852
            z = score / math.sqrt(D) - torch.log(weight + 1e-4)
            z = z.masked_fill(~mask, -torch.inf)
            lse_sp = torch.logsumexp(z, dim=-1) / math.log(2)                            # :848-849
            o_sp = (z.softmax(dim=-1) @ vf).to(torch.bfloat16)
            # --- static window: exact attention + base-2 LSE
            wkf = synth.to_torch_bf16(wk[b][g]).float()
            wvf = synth.to_torch_bf16(wv[b][g]).float()
            zw = (wkf @ qh) / math.sqrt(D)
            lse_w = torch.logsumexp(zw, dim=-1) / math.log(2)
            o_w = (zw.softmax(dim=-1) @ wvf).to(torch.bfloat16)
            # --- flashinfer.merge_state on base-2 LSEs (published definition)
            s = torch.log2(torch.exp2(lse_w.double()) + torch.exp2(lse_sp.double()))
            v = (torch.exp2(lse_w.double() - s) * o_w.double() + torch.exp2(lse_sp.double() - s) * o_sp.double())
            # --- the same thing as one softmax over the union (f64)
            zz = torch.cat([zw.double(), z.double()])
            vv = torch.cat([wvf.double(), vf.double()])
            joint_lse[h] = torch.logsumexp(zz, dim=-1) / math.log(2)
            joint_out[h] = zz.softmax(dim=-1) @ vv
            sp_out[h], sp_lse[h], w_out[h], w_lse[h] = o_sp, lse_sp, o_w, lse_w
            mg_out[h], mg_lse[h] = v.to(torch.bfloat16), s.float()
    bits = lambda t: t.view(torch.int16).numpy().view(np.uint16).copy()
    out[name] = dict(meta=np.array([seed, B, H, Hkv, D, K, L, n, M, c["win_M"], *c["win_rows"]], np.int64),
                     nnz=nnz, sparse_out=bits(sp_out), sparse_lse=sp_lse.numpy().copy(),
                     window_out=bits(w_out), window_lse=w_lse.numpy().copy(),
                     merged_out=bits(mg_out), merged_lse=mg_lse.numpy().copy(),
                     joint_out=joint_out.numpy().astype(np.float32), joint_lse=joint_lse.numpy().astype(np.float32))
    print(f"{name}: nnz {nnz.tolist()} lse window {w_lse[:3].tolist()} sparse {sp_lse[:3].tolist()}")


def run_fill_centre(name, out):
    """models/attnserver.py:133-146 verbatim on CPU tensors (sparse-layer branch of fill): the offloaded keys'
    mean, the centred keys and their norms, for one request.  torch's bf16 mean / norm accumulate in f32 in an
    order of the library's choosing, so next to torch's outputs the fixture lists where they differ from the
    exactly-summed definition (oracle.centre_keys) -- one bf16 ulp at a rounding boundary, by construction."""
    from oracle import oracle as orc

    c = FILL_CENTRE
    k, v = fill_centre_inputs(c)
    T, s_, l_ = c["seq_len"], c["num_sink"], c["num_local"]
    key_cache, value_cache = synth.to_torch_bf16(k), synth.to_torch_bf16(v)
    offload_key = key_cache[s_:T - l_]                                        # :133
    offload_value = value_cache[s_:T - l_]
    offload_key = offload_key.transpose(0, 1).contiguous()                    # :136
    offload_value = offload_value.transpose(0, 1).contiguous()
    avg_k = offload_key.mean(dim=1, keepdim=True)                             # :139
    offload_key = offload_key - avg_k                                         # :142
    kn = offload_key.norm(p=2, dim=-1).float()                                # :143
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    t_avg, t_key, t_kn = bits(avg_k)[:, 0], bits(offload_key), kn.numpy().copy()
    e_avg, e_key, e_val, e_kn = orc.centre_keys(k, v, T, s_, l_)
    assert np.array_equal(e_val, bits(offload_value))
    avg_ties = np.argwhere(t_avg != e_avg)
    # centred keys / norms given TORCH's avg_k must follow the definition exactly, ties in the norm aside
    cen_t = synth.f32_to_bf16_bits((synth.bf16_bits_to_f32(k[s_:T - l_]) - synth.bf16_bits_to_f32(t_avg)[None]).astype(np.float32))
    assert np.array_equal(cen_t.transpose(1, 0, 2), t_key)
    cf = synth.bf16_bits_to_f32(t_key).astype(np.float64)
    kn_exact = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(np.sqrt((cf * cf).sum(-1)).astype(np.float32)))
    kn_ties = np.argwhere(kn_exact != t_kn)
    out[name] = dict(meta=np.array([c["seed"], T, c["Hkv"], c["D"], s_, l_], np.int64),
                     avg_k=t_avg, kn=t_kn, key_sha=np.frombuffer(hashlib.sha256(t_key.tobytes()).digest(), np.uint8),
                     key_head0_tok0=t_key[0, 0].copy(), avg_ties=avg_ties.astype(np.int64),
                     avg_exact_at_ties=e_avg[tuple(avg_ties.T)] if len(avg_ties) else np.zeros((0,), np.uint16),
                     kn_ties=kn_ties.astype(np.int64),
                     kn_exact_at_ties=kn_exact[tuple(kn_ties.T)] if len(kn_ties) else np.zeros((0,), np.float32))
    print(f"{name}: {len(avg_ties)} of {t_avg.size} means and {len(kn_ties)} of {t_kn.size} norms are summation-order ties")


def run_cfg1_skew_sha(name, seed, ref_lsh, ref_attn, out):
    """BASELINE cfg 1 at full size on the CLUSTERED workload (anisotropic clustered keys, heavy-hitter
    queries, ~2 % selected): B = 1, H = 32, Hkv = 8, n = 97 932, M = 98 304, K10 L150.  Key codes by the torch
    restatement of models/attnserver.py:159-168 (ties listed and defined by the exact sign, as run_pipeline), tables
    + retrieve + sparse attention by the compiled reference.  Stored: SHA-256 of the key codes and of (nnz, sorted
    selected ids), nnz, and the reference's outputs (8 KB) -- the big case is pinned without a big file."""
    B, H, Hkv, n, M, D, K, L = 1, 32, 8, 97932, 98304, 128, 10, 150
    keys, kns, vals, W, qb = case_inputs(seed, B, H, Hkv, n, D, K, L, "clustered")
    Wt = synth.to_torch_bf16(W)
    qcodes = torch_qhash(synth.to_torch_bf16(qb), Wt, K, L).contiguous()
    kcodes = torch.stack([torch_khash(synth.to_torch_bf16(keys[b]), Wt, K, L) for b in range(B)])
    kcodes, ties, tie_dots = exact_sign_ties(kcodes, keys, W, K, L)
    assert len(ties) <= 1e-7 * kcodes.numel() * K + 1 and (len(ties) == 0 or np.abs(tie_dots).max() < 1e-5)
    lsh = ref_lsh.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    sc, si = kcodes[0].sort(stable=True)
    lsh.fill(0, 0, sc.contiguous(), si.int().contiguous())
    results = torch.zeros((B * H, M), dtype=torch.int32)
    nnz = torch.zeros((B * H,), dtype=torch.int32)
    lsh.batch_retrieve(0, qcodes, results, nnz)
    srv = ref_attn.SparseAttentionServer()
    srv.alloc(1, H, Hkv, D, B, M)
    srv.fill(0, 0, synth.to_torch_bf16(keys[0]), synth.to_torch_bf16(vals[0]), torch.from_numpy(kns[0]))
    q_t = synth.to_torch_bf16(qb)
    qn = q_t.float().norm(p=2, dim=-1)
    output = torch.zeros((B * H, D), dtype=torch.bfloat16)
    mve = torch.zeros((2, B * H), dtype=torch.float32)
    srv.attention_wrapper(0, K, L, output, mve, q_t, qn, results, nnz)
    nz = nnz.numpy()
    hsh = hashlib.sha256()
    hsh.update(nz.tobytes())
    for h in range(B * H):
        hsh.update(np.sort(results[h, :nz[h]].numpy()).tobytes())
    out[name] = dict(meta=np.array([seed, B, H, Hkv, n, M, D, K, L], np.int64),
                     data=np.array(DATA_KINDS.index("clustered"), np.int64), qcodes=qcodes.numpy().astype(np.int32),
                     kcodes_sha=np.frombuffer(hashlib.sha256(kcodes.numpy().tobytes()).digest(), np.uint8),
                     kcodes_ties=ties, kcodes_tie_dots=tie_dots, nnz=nz.copy(),
                     sha256=np.frombuffer(hsh.digest(), np.uint8),
                     out_bits=output.view(torch.int16).numpy().view(np.uint16).copy(), mve=mve.numpy().copy())
    print(f"{name}: nnz mean {nz.mean():.1f} ({nz.mean() / n * 100:.2f} %) min {nz.min()} max {nz.max()}; {len(ties)} ties")


def run_llama_ops(name, out):
    """The model plumbing the decode-step harness (SURVEY f-3) restates around the path, executed with torch on CPU:
    the RoPE tables exactly as models/llama.py:114-126 builds them (inv_freq = theta^(-2i/D), attention_scaling = 1:
    the default rotary embedding of the HF config the reference loads), rotate_half / apply_rotary_pos_emb verbatim
    from models/utils.py:29-45 on bf16 tensors, and RMSNorm as models/utils.py:47-56 calls it -- flashinfer.rmsnorm,
    a third-party wheel absent from /root/reference (install.sh:4); its published definition is
    out = x / sqrt(mean(x^2) + eps) * weight evaluated in f32 and rounded once to the input dtype."""
    c = LLAMA_OPS
    x, w, q, k = llama_ops_inputs(c)
    D, max_len = c["D"], c["max_len"]
    inv_freq = 1.0 / (c["theta"] ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    position_ids = torch.arange(0, max_len).unsqueeze(0)                                   # llama.py:114
    inv_freq_expanded = inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
    position_ids_expanded = position_ids[:, None, :].float()
    freqs = (inv_freq_expanded.float() @ position_ids_expanded.float()).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos_cache = (emb.cos()[0] * 1.0).to(torch.bfloat16)                                   # :119-124
    sin_cache = (emb.sin()[0] * 1.0).to(torch.bfloat16)

    def rotate_half(t):                                                                    # utils.py:29-33
        t1 = t[..., : t.shape[-1] // 2]
        t2 = t[..., t.shape[-1] // 2:]
        return torch.cat((-t2, t1), dim=-1)

    def apply_rotary_pos_emb(t, cos, sin, position_ids, unsqueeze_dim=1):                  # utils.py:36-45
        cos = cos[position_ids].unsqueeze(unsqueeze_dim)
        sin = sin[position_ids].unsqueeze(unsqueeze_dim)
        return (t * cos) + (rotate_half(t) * sin)

    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()     # noqa: E731
    d = dict(meta=np.array([c["seed"], c["B"], c["H"], c["Hkv"], D, c["hidden"], max_len], np.int64),
             cos_rows=bits(cos_cache[list(c["positions"])]), sin_rows=bits(sin_cache[list(c["positions"])]))
    qt, kt = synth.to_torch_bf16(q), synth.to_torch_bf16(k)
    for p_ in c["positions"]:
        pos = torch.full((c["B"], 1), p_, dtype=torch.long)
        d[f"q_rope_{p_}"] = bits(apply_rotary_pos_emb(qt, cos_cache, sin_cache, pos))
        d[f"k_rope_{p_}"] = bits(apply_rotary_pos_emb(kt, cos_cache, sin_cache, pos))
    xf, wf = synth.to_torch_bf16(x).float(), synth.to_torch_bf16(w).float()
    d["rmsnorm"] = bits((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + c["eps"]) * wf).to(torch.bfloat16))
    out[name] = d
    print(f"{name}: positions {c['positions']}")


def main():
    only = set(sys.argv[1:])            # fixture names to (re)generate; none = all
    ref_lsh, ref_attn = load_ref()
    assert ref_uses_bf16_family(), "fixtures are generated with the __AVX512BF16__ build"
    P = lambda *a, **kw: (lambda name, out: run_pipeline(name, *a, ref_lsh, ref_attn, out, **kw))   # noqa: E731
    Q = lambda *a: (lambda name, out: run_qhash_only(name, *a, out))                                  # noqa: E731
    registry = {
        "qhash_r1_k10_l150": Q(11, 1, 128, 10, 150),
        "qhash_r32_k10_l150": Q(12, 32, 128, 10, 150),
        "qhash_r64_k11_l300": Q(13, 64, 128, 11, 300),
        "qhash_r40_k8_l50": Q(14, 40, 128, 8, 50),
        "qhash_r256_k10_l170": Q(15, 256, 128, 10, 170),
        "lsh_edge": lambda name, out: run_lsh_edge(name, 21, ref_lsh, out),
        "attn_edge": lambda name, out: run_attn_edge(name, 31, ref_attn, out),
        "lsh_small": P(41, 2, 4, 2, 256, 300, 128, 4, 8),
        "cfg0": P(42, 1, 1, 1, 4096, 4288, 128, 10, 150),
        "gqa_32h": P(43, 1, 32, 8, 4096, 4288, 128, 10, 150),
        "b2_k8_l60": P(44, 2, 8, 2, 1500, 1600, 128, 8, 60),
        # BASELINE cfg 2 / cfg 3 head counts at a small sequence: B = 8, H = 32, Hkv = 8 -> 256 query heads
        # (the one-workgroup-per-head / cluster = 1 regime of the decode entry), L = 170 and L = 150
        "cfg2_small": P(45, 8, 32, 8, 2048, 2112, 128, 10, 170),
        "cfg3_small": P(46, 8, 32, 8, 2048, 2112, 128, 10, 150),
        # round 3 -- BASELINE cfg 4's per-GPU geometry and hyper-parameters (70B, TP = 8: 1 kv head, 8 query heads,
        # K = 11, L = 300 -> G = 8, NB = 2048), a G = 8 case with several kv heads and requests
        # (library/sparse_attention/test.py:6-14 has G = 8 in its grid), and the non-isotropic workloads of
        # `skewed` at K = 6 (64 buckets: pieces of hundreds of ids -- slot overflow, second access and
        # chunk pool of the decode kernel) and `clustered` at K10 L150
        "cfg4_small": P(47, 1, 8, 1, 4096, 4160, 128, 11, 300),
        "g8_hkv2": P(48, 2, 16, 2, 1500, 1600, 128, 8, 60),
        "skew_small": P(49, 1, 8, 2, 4096, 4160, 128, 6, 64, data="skewed"),
        "clustered_k10": P(50, 1, 32, 8, 8192, 8256, 128, 10, 150, data="clustered"),
        "cfg1_retrieve_sha": lambda name, out: run_cfg1_retrieve_sha(name, 51, ref_lsh, out),
        "cfg1_skew_sha": lambda name, out: run_cfg1_skew_sha(name, 52, ref_lsh, ref_attn, out),
        "full_dense": lambda name, out: run_full_dense(name, 55, ref_attn, out),
        "window_merge": lambda name, out: run_window_merge(name, out),
        "fill_centre": lambda name, out: run_fill_centre(name, out),
        "llama_ops": lambda name, out: run_llama_ops(name, out),
    }
    unknown = only - set(registry)
    assert not unknown, f"unknown fixtures: {sorted(unknown)}"
    wrote = 0
    for name, fn in registry.items():
        if only and name not in only:
            continue
        cases: dict = {}
        fn(name, cases)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **cases[name])
        wrote += 1
    print("wrote", wrote, "fixtures to", HERE)


if __name__ == "__main__":
    main()
