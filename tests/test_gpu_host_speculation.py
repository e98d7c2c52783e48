"""Host-buffer mode, round 6: the attention launch rides behind the retrieve (csrc/capi.hip: mp_lsh::Spec).  The reference's
caller hands attention_wrapper the SAME pinned query / output / max_value_expsum tensors every step and fills the query tensor
before it calls batch_retrieve (models/attnserver.py:59-66, 273, 299-300): once the library has seen such a call, the next
MP_MEM_HOST batch_retrieve enqueues the paired store's attention launch behind its own kernel, and attention_wrapper only checks
that it is the call that launch assumed.  Everything the launch assumed and the caller may change is a MISS that must be
served correctly: a rewritten query, another query tensor, another ||q||, another layer, an edited row.
Outputs of a hit are those of the device entry within 1 bf16 ulp (the launch computes ||q|| itself; the caller's value is
only compared with it).  Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import cases
from test_gpu_parity import bf16_t, mp  # noqa: F401  (mp: fixture)

pytestmark = pytest.mark.gpu


def _setup(mp, layers=2):
    g = cases.load_golden("gqa_32h")
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    lsh, srv = mp.LSH(), mp.SparseAttentionServer()
    lsh.alloc(K, L, layers, H, Hkv, B, M)
    srv.alloc(layers, H, Hkv, D, B, M)
    for li in range(layers):
        k = np.roll(keys[0], 17 * li, axis=1)
        lsh.fastfill(li, 0, sh.keys(bf16_t(k, "cuda")))
        srv.fill(li, 0, bf16_t(k, "cuda"), bf16_t(np.roll(vals[0], 17 * li, axis=1), "cuda"),
                 torch.from_numpy(np.roll(kns[0], 17 * li, axis=1).copy()).cuda())
    return dict(B=B, H=H, D=D, K=K, L=L, M=M, BH=B * H, qb=qb, sh=sh, lsh=lsh, srv=srv)


def _device_entry(c, layer, q, qn=None):
    BH, M, D = c["BH"], c["M"], c["D"]
    codes, qn0 = c["sh"].query(q)
    qn = qn0 if qn is None else qn
    res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
    nz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
    c["lsh"].batch_retrieve(layer, codes, res, nz)
    out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    c["srv"].attention_wrapper(layer, c["K"], c["L"], out, mve, q, qn, res, nz)
    return codes, out.float().cpu().numpy(), mve.cpu().numpy(), nz.cpu()


def _close(a, b):
    return np.allclose(a, b, rtol=2 ** -7, atol=2e-4)


class Caller:
    """The tensors of models/attnserver.py:59-66, allocated once."""

    def __init__(self, c, pin_results, query_dtype=torch.bfloat16):
        BH, M, D, L = c["BH"], c["M"], c["D"], c["L"]
        self.codes = torch.zeros((BH, L), dtype=torch.int32).pin_memory()
        self.query = torch.zeros((BH, D), dtype=query_dtype).pin_memory()
        self.results = torch.zeros((BH, M), dtype=torch.int32)
        self.nnz = torch.zeros((BH,), dtype=torch.int32)
        if pin_results:
            self.results, self.nnz = self.results.pin_memory(), self.nnz.pin_memory()
        self.out = torch.zeros((BH, D), dtype=torch.bfloat16).pin_memory()
        self.mve = torch.zeros((2, BH), dtype=torch.float32).pin_memory()

    def layer(self, c, layer, q, codes, between=None, qn=None, query=None):
        self.codes.copy_(codes)                                        # :272
        self.query.copy_(q)                                            # :273
        c["lsh"].batch_retrieve(layer, self.codes, self.results, self.nnz)             # :299
        if between is not None:
            between()
        qt = self.query if query is None else query
        c["srv"].attention_wrapper(layer, c["K"], c["L"], self.out, self.mve, qt,
                                   qt.float().norm(p=2, dim=-1) if qn is None else qn, self.results, self.nnz)   # :300
        return self.out.float().numpy().copy(), self.mve.numpy().copy(), self.nnz.clone()


def _counters(L_, reset=False):
    names = ("host_spec_hits", "host_spec_misses", "host_fast_hits")
    got = tuple(L_.get_option(k) for k in names)
    if reset:
        for k in names:
            L_.set_option(k, 0)
    return got


@pytest.mark.parametrize("pin_results", [False, True])
def test_unchanged_caller_is_served_by_the_launch_behind_the_retrieve(mp, pin_results):
    import magicpig_amd._lib as L_

    c = _setup(mp)
    caller = Caller(c, pin_results)
    _counters(L_, reset=True)
    for step in range(6):
        q = bf16_t(np.roll(c["qb"], step, axis=0), "cuda")
        for layer in (0, 1):
            codes, want_out, want_mve, want_nnz = _device_entry(c, layer, q)
            out, mve, nnz = caller.layer(c, layer, q, codes)
            assert torch.equal(nnz, want_nnz)
            assert _close(out, want_out) and np.allclose(mve[1], want_mve[1], atol=1e-3)
    hits, misses, fast = _counters(L_)
    assert fast == 12                      # every call recognised the rows it was handed
    assert hits == 11 and misses == 0      # all but the very first call after alloc (nothing was known about the caller yet)
    # the scores of the last call are there (the launch produced the logits)
    probs = c["srv"].get_score().reshape(c["BH"], c["M"])
    z = int(nnz[0])
    assert z > 0 and float(probs[0, :z].sum()) == pytest.approx(1.0, abs=1e-3)


def test_what_the_launch_assumed_and_the_caller_changed(mp):
    import magicpig_amd._lib as L_

    c = _setup(mp)
    caller = Caller(c, False)
    BH, D = c["BH"], c["D"]
    q0 = bf16_t(c["qb"], "cuda")
    q1 = bf16_t(np.roll(c["qb"], 5, axis=0), "cuda")
    codes0, *_ = _device_entry(c, 0, q0)
    caller.layer(c, 0, q0, codes0)                         # the library now knows the caller's query tensor
    _counters(L_, reset=True)
    # (1) the query tensor is REWRITTEN between the two calls: the attention is that of the new query over the old rows
    codes, _, _, _ = _device_entry(c, 0, q0)
    res_d = torch.zeros((BH, c["M"]), dtype=torch.int32, device="cuda")
    nz_d = torch.zeros((BH,), dtype=torch.int32, device="cuda")
    c["lsh"].batch_retrieve(0, codes, res_d, nz_d)
    o_d = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    m_d = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    c["srv"].attention_wrapper(0, c["K"], c["L"], o_d, m_d, q1, q1.float().norm(p=2, dim=-1), res_d, nz_d)
    out, mve, _ = caller.layer(c, 0, q0, codes, between=lambda: caller.query.copy_(q1))
    assert _counters(L_, reset=True)[:2] == (0, 1)
    assert _close(out, o_d.float().cpu().numpy()) and np.allclose(mve[1], m_d[1].cpu().numpy(), atol=1e-3)
    # (2) ANOTHER query tensor (same bytes)
    other = caller.query.clone().pin_memory()
    _, want_out, want_mve, _ = _device_entry(c, 0, q0)
    out, mve, _ = caller.layer(c, 0, q0, codes, between=lambda: other.copy_(q0), query=other)
    assert _counters(L_, reset=True)[:2] == (0, 1)
    assert _close(out, want_out)
    caller.layer(c, 0, q0, codes)                          # (the library now follows `other`... and back)
    caller.layer(c, 0, q0, codes)
    assert _counters(L_, reset=True)[0] == 1
    # (3) another ||q|| than the query's: the caller's value is what the arithmetic must use
    qn2 = (q0.float().norm(p=2, dim=-1) * 2).cpu()
    _, want_out2, want_mve2, _ = _device_entry(c, 0, q0, qn=qn2.cuda())
    out, mve, _ = caller.layer(c, 0, q0, codes, qn=qn2)
    assert _counters(L_, reset=True)[:2] == (0, 1)
    assert _close(out, want_out2) and np.allclose(mve[1], want_mve2[1], atol=1e-3)
    assert not np.allclose(mve[1], want_mve[1], atol=1e-3)
    # (4) a row edited in place between the calls: served, not the launch's output
    caller.layer(c, 0, q0, codes)
    _counters(L_, reset=True)
    r = int(torch.argmax(caller.nnz))
    z = int(caller.nnz[r])

    def edit():
        have = set(caller.results[r, :z].tolist())
        caller.results[r, z // 2] = next(t for t in range(c["M"] - 200) if t not in have)
    out, _, _ = caller.layer(c, 0, q0, codes, between=edit)
    e_res = res_d.clone()
    e_res[r, :z] = caller.results[r, :z].cuda()
    c["srv"].attention_wrapper(0, c["K"], c["L"], o_d, m_d, q0, q0.float().norm(p=2, dim=-1), e_res, nz_d)
    assert _counters(L_, reset=True)[0] == 0
    assert _close(out, o_d.float().cpu().numpy())
    # (5) the option switches it off
    L_.set_option("host_speculate", 0)
    try:
        caller.layer(c, 0, q0, codes)
        out, _, _ = caller.layer(c, 0, q0, codes)
        assert _counters(L_, reset=True)[:2] == (0, 0)
        assert _close(out, want_out)
    finally:
        L_.set_option("host_speculate", 1)


def test_handles_destroyed_in_either_order(mp):
    """The LSH handle remembers the store it launches for, the store the handles that do: either may go first."""
    for first in ("lsh", "srv"):
        c = _setup(mp, layers=1)
        caller = Caller(c, False)
        q = bf16_t(c["qb"], "cuda")
        codes, *_ = _device_entry(c, 0, q)
        caller.layer(c, 0, q, codes)
        caller.layer(c, 0, q, codes)
        if first == "lsh":
            del c["lsh"]
            c["srv"].clear()
        else:
            del c["srv"]
            c["lsh"].batch_retrieve(0, caller.codes, caller.results, caller.nnz)     # no store to launch for any more
        torch.cuda.synchronize()


def test_row_copy_prefetch_and_the_call_timers(mp):
    """Round 6, second pass on the retrieve call (EXPERIMENTS.md R6-2): the copy of the handed-out rows into the caller's pageable
    tensor prefetches the next row -- an option that changes nothing but time -- and the call's three phases are counted."""
    import magicpig_amd._lib as L_

    c = _setup(mp, layers=1)
    caller = Caller(c, False)
    q = bf16_t(c["qb"], "cuda")
    codes, want_out, want_mve, want_nnz = _device_entry(c, 0, q)
    default = L_.get_option("host_copy_prefetch")
    assert default == 48
    try:
        rows = []
        for lines in (0, 8, 48, 100000):
            L_.set_option("host_copy_prefetch", lines)
            for k in ("host_ret_calls", "host_ret_ns_enqueue", "host_ret_ns_wait", "host_ret_ns_copy"):
                L_.set_option(k, 0)
            caller.results.zero_()
            out, mve, nnz = caller.layer(c, 0, q, codes)
            assert torch.equal(nnz, want_nnz) and _close(out, want_out)
            rows.append(caller.results.clone())
            assert L_.get_option("host_ret_calls") == 1
            assert all(L_.get_option("host_ret_ns_" + k) > 0 for k in ("enqueue", "wait", "copy"))
        for r in rows[1:]:
            assert torch.equal(r, rows[0])
    finally:
        L_.set_option("host_copy_prefetch", default)


def test_f32_query_tensor_is_served_by_the_launch_ahead_too(mp):
    """mp_attn_sparse takes f32 queries as well (MP_DTYPE_F32): the snapshot, the host-side norms and the launch follow the dtype."""
    import magicpig_amd._lib as L_

    c = _setup(mp, layers=1)
    caller = Caller(c, False, query_dtype=torch.float32)
    _counters(L_, reset=True)
    for step in range(4):
        q = bf16_t(np.roll(c["qb"], step, axis=0), "cuda")
        codes, want_out, want_mve, want_nnz = _device_entry(c, 0, q)
        out, mve, nnz = caller.layer(c, 0, q, codes)          # (query.copy_ widens the bf16 rows: the same values in f32)
        assert torch.equal(nnz, want_nnz)
        assert _close(out, want_out) and np.allclose(mve[1], want_mve[1], atol=1e-3)
    hits, misses, fast = _counters(L_)
    assert fast == 4 and hits == 3 and misses == 0
