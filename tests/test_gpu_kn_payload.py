"""Key norms as a payload of the table entries (include/magicpig_hip.h: mp_lsh_get_id_bits; lsh.hip:
lsh_attach_norms_kernel).  The one-launch decode entries pack the bf16 norms the attention store holds into the
bits above the 17-bit token ids the first time a layer is decoded, and read them from LDS afterwards.  Nothing
observable may change: ids, counts and outputs are those of the per-token norm reads, bit for bit."""
import numpy as np
import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu


def bf16_t(bits, device="cpu"):
    return synth.to_torch_bf16(np.ascontiguousarray(bits)).to(device)


@pytest.fixture(scope="module")
def mp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import magicpig_amd
    return magicpig_amd


@pytest.fixture(autouse=True)
def _default_option():
    yield
    if torch.cuda.is_available():
        import magicpig_amd._lib as L
        L.set_option("decode_kn_payload", 1)


def _server(mp, B, H, Hkv, n, M, D, K, L, seed, data="randn"):
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L, data=data)
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0, num_local_tokens=0,
                                    max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    for b in range(B):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[b], "cuda"))
        server.build_table(0, b, n)
        server.attn_server.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"), torch.from_numpy(kns[b]).cuda())
    return server, (keys, kns, vals, W, qb)


def _packed(server, rows=slice(None), n=None):
    """True when some word of the rows [groups][L][:n] carries a payload above the 17-bit id."""
    raw = server.lsh_retriever.get_tables(0, raw=True)[1][rows, :, :n]
    return bool((((raw >> 17) & 0x7fff) != 0).any())


def _decode(server, q):
    out, lse = server.decode(q, 0)
    torch.cuda.synchronize()
    return out.clone(), lse.clone(), server.nnz.clone()


@pytest.mark.parametrize("B,H,Hkv,data", [(1, 32, 8, "randn"), (2, 8, 2, "clustered"), (3, 6, 3, "skewed")])
def test_payload_changes_nothing_observable(mp, B, H, Hkv, data):
    import magicpig_amd._lib as L
    n, M, D, K, Lt = 6000, 6144, 128, 8, 75
    server, (keys, kns, vals, W, qb) = _server(mp, B, H, Hkv, n, M, D, K, Lt, 99, data)
    lsh = server.lsh_retriever
    assert lsh.id_bits(0) == 17
    _, plain = lsh.get_tables(0, raw=True)
    plain = plain.clone()
    assert not _packed(server, n=n)                            # plain ids until the first decode
    gen = torch.Generator(device="cuda").manual_seed(1)
    qs = [torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16) for _ in range(4)]
    L.set_option("decode_kn_payload", 0)
    want = [_decode(server, q) for q in qs]
    assert torch.equal(lsh.get_tables(0, raw=True)[1], plain)   # option off: nothing was packed
    L.set_option("decode_kn_payload", 1)
    got = [_decode(server, q) for q in qs]
    for (o0, l0, z0), (o1, l1, z1) in zip(want, got):
        assert torch.equal(z0, z1) and torch.equal(o0, o1) and torch.equal(l0, l1)
    # the words now carry the norms: id | bf16 bits of the store's norm << 17
    _, raw = lsh.get_tables(0, raw=True)
    _, ids = lsh.get_tables(0)
    assert torch.equal(ids[:, :, :n], plain[:, :, :n])
    kn = server.attn_server.get_key_norm(0).reshape(B * Hkv, M)
    norm_bits = (kn.view(torch.int32) >> 16)                                   # [groups, M]
    expect = torch.gather(norm_bits[:, None, :].expand(-1, Lt, -1), 2, ids[:, :, :n].long())
    assert torch.equal((raw[:, :, :n] >> 17) & 0x7fff, expect)
    # the other consumers of the tables mask the ids: retrieve, get_mask
    codes, _ = server.hasher.query(qs[0].reshape(B * H, D))
    res = torch.zeros((B * H, M), dtype=torch.int32, device="cuda")
    nz = torch.zeros((B * H,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, codes, res, nz)
    assert torch.equal(nz, want[0][2])
    assert int(res.max()) < n
    mask = lsh.get_mask()
    assert torch.equal((mask == 2).sum(-1).reshape(-1).int(), nz.cpu())
    # a rebuilt table holds plain ids again, and the next decode packs again
    server.hash_code_buffer = server.hasher.keys(bf16_t(keys[0], "cuda"))
    server.build_table(0, 0, n)
    assert not _packed(server, slice(0, Hkv), n)
    again = _decode(server, qs[1])
    assert torch.equal(again[0], want[1][0]) and torch.equal(again[2], want[1][2])
    assert _packed(server, slice(0, Hkv), n)


def test_norms_that_cannot_ride_along_and_norms_that_change(mp):
    """f32 norms with mantissa bits below bf16 (a caller may pass any f32 norms to fill) keep their KV group on the
    per-token reads -- the result is the one of the option switched off, not the one of truncated norms; a refill with
    other norms is seen by the next decode."""
    import magicpig_amd._lib as L
    B, H, Hkv, n, M, D, K, Lt = 1, 8, 2, 5000, 5120, 128, 8, 60
    server, (keys, kns, vals, W, qb) = _server(mp, B, H, Hkv, n, M, D, K, Lt, 5)
    gen = torch.Generator(device="cuda").manual_seed(2)
    q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    base = _decode(server, q)                                     # packed, bf16 norms
    odd = torch.from_numpy(kns[0]).cuda() * 1.00390625 + 1e-3     # not bf16 numbers
    odd[1] = torch.from_numpy(kns[0]).cuda()[1]                   # KV group 1 keeps bf16 norms
    server.attn_server.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), odd)
    got = _decode(server, q)
    L.set_option("decode_kn_payload", 0)
    want = _decode(server, q)
    L.set_option("decode_kn_payload", 1)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    G = H // Hkv
    assert not torch.equal(got[0][0, :G], base[0][0, :G])         # group 0: the new norms are in use
    assert torch.equal(got[0][0, G:], base[0][0, G:])             # group 1: unchanged
    assert not _packed(server, slice(0, 1), n) and _packed(server, slice(1, 2), n)   # group 0 carries no payload
    # back to bf16 norms, scaled: picked up by the next decode (the store's version moved)
    server.attn_server.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda() * 2)
    got2 = _decode(server, q)
    L.set_option("decode_kn_payload", 0)
    want2 = _decode(server, q)
    assert torch.equal(got2[0], want2[0]) and torch.equal(got2[1], want2[1])
    assert not torch.equal(got2[0], base[0])


def test_captured_decodes_follow_the_state_of_their_replay(mp):
    """Whether a KV group's payload is used is decided on the device, from words written in stream order: (1) a decode
    captured before any eager one packs nothing (no packing kernels in a caller's graph) and reads the norms per token;
    (2) once an eager decode has packed the words the SAME graph uses them; (3) after a refill of the store alone
    (other norms, tables untouched) or of both the replayed graph follows the new state -- a graph captured in the
    packed state must not read stale payloads."""
    import magicpig_amd._lib as L
    B, H, Hkv, n, M, D, K, Lt = 1, 8, 2, 5000, 5120, 128, 8, 60
    server, (keys, kns, vals, W, qb) = _server(mp, B, H, Hkv, n, M, D, K, Lt, 6)
    gen = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    server.collect_nnz = False
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        server.decode(q, 0)

    def replay():
        graph.replay()
        torch.cuda.synchronize()
        return server.output.clone()

    def eager(option):
        L.set_option("decode_kn_payload", option)
        out = _decode(server, q)[0].reshape(-1, D).clone()
        L.set_option("decode_kn_payload", 1)
        return out

    o1 = replay()
    assert not _packed(server, n=n)                               # (1)
    want = eager(0)
    assert not _packed(server, n=n)
    assert torch.equal(o1, want)
    assert torch.equal(eager(1), want) and _packed(server, n=n)   # packs
    assert torch.equal(replay(), want)                            # (2)
    # (3a) the store alone gets other norms: the packed words are stale, the graph must not use them
    server.attn_server.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda() * 2)
    o3 = replay()
    want3 = eager(0)
    assert torch.equal(o3, want3) and not torch.equal(want3, want)
    assert torch.equal(eager(1), want3) and torch.equal(replay(), want3)      # re-packed with the new norms
    # (3b) a new prompt: tables rebuilt (plain ids again) and store refilled, then only the graph runs
    keys2, kns2, vals2, _, _ = cases.case_inputs(77, B, H, Hkv, n, D, K, Lt)
    server.hash_code_buffer = server.hasher.keys(bf16_t(keys2[0], "cuda"))
    server.build_table(0, 0, n)
    server.attn_server.fill(0, 0, bf16_t(keys2[0], "cuda"), bf16_t(vals2[0], "cuda"), torch.from_numpy(kns2[0]).cuda())
    o4 = replay()
    assert not _packed(server, n=n)
    want4 = eager(0)
    assert torch.equal(o4, want4) and not torch.equal(want4, want3)
    server.collect_nnz = True


@pytest.mark.parametrize("B,H,Hkv,K,Lt,data", [(1, 32, 8, 8, 75, "randn"), (2, 8, 2, 10, 40, "clustered"),
                                               (1, 8, 1, 11, 30, "randn")])
def test_packed_build_writes_what_the_lazy_packing_writes(mp, B, H, Hkv, K, Lt, data):
    """mp_lsh_build_with_norms (LSH.fastfill(..., attn_server=store), what LSHSparseAttnServer.build_table does after
    fill()): the counting sort packs the store's norms into the table words in the same pass.  Bounds, table words
    and decode results are those of the plain build followed by the first decode's lazy packing, bit for bit, and the
    first decode of the layer launches no packing kernel (the words do not change)."""
    n, M, D = 6000, 6144, 128
    lazy, (keys, kns, vals, W, qb) = _server(mp, B, H, Hkv, n, M, D, K, Lt, 41, data)      # tables first, then the store
    eager = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=Lt, batch_size=B, num_sink_tokens=0, num_local_tokens=0,
                                   max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    for b in range(B):                                                                    # the reference's order
        eager.attn_server.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"), torch.from_numpy(kns[b]).cuda())
        codes = eager.hasher.keys(bf16_t(keys[b], "cuda"))
        eager.lsh_retriever.fastfill(0, b, codes, attn_server=eager.attn_server)
    assert _packed(eager, n=n) and not _packed(lazy, n=n)
    gen = torch.Generator(device="cuda").manual_seed(9)
    qs = [torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16) for _ in range(3)]
    before = eager.lsh_retriever.get_tables(0, raw=True)[1].clone()
    for q in qs:
        a, b_ = _decode(eager, q), _decode(lazy, q)
        assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]) and torch.equal(a[2], b_[2])
    eb, et = eager.lsh_retriever.get_tables(0, raw=True)
    lb, lt = lazy.lsh_retriever.get_tables(0, raw=True)
    assert torch.equal(et, before)                       # the first decode found nothing to pack
    # ... and the lazy path arrived at the same words (entries behind the n tokens of a row are not entries: the lazy
    # sweep packs a norm onto their zeros, the sort never writes them)
    assert torch.equal(eb, lb) and torch.equal(et[:, :, :n], lt[:, :, :n])
    # a norm that cannot ride along (not a bf16 number) is refused by the packed build exactly as by the lazy packing
    odd = torch.from_numpy(kns[0]).cuda() * 1.00390625 + 1e-3
    for srv in (eager, lazy):
        srv.attn_server.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), odd)
    eager.lsh_retriever.fastfill(0, 0, eager.hasher.keys(bf16_t(keys[0], "cuda")), attn_server=eager.attn_server)
    a, b_ = _decode(eager, qs[0]), _decode(lazy, qs[0])
    assert torch.equal(a[0], b_[0]) and torch.equal(a[2], b_[2])
    assert not _packed(eager, slice(0, Hkv), n)


def test_norms_written_outside_a_fill_are_never_read_from_stale_table_words(mp):
    """mp_attn_append* and the writable get_key_norm() view change norms without a fill.  An append marks
    the layer's norms "changed outside a fill" on the device, in stream order -- also inside a replayed graph: the decode
    kernel then reads the norms per token (result of the option switched off) until the next fill.  A write through the
    view followed by invalidate_norms() is packed again by the next decode."""
    import magicpig_amd._lib as L
    B, H, Hkv, n, M, D, K, Lt = 1, 8, 2, 5000, 5120, 128, 8, 60
    server, (keys, kns, vals, W, qb) = _server(mp, B, H, Hkv, n, M, D, K, Lt, 8)
    srv = server.attn_server
    gen = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    base = _decode(server, q)                                       # packs the norms into the table words
    assert _packed(server, n=n)

    def both():
        got = _decode(server, q)
        L.set_option("decode_kn_payload", 0)
        want = _decode(server, q)
        L.set_option("decode_kn_payload", 1)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        return got

    # -- an append INTO the indexed range: a selected token's key (and norm) is replaced by a 3x longer one
    probs = srv.get_score().reshape(B * H, M)
    codes, _ = server.hasher.query(q.reshape(B * H, D))
    res = torch.zeros((B * H, M), dtype=torch.int32, device="cuda")
    nz = torch.zeros((B * H,), dtype=torch.int32, device="cuda")
    server.lsh_retriever.batch_retrieve(0, codes, res, nz)
    assert int(nz[0]) > 0
    tok = int(res[0, 0])                                            # selected by head 0 (KV group 0)
    kc = srv.get_key_cache(0)[:, :, tok].clone()                    # [B, Hkv, D]
    vc = srv.get_value_cache(0)[:, :, tok].clone()
    pos = torch.full((B,), tok, dtype=torch.int32, device="cuda")
    srv.append(0, (kc.float() * 3).to(torch.bfloat16), vc, pos)
    after = both()
    assert not torch.equal(after[0][0, 0], base[0][0, 0])           # the new key (norm x 3) is what the kernel saw
    # -- the same under graph replay: append + decode captured once, the appended key changes between replays
    kbuf = (kc.float() * 3).to(torch.bfloat16).clone()
    server.collect_nnz = False
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        srv.append(0, kbuf, vc, pos)
        server.decode(q, 0)
    for scale in (0.5, 2.0):
        kbuf.copy_((kc.float() * scale).to(torch.bfloat16))
        graph.replay()
        torch.cuda.synchronize()
        o_graph = server.output.clone()
        L.set_option("decode_kn_payload", 0)
        want = _decode(server, q)[0].reshape(-1, D)
        L.set_option("decode_kn_payload", 1)
        assert torch.equal(o_graph, want)
    server.collect_nnz = True
    # -- a refill brings the payload back
    srv.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda())
    again = both()
    assert torch.equal(again[0], base[0])
    # -- the writable view + invalidate_norms: doubled norms of KV group 1 reach the next decode, through the payload
    srv.get_key_norm(0)[0, 1].mul_(2.0)
    srv.invalidate_norms(0, 0)
    got = both()
    G = H // Hkv
    assert torch.equal(got[0][0, :G], base[0][0, :G]) and not torch.equal(got[0][0, G:], base[0][0, G:])


def test_a_layer_whose_ids_outgrow_17_bits_goes_back_to_plain_ids(mp):
    """max_length > 2^17 (BASELINE cfg 4: 131 264): the words carry payloads while every id of the layer is below
    2^17; the first build / fill with a wider id strips the layer's other requests' payloads and the layer stays plain
    until clear().  Results never change."""
    B, H, Hkv, D, K, Lt = 2, 2, 1, 128, 8, 6
    n0, n1, M = 5000, (1 << 17) + 40, (1 << 17) + 64
    keys, kns, vals, W, qb = cases.case_inputs(31, 1, H, Hkv, n0, D, K, Lt)
    server = mp.LSHSparseAttnServer(2, H, Hkv, D, K=K, L=Lt, batch_size=B, num_sink_tokens=0, num_local_tokens=0,
                                    max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    lsh = server.lsh_retriever
    assert lsh.id_bits(0) == 17 and lsh.id_bits(1) == 17
    gen = torch.Generator(device="cuda").manual_seed(4)
    big_k = torch.randn((Hkv, n1, D), device="cuda", generator=gen).to(torch.bfloat16)
    big_v = torch.randn((Hkv, n1, D), device="cuda", generator=gen).to(torch.bfloat16)
    for layer in (0, 1):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[0], "cuda"))
        server.build_table(layer, 0, n0)
        server.attn_server.fill(layer, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda())
        server.hash_code_buffer = server.hasher.keys(big_k[:, :n0].contiguous())
        server.build_table(layer, 1, n0)
        server.attn_server.fill(layer, 1, big_k[:, :n0].contiguous(), big_v[:, :n0].contiguous(),
                                big_k[:, :n0].float().norm(dim=-1).to(torch.bfloat16).float())
    q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)

    def dec(layer):
        out, lse = server.decode(q, layer)
        torch.cuda.synchronize()
        return out.clone(), lse.clone(), server.nnz.clone()

    first = [dec(0), dec(1)]
    raw0 = lsh.get_tables(0, raw=True)[1]
    assert bool((((raw0[:, :, :n0] >> 17) & 0x7fff) != 0).any())          # packed, although max_length > 2^17
    # a graph of layer 0's decode captured in the packed, 17-bit state
    server.collect_nnz = False
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        server.decode(q, 0)
    server.collect_nnz = True
    # request 1 of layer 0 grows past 2^17 tokens (device build; then the reference's sorted-rows fill)
    server.hash_code_buffer = server.hasher.keys(big_k)
    codes = server.hash_code_buffer.clone()
    server.build_table(0, 1, n1)
    server.attn_server.fill(0, 1, big_k, big_v, big_k.float().norm(dim=-1).to(torch.bfloat16).float())
    assert lsh.id_bits(0) == 0 and lsh.id_bits(1) == 17
    raw0 = lsh.get_tables(0, raw=True)[1]
    assert int(raw0[:Hkv, :, :n0].min()) >= 0 and int(raw0[:Hkv, :, :n0].max()) < n0      # request 0: stripped
    assert int(raw0[Hkv:, :, :n1].max()) == n1 - 1
    wide = dec(0)
    graph.replay()                       # the id width is read from the device: the frozen launch argument says 17
    torch.cuda.synchronize()
    assert torch.equal(server.output.view_as(wide[0]), wide[0])
    assert torch.equal(wide[0][0], first[0][0][0]) and torch.equal(wide[2][:H], first[0][2][:H])   # request 0 unchanged
    again1 = dec(1)
    assert torch.equal(again1[0], first[1][0]) and torch.equal(again1[2], first[1][2])               # layer 1 untouched
    import magicpig_amd._lib as L
    L.set_option("decode_kn_payload", 0)
    plain = dec(0)
    assert torch.equal(plain[0], wide[0]) and torch.equal(plain[2], wide[2])
    L.set_option("decode_kn_payload", 1)
    # the same through mp_lsh_fill (sorted rows with ids >= 2^17) on layer 1
    sv, si = codes.sort(dim=-1, stable=True)
    lsh.fill(1, 1, sv.contiguous(), si.int().contiguous())
    assert lsh.id_bits(1) == 0
    server.attn_server.fill(1, 1, big_k, big_v, big_k.float().norm(dim=-1).to(torch.bfloat16).float())
    w1 = dec(1)
    assert torch.equal(w1[0], wide[0]) and torch.equal(w1[2], wide[2])      # layer 1 now holds what layer 0 holds
    server.clear()
    assert lsh.id_bits(0) == 17 and lsh.id_bits(1) == 17
