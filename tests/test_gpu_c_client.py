"""The C ABI used from plain C (tests/c_client/client.c: gcc, no HIP headers, no torch, host
buffers) against the CPU oracle: key codes -> tables -> retrieve -> attention on one small case."""
import os
import subprocess

import numpy as np
import pytest

import cases
import oracle
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_matches_oracle(tmp_path):
    from magicpig_amd.build import build

    lib = build()
    exe = str(tmp_path / "client")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_client", "client.c"), "-o", exe,
           "-L", os.path.dirname(lib), "-lmagicpig_hip", "-Wl,-rpath," + os.path.dirname(lib),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)

    B, H, Hkv, n, M, D, K, L = 2, 4, 2, 1500, 1536, 128, 8, 40
    keys, kns, vals, W, qb = cases.case_inputs(77, B, H, Hkv, n, D, K, L)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        np.array([B, H, Hkv, D, K, L, n, M], np.int32).tofile(f)
        np.ascontiguousarray(W).astype(np.uint16).tofile(f)
        np.ascontiguousarray(np.stack(keys)).astype(np.uint16).tofile(f)
        np.ascontiguousarray(np.stack(vals)).astype(np.uint16).tofile(f)
        np.ascontiguousarray(np.stack(kns)).astype(np.float32).tofile(f)
        np.ascontiguousarray(qb).astype(np.uint16).tofile(f)
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "ok arch=gfx950" in r.stdout
    BH = B * H
    raw = np.fromfile(outp, np.uint8)
    o = 0
    def take(count, dt):
        nonlocal o
        a = np.frombuffer(raw, dt, count, o)
        o += count * np.dtype(dt).itemsize
        return a
    codes = take(BH * L, np.int32).reshape(BH, L)
    nnz = take(BH, np.int32)
    results = take(BH * M, np.int32).reshape(BH, M)
    out = take(BH * D, np.uint16).reshape(BH, D)
    mve = take(2 * BH, np.float32).reshape(2, BH)
    assert o == raw.size

    # ---- oracle
    qcodes, qn = oracle.simhash_query(qb, W, K, L)
    assert np.array_equal(codes, qcodes)
    olsh = oracle.LSH()
    olsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(oracle.simhash_keys(keys[b], W, K, L))
        olsh.fill(0, b, sc, si)
    ores = np.zeros((BH, M), np.int32)
    onnz = np.zeros((BH,), np.int32)
    olsh.batch_retrieve(0, qcodes, ores, onnz)
    assert np.array_equal(nnz, onnz)
    ind = np.zeros_like(ores)
    for h in range(BH):
        ind[h, :onnz[h]] = np.sort(ores[h, :onnz[h]])
        assert np.array_equal(results[h, :nnz[h]], ind[h, :onnz[h]])
    osrv = oracle.SparseAttentionServer(exp_mode=2, clamp_cos=1)
    osrv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        osrv.fill(0, b, keys[b], vals[b], kns[b])
    oout = np.zeros((BH, D), np.uint16)
    omve = np.zeros((2, BH), np.float32)
    osrv.attention_wrapper(0, K, L, oout, omve, qb, qn, ind, onnz)
    live = onnz > 0
    assert np.allclose(synth.bf16_bits_to_f32(out)[live], synth.bf16_bits_to_f32(oout)[live], rtol=2 ** -7, atol=2e-4)
    assert np.allclose(mve[1][live], omve[1][live], atol=1e-3)
