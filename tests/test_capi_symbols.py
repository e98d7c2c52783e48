"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/magicpig_hip.h declares; the host mirror fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "magicpig_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_reference_surface():
    syms = declared_symbols()
    for must in ("mp_lsh_alloc", "mp_lsh_fill", "mp_lsh_batch_retrieve", "mp_lsh_clear", "mp_lsh_get_mask",
                 "mp_attn_alloc", "mp_attn_fill", "mp_attn_sparse", "mp_attn_full", "mp_attn_clear",
                 "mp_attn_get_kv", "mp_attn_get_key_norm", "mp_attn_get_score", "mp_attn_invalidate_norms", "mp_simhash_query",
                 "mp_decode_sparse_layer", "mp_merge_state"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from magicpig_amd.build import build

    path = build()
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.mp_arch.restype = ctypes.c_char_p
    assert lib.mp_arch() == b"gfx950"
    # the code object really targets gfx950
    blob = open(path, "rb").read()
    assert b"gfx950" in blob


def test_dropin_module_names():
    import magicpig_amd
    from magicpig_amd.dropin import lsh, sparse_attention_cpu

    assert lsh.LSH is magicpig_amd.LSH
    assert sparse_attention_cpu.SparseAttentionServer is magicpig_amd.SparseAttentionServer
    for m in ("alloc", "fill", "batch_retrieve", "clear", "get_mask", "copy", "fastfill"):
        assert hasattr(lsh.LSH, m)                                   # library/lsh/lsh.cc:316-326
    for m in ("alloc", "fill", "attention", "attention_bf16", "full_attention", "scheduled_attention",
              "attention_wrapper", "attention_wrapper_bf16", "get_key_cache", "get_value_cache",
              "get_key_norm", "get_score", "clear"):
        assert hasattr(sparse_attention_cpu.SparseAttentionServer, m)  # sparse_attention.cc:1243-1263


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import magicpig_amd

    with pytest.raises(magicpig_amd.MagicPigError):
        magicpig_amd.LSH().alloc(4, 8, 1, 4, 2, 1, 128)     # hipMalloc fails loudly, no silent CPU path


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "magicpig_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert not re.search(r"#\s*include[^\n]*oracle", text), f
                assert "mp_oracle" not in text and "libmp_oracle" not in text, f


def test_plain_c_client_compiles_and_links(tmp_path):
    """include/magicpig_hip.h is valid C11 and every call of tests/c_client/client.c resolves
    against the built library (compile + link only: running it needs a GPU, tests/test_gpu_c_client.py)."""
    import shutil
    import subprocess

    from magicpig_amd.build import build

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    lib = build()
    exe = str(tmp_path / "client")
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_client", "client.c"), "-o", exe,
                        "-L", os.path.dirname(lib), "-lmagicpig_hip", "-Wl,-rpath," + os.path.dirname(lib),
                        "-Wl,--allow-shlib-undefined"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)
