"""Compile the reference's OWN C++ for the hot path into oracle/_ref/ and load it.

TEST INFRASTRUCTURE, NOT PRODUCT.  The sources are compiled from where they lie under
/root/reference (nothing is copied into this repository); only the resulting extension
modules land in oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun, where they
serve as the `cpu_baseline` of kind "reference").

Recipe: both libraries are torch C++ extensions made of a handful of files
(library/lsh/lsh.cc; library/sparse_attention/sparse_attention.cc + five vendored FBGEMM
bf16-conversion sources listed in library/sparse_attention/setup.py:39-45), so they are
compiled directly with g++ through torch.utils.cpp_extension.load -- the reference's own
setup.py is not run (it imports py-cpuinfo, absent here).  Deviations from the reference's
flags: no -D_GLIBCXX_USE_CXX11_ABI=0 (torch 2.10 is CXX11-ABI), explicit -O2/-O3 -DNDEBUG
(what setuptools would add).  LSH_THREADS / ATTENTION_THREADS stay at the reference's
compile-time 64 (lsh.h:12, sparse_attention.h:10); callers cap the OpenMP team with
OMP_THREAD_LIMIT instead of patching the source.

Usage:  python oracle/build_ref.py          (build; needs /root/reference)
        from oracle.build_ref import load_ref; lsh_mod, attn_mod = load_ref()
"""
from __future__ import annotations

import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("MAGICPIG_REFERENCE", "/root/reference")
OUT = os.path.join(_HERE, "_ref")


def _cpu_flags() -> set:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def have_reference_sources() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "library/lsh/lsh.cc"))


def build_ref(verbose: bool = False) -> None:
    """JIT-compile the two reference extensions into oracle/_ref/{lsh,sparse_attention_cpu}/."""
    if not have_reference_sources():
        raise RuntimeError(f"reference sources not found under {REF_ROOT}")
    from torch.utils.cpp_extension import load

    flags = _cpu_flags()
    if "avx512f" not in flags:
        raise RuntimeError("the reference needs AVX512F (immintrin _mm512_*); unbuildable on this CPU")
    R = os.path.join(REF_ROOT, "library")
    d = os.path.join(OUT, "lsh")
    os.makedirs(d, exist_ok=True)
    load(name="lsh", sources=[os.path.join(R, "lsh/lsh.cc")],
         extra_cflags=["-O3", "-DNDEBUG", "-mavx512f", "-fopenmp", "-std=c++17"],
         extra_ldflags=["-fopenmp"], build_directory=d, verbose=verbose, is_python_module=False)
    S = os.path.join(R, "sparse_attention")
    F = os.path.join(S, "3rdparty/FBGEMM")
    bf16 = ["-mavx512bf16"] if "avx512_bf16" in flags else []
    d = os.path.join(OUT, "sparse_attention_cpu")
    os.makedirs(d, exist_ok=True)
    load(name="sparse_attention_cpu",
         sources=[os.path.join(S, "sparse_attention.cc")] +
                 [os.path.join(F, "src", f) for f in (
                     "FbgemmBfloat16Convert.cc", "FbgemmBfloat16ConvertAvx2.cc",
                     "FbgemmBfloat16ConvertAvx512.cc", "RefImplementations.cc", "Utils.cc")],
         extra_include_paths=[os.path.join(F, "include"), F],
         extra_cflags=["-O2", "-DNDEBUG", "-mavx512f", "-fopenmp", "-std=c++17"] + bf16,
         extra_ldflags=["-fopenmp"], build_directory=d, verbose=verbose,
         is_python_module=False)
    with open(os.path.join(OUT, "BUILD_INFO.txt"), "w") as f:
        f.write(f"bf16_family={'1' if bf16 else '0'}\n")


def _so(name: str) -> str | None:
    hits = glob.glob(os.path.join(OUT, name, name + "*.so"))
    return hits[0] if hits else None


def ref_available() -> bool:
    """True iff the prebuilt reference modules exist AND this CPU can execute them."""
    if not (_so("lsh") and _so("sparse_attention_cpu")):
        return False
    flags = _cpu_flags()
    need = {"avx512f"}
    try:
        with open(os.path.join(OUT, "BUILD_INFO.txt")) as f:
            if "bf16_family=1" in f.read():
                need.add("avx512_bf16")
    except OSError:
        return False
    return need <= flags


def ref_uses_bf16_family() -> bool:
    try:
        with open(os.path.join(OUT, "BUILD_INFO.txt")) as f:
            return "bf16_family=1" in f.read()
    except OSError:
        return False


def _import_private(name: str):
    """Import oracle/_ref/<name>/<name>.so WITHOUT registering it in sys.modules (the product
    ships drop-in modules of the same names under magicpig_amd/dropin)."""
    import torch  # noqa: F401  (the extension links libtorch)

    path = _so(name)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref():
    """-> (lsh module, sparse_attention_cpu module) of the compiled reference."""
    if not ref_available():
        raise RuntimeError("oracle/_ref is not built (run `python oracle/build_ref.py` where "
                           "/root/reference exists) or this CPU lacks AVX512")
    # Cap the reference's hard-coded 64-thread OpenMP teams at the available cores; must be
    # set before libgomp initialises (64 threads on 8 cores = ~10 ms per call).
    os.environ.setdefault("OMP_THREAD_LIMIT", str(len(os.sched_getaffinity(0))))
    return _import_private("lsh"), _import_private("sparse_attention_cpu")


if __name__ == "__main__":
    build_ref(verbose="-v" in sys.argv)
    print("built:", _so("lsh"), _so("sparse_attention_cpu"))
