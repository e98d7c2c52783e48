"""A/B builds of the library: python scripts/build_variant.py <name> [--only file.hip ...] [-DFLAG ...] ->
magicpig_amd/lib/variants/<name>/libmagicpig_hip.so (git-ignored, shipped by gpurun); use with bench.py --lib or
MP_LIB= for the scripts.  The product build (magicpig_amd/build.py) is untouched.  --only: compile just the named sources with the
flags and link them with the PRODUCT's objects of the others (magicpig_amd/lib/obj; run the product build first)."""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicpig_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    only = []
    while "--only" in flags:
        i = flags.index("--only")
        only.append(flags[i + 1])
        del flags[i:i + 2]
    out = os.path.join(B.LIBDIR, "variants", name)
    os.makedirs(out, exist_ok=True)
    hipcc = B._hipcc()

    def one(src):
        if only and src not in only:
            return os.path.join(B.OBJDIR, src.replace(".hip", ".o"))
        o = os.path.join(out, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc, *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", o], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return o

    with cf.ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(one, B.SOURCES))
    lib = os.path.join(out, "libmagicpig_hip.so")
    subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib)


if __name__ == "__main__":
    main()
