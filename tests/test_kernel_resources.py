"""Register-allocation guard for the gfx950 kernels (CPU only: hipcc cross-compiles, the code objects' metadata is read
back with the ROCm LLVM tools).  A spill reload inside the decode kernel's gather loop waits for every row load issued
before it (EXPERIMENTS.md R3-10: cfg 2 went from 35.6 to 38.9 us), and a 1 024-thread workgroup may use at most 128
VGPRs -- so: no kernel of the library spills a vector register or reserves scratch, and the workgroup-wide kernels stay
inside their register budget."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.elf")
    # llvm-objcopy without an output operand rewrites its input IN PLACE: work on a copy, the objects under
    # magicpig_amd/lib/obj are what the shipped library is linked from and a test must not touch them
    scratch = os.path.join(tmp, "copy.o")
    shutil.copyfile(obj, scratch)
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", scratch, os.path.join(tmp, "out.o")],
                   check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))      # noqa: E731
        out[name] = {k: get(k) for k in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
                                         "vgpr_count", "max_flat_workgroup_size")}
        out[name]["scratch_insts"] = 0
    # scratch instructions per kernel, from the disassembly (a kernel may RESERVE a few bytes of private segment for
    # SGPR spill slots that were all folded into VGPR lanes: no instruction ever touches them)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
        elif cur in out and re.search(r"\bscratch_(load|store)", line):
            out[cur]["scratch_insts"] += 1
    return out


@pytest.mark.skipif(not all(os.path.exists(f"{LLVM}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")),
                    reason="ROCm LLVM tools not found")
def test_no_kernel_spills_vector_registers_or_reserves_scratch(tmp_path):
    from magicpig_amd import build as B
    B.build()
    seen = {}
    for src in B.SOURCES:
        obj = os.path.join(B.OBJDIR, src.replace(".hip", ".o"))
        if src == "capi.hip":           # host code only
            continue
        d = tmp_path / src
        d.mkdir()
        seen.update(_kernels(obj, str(d)))
    assert len(seen) >= 20
    decode = [k for k in seen if "lsh_decode_kernel" in k]
    assert len(decode) >= 10                                   # D = 64 / 128 x window x hash form
    # the one known exception: the key SimHash at head_dim 256 (no BASELINE configuration; prefill side) keeps a whole
    # 256-wide row tile per wave and spills 47 registers in its epilogue
    known = [k for k in seen if "simhash_keys_kernelILi256E" in k]
    for name, r in seen.items():
        if name in known:
            continue
        assert r["vgpr_spill_count"] == 0, (name, r)
        assert r["scratch_insts"] == 0, (name, r)
        # an instantiation of the decode kernel may RESERVE a few bytes nothing touches (SGPR spill slots that were all
        # folded into VGPR lanes: it comes and goes with every edit, EXPERIMENTS.md R3-14); no other kernel reserves any
        if "lsh_decode_kernel" in name:
            assert r["private_segment_fixed_size"] <= 64, (name, r)
        else:
            assert r["private_segment_fixed_size"] == 0, (name, r)
        if r["max_flat_workgroup_size"] >= 1024:
            assert r["vgpr_count"] <= 128, (name, r)
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.skipif(not all(os.path.exists(f"{LLVM}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")),
                    reason="ROCm LLVM tools not found")
def test_decode_kernel_streams_tables_and_rows_past_the_l2(tmp_path):
    """Cache policy of the decode kernel's big read-once streams, read off the disassembly: the K / V row gather
    (16-byte loads) and the table-side loads (direct slots, bucket records, table ids: 4-byte loads) carry `nt`
    (EXPERIMENTS.md R4-15: without it on the table side cfg 3 is 0.65 us per launch slower)."""
    from magicpig_amd import build as B
    B.build()
    d = tmp_path / "lsh"
    d.mkdir()
    obj = os.path.join(B.OBJDIR, "lsh.o")
    fat, co, scratch = str(d / "fat.bin"), str(d / "co.elf"), str(d / "copy.o")
    shutil.copyfile(obj, scratch)
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", scratch, str(d / "out.o")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    seen = 0
    for m in re.finditer(r"<(_ZN2mp17lsh_decode_kernelILi\d+ELi\d+ELb[01]ELi[13]E\w*)>:\n(.*?)\n\n", dis, re.S):
        body = m.group(2)
        # (the LEAN instantiations request rows through a buffer descriptor: a slot past a wave's list costs no request)
        rows = re.findall(r"(?:global|buffer)_load_dwordx4 .* nt\b", body)
        tables = re.findall(r"global_load_dword v\d+, .* nt\b", body)
        assert len(rows) >= 16, (m.group(1), len(rows))          # a 32-token step alone is 16 row loads
        assert len(tables) >= 40, (m.group(1), len(tables))
        seen += 1
    assert seen >= 6
    shutil.rmtree(tmp_path, ignore_errors=True)
