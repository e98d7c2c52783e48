"""scripts/ab_libs.py (two builds of the library, one process each, timed in alternating regions): the parent's side of the
protocol against stand-in workers -- no GPU, no library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

STUB = r"""
import sys, json
name, base, chk = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
print("some runtime warning")            # lines that are neither 'ready' nor JSON are skipped
print("ready", flush=True)
n = 0
log = open(sys.argv[4], "a")
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "quit":
        break
    if cmd == "go":
        n += 1
        log.write(name + "\n"); log.flush()
        print(json.dumps({"us_per_layer": base + n, "checksum": chk}), flush=True)
"""


def test_regions_alternate_and_checksums_are_compared(tmp_path):
    import ab_libs

    log = tmp_path / "order.txt"

    def fake(chks):
        def make_cmd(cfg, lib, data, steps, warmup):
            return [sys.executable, "-c", STUB, lib, {"a": "10", "b": "20"}[lib], str(chks[lib]), str(log)]
        return make_cmd

    r = ab_libs.run("cfg1", ["a", "b"], 3, "randn", 64, 8, make_cmd=fake({"a": 7, "b": 7}))
    assert r["us_per_layer"] == {"a": [11.0, 12.0, 13.0], "b": [21.0, 22.0, 23.0]}
    assert r["outputs_equal"] is True
    assert log.read_text().split() == ["a", "b", "a", "b", "a", "b"]
    log.write_text("")
    r = ab_libs.run("cfg1", ["a", "b"], 1, "randn", 64, 8, make_cmd=fake({"a": 7, "b": 8}))
    assert r["outputs_equal"] is False


def test_variant_names_resolve_to_the_variant_directory():
    import ab_libs

    assert ab_libs.lib_path("product") is None
    assert ab_libs.lib_path("r04head").endswith(os.path.join("lib", "variants", "r04head", "libmagicpig_hip.so"))
    cmd = ab_libs.worker_cmd("cfg3", "r04head", "clustered", 32, 4)
    assert "--ab-worker" in cmd and cmd[cmd.index("--lib") + 1].endswith("libmagicpig_hip.so")
    assert "--lib" not in ab_libs.worker_cmd("cfg3", "product", "randn", 32, 4)
