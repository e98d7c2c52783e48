"""The committed measurement artefacts are self-consistent (CPU only): every bench line under profiles/ carries the
fields the bench contract names, its roofline arithmetic closes (achieved = algorithmic bytes / average launch time,
frac = achieved / peak), the GPU's first layer matched the compiled reference in that run, and
profiles/hbm_traffic_latest.json -- the file bench.py copies `roofline.traffic` from -- is what the PMC passes it cites
say (FETCH_SIZE x 2 + WRITE_SIZE, KB = 1024 B)."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _latest_tag():
    src = json.load(open(os.path.join(PROF, "hbm_traffic_latest.json")))["source"]
    return re.search(r"profiles/(r\d+[a-z]?)_pmc_hbm_traffic", src).group(1)


def test_traffic_json_is_what_the_pmc_passes_say():
    tag = _latest_tag()
    traffic = json.load(open(os.path.join(PROF, "hbm_traffic_latest.json")))
    seen = 0
    for key, suffix in (("lsh_decode_bytes_per_launch", ""), ("lsh_decode_bytes_per_launch_clustered", "_clustered"),
                        ("lsh_decode_bytes_per_launch_byproducts", "_byproducts")):
        for cfg, total in traffic.get(key, {}).items():
            text = open(os.path.join(PROF, f"{tag}_pmc_hbm_traffic_{cfg}{suffix}.md")).read()
            kb = {}
            for m in re.finditer(r"\|\s*void mp::lsh_decode_kernel.*\|\s*(FETCH_SIZE|WRITE_SIZE)\s*\|\s*\d+\s*\|\s*([\d.]+)\s*\|", text):
                kb[m.group(1)] = float(m.group(2))
            assert set(kb) == {"FETCH_SIZE", "WRITE_SIZE"}, (cfg, suffix)
            assert abs((2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024 - total) <= 1e-6 * total, (cfg, suffix)
            seen += 1
    assert seen >= 6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(PROF, f"{_latest_tag()}_bench_cfg*.json"))),
                         ids=lambda p: os.path.basename(p))
def test_bench_lines_close(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["higher_is_better"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    # whole-job throughput = batch / step time; a step is all sparse layers of one token
    batch = d["config"]["global_batch"]
    assert abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0
    assert c["gpu_matches"]["nnz_equal"] is True and c["gpu_matches"]["max_abs_out_diff"] <= 1e-2
    if "cfg0" not in path:
        cfg = re.search(r"(cfg\d)", path).group(1)
        key = "lsh_decode_bytes_per_launch" + ("_clustered" if "clustered" in path else "_byproducts" if "byproducts" in path else "")
        assert r["traffic"] is not None and r["traffic"] >= r["bytes_per_launch"]      # never below the algorithmic bytes
        assert "not measured in this run" in r["traffic_source"]
        # the rocprofv3 average of the same command agrees with the HIP-event time of the line
        stats = open(os.path.join(PROF, os.path.basename(path).replace("_bench_", "_kernel_stats_").replace(".json", ".md"))).read()
        m = re.search(r"\|\s*void mp::lsh_decode_kernel[^|]*\|\s*\d+\s*\|\s*([\d.]+)\s*\|", stats)
        assert m and abs(float(m.group(1)) - r["avg_launch_us"]) <= 0.03 * r["avg_launch_us"], (m and m.group(1), r["avg_launch_us"])
        assert key and cfg


def test_default_line_carries_the_other_configurations():
    """the default single-GPU line (what the round-end driver runs) holds a second and a third
    CONFIGURATION -- cfg 2 and cfg 4's per-GPU share -- each with its own roofline arithmetic, footprint and the
    full-size first layer against the CPU path; the HBM footprint of the headline configuration; no failed leg."""
    path = os.path.join(PROF, f"{_latest_tag()}_bench_driver_style.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert "cfg1" in d["config"]["workload"] and d["n_gpus"] == 1 and "failed_legs" not in d
    legs = dict(d["legs"])
    # round 6: cfg 3's share and the end-to-end step with its split ride in the same line
    assert set(legs) == {"cfg2", "cfg3_share", "cfg4_share", "e2e"}
    e2e = legs.pop("e2e")
    for name in ("cfg1", "cfg2"):
        leg = e2e[name]
        batch = 1 if name == "cfg1" else 8
        assert abs(leg["tokens_per_s"] - batch / (leg["ms_per_step"] * 1e-3)) <= 1e-6 * leg["tokens_per_s"]
        split = leg["split_ms"]
        assert set(split) == {"sparse_attention", "dense_attention", "projections_mlp", "norms_rope_residual", "embed_lm_head"}
        # the parts are re-captured one kind at a time: they add up to the step or less (what the step's own graph overlaps
        # or adds between them), never to more than a few per cent above it
        assert 0.7 * leg["ms_per_step"] <= sum(split.values()) <= 1.05 * leg["ms_per_step"], (name, split)
    for name, leg in legs.items():
        r = leg["roofline"]
        assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) <= 1e-6 * r["achieved"], name
        assert abs(r["frac"] - r["achieved"] / 8000.0) <= 1e-9, name
        assert leg["tokens_per_s"] > 0 and abs(leg["us_per_layer"] * 1e-3 * (75 if name == "cfg4_share" else 30) - leg["ms_per_step"]) < 1e-6
        g = leg["cpu_baseline"]["gpu_matches"]
        assert g["nnz_equal"] is True and g["max_abs_out_diff"] <= 1e-2, name
        f = leg["hbm_bytes_per_layer"]
        assert f["total"] == f["kv"] + f["key_norms"] + f["bounds"] + f["table"] + f["slots"]
    f1 = d["observed"]["hbm_bytes_per_layer"]
    assert f1["slots"] == 32 * 1024 * 8 * 150 * 8 * 4 and f1["slot_bytes"] == 128     # cfg 1: 8 kv groups x L x 2^K x R slots of 128 bytes
    assert legs["cfg2"]["hbm_bytes_per_layer"]["slots"] == 0                             # one workgroup per head: no slots
    for k in ("host_mode", "host_mode_pinned_results"):
        assert d[k]["matches_device_entry"] is True and d[k]["attention_calls_served"]["hits"] >= d[k]["reps"]
