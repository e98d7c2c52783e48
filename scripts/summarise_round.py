"""Markdown table of a round's committed bench lines (profiles/<tag>_bench_<cfg>[_clustered].json), kernel stats
(profiles/<tag>_kernel_stats_*.md) and PMC traffic (profiles/hbm_traffic_latest.json): the rows of DESIGN.md section 5.
usage: python scripts/summarise_round.py r06"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")
rows = [("cfg0", ""), ("cfg1", ""), ("cfg1", "_byproducts"), ("cfg1", "_clustered"), ("cfg2", ""), ("cfg2", "_clustered"), ("cfg3", ""), ("cfg4", "")]
print("| config | tokens/s | us / launch: HIP events; rocprofv3 avg (step / layers) | algorithmic MB | GB/s (of 8 TB/s) | PMC MB (x) | R | nnz / head | CPU reference tokens/s (retrieve + attention us) | host mode us / layer |")
print("|---|---|---|---|---|---|---|---|---|---|")
for cfg, suf in rows:
    f = os.path.join(P, f"{tag}_bench_{cfg}{suf}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, c, hm = d["roofline"], d.get("cpu_baseline", {}), d.get("host_mode") or {}
    ks = os.path.join(P, f"{tag}_kernel_stats_{cfg}{suf}.md")
    kavg = "-"
    if os.path.exists(ks):
        for line in open(ks):
            m = re.match(r"\|\s*void mp::lsh_decode_kernel[^|]*\|\s*\d+\s*\|\s*([\d.]+)\s*\|", line)
            if m:
                kavg = m.group(1)
                break
    tr = r.get("traffic")
    print(f"| {cfg}{suf.replace('_', ' ')} | {d['value']:.0f} | {r['avg_launch_us']:.2f}; {kavg} ({d['sparse_attn_us_per_layer']:.2f}) | "
          f"{r['bytes_per_launch'] / 1e6:.1f} | {r['achieved']:.0f} ({r['frac']:.3f}) | "
          + (f"{tr / 1e6:.1f} ({tr / r['bytes_per_launch']:.2f}x)" if tr else "-") +
          f" | {d['observed']['probed_pieces']['ranges_per_head']} | {d['observed']['nnz_per_head']:.0f} ({100 * d['observed']['selected_fraction']:.2f} %) | "
          f"{c.get('value', 0):.0f} ({c.get('t_retrieve_us', 0):.0f} + {c.get('t_attention_us', 0):.0f}) | {hm.get('us_per_layer') or 0:.0f} |")
    if d.get("value_clustered"):
        print(f"|   same run, second workload (clustered keys, heavy-hitter queries) | {d['value_clustered']:.0f} | ({d['sparse_attn_us_per_layer_clustered']:.2f}) | | | | | {d['observed_clustered']['nnz_per_head']:.0f} | | |")
