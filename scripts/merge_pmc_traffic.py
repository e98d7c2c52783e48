"""Merge the two PMC passes of scripts/profile_round.sh (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 --pmc runs) into
profiles/<tag>_pmc_hbm_traffic_<cfg>[_clustered].md and refresh profiles/hbm_traffic_latest.json, the file bench.py
takes `roofline.traffic` from.  HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB = 1024 B; the x 2 is the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md, calibrated for streaming reads).
usage: python scripts/merge_pmc_traffic.py <tag> [src_dir = gpurun_out]"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
out = {"lsh_decode_bytes_per_launch": {}, "lsh_decode_bytes_per_launch_clustered": {}, "lsh_decode_bytes_per_launch_byproducts": {}}


def avg(path):
    for line in open(path):
        m = re.match(r"\|\s*void mp::lsh_decode_kernel.*\|\s*(FETCH_SIZE|WRITE_SIZE)\s*\|\s*(\d+)\s*\|\s*([\d.]+)\s*\|", line)
        if m:
            return float(m.group(3)), int(m.group(2))
    return None, 0          # (a pass that produced no counters: reported and skipped below)


for cfg in ("cfg1", "cfg2", "cfg3", "cfg4"):
    for suffix, key in (("", "lsh_decode_bytes_per_launch"), ("_clustered", "lsh_decode_bytes_per_launch_clustered"),
                        ("_byproducts", "lsh_decode_bytes_per_launch_byproducts")):      # bench.py --by-products 1
        f = os.path.join(src, f"{tag}_pmc_FETCH_SIZE_{cfg}{suffix}.md")
        w = os.path.join(src, f"{tag}_pmc_WRITE_SIZE_{cfg}{suffix}.md")
        b = os.path.join(src, f"{tag}_bench_{cfg}{suffix}.json")
        if not (os.path.exists(f) and os.path.exists(w)):
            continue
        (fk, _), (wk, _) = avg(f), avg(w)
        if fk is None or wk is None:
            print(cfg + suffix, "SKIPPED: a PMC pass without an lsh_decode_kernel row")
            continue
        total = (2 * fk + wk) * 1024
        out[key][cfg] = total
        alg = None
        if os.path.exists(b):
            alg = json.loads(open(b).read().strip().splitlines()[-1])["roofline"]["bytes_per_launch"]
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic_{cfg}{suffix}.md"), "w") as o:
            o.write(open(f).read().rstrip() + "\n\n" + open(w).read().rstrip() + "\n\n")
            o.write(f"# HBM traffic per launch = FETCH_SIZE x 2 + WRITE_SIZE = {total / 1e6:.2f} MB")
            if alg:
                o.write(f"; algorithmic bytes (SURVEY 8d) {alg / 1e6:.2f} MB -> x{total / alg:.2f}")
            o.write("\n")
        print(cfg + suffix, f"{total / 1e6:.2f} MB", f"x{total / alg:.2f}" if alg else "")
out = {"source": f"profiles/{tag}_pmc_hbm_traffic_cfg{{1,2,3,4}}[_clustered|_byproducts].md (FETCH_SIZE x 2 + WRITE_SIZE, separate "
                 "rocprofv3 --pmc passes, KB = 1024 B)", **out}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"), "w"), indent=1)
