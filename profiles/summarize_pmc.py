"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel average HBM bytes per launch.
usage: python profiles/summarize_pmc.py <fetch_dir> <write_dir> <out.csv> "<header comment>" ["<workload signature>"]
The workload signature (bench.py: "batch=.. size=.. channels=.. norm=.. precision=.. csrc=<digest of the kernel sources>") is
written as a "# workload: ..." line; bench.py attaches roofline.traffic only from a summary whose signature matches its run.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts half of the bytes of 16-B/lane coalesced reads, so
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main(fetch_dir, write_dir, out_path, comment, workload=None):
    agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
    for d in (fetch_dir, write_dir):
        for r in csv.DictReader(open(f"{d}/pmc_counter_collection.csv")):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    lines = ["# " + comment] + (["# workload: " + workload] if workload else []) + ["kernel,launches,avg_FETCH_SIZE_KB,avg_WRITE_SIZE_KB,avg_hbm_MB_per_launch_corrected"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["FETCH_SIZE"][1]):
        f = v["FETCH_SIZE"][1] / max(v["FETCH_SIZE"][0], 1)
        w = v["WRITE_SIZE"][1] / max(v["WRITE_SIZE"][0], 1)
        lines.append(f'"{k}",{v["FETCH_SIZE"][0]},{f:.1f},{w:.1f},{(2 * f + w) * 1024 / 1e6:.1f}')
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
