"""The numbers DESIGN.md section 5, README.md and profiles/README.md quote, read back from the committed evidence of a round:
    python tools/doc_numbers.py [r05]
(one place to look when the evidence is regenerated: the documents are edited by hand from this output)."""
import csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def line(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


b = line(f"{tag}_h3_bench.json")
r = b["roofline"]
print(f"step {b['ms_per_step']:.2f} ms = {b['value']:.3f} volumes/s; fp32 mode {b['fp32_mode']['ms_per_step']:.1f} ms; full chain {b['full_chain']['full_chain_s']:.2f} s")
print(f"dominant kernel: {r['achieved']:.0f} TFLOP/s (frac {r['frac']:.3f}), {r['avg_launch_ms']:.3f} ms x {r['launches_per_step']}, traffic {r['traffic'] / 1e6:.1f} MB vs "
      f"{r['algorithmic_mb_per_launch']:.1f} algorithmic, mfma_busy_frac {r['mfma_busy_frac']}, clock {r['clock_ghz']} GHz")
print("MFMA accountings (dominant / all convs): executed", b["conv_mfma_executed_pct"], "useful", b["conv_mfma_useful_pct"], "algorithmic", b["conv_algorithmic_pct_of_peak"])
s2 = r["secondary"]
print(f"secondary: {s2['achieved']:.0f} TFLOP/s, traffic / algorithmic {s2['traffic_over_algorithmic']}")
print("kinds:", {k: (v["launches_per_step"], round(v["ms_per_step"], 3)) for k, v in b["per_kernel_kind"].items()})
print("cpu baseline:", b["cpu_baseline"]["value"], b["cpu_baseline"]["sample"][-40:])
for f, what in ((f"{tag}_h3_config2_bench.json", "config 2"), (f"{tag}_h3_b1_bench.json", "B = 1"), (f"{tag}_h3_groupnorm_bench.json", "norm=group")):
    d = line(f)
    print(f"{what}: {d['ms_per_step']:.3f} ms, {d['value']:.4f} volumes/s")
for name in (f"{tag}_h3_sq.csv", f"{tag}_h3_kernel_stats.csv", f"{tag}_h3_pmc_hbm.csv"):
    rows = [l for l in open(os.path.join(P, name)) if not l.startswith("#")]
    hdr = next(csv.reader(rows[:1]))
    print(name, "| workload:", next((l.strip() for l in open(os.path.join(P, name)) if l.startswith("# workload")), "-"))
    for row in csv.reader(rows[1:4]):
        d = dict(zip(hdr, row))
        keys = [k for k in ("kernel", "Name", "Calls", "AverageNs", "avg_us_profiled", "mfma_busy_frac", "lds_conflict", "clock_ghz_est", "avg_hbm_MB_per_launch_corrected") if k in d]
        print("   ", {k: d[k][:48] for k in keys})
for name in (f"{tag}_gpu_tests.log", f"{tag}_chain_repeatability.log", f"{tag}_e2e_config5.log"):
    print(name, "|", open(os.path.join(P, name)).read().strip().splitlines()[-1][:160])
