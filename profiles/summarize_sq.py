"""Summarise rocprofv3 --pmc SQ-counter passes into per-kernel averages per launch, next to profiles/summarize_pmc.py (HBM bytes).
usage: python profiles/summarize_sq.py <out.csv> "<header comment>" "<workload signature>" <pass_dir> [<pass_dir> ...]
Each pass_dir holds one *counter_collection.csv of a `rocprofv3 --pmc <up to 8 SQ counters> GRBM_GUI_ACTIVE --kernel-trace` run.
Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* tick in quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs.  Derived columns:
  kernel_cycles   = GRBM_GUI_ACTIVE / 8
  clock_ghz_est   = kernel_cycles / (End - Start timestamp)     (reads high on dispatches shorter than ~0.3 ms; the in-kernel clock of
                    tools/kernel_clock.py is the reference)
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles x 1024 SIMDs)
  valu_per_mfma   = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
  lds_conflict    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac       = SQ_WAIT_ANY / SQ_WAVE_CYCLES,  stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main(out_path, comment, workload, *dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in dirs:
        files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
        assert files, d
        seen = set()
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            key = (d, r["Dispatch_Id"])
            if key not in seen and d == dirs[0]:
                seen.add(key)
                dur[k][0] += 1
                dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    names = sorted({c for v in agg.values() for c in v})
    lines = ["# " + comment, "# workload: " + workload,
             "kernel,launches,avg_us_profiled," + ",".join(names) + ",kernel_cycles,clock_ghz_est,mfma_busy_frac,valu_per_mfma,lds_conflict,wait_frac,stall_frac"]
    rows = []
    for k, v in agg.items():
        avg = {c: v[c][1] / max(v[c][0], 1) for c in names}
        n = dur[k][0]
        us = dur[k][1] / max(n, 1) / 1e3
        cyc = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8
        g = lambda c: avg.get(c, 0.0)
        d = lambda a, b: (a / b) if b else 0.0
        rows.append((us * n, f'"{k}",{n},{us:.1f},' + ",".join(f"{avg[c]:.0f}" for c in names)
                     + f",{cyc:.0f},{d(cyc, us * 1e3):.3f},{d(g('SQ_VALU_MFMA_BUSY_CYCLES'), cyc * 1024):.3f},"
                     + f"{d(g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA'), g('SQ_INSTS_MFMA')):.2f},{d(g('SQ_LDS_BANK_CONFLICT'), g('SQ_LDS_IDX_ACTIVE')):.3f},"
                     + f"{d(g('SQ_WAIT_ANY'), g('SQ_WAVE_CYCLES')):.3f},{d(g('SQ_WAIT_INST_ANY'), g('SQ_WAVE_CYCLES')):.3f}"))
    lines += [r for _, r in sorted(rows, key=lambda t: -t[0])]
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(l[:400] for l in lines[:14]))


if __name__ == "__main__":
    main(*sys.argv[1:])
