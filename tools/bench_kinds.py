"""One line per bench.py run: ms_per_step and the per-kind table.  usage: python tools/bench_kinds.py <tag> [bench.py args...]"""
import json, subprocess, sys, os
tag, args = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args, "--no-cpu-baseline", "--no-fp32-mode", "--no-full-chain"], capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
k = d["per_kernel_kind"]
print(tag, round(d["ms_per_step"], 3), " ".join(f"{x}:{k[x]['launches_per_step']}x={k[x]['ms_per_step']:.3f}" for x in sorted(k)), flush=True)
