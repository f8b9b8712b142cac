#!/bin/bash
# Clock, temperature and socket power while the bench runs a sustained chain of 3000 steps (about 50 s), once with the
# Winograd-x form of the k3 convs (default) and once with the direct kernel only (DM3D_CONV_WINO=0).  Every sample and the
# bench's own progress lines carry wall-clock seconds, so the samples inside the timed region can be told from the rest.
# usage (on the GPU box): bash tools/clocks_under_load.sh gpurun_out/clocks.log
out=${1:-gpurun_out/clocks_under_load.log}
mkdir -p "$(dirname "$out")"; : > "$out"
for wino in 1 0; do
  echo "=== DM3D_CONV_WINO=$wino: python3 bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-fp32-mode --no-full-chain" >> "$out"
  SECONDS=0
  DM3D_CONV_WINO=$wino timeout -k 10 300 python3 bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-fp32-mode \
      --no-full-chain 2>&1 | while IFS= read -r line; do printf '%4d s  %s\n' "$SECONDS" "$line"; done > "$out.bench$wino" &
  pid=$!
  sleep 10
  while kill -0 $pid 2>/dev/null; do
    s=$(rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "junction|sclk|Socket" | sed 's/.*: //' | tr '\n' ' ')
    printf '%4d s  %s\n' "$SECONDS" "$s" >> "$out"
    sleep 3
  done
  wait $pid
  grep -E "model prepared|warm-up step done|timed [0-9]+ steps" "$out.bench$wino" | sed -n '1p;$p' >> "$out"
  grep -E "timed [0-9]+ steps" "$out.bench$wino" >> "$out" || { echo "bench failed" >> "$out"; tail -20 "$out.bench$wino" >> "$out"; exit 1; }
  rm -f "$out.bench$wino"
done
