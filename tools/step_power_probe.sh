#!/usr/bin/env bash
# Socket power and shader clock (rocm-smi, every 0.25 s) while the product sampler runs its real 1000-step chain at B = 32:
# is the STEP held by the board's power cap, as the dense random-operand convolution benchmark is (profiles/HISTORY.md 4g)?
cd "$(dirname "$0")/.."
OUT=gpurun_out/step_power_probe.txt; mkdir -p gpurun_out; : > $OUT
python bench.py --no-cpu-baseline --no-dense-check --repeats 1 > gpurun_out/step_power_probe_line.json 2>/dev/null &
PID=$!
while kill -0 $PID 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -o "Socket Graphics Package Power (W): [0-9.]*\|sclk clock level: [0-9S]: ([0-9]*Mhz)" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.25
done
wait $PID
sort $OUT | uniq -c | sort -k1 -n -r | head -40
