#!/usr/bin/env bash
# Socket power and shader clock (rocm-smi) while the split convolution runs for a few seconds on random / zero operands
# and while a bare MFMA stream runs: is the convolution's clock taken by the power management?  -> gpurun_out/conv_power_probe.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out/conv_power_probe.txt; mkdir -p gpurun_out; : > $OUT
sample() { # label, seconds
  local end=$((SECONDS + $2))
  while [ $SECONDS -lt $end ]; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' | sed "s/^/$1: /" >> $OUT; echo >> $OUT
    sleep 0.3
  done
}
cat > /tmp/conv_loop.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from lion_amd.conv_ops import conv3d_k3
mode, secs = sys.argv[1], float(sys.argv[2])
conv = torch.nn.Conv3d(64, 64, 3, padding=1).cuda()
x = torch.randn(32, 64, 32, 32, 32, device="cuda") if mode == "random" else torch.zeros(32, 64, 32, 32, 32, device="cuda")
with torch.no_grad():
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(200): conv3d_k3(x, conv.weight, conv.bias, split=True)
        torch.cuda.synchronize(); n += 200
    print(f"{mode}: {(time.time() - t0) / n * 1e6:.0f} us per launch over {n} launches", flush=True)
PY
sample idle 1
for mode in random zero; do
  python /tmp/conv_loop.py $mode 5 >> $OUT 2>/dev/null &
  PID=$!
  sleep 1.5   # import + warm-up
  sample conv_$mode 3
  wait $PID
done
( ./tools/exp/mfma_issue_probe 300 2>/dev/null | head -3 >> $OUT ) &
PID=$!
sleep 0.3
sample bare_mfma 2
wait $PID
cat $OUT
