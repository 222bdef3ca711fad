#!/bin/bash
# Samples the GPU's power draw and shader clock (rocm-smi / amd-smi, whichever answers) while the C1 step loop runs: is the step held by the
# chip's power budget?  Output: gpurun_out/power_probe.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
: > $OUT
(python bench.py --steps 1500 --warmup 20 --no-cpu-baseline > gpurun_out/power_probe_bench.json 2> gpurun_out/power_probe_bench.err) &
BP=$!
sleep 25   # imports, engine creation, warm-up
for i in $(seq 1 12); do
  echo "--- sample $i $(date +%s.%N)" >> $OUT
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory|hotspot)" >> $OUT
  sleep 0.5
done
wait $BP
echo "--- idle" >> $OUT
sleep 3
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" >> $OUT
rocm-smi --showmaxpower 2>&1 | grep -iE "power" >> $OUT
python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/power_probe_bench.json') if l.startswith('{')][-1]
print('bench ms/step', d['ms_per_step'])" >> $OUT
cat $OUT
