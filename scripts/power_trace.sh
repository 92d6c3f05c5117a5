#!/bin/bash
# Samples clocks and socket power with rocm-smi while bench.py runs (is the step power/clock limited?).
# usage: bash scripts/power_trace.sh [out_prefix]
out=${1:-gpurun_out/power}
python bench.py --steps 150 --warmup 5 > ${out}_bench.log 2>&1 &
pid=$!
for i in $(seq 1 400); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i $(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Power' | tr -s ' ' | tr '\n' '|')" >> ${out}_smi.log
  sleep 0.25
done
wait $pid
tail -n 1 ${out}_bench.log | cut -c1-200
