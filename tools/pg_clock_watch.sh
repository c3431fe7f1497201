#!/bin/bash
mkdir -p gpurun_out/r5s
python tools/pg_time.py > gpurun_out/r5s/pg_clk_run.txt 2>&1 &
PID=$!
for i in $(seq 1 60); do
  echo "t=$(date +%s.%N) $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1)" >> gpurun_out/r5s/pg_clk.txt
  sleep 0.2
  kill -0 $PID 2>/dev/null || break
done
wait $PID
