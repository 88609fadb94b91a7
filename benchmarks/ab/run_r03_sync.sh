#!/bin/bash
# r03: what a batch-1 step costs beyond its kernel, by the way the host waits (hipSetDeviceFlags)
mkdir -p gpurun_out/r03
for m in auto spin yield; do
  echo "== RAG_SYNC_MODE=$m"
  RAG_SYNC_MODE=$m timeout -k 10 200 python3 benchmarks/step_overhead.py 2>&1 | grep "ms/step\|stream" 
done
