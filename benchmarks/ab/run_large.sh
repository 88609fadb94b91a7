#!/bin/bash
# Large shards: one rank's share of config C4 (12.5M x 768 fp32 = 38.4 GB) and 100M x 384 (153.6 GB) on one GPU.
set -e
mkdir -p gpurun_out/large
B="timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-batch-leg"
$B --rows 12500000 --dim 768 --steps 50 --warmup 5 > gpurun_out/large/c4_shard.json
echo c4 done
$B --rows 100000000 --steps 20 --warmup 3 > gpurun_out/large/100m_nq1.json
echo 100m nq1 done
python3 benchmarks/show.py gpurun_out/large/*.json
