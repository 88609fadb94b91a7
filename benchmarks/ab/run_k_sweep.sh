set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --steps 50 --warmup 5"
rm -f gpurun_out/k_*.json
for k in 10 32 64 65 100 128 200; do $B --k $k > gpurun_out/k_$k.json; done
python benchmarks/show.py gpurun_out/k_*.json
