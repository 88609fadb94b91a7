set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 100 --warmup 10"
cp rag_system_amd/librag_mi355x.so /tmp/lib_default.so
for cfg in default:1024 default:768 t768:256; do
  v=${cfg%%:*}; blocks=${cfg##*:}
  if [ $v = default ]; then cp /tmp/lib_default.so rag_system_amd/librag_mi355x.so; else cp benchmarks/lib_$v.so rag_system_amd/librag_mi355x.so; fi
  O="--opt scan_blocks=$blocks"
  $B $O --k 100 > gpurun_out/td_${v}_b${blocks}_10m_k100.json
  $B $O --rows 1000000 > gpurun_out/td_${v}_b${blocks}_1m_k10.json
  $B $O --rows 1000000 --k 90 > gpurun_out/td_${v}_b${blocks}_1m_k90.json
  $B $O --rows 1250000 > gpurun_out/td_${v}_b${blocks}_1.25m_k10.json
  $B $O --nq 2 > gpurun_out/td_${v}_b${blocks}_10m_nq2.json
  $B $O --storage bf16 --dim 768 > gpurun_out/td_${v}_b${blocks}_bf16.json
  $B $O --rows 12500000 --dim 768 --steps 30 > gpurun_out/td_${v}_b${blocks}_c4shard.json
  $B $O --k 256 > gpurun_out/td_${v}_b${blocks}_10m_k256.json
done
cp /tmp/lib_default.so rag_system_amd/librag_mi355x.so
python benchmarks/show.py gpurun_out/td_*.json | cut -c1-40,96-200
