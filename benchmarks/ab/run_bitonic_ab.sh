set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --steps 50 --warmup 5"
cp rag_system_amd/librag_mi355x.so /tmp/lib_default.so
for v in default never always t16 t32; do
  if [ $v = default ]; then cp /tmp/lib_default.so rag_system_amd/librag_mi355x.so; else cp benchmarks/lib_$v.so rag_system_amd/librag_mi355x.so; fi
  for k in 10 64 100; do $B --k $k > gpurun_out/bt_${v}_$k.json; done
done
cp /tmp/lib_default.so rag_system_amd/librag_mi355x.so
python benchmarks/show.py gpurun_out/bt_*.json
