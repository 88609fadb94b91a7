#!/bin/bash
# Ablation / A-B builds: recompile ONE csrc/*.hip with extra flags and link it with the other
# objects of the normal build into benchmarks/lib_<name>.so (select it with RAG_MI355X_LIB).
#   bash benchmarks/build_variant.sh <name> <file.hip> <flags...>
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
python -m rag_system_amd.build > /dev/null
mkdir -p $root/build/variant
obj=$root/build/variant/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC "$@" -c $root/rag_system_amd/csrc/$src -o $obj 2> /dev/null
others=$(ls $root/build/obj/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/benchmarks/lib_$name.so $obj $others
echo $root/benchmarks/lib_$name.so
