#!/bin/bash
# Build gpurun_ab_<name>.so = the in-tree objects with ONE source recompiled under extra flags (same-box A/B, tools/gpu_ab_libs.sh).
# usage: tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]     (run `python -m sudo_rm_rf_amd.build` first)
set -eu
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/sudo_rm_rf_amd/csrc/build
extra="-fno-slp-vectorize"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $extra "$@" -I$R/include -c $R/sudo_rm_rf_amd/csrc/$src -o /tmp/variant_$name.o 2>&1 | grep -v "argument unused" || true
objs=$(ls $O/*.o | grep -v "/${src%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_ab_$name.so $objs /tmp/variant_$name.o
ls -la $R/gpurun_ab_$name.so
