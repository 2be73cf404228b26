#!/bin/bash
# Builds an A/B variant of libolap_gpu.so with extra compiler flags for join.hip:
#   scripts/build_variant.sh NAME "-DOLAP_RING_NT_LOAD=1 ..."   ->  ab_libs/NAME.so
# Run variants on the GPU box with OLAP_GPU_LIB=$PWD/ab_libs/NAME.so (scripts/ab_libs.sh).
set -e
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/dpu_olap_amd/csrc
mkdir -p $ROOT/ab_libs /tmp/ab_obj
python -m dpu_olap_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I $ROOT/include -I $C $FLAGS -c $C/join.hip -o /tmp/ab_obj/join_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/$NAME.so $C/build/ctx.o $C/build/stage.o $C/build/scan_ops.o /tmp/ab_obj/join_$NAME.o $C/build/gen.o $C/build/arrow_c.o $C/build/group.o -ldl -lpthread
echo $ROOT/ab_libs/$NAME.so
