#!/bin/bash
# Build a variant of the HIP library next to the product one: tools/build_variant.sh <name> "<extra hipcc flags>"
#   -> gym_auv_amd/csrc_<name>/libauv_hip.so   (select it with AUV_HIP_LIB=...; ignored by git, travels with gpurun)
# e.g. tools/build_variant.sh stamps "-DAUV_STAMPS"  (in-kernel phase stamps; the product build never contains one)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; EXTRA=$2
D=$ROOT/gym_auv_amd/csrc_$NAME
mkdir -p $D
cp $ROOT/gym_auv_amd/csrc/*.hip $ROOT/gym_auv_amd/csrc/*.h $ROOT/gym_auv_amd/csrc/Makefile $D/
make -C $D -j8 CXXFLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wall -Wextra -Wno-unused-parameter $EXTRA" all 2>&1 | grep -E "error|Error" || true
ls -la $D/libauv_hip.so
