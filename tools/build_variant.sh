#!/bin/bash
# Build a variant of libwhenet_hip.so with one source recompiled under extra flags (A/B experiments):
#   tools/build_variant.sh <name> <source in csrc> <flags...>  ->  headposeestimation-whenet_amd/lib/variants/lib_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/headposeestimation-whenet_amd
NAME=$1; SRC=$2; shift 2
mkdir -p $P/lib/variants $P/build/variants
O=$P/build/variants/${NAME}_$(echo $SRC | tr . _).o
EXTRA=""; [ "$SRC" = "front2.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -mcode-object-version=5 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden $EXTRA "$@" -c $P/csrc/$SRC -o $O
OBJS=$(ls $P/build/*.o | grep -v "$(echo $SRC | tr . _).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/variants/lib_$NAME.so $OBJS $O -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built $P/lib/variants/lib_$NAME.so
