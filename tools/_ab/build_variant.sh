#!/bin/bash
# tools/_ab/build_variant.sh <name> <file.hip> "<-D flags>": a copy of the library with ONE source rebuilt with extra flags -> tools/_ab/lib_<name>.so
cd $(dirname $0)/../../centernet-pytorch-lightning_amd/csrc
base=$(basename $2 .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $3 -c $2 -o /tmp/${base}_$1.o || exit 1
objs=$(ls _build/*.o | grep -v "_build/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_$1.so /tmp/${base}_$1.o $objs && echo built lib_$1.so
