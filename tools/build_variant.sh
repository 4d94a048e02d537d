#!/bin/bash
# A/B library of the same ABI with one unit rebuilt under extra -D flags:
#   bash tools/build_variant.sh <name> <unit.hip> "<-D flags>"   ->  wav2lip_amd/lib/libw2l_hip_<name>.so   (select with W2L_HIP_LIB)
set -e
NAME=$1; UNIT=$2; DEFS=$3
cd "$(dirname "$0")/../wav2lip_amd/csrc"
mkdir -p build/$NAME
cp build/*.o build/$NAME/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $DEFS -c $UNIT -o build/$NAME/${UNIT%.hip}.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libw2l_hip_$NAME.so build/$NAME/*.o
echo built $NAME
