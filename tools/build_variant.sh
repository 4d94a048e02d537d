#!/bin/bash
# A/B library of the same ABI with one unit rebuilt under extra -D flags:
#   bash tools/build_variant.sh <name> <unit.hip> "<-D flags>" [replaces]  ->  wav2lip_amd/lib/libw2l_hip_<name>.so   (select with W2L_HIP_LIB)
# <unit.hip> is relative to wav2lip_amd/csrc; an experiment unit outside it (a scratch path) names the unit it
# stands in for as the fourth argument (e.g. conv_wino4).  `make -C wav2lip_amd/csrc` must have built build/*.o first.
set -e
NAME=$1; UNIT=$2; DEFS=$3; REPL=${4:-$(basename ${2%.hip})}
cd "$(dirname "$0")/../wav2lip_amd/csrc"
mkdir -p build/$NAME
cp build/*.o build/$NAME/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. $DEFS -c $UNIT -o build/$NAME/$REPL.o 2>build/$NAME/$REPL.log || { tail -30 build/$NAME/$REPL.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libw2l_hip_$NAME.so build/$NAME/*.o
echo built $NAME
