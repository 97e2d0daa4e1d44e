#!/bin/bash
# build a variant of the library for a same-box A/B (tools/gpu_ab_lib.sh): tools/build_variant.sh <tag> "<-D flags>" <unit.hip> [<unit.hip> ...]
# -> pydem_amd/lib/libpydem_hip.so.<tag> = the product objects with the named units recompiled under the extra flags
set -e
TAG=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/.."
python -m pydem_amd.build > /dev/null          # the product objects the variant is linked from
cd pydem_amd
EXCL=""; OBJS=""
for u in "$@"; do
  o=/tmp/variant_${TAG}_${u%.*}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result $FLAGS -c csrc/$u -o $o
  EXCL="$EXCL ${u%.*}.o"; OBJS="$OBJS $o"
done
for o in lib/*.o; do b=$(basename $o); case " $EXCL " in *" $b "*) ;; *) OBJS="$OBJS $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libpydem_hip.so.$TAG $OBJS -L/opt/rocm/lib -lrccl
ls -la lib/libpydem_hip.so.$TAG
