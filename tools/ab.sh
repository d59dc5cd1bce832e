#!/bin/bash
# Developer helper: build libptamd variants with extra -D flags for A/B timing on one GPU box.
# usage: tools/ab.sh name1 "-DFOO=1 -DBAR=2" name2 "..."     -> build/ab/libptamd_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
CSRC=$ROOT/directx-physically-based-raytracer_amd/csrc
OUT=$ROOT/build/ab
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w"
make -s -C $CSRC
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/pt_kernels.hip -o $OUT/pt_kernels_$name.o &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/pt_stream.hip -o $OUT/pt_stream_$name.o &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/pt_bvh.hip -o $OUT/pt_bvh_$name.o &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/pt_api.hip -o $OUT/pt_api_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libptamd_$name.so $OUT/pt_api_$name.o $OUT/pt_bvh_$name.o $OUT/pt_kernels_$name.o $OUT/pt_stream_$name.o $CSRC/pt_skin.o $CSRC/pt_comm.o -ldl &&
    echo built $name ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
rm -f $OUT/*.o
