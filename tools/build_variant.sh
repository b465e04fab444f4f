#!/bin/bash
# Builds a VARIANT of libqdg.so with extra compiler flags (A/B runs of compile-time parameters such
# as the tile size): tools/build_variant.sh NAME -DQDG_TILE=160 ...  ->  quinoa_amd/lib/variants/NAME/libqdg.so
# (git-ignored, travels with gpurun); select it with QDG_LIB=<path> (quinoa_amd/capi.py).
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/quinoa_amd/lib/variants/$name; mkdir -p $out
cd $root/quinoa_amd/csrc || exit 1
for f in qdg_rhs_p1.hip qdg_rhs_p2.hip qdg_kernels.hip qdg_devmesh.hip qdg_api.cpp qdg_meshdata.cpp qdg_partition.cpp qdg_exo.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o $out/${f%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o $out/libqdg.so $out/*.o -ldl && rm -f $out/*.o && echo "built $out/libqdg.so"
