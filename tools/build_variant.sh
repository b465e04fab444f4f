#!/bin/bash
# Build quinoa_amd/lib/libqdg_<name>.so with extra -D flags on the kernel file and the host
# layer (tile constants are shared through qdg_device.hpp); the other objects of the regular
# build are reused.  Usage: tools/build_variant.sh NAME "-DFOO=1 -DBAR"
# Select it at run time with QDG_LIB=quinoa_amd/lib/libqdg_NAME.so (A/B and knock-out runs).
set -e
name=$1; flags=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
obj=$root/quinoa_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $root/quinoa_amd/csrc/qdg_kernels.hip -o $obj/qdg_kernels_$name.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $root/quinoa_amd/csrc/qdg_api.cpp -o $obj/qdg_api_$name.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o $root/quinoa_amd/lib/libqdg_$name.so $obj/qdg_kernels_$name.o \
  $obj/qdg_devmesh.o $obj/qdg_api_$name.o $obj/qdg_meshdata.o $obj/qdg_partition.o $obj/qdg_exo.o -ldl
echo built libqdg_$name.so
