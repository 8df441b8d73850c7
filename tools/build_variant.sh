#!/bin/bash
# A/B builds of the same ABI: tools/build_variant.sh NAME "-DFLAG=0 ..."  ->  libdrs_NAME.so (git-ignored, travels to the GPU box)
# Use: DRS_LIB=$PWD/libdrs_NAME.so python bench.py ...   (compare variants inside ONE gpurun call: the pool's boxes differ)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../diffusionremotesensing_amd/csrc"
mkdir -p build_$NAME
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c "$f" -o build_$NAME/${f%.hip}.o &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../libdrs_$NAME.so build_$NAME/*.o
echo built libdrs_$NAME.so
