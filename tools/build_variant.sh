#!/bin/bash
# A/B builds of the same ABI: tools/build_variant.sh NAME "-DFLAG=0 ..."  ->  variants/libdrs_NAME.so
# variants/ is git-ignored scratch: it travels to the GPU box with a gpurun push while it exists, so delete it when the
# experiment is over (nothing in the product or the tests loads from it).
# Use: DRS_LIB=$PWD/variants/libdrs_NAME.so python bench.py ...   (compare variants inside ONE gpurun call: the pool's boxes differ)
set -e
NAME=$1; FLAGS=$2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/variants
mkdir -p $OUT/obj_$NAME
cd $ROOT/diffusionremotesensing_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c "$f" -o $OUT/obj_$NAME/${f%.hip}.o &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdrs_$NAME.so $OUT/obj_$NAME/*.o
rm -rf $OUT/obj_$NAME
echo built variants/libdrs_$NAME.so
