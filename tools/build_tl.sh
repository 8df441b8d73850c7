#!/bin/bash
# diagnostic build with in-kernel phase stamps (s_memtime) in the SP wave-specialised kernel: libdrs_tl.so (git-ignored).
# Use: DRS_LIB=$PWD/libdrs_tl.so DRS_CONCURRENT=0 python tools/per_op_table.py --iters 1
set -e
cd "$(dirname "$0")/../diffusionremotesensing_amd/csrc"
mkdir -p build_tl
for f in *.hip; do
  o=build_tl/${f%.hip}.o
  stale=0
  for h in *.h *.inc; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRS_SP_TIMELINE -c "$f" -o "$o" &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../libdrs_tl.so build_tl/*.o
echo built libdrs_tl.so
