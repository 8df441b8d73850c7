#!/bin/bash
# diagnostic build with in-kernel phase stamps (s_memtime) in the SP wave-specialised kernels: variants/libdrs_tl.so
# Use: DRS_LIB=$PWD/variants/libdrs_tl.so DRS_CONCURRENT=0 python tools/per_op_table.py --iters 1
exec "$(dirname "$0")/build_variant.sh" tl "-DDRS_SP_TIMELINE"
