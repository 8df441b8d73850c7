// LDS-tiled implicit-GEMM tap-convolution on the matrix cores (placeholder until the MFMA kernels land).
#include "drs_common.h"

bool drs_tapconv_mfma_supported(const TapConv& d, int impl) {
  (void)d; (void)impl;
  return false;
}
int drs_launch_tapconv_mfma(const TapConv& d, int impl, hipStream_t s) {
  (void)d; (void)impl; (void)s;
  DrsErr::set("tapconv_mfma: not built");
  return DRS_ERR_ARG;
}
