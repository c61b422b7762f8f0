// vsyn_fused.h — fused single-pass synthesis kernels (speed path).  Placeholder until the wave-level
// kernels land: reports "unsupported" so every submit takes the staged kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "vsyn_device.h"

struct FusedTables {
  void* d_tables = nullptr;
};

static inline hipError_t fused_tables_create(const ConstHeader&, const uint8_t*, FusedTables*) { return hipSuccess; }
static inline void fused_tables_destroy(FusedTables*) {}
static inline bool fused_supported(const ConstHeader&) { return false; }
static inline const char* fused_kernel_name(const ConstHeader&) { return "none"; }
static inline const char* fused_imdct_kernel_name(uint32_t) { return "vsyn_imdct_plain_kernel"; }
static inline hipError_t fused_launch(const ConstHeader&, const uint8_t*, const FusedTables&, uint32_t, uint32_t, const vsyn_segment*,
                                      uint32_t, const PktInfo*, const SegInfo*, const float*, const uint16_t*, float*, uint64_t,
                                      float*, DevStatus*, hipStream_t) {
  return hipErrorNotSupported;
}
static inline hipError_t fused_imdct_launch(const ConstHeader&, const uint8_t*, const FusedTables&, int, uint32_t, uint32_t,
                                            const float*, float*, hipStream_t, bool* done) {
  *done = false;
  return hipSuccess;
}
