// tools/fft_xchg_bench.hip — A/B microbenchmark for the two index exchanges of the wave-level FFT-512 (vsyn_fused.h):
//   variant 0  through a wave-private padded LDS image (what fft512_wave does),
//   variant 1  in registers: v_permlane32_swap / v_permlane16_swap for lane bits 5 and 4, DPP row shifts with bank masks
//              for lane bits 3 and 2, DPP quad_perm + select for lane bits 1 and 0 (no LDS traffic, no LDS round trip).
// Both move the same values to the same places, so the outputs must be bit-identical; the run checks that and times a
// loop of FFTs at the fused kernel's occupancy (8-wave workgroups, two per CU).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/fft_xchg_bench tools/fft_xchg_bench.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Tw {
  float2 tw1[8][64];
  float2 tw2[8][8];
};

__device__ __forceinline__ float2 f2(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return f2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return f2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return f2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return f2(a.y, -a.x); }
__device__ __forceinline__ void dft8(float2 (&x)[8]) {
  const float h = 0.70710678118654752440f;
  float2 a0 = cadd(x[0], x[4]), a1 = cadd(x[1], x[5]), a2 = cadd(x[2], x[6]), a3 = cadd(x[3], x[7]);
  float2 b0 = csub(x[0], x[4]), b1 = csub(x[1], x[5]), b2 = csub(x[2], x[6]), b3 = csub(x[3], x[7]);
  b1 = f2((b1.x + b1.y) * h, (b1.y - b1.x) * h);
  b2 = mul_mi(b2);
  b3 = f2((b3.y - b3.x) * h, -(b3.x + b3.y) * h);
  float2 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), c3 = mul_mi(csub(a1, a3));
  x[0] = cadd(c0, c2); x[4] = csub(c0, c2); x[2] = cadd(c1, c3); x[6] = csub(c1, c3);
  float2 d0 = cadd(b0, b2), d1 = csub(b0, b2), d2 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
  x[1] = cadd(d0, d2); x[5] = csub(d0, d2); x[3] = cadd(d1, d3); x[7] = csub(d1, d3);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(float& a, float& b) {  // a[32..63] <-> b[0..31]
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {  // odd rows of a <-> even rows of b
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// lo/hi: registers whose index differs in the bit being transposed with lane bit 3 (SH = 8) or 2 (SH = 4)
template <int SH>
__device__ __forceinline__ void swap_row(float& lo, float& hi) {
  constexpr int shr = 0x110 + SH, shl = 0x100 + SH;
  constexpr int up = SH == 8 ? 0xC : 0xA, dn = SH == 8 ? 0x3 : 0x5;
  const unsigned l = __float_as_uint(lo), h = __float_as_uint(hi);
  const unsigned nl = __builtin_amdgcn_update_dpp(l, h, shr, 0xF, up, false);  // lanes with the bit set: hi of lane - SH
  const unsigned nh = __builtin_amdgcn_update_dpp(h, l, shl, 0xF, dn, false);  // lanes with the bit clear: lo of lane + SH
  lo = __uint_as_float(nl);
  hi = __uint_as_float(nh);
}
template <int BIT>  // lane bit 1 or 0: quad_perm + select
__device__ __forceinline__ void swap_quad(float& lo, float& hi, bool bitset) {
  constexpr int qp = BIT == 1 ? 0x4E : 0xB1;
  const unsigned l = __float_as_uint(lo), h = __float_as_uint(hi);
  const unsigned th = __builtin_amdgcn_mov_dpp(h, qp, 0xF, 0xF, false);
  const unsigned tl = __builtin_amdgcn_mov_dpp(l, qp, 0xF, 0xF, false);
  lo = __uint_as_float(bitset ? th : l);
  hi = __uint_as_float(bitset ? h : tl);
}

template <int VARIANT>
__device__ __forceinline__ void fft512(float2 (&z)[8], float2* __restrict__ xb, const Tw* __restrict__ T, uint32_t lane) {
  const uint32_t c = lane & 7u, hi = lane >> 3;
  dft8(z);
#pragma unroll
  for (int t = 1; t < 8; ++t) z[t] = cmulf(z[t], T->tw1[t][lane]);
  if (VARIANT == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) xb[t * 72 + lane] = z[t];
#pragma unroll
    for (int a = 0; a < 8; ++a) z[a] = xb[hi * 72 + a * 8 + c];
  } else {
    // register bit 2 <-> lane bit 5, bit 1 <-> lane bit 4, bit 0 <-> lane bit 3
#pragma unroll
    for (int i = 0; i < 4; ++i) { swap32(z[i].x, z[i + 4].x); swap32(z[i].y, z[i + 4].y); }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (!(i & 2)) { swap16(z[i].x, z[i + 2].x); swap16(z[i].y, z[i + 2].y); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { swap_row<8>(z[i].x, z[i + 1].x); swap_row<8>(z[i].y, z[i + 1].y); }
  }
  dft8(z);
#pragma unroll
  for (int a = 1; a < 8; ++a) z[a] = cmulf(z[a], T->tw2[a][c]);
  if (VARIANT == 0) {
#pragma unroll
    for (int a = 0; a < 8; ++a) xb[a * 65 + lane] = z[a];
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = xb[c * 65 + hi * 8 + k];
  } else {
    const bool b1 = (lane & 2u) != 0, b0 = (lane & 1u) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { swap_row<4>(z[i].x, z[i + 4].x); swap_row<4>(z[i].y, z[i + 4].y); }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (!(i & 2)) { swap_quad<1>(z[i].x, z[i + 2].x, b1); swap_quad<1>(z[i].y, z[i + 2].y, b1); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { swap_quad<0>(z[i].x, z[i + 1].x, b0); swap_quad<0>(z[i].y, z[i + 1].y, b0); }
  }
  dft8(z);
}

#define WAVES 8
template <int VARIANT>
__global__ void __launch_bounds__(WAVES * 64, 4) fft_loop(const Tw* __restrict__ Tg, float2* __restrict__ io, int iters) {
  __shared__ Tw s_t;
  __shared__ float2 s_x[VARIANT == 0 ? WAVES : 1][576];
  {
    const uint4* src = (const uint4*)Tg;
    uint4* dst = (uint4*)&s_t;
    for (uint32_t i = threadIdx.x; i < sizeof(Tw) / 16; i += WAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  float2* xb = s_x[VARIANT == 0 ? wave : 0];
  float2* p = io + ((size_t)blockIdx.x * WAVES + wave) * 512 + lane;
  float2 z[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) z[t] = p[64 * t];
  uint32_t lane_v = lane;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(lane_v));  // as in the fused kernel: keeps the table loads inside the loop
    fft512<VARIANT>(z, xb, &s_t, lane_v);
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = f2(z[t].x * 0.0441941738f, z[t].y * 0.0441941738f);  // 1/sqrt(512): keep the magnitude
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) p[64 * t] = z[t];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;
  const int blocks = 512;  // 2 per CU
  const size_t n = (size_t)blocks * WAVES * 512;
  std::vector<float2> h(n), out[2];
  srand(1);
  for (auto& v : h) v = make_float2((rand() % 2001 - 1000) * 1e-3f, (rand() % 2001 - 1000) * 1e-3f);
  Tw tw;
  memset(&tw, 0, sizeof(tw));
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 8; ++k) {
      const double a = -2.0 * M_PI * (double)((l * k) & 511) / 512.0;
      tw.tw1[k][l] = make_float2((float)cos(a), (float)sin(a));
    }
  for (int a = 0; a < 8; ++a)
    for (int c = 0; c < 8; ++c) {
      const double w = -2.0 * M_PI * (double)((8 * c * a) & 511) / 512.0;
      tw.tw2[a][c] = make_float2((float)cos(w), (float)sin(w));
    }
  Tw* d_tw;
  float2* d_io;
  CHECK(hipMalloc((void**)&d_tw, sizeof(Tw)));
  CHECK(hipMemcpy(d_tw, &tw, sizeof(Tw), hipMemcpyHostToDevice));
  CHECK(hipMalloc((void**)&d_io, n * sizeof(float2)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int v = 0; v < 2; ++v) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipMemcpy(d_io, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
      CHECK(hipEventRecord(e0, 0));
      if (v == 0) fft_loop<0><<<blocks, WAVES * 64>>>(d_tw, d_io, iters);
      else fft_loop<1><<<blocks, WAVES * 64>>>(d_tw, d_io, iters);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = ms < best ? ms : best;
    }
    out[v].resize(n);
    CHECK(hipMemcpy(out[v].data(), d_io, n * sizeof(float2), hipMemcpyDeviceToHost));
    const double ffts = (double)blocks * WAVES * iters;
    printf("variant %d (%s): %.3f ms for %d FFT-512 per wave, %d waves: %.1f ns per wave-FFT, %.2f G FFT/s\n", v, v ? "registers" : "LDS image", best,
           iters, blocks * WAVES, best * 1e6 / iters, ffts / best / 1e6);
  }
  size_t diff = 0;
  for (size_t i = 0; i < n; ++i)
    if (memcmp(&out[0][i], &out[1][i], sizeof(float2))) ++diff;
  printf("bitwise differences between the variants: %zu of %zu\n", diff, n);
  return diff ? 2 : 0;
}
