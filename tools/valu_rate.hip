// tools/valu_rate.hip — microbenchmark: VALU issue rate on gfx950 for the instruction mix of the fused kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a[8];
  float2_ p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2_{a[i], a[i] + 1.f}; }
  const float c = 1.0001f, d = 0.5f;
  const float2_ pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);              // v_fma_f32
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pc, pd); // v_pk_fma_f32
      if (MODE == 2) a[i] = a[i] + c;                                // v_add_f32
      if (MODE == 3) p[i] = p[i] + pc;                               // v_pk_add_f32
      if (MODE == 4) p[i] = p[i] * pc;                               // v_pk_mul_f32
      if (MODE == 5) a[i] = __builtin_floorf(a[i] * c);              // v_mul + v_floor
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int ops_per_iter_per_lane) {
  float* out;
  hipMalloc(&out, 256 * 4096 * 4);
  const int iters = 4096;
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
    int blocks = 256 * wps;                // 256 CUs x (wps*4 waves = wps blocks of 256 threads)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 8 * (MODE == 5 ? 2 : 1);       // wave-instructions
    double per_simd_cyc = ms * 1e-3 * 2.4e9 / (winst / 1024.0);                  // cycles per wave-instr per SIMD @2.4GHz
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.1f Gflop-equiv/s\n", name, wps, ms, per_simd_cyc,
           winst * 64 * ops_per_iter_per_lane / (ms * 1e-3) / 1e9);
  }
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 2);
  run<1>("v_pk_fma_f32", 4);
  run<2>("v_add_f32", 1);
  run<3>("v_pk_add_f32", 2);
  run<4>("v_pk_mul_f32", 2);
  run<5>("mul+floor", 1);
  return 0;
}
