// tools/fetch_calib.hip — calibrates rocprofv3 FETCH_SIZE on gfx950 for the fused kernel's access shapes:
// streams a 512 MiB buffer once with 8-byte-per-lane loads (global_load_dwordx2, what the fused kernel issues) and once
// with 16-byte-per-lane loads. Run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename V>
__global__ void __launch_bounds__(256) stream_read(const V* __restrict__ in, float* __restrict__ out, size_t n) {
  float acc = 0.f;
  // each wave reads 8 consecutive 64-lane rows (like one packet's residue block), then jumps
  for (size_t base = (size_t)blockIdx.x * 256 + threadIdx.x; base < n; base += (size_t)gridDim.x * 256) {
    V v = in[base];
    acc += ((const float*)&v)[0];
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const size_t bytes = 512ull << 20;
  void* buf;
  float* out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    stream_read<float2><<<256 * 16, 256>>>((const float2*)buf, out, bytes / 8);
    stream_read<float4><<<256 * 16, 256>>>((const float4*)buf, out, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel launch\n", bytes);
  return 0;
}
