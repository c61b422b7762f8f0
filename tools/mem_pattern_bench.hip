// tools/mem_pattern_bench.hip — what can HBM deliver for the ACCESS PATTERN of the fused synthesis kernel, with no arithmetic?
// One wavefront per (stream, run, channel) walks R consecutive 4 KiB blocks of its own region (8 KiB apart: the other channel's
// block lies between), reading each block once (one block ahead) and writing one 4 KiB block per step into its own output plane,
// 16 waves per CU — exactly the fused kernel's geometry (64 streams x 32 runs x 2 channels = 4096 waves). Variants:
//   width   8 or 16 bytes per lane per instruction (global_load/store_dwordx2 vs dwordx4)
//   layout  "runs": the kernel's geometry;  "linear": the same bytes as one flat copy (wave w handles blocks w, w + W, ...)
//   pace    0: as fast as memory allows;  N > 0: N dependent FMAs per lane between load and store (a wave then issues its next
//           request only every ~N*4 cycles, as a computing wave does)
//   hipcc --offload-arch=gfx950 -O3 -o tools/mem_pattern_bench tools/mem_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t BLK = 1024;  // floats per block (one channel of one long packet)

template <int WIDTH, bool LINEAR, bool WRITE, int NT = 0, int ORDER = 0>
__global__ void __launch_bounds__(512, 4) walk(const float* __restrict__ in, float* __restrict__ out, uint32_t streams, uint32_t runs, uint32_t R,
                                                uint32_t ppk, uint64_t plane, int pace) {
  const uint32_t wave = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t nwaves = streams * runs * 2;
  if (wave >= nwaves) return;
  uint32_t c = wave & 1u, r = (wave >> 1) % runs, s = (wave >> 1) / runs;
  if (ORDER == 1) { c = wave & 1u; s = (wave >> 1) % streams; r = (wave >> 1) / streams; }        // (r, s, c): a workgroup = 4 streams, same run index
  else if (ORDER == 2) { r = wave % runs; s = (wave / runs) % streams; c = wave / (runs * streams); }  // (c, s, r)
  else if (ORDER == 3) { r = wave % runs; c = (wave / runs) & 1u; s = wave / (2u * runs); }          // (s, c, r): a workgroup = 8 consecutive runs of one channel
  constexpr int V = WIDTH / 4;            // floats per lane per instruction
  constexpr int NI = BLK / (64 * V);      // instructions per block
  typedef float vec __attribute__((ext_vector_type(V)));
  auto src_of = [&](uint32_t q) -> const vec* {
    const size_t blk = LINEAR ? (size_t)wave + (size_t)q * nwaves : ((size_t)s * ppk + (size_t)r * R + q) * 2 + c;
    return (const vec*)(in + blk * BLK) + lane;
  };
  auto dst_of = [&](uint32_t q) -> vec* {
    const size_t off = LINEAR ? ((size_t)wave + (size_t)q * nwaves) * BLK : ((size_t)s * 2 + c) * plane + ((size_t)r * R + q) * BLK;
    return (vec*)(out + off) + lane;
  };
  vec cur[NI], nxt[NI];
  {
    const vec* p = src_of(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) cur[i] = (NT & 1) ? __builtin_nontemporal_load(p + 64 * i) : p[64 * i];
  }
  for (uint32_t q = 0; q < R; ++q) {
    const vec* p = src_of(q + 1 < R ? q + 1 : q);
#pragma unroll
    for (int i = 0; i < NI; ++i) nxt[i] = (NT & 1) ? __builtin_nontemporal_load(p + 64 * i) : p[64 * i];
    float acc = cur[0][0];
    for (int k = 0; k < pace; ++k) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
    if (WRITE) {
      vec* d = dst_of(q);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        vec v = cur[i];
        v[0] += acc * 1e-30f;
        if (NT & 2) __builtin_nontemporal_store(v, d + 64 * i);
        else d[64 * i] = v;
      }
    } else if (acc == 1234.5f) {
      out[wave] = acc;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
  }
}


template <int POL>
__device__ __forceinline__ float2 pol_load(const float2* p) {
  float2 v;
  if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off nt sc1" : "=v"(v) : "v"(p) : "memory");
  else if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if (POL == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int POL>
__device__ __forceinline__ void pol_store(float2* p, float2 v) {
  if (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  else if (POL == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// the kernel's geometry, 8 bytes per lane, cache policy of loads (LP) and stores (SP) spelled out
template <int LP, int SP>
__global__ void __launch_bounds__(512, 4) walk_pol(const float* __restrict__ in, float* __restrict__ out, uint32_t streams, uint32_t runs, uint32_t R,
                                                    uint32_t ppk, uint64_t plane) {
  const uint32_t wave = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t nwaves = streams * runs * 2;
  if (wave >= nwaves) return;
  const uint32_t c = wave & 1u, r = (wave >> 1) % runs, s = (wave >> 1) / runs;
  auto src_of = [&](uint32_t q) { return (const float2*)(in + (((size_t)s * ppk + (size_t)r * R + q) * 2 + c) * BLK) + lane; };
  auto dst_of = [&](uint32_t q) { return (float2*)(out + ((size_t)s * 2 + c) * plane + ((size_t)r * R + q) * BLK) + lane; };
  float2 cur[8], nxt[8];
  {
    const float2* p = src_of(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = pol_load<LP>(p + 64 * i);
  }
  for (uint32_t q = 0; q < R; ++q) {
    const float2* p = src_of(q + 1 < R ? q + 1 : q);
#pragma unroll
    for (int i = 0; i < 8; ++i) nxt[i] = pol_load<LP>(p + 64 * i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    float2* d = dst_of(q);
#pragma unroll
    for (int i = 0; i < 8; ++i) pol_store<SP>(d + 64 * i, cur[i]);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
  }
}
template <int LP, int SP>
static void run_pol(const char* name, const float* in, float* out, uint32_t streams, uint32_t runs, uint32_t R, uint64_t plane) {
  const uint32_t ppk = runs * R, nwaves = streams * runs * 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    walk_pol<LP, SP><<<(nwaves + 7) / 8, 512>>>(in, out, streams, runs, R, ppk, plane);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)nwaves * R * BLK * 4 * 2;
  printf("%-44s: %.3f ms  %.2f TB/s (read + write)\n", name, best, bytes / best / 1e9);
}

// one wave per (stream, run), both channels, PK packets (PK x 8 KiB contiguous) per step, one step ahead; 8-byte accesses
template <int PK>
__global__ void __launch_bounds__(512) walk_wide(const float* __restrict__ in, float* __restrict__ out, uint32_t streams, uint32_t runs, uint32_t R,
                                                 uint32_t ppk, uint64_t plane) {
  const uint32_t wave = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (wave >= streams * runs) return;
  const uint32_t r = wave % runs, s = wave / runs;
  constexpr int NI = PK * 16;
  typedef float vec __attribute__((ext_vector_type(2)));
  vec cur[NI], nxt[NI];
  auto load = [&](uint32_t q, vec (&b)[NI]) {
    const vec* p = (const vec*)(in + (((size_t)s * ppk + (size_t)r * R + (q < R ? q : R - PK)) * 2) * BLK) + lane;
#pragma unroll
    for (int i = 0; i < NI; ++i) b[i] = p[64 * i];
  };
  load(0, cur);
  for (uint32_t q = 0; q < R; q += PK) {
    load(q + PK, nxt);
#pragma unroll
    for (int k = 0; k < PK; ++k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        vec* d = (vec*)(out + ((size_t)s * 2 + c) * plane + ((size_t)r * R + q + k) * BLK) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[64 * i] = cur[k * 16 + c * 8 + i];
      }
#pragma unroll
    for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
  }
}

template <int PK>
static void run_wide(const char* name, const float* in, float* out, uint32_t streams, uint32_t runs, uint32_t R, uint64_t plane) {
  const uint32_t ppk = runs * R, nwaves = streams * runs;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    walk_wide<PK><<<(nwaves + 7) / 8, 512>>>(in, out, streams, runs, R, ppk, plane);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)nwaves * R * BLK * 4 * 4;
  printf("%-44s: %.3f ms  %.2f TB/s (read + write)\n", name, best, bytes / best / 1e9);
}

// one wave per (stream, run): BOTH channels (8 KiB contiguous per step in, 2 x 4 KiB out), DEPTH steps ahead
template <int WIDTH, int DEPTH>
__global__ void __launch_bounds__(512) walk_pair(const float* __restrict__ in, float* __restrict__ out, uint32_t streams, uint32_t runs, uint32_t R,
                                                 uint32_t ppk, uint64_t plane, uint32_t wpb) {
  const uint32_t wave = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if ((threadIdx.x >> 6) >= wpb || wave >= streams * runs) return;
  const uint32_t r = wave % runs, s = wave / runs;
  constexpr int V = WIDTH / 4;
  constexpr int NI = 2 * BLK / (64 * V);
  typedef float vec __attribute__((ext_vector_type(V)));
  vec buf[DEPTH + 1][NI];
  auto load = [&](uint32_t q, vec (&b)[NI]) {
    const vec* p = (const vec*)(in + (((size_t)s * ppk + (size_t)r * R + (q < R ? q : R - 1)) * 2) * BLK) + lane;
#pragma unroll
    for (int i = 0; i < NI; ++i) b[i] = p[64 * i];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, buf[d]);
  for (uint32_t q = 0; q < R; q += DEPTH + 1) {
#pragma unroll
    for (int ph = 0; ph <= DEPTH; ++ph) {  // rotating buffers, fully unrolled so that every index is static
      if (q + ph >= R) break;
      load(q + ph + DEPTH, buf[(ph + DEPTH) % (DEPTH + 1)]);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        vec* d = (vec*)(out + ((size_t)s * 2 + c) * plane + ((size_t)r * R + q + ph) * BLK) + lane;
#pragma unroll
        for (int i = 0; i < NI / 2; ++i) d[64 * i] = buf[ph][c * (NI / 2) + i];
      }
    }
  }
}

template <int WIDTH, int DEPTH>
static void run_pair(const char* name, const float* in, float* out, uint32_t streams, uint32_t runs, uint32_t R, uint64_t plane, uint32_t wpb) {
  const uint32_t ppk = runs * R, nwaves = streams * runs;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    walk_pair<WIDTH, DEPTH><<<(nwaves + wpb - 1) / wpb, 512>>>(in, out, streams, runs, R, ppk, plane, wpb);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)nwaves * R * BLK * 4 * 4;
  printf("%-44s: %.3f ms  %.2f TB/s (read + write)\n", name, best, bytes / best / 1e9);
}

template <int WIDTH, bool LINEAR, bool WRITE, int NT = 0, int ORDER = 0>
static void run(const char* name, const float* in, float* out, uint32_t streams, uint32_t runs, uint32_t R, uint64_t plane, int pace) {
  const uint32_t ppk = runs * R, nwaves = streams * runs * 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    walk<WIDTH, LINEAR, WRITE, NT, ORDER><<<(nwaves + 7) / 8, 512>>>(in, out, streams, runs, R, ppk, plane, pace);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)nwaves * R * BLK * 4 * (WRITE ? 2 : 1);
  printf("%-34s pace %4d: %.3f ms  %.2f TB/s (%s)\n", name, pace, best, bytes / best / 1e9, WRITE ? "read + write" : "read only");
}

int main(int argc, char** argv) {
  const uint32_t streams = 64, runs = 32, R = 32;
  const uint64_t plane = (uint64_t)runs * R * BLK + 64;
  const size_t n_in = (size_t)streams * runs * R * 2 * BLK, n_out = (size_t)streams * 2 * plane;
  float *in, *out;
  CHECK(hipMalloc((void**)&in, n_in * 4));
  CHECK(hipMalloc((void**)&out, n_out * 4));
  CHECK(hipMemset(in, 0, n_in * 4));
  CHECK(hipMemset(out, 0, n_out * 4));
  if (argc > 1) {
    // counter mode (round 3): `mem_pattern_bench <run length>` launches ONLY the streaming pattern at that run length (same bytes, 32 / R
    // times the waves) so that a `rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum ...` pass sees one kernel shape; "flat" = the linear sweep
    if (!strcmp(argv[1], "flat")) {
      run<16, true, true, 3>("linear, 16 B/lane, nt both", in, out, streams, runs, R, plane, 0);
      return 0;
    }
    const uint32_t r = (uint32_t)atoi(argv[1]);
    if (r < 1 || r > 32 || (32 % r)) return 2;
    run<8, false, true, 3>("runs of argv[1], nt both", in, out, streams, runs * (32 / r), r, plane, 0);
    return 0;
  }
  // cache policies spelled out (loads / stores): 0 default, 1 nt, 2 nt sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc0
  run_pol<0, 0>("policy: default / default", in, out, streams, runs, R, plane);
  run_pol<1, 1>("policy: nt / nt", in, out, streams, runs, R, plane);
  run_pol<2, 2>("policy: nt sc1 / nt sc1", in, out, streams, runs, R, plane);
  run_pol<3, 3>("policy: sc0 sc1 nt / sc0 sc1 nt", in, out, streams, runs, R, plane);
  run_pol<1, 2>("policy: nt / nt sc1", in, out, streams, runs, R, plane);
  run_pol<1, 3>("policy: nt / sc0 sc1 nt", in, out, streams, runs, R, plane);
  run_pol<2, 1>("policy: nt sc1 / nt", in, out, streams, runs, R, plane);
  run_pol<3, 1>("policy: sc0 sc1 nt / nt", in, out, streams, runs, R, plane);
  run_pol<4, 4>("policy: sc1 / sc1", in, out, streams, runs, R, plane);
  run_pol<5, 5>("policy: sc0 / sc0", in, out, streams, runs, R, plane);
  run_pol<0, 1>("policy: default / nt", in, out, streams, runs, R, plane);
  run_pol<1, 0>("policy: nt / default", in, out, streams, runs, R, plane);
  // round 3: which (stream, run, channel) a wave gets — the kernel's order is (s, r, c): a workgroup = 4 consecutive runs x 2 channels
  run<8, false, true, 3, 0>("order (s, r, c) [the kernel's], nt both", in, out, streams, runs, R, plane, 0);
  run<8, false, true, 3, 1>("order (r, s, c), nt both", in, out, streams, runs, R, plane, 0);
  run<8, false, true, 3, 2>("order (c, s, r), nt both", in, out, streams, runs, R, plane, 0);
  run<8, false, true, 3, 3>("order (s, c, r), nt both", in, out, streams, runs, R, plane, 0);
  // other run lengths (same bytes): more, shorter runs = the resident waves cover a smaller address window at any time
  run<8, false, true, 3>("runs of 16 (x2 waves), nt both", in, out, streams, 2 * runs, R / 2, plane, 0);
  run<8, false, true, 3>("runs of 8 (x4 waves), nt both", in, out, streams, 4 * runs, R / 4, plane, 0);
  run<8, false, true, 3>("runs of 4 (x8 waves), nt both", in, out, streams, 8 * runs, R / 8, plane, 0);
  run<8, false, true, 3>("runs of 2 (x16 waves), nt both", in, out, streams, 16 * runs, R / 16, plane, 0);
  run<8, false, true, 3>("runs of 1 (x32 waves), nt both", in, out, streams, 32 * runs, R / 32, plane, 0);
  for (int pace : {0}) {
    run<8, false, true>("runs,   8 B/lane", in, out, streams, runs, R, plane, pace);
    run<8, false, true, 1>("runs,   8 B/lane, nt loads", in, out, streams, runs, R, plane, pace);
    run<8, false, true, 2>("runs,   8 B/lane, nt stores", in, out, streams, runs, R, plane, pace);
    run<8, false, true, 3>("runs,   8 B/lane, nt both", in, out, streams, runs, R, plane, pace);
    run<16, false, true, 3>("runs,   16 B/lane, nt both", in, out, streams, runs, R, plane, pace);
    run<16, true, true, 3>("linear, 16 B/lane, nt both", in, out, streams, runs, R, plane, pace);
    run<16, false, true>("runs,   16 B/lane", in, out, streams, runs, R, plane, pace);
    run<8, true, true>("linear, 8 B/lane", in, out, streams, runs, R, plane, pace);
    run<16, true, true>("linear, 16 B/lane", in, out, streams, runs, R, plane, pace);
  }
  // both channels per wave; 2048 waves x 32 steps (8 waves/CU: 4 waves per 512-thread block slot... wpb = waves used per block)
  run_pair<8, 1>("pair 8 B, depth 1, 2048 waves (8/CU)", in, out, streams, runs, R, plane, 8);
  run_pair<16, 1>("pair 16 B, depth 1, 2048 waves (8/CU)", in, out, streams, runs, R, plane, 8);
  run_pair<8, 2>("pair 8 B, depth 2, 2048 waves (8/CU)", in, out, streams, runs, R, plane, 8);
  run_pair<16, 2>("pair 16 B, depth 2, 2048 waves (8/CU)", in, out, streams, runs, R, plane, 8);
  // 4096 waves x 16 steps (16 waves/CU)
  run_pair<8, 1>("pair 8 B, depth 1, 4096 waves (16/CU)", in, out, streams, 2 * runs, R / 2, plane, 8);
  run_pair<16, 1>("pair 16 B, depth 1, 4096 waves (16/CU)", in, out, streams, 2 * runs, R / 2, plane, 8);
  run_pair<8, 2>("pair 8 B, depth 2, 4096 waves (16/CU)", in, out, streams, 2 * runs, R / 2, plane, 8);
  run_wide<1>("wide: 8 KiB per step, 2048 waves", in, out, streams, runs, R, plane);
  run_wide<2>("wide: 16 KiB per step, 2048 waves", in, out, streams, runs, R, plane);
  run_wide<4>("wide: 32 KiB per step, 2048 waves", in, out, streams, runs, R, plane);
  run_wide<2>("wide: 16 KiB per step, 4096 waves", in, out, streams, 2 * runs, R / 2, plane);
  run_wide<1>("wide: 8 KiB per step, 1024 waves", in, out, streams, runs / 2, R * 2, plane);
  run_wide<4>("wide: 32 KiB per step, 1024 waves", in, out, streams, runs / 2, R * 2, plane);
  // 3072 waves (12/CU)
  run_pair<8, 1>("pair 8 B, depth 1, 3072 waves (12/CU)", in, out, streams, 48, 21, plane, 8);
  return 0;
}
