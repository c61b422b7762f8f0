// vsyn_pcm.h — PCM post-stage: planar f32 -> interleaved int16 / f32, what the consumer of gotPcmData does next
// (reference hand-off point: src/ParseOggVorbis.hpp:1045-1054; conversion rule: tests/libvorbis-standalone/vorbis_vorbisfile.c:2026-2029).
#pragma once
#include "vsyn_device.h"

// thread = 4 consecutive frames of one segment, all channels: 16-byte loads per channel plane, one 16-byte (stereo s16) or
// wider interleaved store; pure streaming, HBM-bound.
__device__ __forceinline__ int pcm_s16(float x) {
  // ov_read (vorbis_vorbisfile.c:2026-2028): vorbis_ftoi(x * 32768.f) — the f32 product rounded to the nearest-even integer
  // (cvtsd2si of the promoted product, os.h:156-158) —, then clamped
  const float r = __builtin_rintf(x * 32768.f);
  return (int)fminf(fmaxf(r, -32768.f), 32767.f);
}

template <int FORMAT>
__global__ void __launch_bounds__(256) vsyn_pcm_interleave_kernel(const uint8_t* __restrict__ cb, const SegInfo* __restrict__ sinfo, uint32_t S,
                                                                  const float* __restrict__ pcm, uint64_t plane_stride, void* __restrict__ outv,
                                                                  uint64_t out_stride, uint32_t* __restrict__ frames_out) {
  const uint32_t g = blockIdx.y;
  if (g >= S) return;
  const uint32_t C = hdr_of(cb)->channels;
  const uint32_t frames = (uint32_t)min((uint64_t)sinfo[g].total_emit, min(plane_stride, out_stride));
  if (blockIdx.x == 0 && threadIdx.x == 0 && frames_out) frames_out[g] = frames;
  const uint32_t f0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
  if (f0 >= frames) return;
  const float* src = pcm + (size_t)g * C * plane_stride + f0;
  const uint32_t n = min(4u, frames - f0);
  if (C == 2 && n == 4 && ((plane_stride & 3u) == 0) && (((uintptr_t)pcm & 15u) == 0)) {
    const float4 l = *(const float4*)src, r = *(const float4*)(src + plane_stride);
    if (FORMAT == VSYN_PCM_S16) {
      int16_t* o = (int16_t*)outv + ((size_t)g * out_stride + f0) * 2u;
      const uint32_t w0 = (uint32_t)(uint16_t)pcm_s16(l.x) | ((uint32_t)(uint16_t)pcm_s16(r.x) << 16);
      const uint32_t w1 = (uint32_t)(uint16_t)pcm_s16(l.y) | ((uint32_t)(uint16_t)pcm_s16(r.y) << 16);
      const uint32_t w2 = (uint32_t)(uint16_t)pcm_s16(l.z) | ((uint32_t)(uint16_t)pcm_s16(r.z) << 16);
      const uint32_t w3 = (uint32_t)(uint16_t)pcm_s16(l.w) | ((uint32_t)(uint16_t)pcm_s16(r.w) << 16);
      if (((uintptr_t)o & 15u) == 0) *(uint4*)o = make_uint4(w0, w1, w2, w3);
      else { ((uint32_t*)o)[0] = w0; ((uint32_t*)o)[1] = w1; ((uint32_t*)o)[2] = w2; ((uint32_t*)o)[3] = w3; }
    } else {
      float* o = (float*)outv + ((size_t)g * out_stride + f0) * 2u;
      if (((uintptr_t)o & 15u) == 0) {
        *(float4*)o = make_float4(l.x, r.x, l.y, r.y);
        *(float4*)(o + 4) = make_float4(l.z, r.z, l.w, r.w);
      } else {
        o[0] = l.x; o[1] = r.x; o[2] = l.y; o[3] = r.y; o[4] = l.z; o[5] = r.z; o[6] = l.w; o[7] = r.w;
      }
    }
    return;
  }
  for (uint32_t k = 0; k < n; ++k)
    for (uint32_t c = 0; c < C; ++c) {
      const float x = src[(size_t)c * plane_stride + k];
      const size_t idx = ((size_t)g * out_stride + f0 + k) * C + c;
      if (FORMAT == VSYN_PCM_S16) ((int16_t*)outv)[idx] = (int16_t)pcm_s16(x);
      else ((float*)outv)[idx] = x;
    }
}


// Per-(segment, channel) digest: sum of |x| over the segment's emitted frames, in double, in a fixed order (thread t adds
// samples t, t + 256, ...; then a fixed LDS tree): the same PCM always gives the same bits, whatever else runs on the GPU.
__global__ void __launch_bounds__(256) vsyn_pcm_abs_sum_kernel(const uint8_t* __restrict__ cb, const SegInfo* __restrict__ sinfo, uint32_t S,
                                                               const float* __restrict__ pcm, uint64_t plane_stride, double* __restrict__ out) {
  __shared__ double s_part[256];
  const uint32_t C = hdr_of(cb)->channels, u = blockIdx.x, g = u / C, t = threadIdx.x;
  if (g >= S) return;
  const uint32_t frames = (uint32_t)min((uint64_t)sinfo[g].total_emit, plane_stride);
  const float* x = pcm + (size_t)u * plane_stride;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;  // four chains per thread: the adds of one chain are dependent
  uint32_t i = t;
  for (; i + 768u < frames; i += 1024u) {
    a0 += (double)fabsf(x[i]);
    a1 += (double)fabsf(x[i + 256u]);
    a2 += (double)fabsf(x[i + 512u]);
    a3 += (double)fabsf(x[i + 768u]);
  }
  for (; i < frames; i += 256u) a0 += (double)fabsf(x[i]);
  s_part[t] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (uint32_t d = 128; d; d >>= 1) {
    if (t < d) s_part[t] += s_part[t + d];
    __syncthreads();
  }
  if (t == 0) out[u] = s_part[0];
}
