// vsyn_device.h — device-side data layout shared by the host layer and the kernels (gfx950 only).
//
// HBM layout
//   constant block  one allocation per handle: ConstHeader, then FloorConst[], MapConst[], the inverse-dB
//                   table, and per blocksize: pre/post twiddles (n/4 float2 each), FFT twiddles, windows.
//                   < 200 KB for 256/2048; read-only, L2/MALL resident; this is the block a multi-GPU job
//                   broadcasts once.
//   per batch       packets [P] (16 B), segments [S] (24 B), ys u16 [P][C][ys_stride], residue f32 packed,
//                   pcm f32 [S][C][plane_stride]  — caller owned.
//   workspace       PktInfo [P] (32 B), SegInfo [S], unwrapped floor posts u16 [P][C][ys_stride];
//                   staged path only: after_envelope f32 (residue-shaped) and pcm_after_mdct f32 (2x).
//   stream state    StreamState [max_streams][2] (tagged records, see below) + overlap carry f32 [2][max_streams][C][blocksize1/2]
//                   (double buffered: a submit reads one half and writes the other).
#pragma once
#include <stdint.h>

#include "../../include/vorbis_synth_hip.h"

#define VSYN_SCHED_GROUPS 64
#define VSYN_MAX_TABLES 64 /* Vorbis I: floor/mapping/mode counts are 6-bit fields (hpp:923,941,950) */

struct FloorConst {               // one floor-1 configuration (VorbisFloor1, hpp:416-471) + precomputed neighbours
  uint32_t mult, posts, range;    // range = {256,128,86,64}[mult-1]  (hpp:486-492)
  uint16_t xs[VSYN_MAX_POSTS + 1];         // header order
  uint16_t xs_sorted[VSYN_MAX_POSTS + 1];  // ascending (hpp:459-469)
  uint8_t sorted_idx[VSYN_MAX_POSTS + 3];  // sorted position -> header index
  uint8_t lo[VSYN_MAX_POSTS + 3];          // low_neighbor(xs,i)  (Utils.hpp:60-87), header indices
  uint8_t hi[VSYN_MAX_POSTS + 3];          // high_neighbor(xs,i) (Utils.hpp:91-118)
  struct PostK {                           // per-post constants of step 1, one 16-byte load per post
    uint16_t lo, hi;                       // neighbour header indices
    uint16_t dxi, adx;                     // xs[i] - xs[lo], xs[hi] - xs[lo]
    float inv_adx;                         // 1 / adx
    uint32_t idx;                          // sched[]: header index of the post; pk[] of a floor of <= 32 posts: (1 << lo) | (1 << hi) | (1 << i)
  } pk[VSYN_MAX_POSTS + 1];
  // Step 1 as a schedule of GROUPS of up to four mutually independent posts (same depth in the neighbour tree: a post depends on
  // its two neighbours only): a lane that unwraps a row works on four posts at once instead of one (vsyn_prep.h). Unused entries of
  // a group repeat its last post. At most posts - 2 <= 63 groups.
  uint32_t ngroups, pad_[3];
  PostK sched[VSYN_SCHED_GROUPS][4];
};

struct MapConst {                 // VorbisMapping (hpp:765-814), synthesis-relevant part
  uint32_t ncoup;
  uint8_t chfloor[VSYN_MAX_CHANNELS];
  uint16_t coup[2 * 256];         // (magnitude, angle) pairs, header order
};

struct ConstHeader {
  uint32_t channels, bs[2], lg[2], ys_stride, num_floors, num_mappings, num_modes, max_streams;
  uint32_t off_floor, off_map, off_invdb;          // byte offsets from the block base
  uint32_t off_pre[2], off_post[2], off_fft[2], off_win[2];
  uint8_t mode_blockflag[VSYN_MAX_TABLES], mode_mapping[VSYN_MAX_TABLES];
  uint32_t total_bytes;
};

struct PktInfo {                  // written by the layout kernel, 32 bytes
  uint64_t res_off;               // float index of the packet's residue block
  uint32_t out_pos;               // first emitted sample, relative to the segment's plane
  uint32_t emit;                  // forwardReadyPcm num_frames (hpp:1019-1059), 0 on error
  uint32_t used;                  // floor_output_used after nonzero propagate (hpp:1174-1180)
  uint32_t own;                   // channels whose own floor curve was decoded
  uint16_t n;                     // blocksize
  uint8_t lng, widx;              // block flag; window table index prev + 2*next (hpp:874-886)
  uint8_t mapping, bad;
  uint16_t pad;
};

struct SegInfo {
  uint32_t has_carry;             // overlap carry-in valid (previous submit left a block for this stream)
  uint32_t carry_n;               // blocksize of that block
  uint32_t parity_in;             // which carry half to read; the other half is written
  uint32_t total_emit;
};

struct StreamState {              // VorbisStreamDecodeState (hpp:975-1115) reduced to what crosses a batch boundary
  uint64_t abs_total_pos;
  uint32_t has_prev, prev_n, parity;
  uint32_t tag;                   // submit number that wrote the record (0: never written)
};
// Two records per stream slot. A submit READS the newer record written by an earlier submit and WRITES the other one, tagged with its
// own number: every wave of a submit — whenever it starts, the segment's last wave may long have finished — finds the state the
// stream had BEFORE this submit (a record tagged with the reader's own submit number is not a candidate, and the record being
// overwritten is by construction the older one: even a torn read of it loses the comparison). Submit numbers start at 1 and skip 0.

struct DevStatus {
  uint32_t flags, first_bad_packet;
};

#ifdef __HIPCC__
// Streaming accesses: the residue rows are read once and the PCM is written once, so both carry the non-temporal hint (do not keep
// the line in L2 / the memory-side cache). Measured on the steady synthesis kernel: stores alone -0.8 %, loads and stores -3.7 %
// (0.2462 -> 0.2372 ms); on the bare access pattern (tools/mem_pattern_bench.hip) 5.06 -> 5.18 TB/s.
#define VSYN_CPOL_NT 2  /* cache-policy immediate of the buffer / global load builtins: non-temporal */
typedef float vsyn_nt_f2 __attribute__((ext_vector_type(2)));
typedef float vsyn_nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 stream_load2(const float2* p) {
  const vsyn_nt_f2 v = __builtin_nontemporal_load((const vsyn_nt_f2*)p);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void stream_store2(float* p, float a, float b) { __builtin_nontemporal_store(vsyn_nt_f2{a, b}, (vsyn_nt_f2*)p); }
__device__ __forceinline__ void stream_store4(float* p, float a, float b, float c, float d) {
  __builtin_nontemporal_store(vsyn_nt_f4{a, b, c, d}, (vsyn_nt_f4*)p);
}
#endif
#ifdef __HIPCC__
__device__ __forceinline__ const ConstHeader* hdr_of(const uint8_t* cb) { return (const ConstHeader*)cb; }
__device__ __forceinline__ const FloorConst* floor_of(const uint8_t* cb, uint32_t f) {
  return (const FloorConst*)(cb + hdr_of(cb)->off_floor) + f;
}
__device__ __forceinline__ const MapConst* map_of(const uint8_t* cb, uint32_t m) {
  return (const MapConst*)(cb + hdr_of(cb)->off_map) + m;
}
__device__ __forceinline__ const float* invdb_of(const uint8_t* cb) { return (const float*)(cb + hdr_of(cb)->off_invdb); }
__device__ __forceinline__ const float2* pre_of(const uint8_t* cb, int b) { return (const float2*)(cb + hdr_of(cb)->off_pre[b]); }
__device__ __forceinline__ const float2* post_of(const uint8_t* cb, int b) { return (const float2*)(cb + hdr_of(cb)->off_post[b]); }
__device__ __forceinline__ const float2* fft_of(const uint8_t* cb, int b) { return (const float2*)(cb + hdr_of(cb)->off_fft[b]); }
__device__ __forceinline__ const float* win_of(const uint8_t* cb, int b, int widx) {
  return (const float*)(cb + hdr_of(cb)->off_win[b]) + (size_t)widx * hdr_of(cb)->bs[b];
}

__device__ __forceinline__ void raise_status(DevStatus* st, uint32_t flag, uint32_t pkt) {
  atomicOr(&st->flags, flag);
  atomicMin(&st->first_bad_packet, pkt);
}

// the state a stream slot had before submit `epoch`; *slot = the record it was read from (state_write takes the other one)
__device__ __forceinline__ StreamState state_read(const StreamState* __restrict__ st, uint32_t stream, uint32_t epoch, uint32_t* slot) {
  const StreamState a = st[2u * stream], b = st[2u * stream + 1u];
  const uint32_t da = (a.tag == 0u || a.tag == epoch) ? 0xFFFFFFFFu : epoch - a.tag;
  const uint32_t db = (b.tag == 0u || b.tag == epoch) ? 0xFFFFFFFFu : epoch - b.tag;
  if (da <= db) {
    *slot = 0u;
    if (da == 0xFFFFFFFFu) return StreamState{0, 0, 0, 0, 0};
    return a;
  }
  *slot = 1u;
  return b;
}
__device__ __forceinline__ void state_write(StreamState* __restrict__ st, uint32_t stream, uint32_t read_slot, StreamState ns, uint32_t epoch) {
  ns.tag = epoch;
  st[2u * stream + (read_slot ^ 1u)] = ns;
}
#endif
