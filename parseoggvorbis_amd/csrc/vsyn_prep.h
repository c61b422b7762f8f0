// vsyn_prep.h — vsyn_prep_kernel: the whole preparation of a batch in ONE kernel without any inter-workgroup dependency — what
// vsyn_layout_kernel + vsyn_floor_unwrap_kernel (vsyn_staged.h) do as two kernels chained by a scan.
//
// Why (round 3): the two pre-kernels are ~35 us of dependent latency when they run in front of the synthesis kernel; overlapped with
// the previous submit's synthesis kernel on a second queue they cost an event record and a cross-queue wait per submit (12-19 us
// between two synthesis kernels; tools/stream_sync_bench.hip: no cheaper ordering primitive, hipExtAnyOrderLaunch is ignored on this
// part) and their waves compete with an exact-fit synthesis grid for wave slots (+6 us on the synthesis kernel). This kernel is short
// enough (~17 us for 65 536 stereo packets) to simply run IN FRONT of the synthesis kernel on the caller's stream: a submit is two
// launches on one queue, no second queue, no events. Who gets it (vorbis_synth_hip.hip, submit_device_impl): every submit whose runs all
// go to fused kernels, whose segments have at most PREP_MAX_SEG_PACKETS packets and which carries no residue-VQ stage — with or without
// VSYN_SUBMIT_INPUTS_READY (profiles/r03_experiments/batch_preparation_ab.txt: 1-4 % faster per step than the hidden pre-kernels).
//
// Two kinds of workgroup per (segment, chunk of whole runs, about one (packet, channel) row per thread), running side by side:
//   LAYOUT (threads <-> packets)
//   1. the scan's running values in front of the chunk — absolute position (granule-aware, hpp:1028-1044), residue offset, block size —
//      by a reduction over the segment's EARLIER descriptors (each thread a contiguous piece, one block scan): every workgroup does
//      that for itself instead of waiting for a predecessor, which is what makes the kernel dependency-free (a segment has at most
//      PREP_MAX_SEG_PACKETS packets here; longer segments keep the layout kernel's chunked scan);
//   2. PktInfo of the chunk's packets by a block scan (pkt_step_core, shared with the layout kernel), emit_len, the class of each run
//      from a bitmap of its packets' block flags;
//   3. the thread of a segment's last packet leaves the stream state for the next submit (tagged records, vsyn_device.h) and the
//      segment's SegInfo.
//   FLOOR (threads <-> (packet, channel) rows)
//   4. floor-1 step 1 (hpp:521-559, prep_unwrap_rows): which floor, and whether the channel carries a curve, follow from the packet's
//      own descriptor — nothing the scan produces is needed; floors of up to 32 posts as a branch-free chain over a register array,
//      longer ones with the row's posts in LDS, four independent posts at a time.
#pragma once
#include <hip/hip_runtime.h>

#include "vsyn_device.h"
#include "vsyn_staged.h"

#define PREP_MAX_SEG_PACKETS 4096u
#define PREP_WAVES 4  // waves per workgroup
#define PREP_THREADS (PREP_WAVES * 64)

struct PrepCtx {  // launch arguments (all wave-uniform)
  const uint8_t* cb;
  const vsyn_packet* packets;
  const vsyn_segment* segs;
  const uint16_t* ys;
  uint16_t* fy;
  PktInfo* info;
  SegInfo* sinfo;
  StreamState* state;
  DevStatus* status;
  uint32_t* emit_len;
  uint8_t* run_cls;
  uint64_t plane_stride, long_modes;
  uint32_t S, R, runs_per_seg, fused_ok, P, epoch;
  uint32_t chunk_runs, chunks_per_seg;  // a workgroup takes chunk_runs whole runs of one segment
};

// -DPREP_STAMPS (diagnostic builds): cycles per phase of the preparation, summed per wave, printed by vsyn_destroy
#ifdef PREP_STAMPS
#define PREP_NSTAMPS 8
__device__ unsigned long long g_prep_stamps[8192][PREP_NSTAMPS];
#define PSTAMP(i)                                                \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    const unsigned long long t_ = __builtin_readcyclecounter();  \
    pst_acc[i] += t_ - pst_last;                                 \
    pst_last = t_;                                               \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(3))) uint16_t prep_lds_u16;
typedef __attribute__((address_space(3))) uint32_t prep_lds_u32;

// inclusive wave scan of (AbsScan, residue floats): lane l gets the combination of lanes 0..l
__device__ __forceinline__ void prep_wave_scan(AbsScan& inc, uint64_t& rinc, const uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    AbsScan o;
    o.val = __shfl_up(inc.val, d);
    o.set = __shfl_up(inc.set, d);
    const uint64_t ro = __shfl_up(rinc, d);
    if ((int)lane >= d) {
      inc = abs_combine(o, inc);
      rinc += ro;
    }
  }
}
__device__ __forceinline__ uint64_t prep_readlane64(uint64_t v, uint32_t l) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l) << 32);
}

// Floor-1 step 1 (hpp:521-559) of the rows held by the lanes with `act` set: row = (packet p, channel c), floor number fl_id (any mix
// of floors: rows of one floor are processed together so that the schedule comes through the scalar unit). Same arithmetic and same
// output rows as vsyn_floor_unwrap_kernel.
// Floors of more than 32 posts: the row's posts live in LDS, rowbuf[post][thread] (32-bit amplitudes, the thread's own column:
// conflict-free, no synchronisation), and are worked on in GROUPS of four mutually independent posts (FloorConst::sched): twelve LDS
// reads in flight together, four chains of arithmetic side by side, four writes. Evaluation order differs from the header's, the
// values do not: a post depends on its two neighbours only, and those sit in earlier groups. Shorter floors: a register array.
__device__ __forceinline__ void prep_unwrap_rows(const PrepCtx& A, const bool act, const uint32_t fl_id, const uint32_t p, const size_t gid,
                                                 const uint32_t stride, prep_lds_u32* rowbuf) {
  const uint8_t* __restrict__ cb = A.cb;
  const FloorConst* const floors = (const FloorConst*)(cb + hdr_of(cb)->off_floor);
  typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
  typedef const __attribute__((address_space(4))) u32x16* const_grp;
  typedef const __attribute__((address_space(4))) uint32_t* kptr;
  const uint32_t NT = blockDim.x;
  prep_lds_u32* const col = rowbuf + threadIdx.x;  // post i of this thread's row at col[i * NT]
  uint64_t todo = __ballot(act);
  while (todo) {
    const uint32_t f = __builtin_amdgcn_readlane(fl_id, (uint32_t)__builtin_ctzll(todo));
    const bool mine = act && fl_id == f;
    todo &= ~__ballot(mine);
    const FloorConst* fc = floors + f;
    const uint32_t posts = *(kptr)(uintptr_t)&fc->posts, range = *(kptr)(uintptr_t)&fc->range, mult = *(kptr)(uintptr_t)&fc->mult;
    const uint32_t ngroups = *(kptr)(uintptr_t)&fc->ngroups;
    if (!mine) continue;  // (divergent from here on: the lanes of this floor)
    if (posts <= 32u) {
      // Up to 32 posts (every floor libvorbis writes for the common modes): the row in a register array indexed by the wave-uniform
      // neighbour numbers, one post at a time in header order, the per-post constants by scalar loads of four posts, one load ahead.
      // Measured on config 3 (29 posts, two waves per SIMD): 620 cycles per post with round 3's first form of the chain (~70 instructions
      // per post, a wave vote and two divergent branches in it), the kernel 20.7 us; 19.3 us with the form below. The grouped forms
      // further down / with this array cost 1.7x / 2x per post (an access to the array is an s_set_gpr_idx mode switch; four chains
      // side by side do not make up for it).
      uint32_t fr[32];
      const uint2* in8 = (const uint2*)(A.ys + gid * stride);
      uint2 win[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        win[j] = make_uint2(0u, 0u);
        if (j * 4 < posts) win[j] = in8[j];
      }
      uint32_t flags = 3;
      bool bad = false;
      // The chain runs on the float form of the prediction's division (exact while |dy| * dx < 2^21: always, for amplitudes in range) and
      // only NOTES a larger product; a row that saw one (absurd coded values) is redone with the integer division afterwards. No
      // wave-level vote and no divergent branch inside the chain: ~50 instead of ~70 instructions per post, and the chain is what this
      // kernel's time is.
      typedef const __attribute__((address_space(4))) u32x16* const_pk4;
      for (int attempt = 0; attempt < 2; ++attempt) {
        const bool exact = attempt == 1;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          fr[4 * j + 0] = win[j].x & 0xFFFFu;
          fr[4 * j + 1] = win[j].x >> 16;
          fr[4 * j + 2] = win[j].y & 0xFFFFu;
          fr[4 * j + 3] = win[j].y >> 16;
        }
        flags = 3;
        bad = false;
        uint32_t big_acc = 0;
        auto step = [&](const uint32_t i, const uint32_t kx, const uint32_t ky, const uint32_t kz, const uint32_t touched) {
          const uint32_t lo = kx & 0xFFFFu, hi = kx >> 16;
          const uint32_t val = fr[i], ylo = fr[lo], yhi = fr[hi];
          const uint32_t dxi = ky & 0xFFFFu, adx = ky >> 16;
          const bool up = yhi >= ylo;
          const uint32_t ady = up ? yhi - ylo : ylo - yhi;
          const uint32_t prod = ady * dxi;
          uint32_t off;
          if (exact) {
            off = prod / adx;
          } else {
            off = (uint32_t)(((float)prod + 0.5f) * __uint_as_float(kz));  // == prod / adx while prod < 2^21 (vsyn_staged.h, predict_post)
            big_acc |= prod >> 21;
          }
          const uint32_t predicted = up ? ylo + off : ylo - off;
          const bool ok = predicted <= range;  // hpp:536
          const uint32_t pr = ok ? predicted : 0u;
          // hpp:540-556 without branches: m = the smaller room; beyond 2 m the value counts linearly from the nearer edge (d resp. -d - 1),
          // below it it is the zig-zag code of the offset (even: + val / 2, odd: - (val + 1) / 2 = ~(val >> 1))
          const uint32_t high_room = range - pr;
          const uint32_t m = min(high_room, pr);
          const uint32_t d = val - m;
          const uint32_t dbig = high_room > pr ? d : ~d;
          const uint32_t dsmall = (val >> 1) ^ (0u - (val & 1u));
          const uint32_t delta = val >= 2u * m ? dbig : dsmall;
          const uint32_t fn = val == 0 ? pr : pr + delta;
          flags |= val != 0 ? touched : 0u;  // (1 << lo) | (1 << hi) | (1 << i), from the table
          bad = bad || !ok;
          fr[i] = bad ? 0u : fn;  // after the first out-of-range prediction the row is dropped; keep the chain tame
        };
        u32x16 kn = *(const_pk4)(uintptr_t)&fc->pk[2];
        for (uint32_t i = 2; i < posts; i += 4) {
          const u32x16 kq = kn;
          kn = *(const_pk4)(uintptr_t)&fc->pk[i + 4];  // (pk[] has 66 entries)
          step(i, kq[0], kq[1], kq[2], kq[3]);
          if (i + 1 < posts) step(i + 1, kq[4], kq[5], kq[6], kq[7]);
          if (i + 2 < posts) step(i + 2, kq[8], kq[9], kq[10], kq[11]);
          if (i + 3 < posts) step(i + 3, kq[12], kq[13], kq[14], kq[15]);
        }
        if (exact || !__any(big_acc != 0u)) break;
        if (big_acc == 0u) break;  // (only the rows that saw a large product are redone)
      }
      uint2* out8 = (uint2*)(A.fy + gid * stride);
      if (bad) raise_status(A.status, VSYN_ST_FLOOR_RANGE, p);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        if (j * 4 >= posts) break;
        uint32_t w[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t i = 4 * j + e;
          uint32_t v = fr[i] * mult;  // hpp:573,578
          if (v > 0x7FFFu || fr[i] > 0x7FFFu) v = 0x7FFFu;
          w[e] = i < posts ? (bad ? 0x8000u : (v | (((flags >> i) & 1u) << 15))) : 0u;
        }
        out8[j] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
      }
      continue;
    }
    {
      const uint2* in8 = (const uint2*)(A.ys + gid * stride);
      for (uint32_t j = 0; j * 4 < posts; ++j) {
        const uint2 w = in8[j];
        col[(4 * j + 0) * NT] = w.x & 0xFFFFu;
        col[(4 * j + 1) * NT] = w.x >> 16;
        col[(4 * j + 2) * NT] = w.y & 0xFFFFu;
        col[(4 * j + 3) * NT] = w.y >> 16;
      }
    }
    uint64_t flags_lo = 3;
    uint32_t flag_64 = 0;
    bool bad = false;
    u32x16 kn = *(const_grp)(uintptr_t)&fc->sched[0][0];
    for (uint32_t gi = 0; gi < ngroups; ++gi) {
      const u32x16 kq = kn;
      kn = *(const_grp)(uintptr_t)&fc->sched[gi + 1u < VSYN_SCHED_GROUPS ? gi + 1u : gi][0];  // one group ahead
      uint32_t val[4], ylo[4], yhi[4], fn[4], prod[4], off[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        val[e] = col[kq[4 * e + 3] * NT];
        ylo[e] = col[(kq[4 * e] & 0xFFFFu) * NT];
        yhi[e] = col[(kq[4 * e] >> 16) * NT];
      }
      bool any_big = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t dxi = kq[4 * e + 1] & 0xFFFFu;
        const bool up = yhi[e] >= ylo[e];
        const uint32_t ady = up ? yhi[e] - ylo[e] : ylo[e] - yhi[e];
        prod[e] = ady * dxi;
        // off = (|dy| * dxi) / adx exactly, as floor((prod + 0.5) * (1 / adx)), while prod < 2^21 (always, for in-range amplitudes):
        // prod + 0.5 is exact and the product's rounding stays inside the 0.5 / adx guard band (vsyn_staged.h, predict_post)
        off[e] = (uint32_t)(((float)prod[e] + 0.5f) * __uint_as_float(kq[4 * e + 2]));
        any_big = any_big || prod[e] >= (1u << 21) || ady >= 65536u;
      }
      if (__any(any_big)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool up = yhi[e] >= ylo[e];
          const uint32_t ady = up ? yhi[e] - ylo[e] : ylo[e] - yhi[e];
          if (prod[e] >= (1u << 21) || ady >= 65536u) off[e] = prod[e] / (kq[4 * e + 1] >> 16);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool up = yhi[e] >= ylo[e];
        const uint32_t predicted = up ? ylo[e] + off[e] : ylo[e] - off[e];
        const bool ok = predicted <= range;  // hpp:536
        const uint32_t pr = ok ? predicted : 0u;
        const uint32_t high_room = range - pr, low_room = pr;
        const uint32_t room = min(high_room, low_room) * 2;
        const uint32_t big = high_room > low_room ? val[e] - low_room + pr : pr - val[e] + high_room - 1;
        const uint32_t small = (val[e] & 1u) ? pr - (val[e] + 1) / 2 : pr + val[e] / 2;
        fn[e] = val[e] == 0 ? pr : (val[e] >= room ? big : small);
        const uint32_t lo = kq[4 * e] & 0xFFFFu, hi = kq[4 * e] >> 16, i = kq[4 * e + 3];  // lo, hi < i <= 64
        const uint64_t touched = (1ull << lo) | (1ull << hi) | (i < 64u ? 1ull << i : 0ull);
        flags_lo |= val[e] != 0 ? touched : 0ull;
        flag_64 |= (val[e] != 0 && i >= 64u) ? 1u : 0u;
        bad = bad || !ok;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) col[kq[4 * e + 3] * NT] = bad ? 0u : fn[e];  // (a row with an out-of-range prediction is dropped: keep the chain tame)
    }
    uint2* out8 = (uint2*)(A.fy + gid * stride);
    if (bad) raise_status(A.status, VSYN_ST_FLOOR_RANGE, p);
    for (uint32_t j = 0; j * 4 < posts; ++j) {
      uint32_t w[4];
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t i = 4 * j + e;
        const uint32_t fv = i < posts ? col[i * NT] : 0u;
        uint32_t v = fv * mult;  // hpp:573,578
        if (v > 0x7FFFu || fv > 0x7FFFu) v = 0x7FFFu;  // wrapped / absurd amplitude: renders >= 256 -> FLOOR_VALUE later
        const uint32_t fl = i < 64 ? (uint32_t)((flags_lo >> i) & 1ull) : (i == 64 ? flag_64 : 0u);
        w[e] = i < posts ? (bad ? 0x8000u : (v | (fl << 15))) : 0u;
      }
      out8[j] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
    }
  }
}

// Inclusive scan over the PREP_THREADS threads of a workgroup (wave scans + the wave totals through LDS); *total = the whole block.
// Two barriers; s_abs / s_res: PREP_WAVES entries each.
__device__ __forceinline__ void prep_block_scan(AbsScan& inc, uint64_t& rinc, AbsScan* s_abs, uint64_t* s_res, AbsScan* total, uint64_t* rtotal) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  prep_wave_scan(inc, rinc, lane);
  __syncthreads();  // (the arrays may still be read from a previous scan)
  if (lane == 63u) {
    s_abs[wv] = inc;
    s_res[wv] = rinc;
  }
  __syncthreads();
  AbsScan pre = {0, 0}, tot = {0, 0};
  uint64_t rpre = 0, rtot = 0;
  const uint32_t nw = blockDim.x >> 6;
#pragma unroll
  for (uint32_t w = 0; w < PREP_WAVES; ++w) {
    if (w >= nw) break;
    const AbsScan a = s_abs[w];
    const uint64_t r = s_res[w];
    if (w < wv) {
      pre = abs_combine(pre, a);
      rpre += r;
    }
    tot = abs_combine(tot, a);
    rtot += r;
  }
  inc = abs_combine(pre, inc);
  rinc += rpre;
  *total = tot;
  *rtotal = rtot;
}

// One workgroup = one (segment, chunk of chunk_runs whole runs). Threads <-> packets for the scan (coalesced descriptor loads), threads
// <-> (packet, channel) rows for floor-1 step 1: chunk_runs is chosen so that a chunk has about PREP_THREADS rows, i.e. every lane of
// every wave unwraps exactly one row (the unwrap is the VALU-bound half of this kernel: ~60 instructions per post and row).
__global__ void __launch_bounds__(PREP_THREADS) vsyn_prep_kernel(const PrepCtx A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_rows[];  // [posts of the longest floor, rounded up to 4][PREP_THREADS]: prep_unwrap_rows
  __shared__ AbsScan s_abs[PREP_WAVES];
  __shared__ uint64_t s_res[PREP_WAVES];
  __shared__ uint32_t s_pk[PREP_THREADS];      // block size of each packet of the pass (a packet needs its predecessor's)
  __shared__ uint32_t s_longbits[PREP_MAX_SEG_PACKETS / 32 + 2];  // bit 0: the packet in front of the chunk, bit 1 + i: packet cs + i — set = a valid long block
  __shared__ uint32_t s_last_n;
  const uint8_t* __restrict__ cb = A.cb;
  const ConstHeader* H = hdr_of(cb);
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t NT = blockDim.x;  // 64 ... PREP_THREADS: the host sizes the workgroup to the rows a (segment, chunk) can have
  const uint32_t C = H->channels;
  // Two kinds of workgroup per (segment, chunk), dealt alternately so that both start at once: even blockIdx = the LAYOUT of the chunk's
  // packets (scan, PktInfo, run classes, stream state), odd = FLOOR-1 STEP 1 of their rows ("even / odd" in the sense spelled out below). The rows need nothing the scan produces —
  // which floor and whether the channel carries a curve follow from the packet's own descriptor — so the two halves of the preparation,
  // ~13 k and ~16 k cycles of dependent work, run side by side instead of one behind the other.
  // Dealing: consecutive workgroups go to consecutive XCDs, and inside an XCD consecutive ones to consecutive CUs (32 of them) — a plain
  // even / odd split would put every floor workgroup on four of the eight XCDs, and one by the XCD-local index j alone every floor
  // workgroup on every other CU. So: the pair (2m, 2m + 1) of XCD-local indices serves chunk 8 m + xcd, and which of the two is the
  // floor workgroup flips with j / 32: a CU's four workgroups j = k, k + 32, k + 64, k + 96 are two of each kind.
  const uint32_t xj = blockIdx.x >> 3;
  const uint32_t role = (xj ^ (xj >> 5)) & 1u, bid = ((xj >> 1) << 3) | (blockIdx.x & 7u);
  const uint32_t g = bid / A.chunks_per_seg, ch = bid % A.chunks_per_seg;
  if (g >= A.S) return;
#ifdef PREP_STAMPS
  unsigned long long pst_acc[PREP_NSTAMPS] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pst_last = __builtin_readcyclecounter();
  const unsigned long long pst_t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock: when this wave started
#define PSTAMP_FLUSH()                                                                                              \
  do {                                                                                                              \
    const uint32_t unit_ = blockIdx.x * PREP_WAVES + wave;                                                          \
    if (lane == 0 && unit_ < 8192) {                                                                                \
      for (int i_ = 0; i_ < 5; ++i_) g_prep_stamps[unit_][i_] = pst_acc[i_];                                        \
      g_prep_stamps[unit_][5] = pst_t0;                                                                             \
      g_prep_stamps[unit_][6] = __builtin_amdgcn_s_memrealtime();                                                   \
      g_prep_stamps[unit_][7] = 1ull + role;                                                                        \
    }                                                                                                               \
  } while (0)
#else
#define PSTAMP_FLUSH() do { } while (0)
#endif
  const vsyn_segment sg = A.segs[g];
  const uint32_t num = sg.num_packets;
  const uint32_t R = A.R;
  const uint32_t run0 = ch * A.chunk_runs;                          // first run of the chunk
  const uint32_t run1 = min(A.runs_per_seg, run0 + A.chunk_runs);   // one past its last run (of the batch's grid of runs)
  const uint32_t cs = run0 * R;                                     // first packet of the chunk
  uint8_t* const cls_row = A.run_cls + (size_t)g * A.runs_per_seg;
  if (sg.stream >= H->max_streams || (uint64_t)sg.first_packet + sg.num_packets > A.P || (sg.residue_off & 3)) {
    if (role) return;
    for (uint32_t r = run0 + t; r < run1; r += NT) cls_row[r] = 0xFFu;
    if (ch == 0u && t == 0u) {
      raise_status(A.status, VSYN_ST_BAD_SEGMENT, sg.first_packet < A.P ? sg.first_packet : 0);
      A.sinfo[g] = SegInfo{0, 0, 0, 0};
    }
    return;
  }
  const uint32_t ce = min(num, run1 * R);  // one past the chunk's last packet (<= cs: the chunk lies beyond the segment's end)
  const vsyn_packet* const spk = A.packets + sg.first_packet;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t num_modes = H->num_modes, bs0 = H->bs[0], bs1 = H->bs[1];
  const uint64_t long_modes = A.long_modes;
#define PREP_IS_LONG(m) ((m) < num_modes && (m) < 64u && ((long_modes >> (m)) & 1ull))
#define PREP_N_OF_MODE(m) (PREP_IS_LONG(m) ? bs1 : bs0)
  if (role) {
    // ---- floor-1 step 1 of the chunk's rows: thread <-> row (packet base + t / C, channel t % C) ---------------------------------------
    const uint32_t ce = min(num, run1 * R);
    if (cs >= ce) return;
    const vsyn_packet* const spk = A.packets + sg.first_packet;
    const uint32_t stride = __builtin_amdgcn_readfirstlane(H->ys_stride);
    const MapConst* const maps = (const MapConst*)(cb + H->off_map);
    const uint32_t chan_mask = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);
    const uint32_t num_modes = H->num_modes;
    const uint32_t ppp = NT / C;
    for (uint32_t base = cs; base < ce; base += ppp) {
      const uint32_t pe = min(ce, base + ppp);
      const uint32_t rp = t / C, c = t - rp * C;
      const bool row_ok = rp < pe - base;
      uint32_t kmode = 0xFFFFFFFFu, fused_mask = 0;
      if (row_ok) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = *(const u32x2*)(spk + base + rp);  // mode | window flags, floor_used
        kmode = w.x & 0xFFu;
        fused_mask = w.y;
      }
      const bool mode_ok = row_ok && kmode < num_modes;
      // which floor: mode -> mapping -> the channel's floor, one DISTINCT mode of the wave at a time through the scalar unit
      uint32_t fl_id = 0xFFFFFFFFu;
      {
        typedef const __attribute__((address_space(4))) uint32_t* kptr;
        uint64_t todo = __ballot(mode_ok);
        while (todo) {
          const uint32_t m = __builtin_amdgcn_readlane(kmode, (uint32_t)__builtin_ctzll(todo));
          const bool mine = mode_ok && kmode == m;
          todo &= ~__ballot(mine);
          const uint32_t mw = *(kptr)(uintptr_t)((const uint8_t*)H->mode_mapping + (m & ~3u));
          const uint32_t mp = (mw >> (8u * (m & 3u))) & 0xFFu;
          if (mine) fl_id = maps[mp].chfloor[c];
        }
      }
      // a row is unwrapped when its packet names a valid mode and carries a curve for the channel (hpp:1159); packets the scan will
      // flag for their granule or for plane overflow are unwrapped too — the reference decodes the floor before it gets there as well
      const bool act = mode_ok && (((fused_mask & chan_mask) >> c) & 1u);
      const uint32_t prow = sg.first_packet + base + rp;
      PSTAMP(3);  // floor role: descriptors, floor numbers
      prep_unwrap_rows(A, act, fl_id, prow, (size_t)prow * C + c, stride, (prep_lds_u32*)s_rows);
      PSTAMP(4);  // floor role: the chains
    }
    PSTAMP_FLUSH();
    return;
  }
  uint32_t st_slot;
  const StreamState st0 = state_read(A.state, sg.stream, A.epoch, &st_slot);
  const bool reset = (sg.flags & VSYN_SEG_RESET) != 0;
  const uint32_t carry_n = (!reset && st0.has_prev) ? st0.prev_n : 0;
  const int64_t abs0 = reset ? 0 : (int64_t)st0.abs_total_pos;
  SegInfo si;
  si.has_carry = carry_n ? 1u : 0u;
  si.carry_n = carry_n;
  si.parity_in = reset ? 0u : st0.parity;
  si.total_emit = 0;
  if (cs >= num) {  // the chunk lies beyond the segment's end (an empty segment: chunk 0 keeps its records in order)
    for (uint32_t r = run0 + t; r < run1; r += NT) cls_row[r] = 0xFFu;
    if (num == 0u && ch == 0u && t == 0u) {
      A.sinfo[g] = si;
      if (reset) state_write(A.state, sg.stream, st_slot, StreamState{0, 0, 0, 0, 0}, A.epoch);
    }
    return;
  }
  PSTAMP(0);  // header fields, stream state

  // ---- 1. the scan's values in front of the chunk: every workgroup reduces the segment's earlier descriptors for itself (a thread a
  //         contiguous piece, one block scan) instead of waiting for a predecessor — that is what makes the kernel dependency-free ------
  AbsScan cin = {0, 0};
  uint64_t cres = 0;
  uint32_t prev_n_in = carry_n;
  bool halo_long = !carry_n;  // the block in front of the chunk is a valid long one (or there is none at all)
  if (cs > 0u) {
    const uint32_t per = (cs + NT - 1u) / NT;
    const uint32_t b = min(cs, t * per), e = min(cs, b + per);
    AbsScan agg = {0, 0};
    uint64_t res = 0;
    uint32_t prev_n = 0;
    if (b < e) prev_n = b == 0u ? carry_n : PREP_N_OF_MODE((uint32_t)spk[b - 1u].mode);
    constexpr uint32_t KEEP = 8;
    for (uint32_t base = b; base < e; base += KEEP) {
      u32x4 kq[KEEP];
#pragma unroll
      for (uint32_t j = 0; j < KEEP; ++j)
        if (base + j < e) kq[j] = *(const u32x4*)(spk + base + j);
#pragma unroll
      for (uint32_t j = 0; j < KEEP; ++j)
        if (base + j < e) {
          const uint32_t md = kq[j].x & 0xFFu;
          const uint32_t n = PREP_N_OF_MODE(md);
          const int64_t gran = (int64_t)((uint64_t)kq[j].z | ((uint64_t)kq[j].w << 32));
          AbsScan el;
          el.set = gran >= 0;
          el.val = el.set ? gran : (prev_n ? (int64_t)(prev_n / 4 + n / 4) : 0);
          agg = abs_combine(agg, el);
          res += (uint64_t)C * (n / 2);
          prev_n = n;
        }
    }
    prep_block_scan(agg, res, s_abs, s_res, &cin, &cres);
    const uint32_t hm = __builtin_amdgcn_readfirstlane((uint32_t)spk[cs - 1u].mode);
    prev_n_in = PREP_N_OF_MODE(hm);
    halo_long = PREP_IS_LONG(hm);
  }
  PSTAMP(1);  // scan in front of the chunk

  const uint32_t stride = __builtin_amdgcn_readfirstlane(H->ys_stride);
  const MapConst* const maps = (const MapConst*)(cb + H->off_map);
  const uint32_t chan_mask = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);
  // packets per pass: as many whole packets as give at most PREP_THREADS rows (a stream of more than PREP_THREADS channels — there is
  // none: VSYN_MAX_CHANNELS is 32 — would need rows of one packet spread over passes)
  const uint32_t ppp = NT / C;
  for (uint32_t w = t; w < (ce - cs + 1u + 31u) / 32u; w += NT) s_longbits[w] = 0u;
  __syncthreads();
  if (t == 0u && halo_long) s_longbits[0] = 1u;
  for (uint32_t base = cs; base < ce; base += ppp) {
    // ---- 2. thread <-> packet base + t: PktInfo by a block scan ---------------------------------------------------------------------
    const uint32_t pe = min(ce, base + ppp);  // packets [base, pe) in this pass
    const uint32_t q = base + t;
    const bool valid = q < pe;
    vsyn_packet k = {};
    if (valid) {
      const u32x4 w = *(const u32x4*)(spk + q);
      k.mode = (uint8_t)(w.x & 0xFFu);
      k.prev_long = (uint8_t)((w.x >> 8) & 0xFFu);
      k.next_long = (uint8_t)((w.x >> 16) & 0xFFu);
      k.floor_used = w.y;
      k.granule = (int64_t)((uint64_t)w.z | ((uint64_t)w.w << 32));
    }
    const bool mode_ok = valid && k.mode < num_modes;
    const uint32_t kmode = k.mode;
    const uint32_t lng = (valid && PREP_IS_LONG(kmode)) ? 1u : 0u;
    const uint32_t n = lng ? bs1 : bs0;
    // block size in front of each packet: the previous thread's (across the wave boundary through LDS)
    __syncthreads();  // s_pk of the previous pass is no longer read
    s_pk[t] = n;
    __syncthreads();
    const uint32_t prev_n = t == 0u ? prev_n_in : s_pk[t - 1u];
    AbsScan inc = {0, 0};
    uint64_t rinc = 0;
    if (valid) {
      inc.set = k.granule >= 0;
      inc.val = inc.set ? k.granule : (prev_n ? (int64_t)(prev_n / 4 + n / 4) : 0);
      rinc = (uint64_t)C * (n / 2);
    }
    const AbsScan own_el = inc;
    const uint64_t own_res = rinc;
    AbsScan tot;
    uint64_t rtot;
    prep_block_scan(inc, rinc, s_abs, s_res, &tot, &rtot);
    // exclusive prefix = (everything in front of the pass) o (inclusive of the threads before this one)
    //   inclusive = ex o own  =>  for the two components:  residue: ex = inc - own;  AbsScan: recompute from the neighbour
    const uint64_t rex = rinc - own_res;
    AbsScan ex;
    {
      // the inclusive value of thread t - 1: shuffle inside the wave, LDS across the wave boundary (s_abs holds the wave totals, and
      // prep_block_scan's result already includes the earlier waves)
      ex.val = __shfl_up(inc.val, 1);
      ex.set = __shfl_up(inc.set, 1);
      if (lane == 0u) {
        AbsScan pre = {0, 0};
        for (uint32_t w = 0; w < wave; ++w) pre = abs_combine(pre, s_abs[w]);
        ex = pre;
      }
    }
    (void)own_el;
    ex = abs_combine(cin, ex);
    const int64_t abs_before = ex.set ? ex.val : abs0 + ex.val;
    const uint64_t res_off = sg.residue_off + cres + rex;
    const uint32_t p = sg.first_packet + q;
    // mode -> mapping and the nonzero propagate over the mapping's coupling steps (hpp:1174-1180), one DISTINCT mode of the wave at a
    // time: everything about a mode then comes through the scalar unit (a wave rarely sees more than two modes) instead of three
    // dependent per-lane loads
    uint32_t mapping = 0, own = k.floor_used & chan_mask, used = own;
    {
      typedef const __attribute__((address_space(4))) uint32_t* kptr;
      uint64_t todo = __ballot(mode_ok);
      while (todo) {
        const uint32_t m = __builtin_amdgcn_readlane(kmode, (uint32_t)__builtin_ctzll(todo));
        const bool mine = mode_ok && kmode == m;
        todo &= ~__ballot(mine);
        const uint32_t mw = *(kptr)(uintptr_t)((const uint8_t*)H->mode_mapping + (m & ~3u));
        const uint32_t mp = (mw >> (8u * (m & 3u))) & 0xFFu;
        const MapConst* mc = maps + mp;
        const uint32_t ncoup = *(kptr)(uintptr_t)&mc->ncoup;
        uint32_t u = used;
        for (uint32_t i = 0; i < ncoup; ++i) {
          const uint32_t pair = *(kptr)(uintptr_t)&mc->coup[2 * i];  // (magnitude, angle)
          const uint32_t ma = pair & 0xFFFFu, an = pair >> 16;
          if (((u >> ma) | (u >> an)) & 1u) u |= (1u << ma) | (1u << an);
        }
        if (mine) {
          mapping = mp;
          used = u;
        }
      }
    }
    if (valid) {
      const PktStep ps = pkt_step_core(k, mode_ok, lng, mapping, n, prev_n, abs_before, abs0, res_off, A.plane_stride, own, used);
      if (ps.raise) {
        if (ps.raise & VSYN_ST_BAD_MODE) raise_status(A.status, VSYN_ST_BAD_MODE, p);
        if (ps.raise & VSYN_ST_GRANULE) raise_status(A.status, VSYN_ST_GRANULE, p);
        if (ps.raise & VSYN_ST_PLANE_OVERFLOW) raise_status(A.status, VSYN_ST_PLANE_OVERFLOW, p);
      }
      A.info[p] = ps.pi;
      if (A.emit_len) A.emit_len[p] = ps.pi.emit;
      if (q == num - 1u) {  // the segment's last packet: stream state for the next submit, SegInfo for this one's consumers
        SegInfo so = si;
        so.total_emit = (uint32_t)(ps.abs_after - abs0);
        A.sinfo[g] = so;
        StreamState ns;
        ns.abs_total_pos = (uint64_t)ps.abs_after;
        ns.has_prev = 1;
        ns.prev_n = n;
        ns.parity = si.parity_in ^ 1u;
        ns.tag = 0;
        state_write(A.state, sg.stream, st_slot, ns, A.epoch);
      }
      if (mode_ok && lng) atomicOr(&s_longbits[(q - cs + 1u) >> 5], 1u << ((q - cs + 1u) & 31u));
      if (q == pe - 1u) s_last_n = n;
    }
    __syncthreads();  // s_last_n is in place; everyone has read its predecessor's block size
    PSTAMP(2);  // descriptors, block scan, PktInfo
    // carry the scan into the next pass (s_last_n was written before the barriers above)
    cin = abs_combine(cin, tot);
    cres += rtot;
    prev_n_in = s_last_n;
  }
  // ---- 4. which kernel takes each run of the chunk (vsyn_staged.h, run_class): 1 = every packet the run touches, its one-packet halo
  //         included, is a valid long block and there is no carry-in of another size; 2 = anything else a fused kernel covers ----------
  __syncthreads();
  {
    const uint32_t nruns_seg = (num + R - 1u) / R;
    for (uint32_t r = run0 + t; r < run1; r += NT) {
      uint32_t cls = 0xFFu;  // no such run
      if (r < nruns_seg) {
        const uint32_t first = r * R - cs, end = min(num, (r + 1u) * R) - cs;  // bits first (the halo) .. end (the run's last packet)
        bool all_long = true;
        for (uint32_t w = first >> 5; all_long && w <= (end >> 5); ++w) {
          uint32_t need = 0xFFFFFFFFu;
          if (w == (first >> 5)) need &= 0xFFFFFFFFu << (first & 31u);
          if (w == (end >> 5)) need &= 0xFFFFFFFFu >> (31u - (end & 31u));
          all_long = (s_longbits[w] & need) == need;
        }
        cls = (all_long && (A.fused_ok & 1u)) ? 1u : ((A.fused_ok & 2u) ? 2u : 0u);
      }
      cls_row[r] = (uint8_t)cls;
    }
  }
  PSTAMP_FLUSH();
#undef PREP_IS_LONG
#undef PREP_N_OF_MODE
}
