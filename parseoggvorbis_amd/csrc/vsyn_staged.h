// vsyn_staged.h — layout / floor-unwrap kernels (used by every path) and the STAGED synthesis kernels:
// any blocksize 64..8192, any channel count / coupling list, every intermediate materialised in HBM so it
// can be handed out as a debug tap (the reference's push_data_* hooks).  Correctness-first; the fused
// kernels in vsyn_fused.h are the speed path.  Compiled with -ffp-contract=off: every a*b+c below is two
// roundings unless written as __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>

#include "vsyn_device.h"

// ------------------------------------------------------------------------------------------------
// K0  layout: one workgroup per segment.  Restates the integer bookkeeping of
// VorbisStreamDecodeState::advancePcmOffsetBeginAudioPacket / forwardReadyPcm (hpp:1019-1067) as a scan:
//   natural frames L_q = n_{q-1}/4 + n_q/4 (0 for a stream's first packet, hpp:1021-1027)
//   abs_after_q = granule_q if granule_q >= 0 (hpp:1028-1044,1056-1057) else abs_before_q + L_q
// plus the residue offset prefix sum and the nonzero propagate (hpp:1174-1180).
// ------------------------------------------------------------------------------------------------
struct AbsScan {
  int64_t val;
  int set;
};
__device__ __forceinline__ AbsScan abs_combine(AbsScan a, AbsScan b) {
  AbsScan r;
  if (b.set) return b;
  r.set = a.set;
  r.val = a.val + b.val;
  return r;
}

// Shared by the layout kernel (which builds the staged list) and the fused kernel (which skips non-fast runs):
// both must take the same decision.
// Which kernel takes a run of packets [qa, qb) of a segment: 1 = fused long-run kernel (every packet it touches, its
// one-packet halo included, is a long block and there is no carry-in from an earlier submit), 2 = fused mixed-block
// kernel (any mix of valid short/long blocks, carry-in allowed), 0 = staged work list. Shared by the layout kernel
// (which builds the list) and the fused kernels (which skip what is not theirs): all three must agree.
__device__ __forceinline__ uint32_t run_class(const ConstHeader* H, const vsyn_packet* __restrict__ spk, uint32_t qa, uint32_t qb,
                                              uint32_t carry_n, uint32_t ok_mask) {
  if (!ok_mask) return 0;
  bool all_long = !(qa == 0 && carry_n), all_valid = true;
  for (uint32_t q = qa ? qa - 1 : 0; q < qb; ++q) {
    const uint32_t m = spk[q].mode;
    const bool valid = m < H->num_modes;
    all_valid = all_valid && valid;
    all_long = all_long && valid && H->mode_blockflag[m];
  }
  if (all_long && (ok_mask & 1u)) return 1;
  if (ok_mask & 2u) return 2;  // the mixed kernel also steps over packets with an invalid mode (PktInfo.bad)
  (void)all_valid;
  return 0;
}

// The same decision from a bitmap in LDS (bit = a valid long block): bits [first, end) are the run's packets, bit first - 1 its
// one-packet halo (for the segment's first run there is none: the carry-in decides). Classifying a run is then a few LDS words
// instead of a serial walk over R+1 descriptors in global memory (which was most of the layout kernel's time, and its
// workgroups sit on wave slots the synthesis kernel of the previous submit is waiting for).
__device__ __forceinline__ uint32_t run_class_bits(const uint32_t* bits, uint32_t first, uint32_t end, uint32_t carry_n, uint32_t ok_mask, bool seg_start) {
  if (!ok_mask) return 0;
  bool all_long = !(seg_start && carry_n);
  const uint32_t lo = seg_start ? first : first - 1u;  // one-packet halo
  for (uint32_t w = lo >> 5; all_long && w <= ((end - 1u) >> 5); ++w) {
    uint32_t need = 0xFFFFFFFFu;
    if (w == (lo >> 5)) need &= 0xFFFFFFFFu << (lo & 31u);
    if (w == ((end - 1u) >> 5)) need &= 0xFFFFFFFFu >> (31u - ((end - 1u) & 31u));
    all_long = (bits[w] & need) == need;
  }
  if (all_long && (ok_mask & 1u)) return 1;
  if (ok_mask & 2u) return 2;
  return 0;
}

// One packet of the scan, shared by the layout kernel and the in-wave preparation of the fused kernels (vsyn_prep.h): everything
// PktInfo holds, from the packet's descriptor, the block size in front of it and the scan's running values (hpp:1019-1067, 1174-1180).
struct PktStep {
  PktInfo pi;
  int64_t abs_after;
  uint32_t raise;  // VSYN_ST_* to raise for this packet (0: none)
};
// (own / used: floor_output_used before and after the nonzero propagate of hpp:1174-1180, see coupling_propagate)
__device__ __forceinline__ PktStep pkt_step_core(const vsyn_packet& k, bool mode_ok, uint32_t lng, uint32_t mapping, uint32_t n, uint32_t prev_n,
                                                 int64_t abs_before, int64_t abs0, uint64_t res_off, uint64_t plane_stride, uint32_t own, uint32_t used) {
  PktStep r;
  PktInfo pi = {};
  r.raise = mode_ok ? 0u : (uint32_t)VSYN_ST_BAD_MODE;
  const uint32_t L = prev_n ? prev_n / 4 + n / 4 : 0;
  int64_t abs_after = abs_before + L;
  uint32_t emit = L;
  bool bad = !mode_ok;
  if (k.granule >= 0) {
    // hpp:1029 (position already past the page granule) and hpp:1041 (packets cannot reach it)
    if (k.granule < abs_before || k.granule > abs_before + (int64_t)L) {
      r.raise |= VSYN_ST_GRANULE;
      bad = true;
      emit = 0;
    } else {
      emit = (uint32_t)(k.granule - abs_before);
    }
    abs_after = k.granule;
  }
  const int64_t rel = abs_before - abs0;
  if (rel < 0 || (uint64_t)rel + emit > plane_stride) {
    if (emit) r.raise |= VSYN_ST_PLANE_OVERFLOW;
    if (emit) bad = true;
    emit = 0;
  }
  pi.res_off = res_off;
  pi.out_pos = rel < 0 ? 0u : (uint32_t)rel;
  pi.emit = emit;
  pi.n = (uint16_t)n;
  pi.lng = (uint8_t)lng;
  pi.widx = lng ? (uint8_t)((k.prev_long ? 1 : 0) | (k.next_long ? 2 : 0)) : 0;
  pi.mapping = (uint8_t)mapping;
  pi.bad = bad ? 1 : 0;
  pi.own = own;
  pi.used = used;
  r.pi = pi;
  r.abs_after = abs_after;
  return r;
}
__device__ __forceinline__ PktStep pkt_step(const MapConst* __restrict__ maps, const vsyn_packet& k, bool mode_ok, uint32_t lng, uint32_t mapping,
                                            uint32_t n, uint32_t prev_n, int64_t abs_before, int64_t abs0, uint64_t res_off,
                                            uint64_t plane_stride, uint32_t C) {
  const uint32_t chan_mask = C >= 32 ? 0xFFFFFFFFu : ((1u << C) - 1u);
  uint32_t own = k.floor_used & chan_mask, used = own;
  const MapConst* mc = maps + mapping;
  for (uint32_t i = 0; i < mc->ncoup; ++i) {  // hpp:1175-1180
    uint32_t m = mc->coup[2 * i], a = mc->coup[2 * i + 1];
    if (((used >> m) | (used >> a)) & 1u) used |= (1u << m) | (1u << a);
  }
  return pkt_step_core(k, mode_ok, lng, mapping, n, prev_n, abs_before, abs0, res_off, plane_stride, own, used);
}

// A segment is scanned in CHUNKS of chunk_packets packets (a multiple of the run length R, so that a run never straddles two
// chunks), one workgroup per (segment, chunk). With one chunk per segment — every segment of the batch is at most
// LAYOUT_CHUNK_PACKETS long: the usual case — nothing below about look-back runs. Longer segments (ONE long stream is the second
// partitioning of SURVEY 8e) chain their chunks with a decoupled look-back: a chunk publishes the aggregate of its own packets, walks
// back over its predecessors' records until it meets one that already holds an inclusive prefix, and publishes its own inclusive
// prefix. The serial scan of one workgroup per segment made 1 x 65 536 packets run at half the rate of 64 x 1024.
// Block size LAYOUT_THREADS, or one wavefront per chunk (LAYOUT_THREADS_SHORT) when no chunk of the batch has more than
// LAYOUT_SHORT_PACKETS packets: a batch of thousands of short streams would otherwise put four nearly idle waves per segment
// on the chip, next to the synthesis kernel. Dynamic LDS: layout_lds_bytes().
#define LAYOUT_THREADS 256
#define LAYOUT_THREADS_SHORT 64
#define LAYOUT_SHORT_PACKETS 256u
#define LAYOUT_CHUNK_PACKETS 4096u
struct LayoutChunk {          // look-back record of one (segment, chunk); 48 bytes
  uint32_t flag;              // (epoch & 2^30-1) * 4 + 1: aggregate valid, + 2: inclusive prefix valid, + 3: chain lost (no clearing between submits)
  uint32_t pad;
  int64_t agg_val, inc_val;   // AbsScan of the chunk's own packets / of everything up to its end
  uint64_t agg_res, inc_res;  // residue floats likewise
  uint32_t agg_set, inc_set;
};
static inline __host__ __device__ size_t layout_lds_bytes(uint32_t threads, uint32_t chunk_packets) {
  return (size_t)threads * (sizeof(AbsScan) + sizeof(uint64_t)) + (size_t)((chunk_packets + 1u + 31u) / 32u) * 4u + 16u;
}
__global__ void __launch_bounds__(LAYOUT_THREADS)
vsyn_layout_kernel(const uint8_t* __restrict__ cb, uint32_t P, const vsyn_packet* __restrict__ pk, uint32_t S,
                   const vsyn_segment* __restrict__ segs, uint64_t plane_stride, PktInfo* __restrict__ info,
                   SegInfo* __restrict__ sinfo, StreamState* __restrict__ state, uint32_t* __restrict__ emit_len,
                   DevStatus* __restrict__ status, uint32_t R, uint32_t fused_ok, uint32_t* __restrict__ staged_list,
                   uint32_t* __restrict__ staged_count, uint32_t* __restrict__ next_count, uint32_t* __restrict__ seg_of_pkt,
                   uint8_t* __restrict__ run_cls, uint32_t runs_per_seg, uint32_t chunk_packets, uint32_t chunks_per_seg,
                   LayoutChunk* __restrict__ chunks, uint32_t epoch) {
  const ConstHeader* H = hdr_of(cb);
  const uint32_t g = blockIdx.x / chunks_per_seg, ch = blockIdx.x % chunks_per_seg, t = threadIdx.x, NT = blockDim.x;
  if (g >= S) return;
  if (blockIdx.x == 0 && t == 0) *next_count = 0;  // the other submit parity's list counter (no memset node needed)
  const vsyn_segment sg = segs[g];
  const bool seg_ok = sg.stream < H->max_streams && (uint64_t)sg.first_packet + sg.num_packets <= P && (sg.residue_off & 3) == 0;
  const uint32_t cs = ch * chunk_packets;  // first packet of this chunk
  const uint32_t runs_per_chunk = chunk_packets / R;
  if (!seg_ok) {
    if (t == 0 && ch == 0) {
      raise_status(status, VSYN_ST_BAD_SEGMENT, sg.first_packet < P ? sg.first_packet : 0);
      sinfo[g] = SegInfo{0, 0, 0, 0};
    }
    for (uint32_t r = ch * runs_per_chunk + t; r < min(runs_per_seg, (ch + 1u) * runs_per_chunk); r += NT) run_cls[(size_t)g * runs_per_seg + r] = 0xFFu;
    // mark every packet we may safely touch as bad so later kernels skip it
    if ((uint64_t)sg.first_packet + sg.num_packets <= P)
      for (uint32_t q = cs + t; q < min(sg.num_packets, cs + chunk_packets); q += NT) {
        PktInfo pi = {};
        pi.bad = 1;
        pi.n = (uint16_t)H->bs[0];
        info[sg.first_packet + q] = pi;
      }
    return;
  }
  const uint32_t C = H->channels, num = sg.num_packets;
  // runs of this chunk that lie beyond the segment's end
  if (cs >= num && !(ch == 0)) {
    for (uint32_t r = ch * runs_per_chunk + t; r < min(runs_per_seg, (ch + 1u) * runs_per_chunk); r += NT) run_cls[(size_t)g * runs_per_seg + r] = 0xFFu;
    return;
  }
  const uint32_t cn = min(num, cs + chunk_packets) - min(num, cs);  // packets of this chunk
  const bool last_chunk = cs + cn >= num;
  uint32_t st_slot;
  const StreamState st0 = state_read(state, sg.stream, epoch, &st_slot);
  const bool reset = (sg.flags & VSYN_SEG_RESET) != 0;
  const uint32_t carry_n = (!reset && st0.has_prev) ? st0.prev_n : 0;
  const int64_t abs0 = reset ? 0 : (int64_t)st0.abs_total_pos;

  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // layout_lds_bytes(): scan arrays [NT], then the bitmap
  AbsScan* s_abs = (AbsScan*)s_dyn;
  uint64_t* s_res = (uint64_t*)(s_dyn + (size_t)NT * sizeof(AbsScan));
  // bit 0: the packet in front of the chunk (the first run's halo), bit 1 + i: packet cs + i — set = a valid long block
  uint32_t* s_longbits = (uint32_t*)(s_dyn + (size_t)NT * (sizeof(AbsScan) + sizeof(uint64_t)));
  for (uint32_t w = t; w < (cn + 1u + 31u) / 32u; w += NT) s_longbits[w] = 0u;
  __shared__ int64_t s_abs_end;
  __shared__ uint32_t s_last_n;
  __shared__ AbsScan s_chunk_ex;   // exclusive prefix of this chunk (everything before it in the segment)
  __shared__ uint64_t s_chunk_rex;
  __shared__ uint32_t s_chunk_lost;

  const uint32_t per = (cn + NT - 1) / NT;
  const uint32_t qb = min(cn, t * per), qe = min(cn, qb + per);  // this thread's packets, relative to the chunk
  const vsyn_packet* spk = pk + sg.first_packet;                 // (segment-relative indexing: spk[cs + i])

  // mode -> (block flag, mapping) from LDS, and — when a thread has few packets — its descriptors loaded up front in one
  // burst and kept in registers for both passes: the kernel is a chain of small dependent loads otherwise, and while it
  // runs its workgroups hold wave slots of 64 CUs
  __shared__ uint8_t s_bf[VSYN_MAX_TABLES], s_mm[VSYN_MAX_TABLES];
  if (t < VSYN_MAX_TABLES) {
    s_bf[t] = H->mode_blockflag[t];
    s_mm[t] = H->mode_mapping[t];
  }
  constexpr uint32_t KEEP = 8;
  const bool keep = per <= KEEP;  // then pass B reuses what pass A loaded
  vsyn_packet kq[KEEP];
  const uint32_t num_modes = H->num_modes, bs0 = H->bs[0], bs1 = H->bs[1];
  uint32_t carry_prev_mode = 0xFFFFFFFFu;  // mode of the packet before this thread's first one (for the block size in front of it)
  if (cs + qb > 0 && qb < cn) carry_prev_mode = spk[cs + qb - 1].mode;
  __syncthreads();
  auto n_of_mode = [&](uint32_t m) -> uint32_t { return (m < num_modes && s_bf[m]) ? bs1 : bs0; };
  if (t == 0 && cs > 0) {  // the halo bit of the chunk's first run
    const uint32_t m = spk[cs - 1].mode;
    if (m < num_modes && s_bf[m]) atomicOr(&s_longbits[0], 1u);
  }
  const uint32_t prev_n0 = cs + qb == 0 ? carry_n : (qb < cn ? n_of_mode(carry_prev_mode) : 0);

  // pass A: per-thread aggregate
  AbsScan agg = {0, 0};
  uint64_t res = 0;
  {
    uint32_t prev_n = prev_n0;
    auto step_a = [&](const vsyn_packet& k) {
      const uint32_t n = n_of_mode(k.mode);
      AbsScan e;
      e.set = k.granule >= 0;
      e.val = e.set ? k.granule : (prev_n ? (int64_t)(prev_n / 4 + n / 4) : 0);
      agg = abs_combine(agg, e);
      res += (uint64_t)C * (n / 2);
      prev_n = n;
    };
    // KEEP descriptors per burst: one exposed round trip per burst instead of one per packet
    for (uint32_t base = qb; base < qe; base += KEEP) {
#pragma unroll
      for (uint32_t j = 0; j < KEEP; ++j)
        if (base + j < qe) kq[j] = spk[cs + base + j];
#pragma unroll
      for (uint32_t j = 0; j < KEEP; ++j)
        if (base + j < qe) step_a(kq[j]);
    }
  }
  // exclusive scan over the thread aggregates: wave-level shuffles, then the wave totals through LDS
  AbsScan chunk_agg;
  uint64_t chunk_res;
  {
    const uint32_t lane = t & 63u, wv = t >> 6, NW = NT >> 6;
    AbsScan inc = agg;
    uint64_t rinc = res;
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
    if (lane == 63) {
      s_abs[wv] = inc;
      s_res[wv] = rinc;
    }
    __syncthreads();
    AbsScan pre = {0, 0};
    uint64_t rpre = 0;
    for (uint32_t w = 0; w < wv; ++w) {
      pre = abs_combine(pre, s_abs[w]);
      rpre += s_res[w];
    }
    chunk_agg = pre;
    chunk_res = rpre;
    for (uint32_t w = wv; w < NW; ++w) {  // (every thread: the chunk's total)
      chunk_agg = abs_combine(chunk_agg, s_abs[w]);
      chunk_res += s_res[w];
    }
    // exclusive = (prefix of earlier waves) o (inclusive of lane-1)
    AbsScan ex;
    ex.val = __shfl_up(inc.val, 1);
    ex.set = __shfl_up(inc.set, 1);
    uint64_t rex = __shfl_up(rinc, 1);
    if (lane == 0) {
      ex.val = 0;
      ex.set = 0;
      rex = 0;
    }
    ex = abs_combine(pre, ex);
    rex += rpre;
    __syncthreads();
    s_abs[t] = ex;
    s_res[t] = rex;
  }
  // chunk chain (segments longer than one chunk only): decoupled look-back over the predecessors' records
  if (chunks_per_seg > 1) {
    if (t == 0) {
      const uint32_t ep = epoch & 0x3FFFFFFFu;  // the tag shares its word with two state bits
      LayoutChunk* mine = chunks + (size_t)g * chunks_per_seg + ch;
      AbsScan ex = {0, 0};
      uint64_t rex = 0;
      bool lost = false;  // a predecessor's record never arrived within the bound
      if (ch > 0) {
        mine->agg_val = chunk_agg.val;
        mine->agg_set = (uint32_t)chunk_agg.set;
        mine->agg_res = chunk_res;
        __hip_atomic_store(&mine->flag, ep * 4u + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        AbsScan run = {0, 0};  // aggregate of the chunks between the record in hand and this chunk
        uint64_t rrun = 0;
        for (uint32_t pc = ch; pc-- > 0;) {
          LayoutChunk* pr = chunks + (size_t)g * chunks_per_seg + pc;
          uint32_t f;
          uint32_t spins = 0;
          while (((f = __hip_atomic_load(&pr->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 2) != ep || (f & 3u) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 24)) {  // (bounded: a predecessor that never arrives cannot hang the device; the batch is flagged)
              lost = true;
              break;
            }
          }
          if ((f & 3u) == 3u) lost = true;  // a predecessor gave up: so does everything behind it, at once
          if (lost) break;
          if ((f & 3u) == 2u) {
            AbsScan a = {pr->inc_val, (int)pr->inc_set};
            ex = abs_combine(a, run);
            rex = pr->inc_res + rrun;
            break;
          }
          AbsScan a = {pr->agg_val, (int)pr->agg_set};
          run = abs_combine(a, run);
          rrun += pr->agg_res;
        }
      }
      if (lost) raise_status(status, VSYN_ST_BAD_SEGMENT, sg.first_packet);
      s_chunk_lost = lost ? 1u : 0u;
      const AbsScan incl = abs_combine(ex, chunk_agg);
      mine->inc_val = incl.val;
      mine->inc_set = (uint32_t)incl.set;
      mine->inc_res = rex + chunk_res;
      __hip_atomic_store(&mine->flag, ep * 4u + (lost ? 3u : 2u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      s_chunk_ex = ex;
      s_chunk_rex = rex;
    }
  } else if (t == 0) {
    s_chunk_ex = AbsScan{0, 0};
    s_chunk_rex = 0;
    s_chunk_lost = 0u;
  }
  __syncthreads();
  if (s_chunk_lost) {
    // the chain broke in front of this chunk (flagged above): its positions are unknown — mark its packets bad and its runs absent
    // instead of laying them out from a record that was never validated
    for (uint32_t r = ch * runs_per_chunk + t; r < min(runs_per_seg, (ch + 1u) * runs_per_chunk); r += NT) run_cls[(size_t)g * runs_per_seg + r] = 0xFFu;
    for (uint32_t q = cs + t; q < cs + cn; q += NT) {
      PktInfo pi = {};
      pi.bad = 1;
      pi.n = (uint16_t)H->bs[0];
      info[sg.first_packet + q] = pi;
      if (emit_len) emit_len[sg.first_packet + q] = 0;
    }
    if (t == 0 && last_chunk) sinfo[g] = SegInfo{0, 0, 0, 0};
    return;
  }

  // pass B
  {
    AbsScan pre = abs_combine(s_chunk_ex, s_abs[t]);
    int64_t abs_before = pre.set ? pre.val : abs0 + pre.val;
    uint64_t res_off = sg.residue_off + s_chunk_rex + s_res[t];
    uint32_t prev_n = prev_n0;
    auto step_b = [&](uint32_t ql, const vsyn_packet& k) {  // ql: index inside the chunk
      const uint32_t q = cs + ql;
      const uint32_t p = sg.first_packet + q;
      const bool mode_ok = k.mode < num_modes;
      const uint32_t lng = mode_ok && s_bf[k.mode] ? 1u : 0u;
      const uint32_t n = lng ? bs1 : bs0;
      const PktStep ps = pkt_step(map_of(cb, 0), k, mode_ok, lng, mode_ok ? s_mm[k.mode] : 0u, n, prev_n, abs_before, abs0, res_off, plane_stride, C);
      // (one flag at a time, in the order the checks are made: first_bad_packet is a minimum over packets, the flags an OR)
      if (ps.raise & VSYN_ST_BAD_MODE) raise_status(status, VSYN_ST_BAD_MODE, p);
      if (ps.raise & VSYN_ST_GRANULE) raise_status(status, VSYN_ST_GRANULE, p);
      if (ps.raise & VSYN_ST_PLANE_OVERFLOW) raise_status(status, VSYN_ST_PLANE_OVERFLOW, p);
      info[p] = ps.pi;
      if (mode_ok && lng) atomicOr(&s_longbits[(ql + 1u) >> 5], 1u << ((ql + 1u) & 31u));
      seg_of_pkt[p] = g;
      if (emit_len) emit_len[p] = ps.pi.emit;
      if (q == num - 1) {
        s_abs_end = ps.abs_after;
        s_last_n = n;
      }
      abs_before = ps.abs_after;
      res_off += (uint64_t)C * (n / 2);
      prev_n = n;
    };
    for (uint32_t base = qb; base < qe; base += KEEP) {
      if (!keep) {
#pragma unroll
        for (uint32_t j = 0; j < KEEP; ++j)
          if (base + j < qe) kq[j] = spk[cs + base + j];
      }
#pragma unroll
      for (uint32_t j = 0; j < KEEP; ++j)
        if (base + j < qe) step_b(base + j, kq[j]);
    }
  }
  __syncthreads();

  // pass C: classify the chunk's runs of R packets. A run goes to the fused long-block kernel iff every packet it touches
  // (its one-packet halo included) is a valid long block and any carry-in is a long block; all other runs are
  // appended to the staged work list (entry = packet index | emit << 31; halo packets carry emit = 0).
  {
    const uint32_t nruns = (num + R - 1) / R;
    for (uint32_t r = ch * runs_per_chunk + t; r < min(runs_per_seg, (ch + 1u) * runs_per_chunk); r += NT) {
      const uint32_t qa = r * R, qb2 = min(num, qa + R);
      // (bitmap positions: packet q of the chunk at bit q - cs + 1; the halo of the chunk's first run at bit 0)
      const uint32_t cls = r >= nruns ? 0xFFu  // 0xFF: no such run
                                      : run_class_bits(s_longbits, qa - cs + 1u, qb2 - cs + 1u, qa == 0 ? carry_n : 0u, fused_ok, qa == 0);
      run_cls[(size_t)g * runs_per_seg + r] = (uint8_t)cls;  // the fused kernels read this instead of re-deriving it
      if (r >= nruns || cls) continue;
      const bool prev_fast = qa > 0 && run_class(H, spk, qa - R, qa, carry_n, fused_ok) != 0;
      const uint32_t cnt = (qb2 - qa) + (prev_fast ? 1u : 0u);
      uint32_t at = atomicAdd(staged_count, cnt);
      if (prev_fast) staged_list[at++] = sg.first_packet + qa - 1;  // IMDCT only: the overlap of `qa` reads its block
      for (uint32_t q = qa; q < qb2; ++q) staged_list[at++] = (sg.first_packet + q) | 0x80000000u;
    }
  }
  if (t == 0 && (last_chunk || num == 0)) {
    SegInfo si;
    si.has_carry = carry_n ? 1u : 0u;
    si.carry_n = carry_n;
    si.parity_in = reset ? 0u : st0.parity;
    if (num == 0) {
      si.total_emit = 0;
      sinfo[g] = si;
      if (reset) state_write(state, sg.stream, st_slot, StreamState{0, 0, 0, 0, 0}, epoch);
      return;
    }
    si.total_emit = (uint32_t)(s_abs_end - abs0);
    sinfo[g] = si;
    StreamState ns;
    ns.abs_total_pos = (uint64_t)s_abs_end;
    ns.has_prev = 1;
    ns.prev_n = s_last_n;
    ns.parity = si.parity_in ^ 1u;
    ns.tag = 0;
    state_write(state, sg.stream, st_slot, ns, epoch);
  }
}

// ------------------------------------------------------------------------------------------------
// K1  floor-1 step 1 (amplitude value synthesis), hpp:521-559, one thread per (packet, channel).
// Serial over <=65 posts, parallel over the batch. Neighbour indices come precomputed from the setup
// (Utils.hpp:60-118 depend on xs only).  uint32 wrap-around semantics as in the reference (y_t = uint32_t).
// Output row: final_y * multiplier (saturated to 15 bits) | step2_flag << 15, header order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t render_point_u32(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t X) {
  // Utils.hpp:122-137
  uint32_t adx = x1 - x0;
  bool up = y1 >= y0;
  uint32_t ady = up ? y1 - y0 : y0 - y1;
  uint32_t off = (ady * (X - x0)) / adx;
  return up ? y0 + off : y0 - off;
}

// predicted = render_point(xs[lo], fy[lo], xs[hi], fy[hi], xs[i]) with the post geometry folded into the
// per-floor constants dxi = xs[i]-xs[lo] and inv = 1/(xs[hi]-xs[lo]):  off = (|dy| * dxi) / adx  exactly, via
// floor((|dy|*dxi + 0.5) * inv) while |dy|*dxi < 2^24 (always, for in-range amplitudes); else the integer divide.
__device__ __forceinline__ uint32_t predict_post(uint32_t ylo, uint32_t yhi, uint32_t dxi, uint32_t adx, float inv) {
  const bool up = yhi >= ylo;
  const uint32_t ady = up ? yhi - ylo : ylo - yhi;
  uint32_t off;
  const uint32_t prod = ady * dxi;
  // float path only where it is provably the integer quotient: prod + 0.5 exact (< 2^23) and the rounding of the
  // product (<= 2.4e-7 * prod/adx) stays inside the 0.5/adx guard band (prod < 2^21); in-range amplitudes give
  // prod <= 255 * 4096
  if (ady < 65536u && prod < (1u << 21)) off = (uint32_t)(((float)prod + 0.5f) * inv);
  else off = prod / adx;
  return up ? ylo + off : ylo - off;
}

#define UNWRAP_THREADS 128
#define UNWRAP_REG_POSTS 32
__global__ void __launch_bounds__(UNWRAP_THREADS)
vsyn_floor_unwrap_kernel(const uint8_t* __restrict__ cb, uint32_t P, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                         const PktInfo* __restrict__ info, const uint16_t* __restrict__ ys, uint16_t* __restrict__ fy_out,
                         DevStatus* __restrict__ status) {
  // list == nullptr: rows are all (packet, channel) pairs; else rows come from the staged work list
  __shared__ uint32_t s_fy[VSYN_MAX_POSTS][UNWRAP_THREADS];  // post-major: thread-contiguous, conflict-free
  const ConstHeader* H = hdr_of(cb);
  const uint32_t C = H->channels, stride = H->ys_stride, t = threadIdx.x;
  const uint32_t rows = (list ? *count : P) * C;
  for (uint32_t base = blockIdx.x * UNWRAP_THREADS; base < rows; base += gridDim.x * UNWRAP_THREADS) {
    const uint32_t row = base + t;
    if (row >= rows) continue;
    const uint32_t p = list ? (list[row / C] & 0x7FFFFFFFu) : row / C, c = row % C;
    const uint32_t gid = p * C + c;
    const PktInfo pi = info[p];
    if (pi.bad || !((pi.own >> c) & 1u)) continue;
    const FloorConst* fc = floor_of(cb, map_of(cb, pi.mapping)->chfloor[c]);
    uint16_t* out = fy_out + (size_t)gid * stride;
    const uint32_t posts = fc->posts, range = fc->range, mult = fc->mult;
    // When every active row of the wavefront uses the same floor (always, unless block sizes alternate inside the 64 rows) the
    // per-post constants are wave-uniform: read them through the scalar unit (constant address space -> s_load, scalar cache)
    // instead of 64 identical per-lane loads per post.
    const uint64_t fc_bits = (uint64_t)(uintptr_t)fc;
    const uint64_t fc_first = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)fc_bits) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(fc_bits >> 32)) << 32);
    const bool fc_uniform = __all(fc_bits == fc_first);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x4* const_pk;
#ifndef VSYN_NO_UNWRAP_REGS
    // Up to 32 posts (every floor libvorbis writes for the common modes): the row lives in a per-thread register array
    // indexed by the wave-uniform neighbour indices (s_set_gpr_idx / v_movrel), so the serial chain over the posts is a
    // few dozen VALU cycles per post instead of three dependent LDS round trips. 32 is where the compiler still keeps the
    // array in registers; longer floors take the LDS path below.
    if (fc_uniform && __builtin_amdgcn_readfirstlane(posts) <= UNWRAP_REG_POSTS) {
      const FloorConst* fcu = (const FloorConst*)(uintptr_t)fc_first;
      const uint32_t posts_u = __builtin_amdgcn_readfirstlane(posts);
      uint32_t f[UNWRAP_REG_POSTS];
      const uint2* in8 = (const uint2*)(ys + (size_t)gid * stride);
#pragma unroll
      for (uint32_t j = 0; j < UNWRAP_REG_POSTS / 4; ++j) {
        uint2 w = make_uint2(0u, 0u);
        if (j * 4 < posts_u) w = in8[j];
        f[4 * j + 0] = w.x & 0xFFFFu;
        f[4 * j + 1] = w.x >> 16;
        f[4 * j + 2] = w.y & 0xFFFFu;
        f[4 * j + 3] = w.y >> 16;
      }
      uint32_t flags = 3;
      bool bad = false;
      u32x4 kn = *(const_pk)(uintptr_t)&fcu->pk[2];
      for (uint32_t i = 2; i < posts_u; ++i) {
        const u32x4 kq = kn;
        kn = *(const_pk)(uintptr_t)&fcu->pk[i + 1];  // next post's constants while this one computes (pk[] has 65 entries)
        const uint32_t lo = kq.x & 0xFFFFu, hi = kq.x >> 16;
        const uint32_t val = f[i], ylo = f[lo], yhi = f[hi];
        const uint32_t dxi = kq.y & 0xFFFFu, adx = kq.y >> 16;
        const bool up = yhi >= ylo;
        const uint32_t ady = up ? yhi - ylo : ylo - yhi;
        const uint32_t prod = ady * dxi;
        uint32_t off = (uint32_t)(((float)prod + 0.5f) * __uint_as_float(kq.z));
        if (__any(prod >= (1u << 21))) off = prod >= (1u << 21) ? prod / adx : off;
        const uint32_t predicted = up ? ylo + off : ylo - off;
        const bool ok = predicted <= range;  // hpp:536
        const uint32_t pr = ok ? predicted : 0u;
        const uint32_t high_room = range - pr, low_room = pr;
        const uint32_t room = min(high_room, low_room) * 2;
        const uint32_t big = high_room > low_room ? val - low_room + pr : pr - val + high_room - 1;
        const uint32_t small = (val & 1u) ? pr - (val + 1) / 2 : pr + val / 2;
        const uint32_t fn = val == 0 ? pr : (val >= room ? big : small);
        const uint32_t touched = (1u << lo) | (1u << hi) | (1u << i);  // lo, hi < i < 32
        flags |= val != 0 ? touched : 0u;
        bad = bad || !ok;
        f[i] = bad ? 0u : fn;  // after the first out-of-range prediction the row is dropped; keep the chain tame
      }
      uint2* out8 = (uint2*)out;
      if (bad) raise_status(status, VSYN_ST_FLOOR_RANGE, p);
#pragma unroll
      for (uint32_t j = 0; j < UNWRAP_REG_POSTS / 4; ++j) {
        if (j * 4 >= posts_u) break;
        uint32_t w[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t i = 4 * j + e;
          uint32_t v = f[i] * mult;  // hpp:573,578
          if (v > 0x7FFFu || f[i] > 0x7FFFu) v = 0x7FFFu;
          w[e] = i < posts_u ? (bad ? 0x8000u : (v | (((flags >> i) & 1u) << 15))) : 0u;
        }
        out8[j] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
      }
      continue;
    }
#endif
    {  // whole coded row up front (16-byte loads; rows are 8-byte aligned multiples of 4 posts), values parked in LDS
      const uint2* in8 = (const uint2*)(ys + (size_t)gid * stride);
      for (uint32_t j = 0; j * 4 < posts; ++j) {
        const uint2 w = in8[j];
        s_fy[4 * j + 0][t] = w.x & 0xFFFFu;
        if (4 * j + 1 < VSYN_MAX_POSTS) s_fy[4 * j + 1][t] = w.x >> 16;
        if (4 * j + 2 < VSYN_MAX_POSTS) s_fy[4 * j + 2][t] = w.y & 0xFFFFu;
        if (4 * j + 3 < VSYN_MAX_POSTS) s_fy[4 * j + 3][t] = w.y >> 16;
      }
    }
    uint64_t flags_lo = 3;
    uint32_t flag_64 = 0;
    bool bad = false;
    // one post: kq = its constants (neighbour indices, dx, 1/adx). Selects, no branches (the body ran as ~120 instructions
    // of exec-mask juggling per post; at two waves per SIMD every instruction of it costs ~8 cycles).
    auto step = [&](uint32_t i, const uint4 kq) -> bool {
      const uint32_t lo = kq.x & 0xFFFFu, hi = kq.x >> 16;
      const uint32_t val = s_fy[i][t];  // coded value; overwritten below by the amplitude
      const uint32_t ylo = s_fy[lo][t], yhi = s_fy[hi][t];
      const uint32_t dxi = kq.y & 0xFFFFu, adx = kq.y >> 16;
      // predict_post() with its rare integer-divide path taken only if some lane needs it (wave-uniform branch)
      const bool up = yhi >= ylo;
      const uint32_t ady = up ? yhi - ylo : ylo - yhi;
      const uint32_t prod = ady * dxi;
      uint32_t off = (uint32_t)(((float)prod + 0.5f) * __uint_as_float(kq.z));
      if (__any(prod >= (1u << 21))) off = prod >= (1u << 21) ? prod / adx : off;
      const uint32_t predicted = up ? ylo + off : ylo - off;
      const bool ok = predicted <= range;  // hpp:536
      const uint32_t pr = ok ? predicted : 0u;
      const uint32_t high_room = range - pr, low_room = pr;
      const uint32_t room = min(high_room, low_room) * 2;
      const uint32_t big = high_room > low_room ? val - low_room + pr : pr - val + high_room - 1;
      const uint32_t small = (val & 1u) ? pr - (val + 1) / 2 : pr + val / 2;
      const uint32_t f = val == 0 ? pr : (val >= room ? big : small);
      const uint64_t touched = (1ull << lo) | (1ull << hi) | (i < 64 ? 1ull << i : 0ull);  // lo, hi < i <= 64
      flags_lo |= val != 0 ? touched : 0ull;
      flag_64 |= (val != 0 && i >= 64) ? 1u : 0u;
      s_fy[i][t] = f;
      return ok;
    };
    if (fc_uniform) {
      const FloorConst* fcu = (const FloorConst*)(uintptr_t)fc_first;
      const uint32_t posts_u = __builtin_amdgcn_readfirstlane(posts);
      for (uint32_t i = 2; i < posts_u && !bad; ++i) {
        const u32x4 w = *(const_pk)(uintptr_t)&fcu->pk[i];
        if (!step(i, make_uint4(w.x, w.y, w.z, w.w))) bad = true;
      }
    } else {
      for (uint32_t i = 2; i < posts && !bad; ++i)
        if (!step(i, *(const uint4*)&fc->pk[i])) bad = true;
    }
    if (bad) {
      raise_status(status, VSYN_ST_FLOOR_RANGE, p);
      for (uint32_t i = 0; i < posts; ++i) out[i] = 0x8000;  // flat zero curve, all flagged: harmless
      continue;
    }
    uint2* out8 = (uint2*)out;
    for (uint32_t j = 0; j * 4 < posts; ++j) {  // 8-byte stores, 4 posts each (row stride is a multiple of 4)
      uint32_t w[4];
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e) {
        const uint32_t i = 4 * j + e;
        const uint32_t f = i < posts ? s_fy[i < VSYN_MAX_POSTS ? i : 0][t] : 0u;
        uint32_t v = f * mult;  // hpp:573,578
        if (v > 0x7FFFu || f > 0x7FFFu) v = 0x7FFFu;  // wrapped / absurd amplitude: renders >= 256 -> FLOOR_VALUE later
        const uint32_t fl = i < 64 ? (uint32_t)((flags_lo >> i) & 1ull) : (i == 64 ? flag_64 : 0u);
        w[e] = i < posts ? (v | (fl << 15)) : 0u;
      }
      out8[j] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
    }
  }
}

// floor-1 step 2 for ONE bin (hpp:563-589): value of the piecewise-linear integer curve at x.
// render_line's DDA (Utils.hpp:143-183) equals render_point per x (tests/test_oracle_vs_ref.py::test_render_helpers).
// Generic linear walk over the sorted posts; the fused kernels use a precomputed segment table instead.
__device__ __forceinline__ uint32_t floor1_curve_at(const FloorConst* fc, const uint16_t* __restrict__ fyrow, uint32_t x) {
  uint32_t lx = 0, ly = fyrow[fc->sorted_idx[0]] & 0x7FFFu;
  for (uint32_t s = 1; s < fc->posts; ++s) {
    const uint32_t v = fyrow[fc->sorted_idx[s]];
    if (!(v >> 15)) continue;
    const uint32_t hx = fc->xs_sorted[s], hy = v & 0x7FFFu;
    if (x < hx) return render_point_u32(lx, ly, hx, hy, x);
    lx = hx;
    ly = hy;
  }
  return ly;  // flat extension, hpp:583-584
}

// ------------------------------------------------------------------------------------------------
// K2 (staged)  inverse coupling (hpp:1213-1241) + floor product (hpp:1243-1255): one thread per bin,
// all channels of the bin handled by the same thread (couplings chain across channels), in place in `env`.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inverse_couple(float& m, float& a) {  // hpp:1220-1239 as selects (see couple2)
  const float d = m > 0.f ? a : -a;
  const float x = m - d, y = m + d;
  const bool ap = a > 0.f;
  a = ap ? x : m;
  m = ap ? m : y;
}

__global__ void __launch_bounds__(256)
vsyn_spectrum_kernel(const uint8_t* __restrict__ cb, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                     const PktInfo* __restrict__ info, const float* __restrict__ residue, const uint16_t* __restrict__ fy,
                     float* __restrict__ env, uint16_t* __restrict__ curve_tap, DevStatus* __restrict__ status) {
  const ConstHeader* H = hdr_of(cb);
  const uint32_t total = *count;
  for (uint32_t li = blockIdx.x; li < total; li += gridDim.x) {  // grid-stride over the staged work list
  const uint32_t p = list[li] & 0x7FFFFFFFu;
  const PktInfo pi = info[p];
  if (pi.bad) continue;
  const uint32_t n2 = pi.n / 2u, C = H->channels;
  for (uint32_t i = threadIdx.x; i < n2; i += 256) {
  const float* src = residue + pi.res_off;
  float* dst = env + pi.res_off;
  for (uint32_t c = 0; c < C; ++c) dst[(size_t)c * n2 + i] = src[(size_t)c * n2 + i];
  const MapConst* mc = map_of(cb, pi.mapping);
  for (uint32_t k = mc->ncoup; k > 0; --k) {  // reverse order, hpp:1214
    const uint32_t m = mc->coup[2 * (k - 1)], a = mc->coup[2 * (k - 1) + 1];
    float mv = dst[(size_t)m * n2 + i], av = dst[(size_t)a * n2 + i];
    inverse_couple(mv, av);
    dst[(size_t)m * n2 + i] = mv;
    dst[(size_t)a * n2 + i] = av;
  }
  const float* invdb = invdb_of(cb);
  for (uint32_t c = 0; c < C; ++c) {
    if (!((pi.used >> c) & 1u)) continue;
    float f = 0.f;  // propagated-but-undecoded floor: floor_outputs stays zero (hpp:1159, 1169)
    if ((pi.own >> c) & 1u) {
      const FloorConst* fc = floor_of(cb, mc->chfloor[c]);
      const uint32_t v = floor1_curve_at(fc, fy + ((size_t)p * C + c) * H->ys_stride, i);
      if (curve_tap) curve_tap[pi.res_off + (size_t)c * n2 + i] = (uint16_t)min(v, 65535u);  // "floor1 floor", hpp:585
      if (v >= 256u) {  // hpp:587
        raise_status(status, VSYN_ST_FLOOR_VALUE, p);
        f = 0.f;
      } else {
        f = invdb[v];
      }
    }
    dst[(size_t)c * n2 + i] *= f;
  }
  }
  }
}

// ------------------------------------------------------------------------------------------------
// K3 (staged)  IMDCT for any n = 64..8192: one workgroup per block, radix-2 Stockham FFT of n/4 complex
// points in LDS.  Computes what mdct_backward (mdct.cpp:433-527) computes, by a different factorisation:
//   c[k]  = (X[2k] + i X[M-1-2k]) * exp(-i pi (4k+1) / (4M)),  M = n/2, k < n/4
//   Cf    = FFT_{n/4}(c);  d[m] = Cf[m] * exp(-i pi m / M)
//   u[2m] = Re d[m], u[M-1-2m] = -Im d[m]                     (u = DCT-IV of X)
//   y[i]  = u[i+M/2] (i < M/2);  -u[3M/2-1-i] (M/2 <= i < 3M/2);  -u[i-3M/2] (i >= 3M/2)
// Twiddles are computed in double on the host and stored as f32 (like mdct_init, mdct.cpp:101-110).
// Results agree with the reference to ~1e-7 * |output| (gate: 1e-5 abs at |pcm| <= 1, compare-debug-out.py:90).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

__device__ void imdct_block_lds(const uint8_t* __restrict__ cb, int b, uint32_t n, const float* __restrict__ X,
                                float* __restrict__ y, float2* bufA, float2* bufB) {
  const uint32_t M = n / 2, N4 = n / 4, T = blockDim.x, t = threadIdx.x;
  const float2* pre = pre_of(cb, b);
  const float2* post = post_of(cb, b);
  const float2* tw = fft_of(cb, b);
  for (uint32_t k = t; k < N4; k += T) bufA[k] = cmul(make_float2(X[2 * k], X[M - 1 - 2 * k]), pre[k]);
  __syncthreads();
  float2* src = bufA;
  float2* dst = bufB;
  for (uint32_t Ns = 1; Ns < N4; Ns <<= 1) {
    const uint32_t tstride = N4 / (2 * Ns);
    for (uint32_t j = t; j < N4 / 2; j += T) {
      const uint32_t k = j & (Ns - 1);
      const float2 v0 = src[j];
      const float2 v1 = cmul(src[j + N4 / 2], tw[k * tstride]);
      const uint32_t o = ((j - k) << 1) + k;
      dst[o] = make_float2(v0.x + v1.x, v0.y + v1.y);
      dst[o + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* tmp = src;
    src = dst;
    dst = tmp;
  }
  float* u = (float*)dst;  // M floats == N4 float2
  for (uint32_t m = t; m < N4; m += T) {
    const float2 d = cmul(src[m], post[m]);
    u[2 * m] = d.x;
    u[M - 1 - 2 * m] = -d.y;
  }
  __syncthreads();
  for (uint32_t i = t; i < n; i += T) {
    float v;
    if (i < M / 2) v = u[i + M / 2];
    else if (i < 3 * M / 2) v = -u[3 * M / 2 - 1 - i];
    else v = -u[i - 3 * M / 2];
    y[i] = v;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256)
vsyn_imdct_staged_kernel(const uint8_t* __restrict__ cb, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                         const PktInfo* __restrict__ info, const float* __restrict__ env, float* __restrict__ blk) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const ConstHeader* H = hdr_of(cb);
  const uint32_t C = H->channels, total = *count * C;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    const uint32_t p = list[w / C] & 0x7FFFFFFFu, c = w % C;
    const PktInfo pi = info[p];
    if (pi.bad) continue;
    const uint32_t n = pi.n;
    float2* bufA = (float2*)lds_raw;
    float2* bufB = bufA + n / 4;
    imdct_block_lds(cb, pi.lng, n, env + pi.res_off + (size_t)c * (n / 2), blk + 2 * pi.res_off + (size_t)c * n, bufA, bufB);
  }
}

// plain [count][n/2] -> [count][n] (BASELINE config 2), generic version
__global__ void __launch_bounds__(256)
vsyn_imdct_plain_kernel(const uint8_t* __restrict__ cb, int b, uint32_t n, uint32_t count, const float* __restrict__ in,
                        float* __restrict__ out) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  float2* bufA = (float2*)lds_raw;
  float2* bufB = bufA + n / 4;
  for (uint32_t blk = blockIdx.x; blk < count; blk += gridDim.x)
    imdct_block_lds(cb, b, n, in + (size_t)blk * (n / 2), out + (size_t)blk * n, bufA, bufB);
}

// ------------------------------------------------------------------------------------------------
// K4 (staged)  window + overlap-add + PCM hand-off (hpp:1008-1059) as a gather: emitted sample s of packet q
//   = fl( fl(prev[n_prev/2 + s] * w_prev[n_prev/2 + s]) + fl(cur[j] * w_cur[j]) ),  j = n_cur/2 - L + s,
// each term present only where its block covers the sample (tests/test_oracle_vs_ref.py pins this against
// the reference's sliding buffer).  Mul and add are rounded separately, as `buf += pcm*window` is upstream.
// The last packet of a segment also stores its windowed right half as the stream's carry.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vsyn_overlap_kernel(const uint8_t* __restrict__ cb, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                    const PktInfo* __restrict__ info, const vsyn_segment* __restrict__ segs, const SegInfo* __restrict__ sinfo,
                    const uint32_t* __restrict__ seg_of_pkt, const float* __restrict__ blk, float* __restrict__ pcm,
                    uint64_t plane_stride, float* __restrict__ carry) {
  const ConstHeader* H = hdr_of(cb);
  const uint32_t C = H->channels, total = *count * C;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
  const uint32_t entry = list[w / C], c = w % C;
  if (!(entry >> 31)) continue;  // halo entry: block only
  const uint32_t p = entry & 0x7FFFFFFFu;
  const uint32_t g = seg_of_pkt[p];
  const PktInfo pi = info[p];
  if (pi.bad) continue;
  for (uint32_t s = threadIdx.x; s < H->bs[1] / 2; s += 256) {
  const vsyn_segment sg = segs[g];
  const SegInfo si = sinfo[g];
  const uint32_t q = p - sg.first_packet, n = pi.n, half1 = H->bs[1] / 2;
  const float* cur = blk + 2 * pi.res_off + (size_t)c * n;
  const float* wc = win_of(cb, pi.lng, pi.widx);
  const size_t carry_half = (size_t)H->max_streams * C * half1;

  // prev block (or carry)
  uint32_t n_prev = 0;
  const float* prev = nullptr;
  const float* wp = nullptr;
  const float* cin = nullptr;
  if (q > 0) {
    const PktInfo pp = info[p - 1];
    if (!pp.bad) {
      n_prev = pp.n;
      prev = blk + 2 * pp.res_off + (size_t)c * n_prev;
      wp = win_of(cb, pp.lng, pp.widx);
    }
  } else if (si.has_carry) {
    n_prev = si.carry_n;
    cin = carry + si.parity_in * carry_half + ((size_t)sg.stream * C + c) * half1;
  }
  if (s < pi.emit && n_prev) {
    const uint32_t L = n_prev / 4 + n / 4;
    float acc = 0.f;
    const uint32_t ip = n_prev / 2 + s;
    if (ip < n_prev) acc = acc + (cin ? cin[s] : prev[ip] * wp[ip]);
    const int32_t j = (int32_t)(n / 2) - (int32_t)L + (int32_t)s;
    if (j >= 0) acc = acc + cur[j] * wc[j];
    pcm[((size_t)g * C + c) * plane_stride + pi.out_pos + s] = acc;
  }
  if (q == sg.num_packets - 1 && s < n / 2) {
    float* cout = carry + (si.parity_in ^ 1u) * carry_half + ((size_t)sg.stream * C + c) * half1;
    cout[s] = cur[n / 2 + s] * wc[n / 2 + s];
  }
  }
  }
}

