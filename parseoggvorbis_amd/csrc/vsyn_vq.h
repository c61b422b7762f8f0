// vsyn_vq.h — residue VQ stage: the data-parallel half of VorbisResidue::decode on the device.
//
// Reference: src/ParseOggVorbis.hpp:670-762.  The host keeps the bit-serial half (one classification word per partition
// group, one codebook entry number per vector) and ships those numbers; this kernel does what the reference does with
// them — look the entry's value vector up (lookup_table_, hpp:367-374) and add it into the residue vector, pass after
// pass (hpp:737-753), then de-interleave format 2 (hpp:687-693) — and writes "after_residue" in the packing the
// synthesis kernels read.
//
// One wavefront per packet (as many single-wave workgroups as are resident at once walk the packets; the per-residue tables a
// wave stages in LDS survive from packet to packet while the residue in use stays the same). Per packet:
//   descriptors PktInfo, vsyn_vq_packet, the mapping's submap records and the residue header are wave-uniform: scalar loads.
//   entries     the packet's entry numbers (<= VQ_ENT_CAP) go to LDS with direct-to-LDS loads (global_load_lds_dwordx4, no
//               registers), issued first and waited for only before the accumulate phase.
//   scan        how many entries each (pass, partition, vector) slot consumes, and the exclusive scan of that in decode order
//               -> where each slot's first entry sits.  Lane = one (partition, vector) pair with its 8 passes; two passes share
//               a 32-bit word, so four DPP scans give the eight per-pass prefix sums.  The classification bytes are read once,
//               here, and kept in LDS.
//   accumulate  every ELEMENT belongs to exactly one partition and receives at most one value per pass, so a lane that owns
//               VQ_GROUP consecutive elements of a partition adds their up-to-8 contributions in pass order in registers —
//               the same sequence of f32 additions as the reference —, then stores each element once (format 2 de-interleaved,
//               16-byte stores).  No atomics.  Only passes in which the residue has a codebook at all are visited (a scalar loop
//               over the set bits); every power-of-two vector length >= 2 goes through one route (four 8-byte reads per group:
//               a pair of consecutive elements never straddles a vector), length 1 through eight 4-byte reads; which routes a
//               residue can need at all is a setup fact (VqResidue::kinds). Format 0, other lengths and partition tails go
//               through one out-of-line per-element routine.
//   tables      from the pool in global memory through a buffer descriptor (32-bit offsets, hardware bounds check: an out-of-range
//               entry number raises the status and can only read other table data or zeros, never outside the pool) — or, for a
//               setup whose tables fit VQ_IMG_MAX_BYTES, from one copy per workgroup in LDS (vsyn_residue_vq_kernel<true>:
//               workgroups of several waves, one barrier after staging the copy, then every wave on its own as above).
//   zero fill   the bins outside the partitions are two runs of consecutive bins per channel.
// A single wave needs no barriers: its LDS operations execute in order.
// Measured (bench.py --workload config3_vq, 65 536 stereo long packets, kernel time; DESIGN.md f-1 has the full record): tables in LDS
// (synthetic books, 15 KB) 0.225-0.245 ms; the stereo fixture's books (243 KB, global memory) 0.19 ms. Round 1's form of the
// accumulate phase (one exec-masked variant per vector length, lane-mask error tracking) 0.285 ms; its predecessor (entries and
// classifications read from global memory inside the accumulate loop, per-pass LDS book records, zero fill element by element)
// 0.48 ms; earlier alternatives: workgroup of 128/256 threads per packet 0.67-0.88 ms; separate scan + accumulate kernels (thread =
// element group, no per-packet loop) 0.69 ms; thread = one entry of a pass with f32 accumulators in LDS 0.69-0.77 ms; forcing 8
// waves/SIMD (64 VGPRs, spills) 0.53 ms. The kernel is bound by its instruction count (~2900 per packet, ~1200 of them on the CU's
// one scalar unit), not by memory: per-wave time grows linearly with the waves per CU. Measured and dropped in round 2: a two-stage
// packet pipeline with double-buffered LDS, descriptors one packet ahead by vector load, several groups per lane before any store,
// lattice books as half vectors in LDS (-DVQ_STAMPS and profiles/r02_experiments/vq_* hold the phase times and A/B runs).
#pragma once
#include "vsyn_device.h"

#define VQ_THREADS 64      /* one wavefront per packet: every hand-off is wave-local, no barriers */
#define VQ_GROUP 8
#define VQ_MAX_SLOTS 8192  /* (pass, partition, channel) slots of one submap vector; vsyn_attach_vq enforces it. The kernel's
                              dynamic LDS is sized to the largest count the attached setup can produce (fixtures: 400) */
#define VQ_IMG_MAX_BYTES 49152u   /* LDS table image (one copy per workgroup): setups whose tables exceed this keep them in global memory */
#define VQ_ENT_CAP 2048    /* a packet with at most this many entry numbers has them staged in LDS by direct-to-LDS loads
                              (stereo long blocks of the fixture: ~1400); longer ones read them from global memory */

struct VqBook {       // 16 bytes
  uint32_t dims, entries;
  uint32_t table_off;  // float index into the pool; 0xFFFFFFFF: no value table
  uint32_t lds_off;    // float index into the LDS table image (VqHeader::off_img); 0xFFFFFFFF: not in the image
};
struct VqResidue {
  uint32_t type, begin, end, psize, nclass, classwords;
  uint32_t pass_mask;  // bit k: some class has a codebook in pass k
  uint32_t kinds;      // VQ_KIND_*: the routes of the accumulate phase this residue can need
  int16_t books[64 * 8];
};
struct VqMap {        // all fields the per-packet code reads are dwords: they come in through the scalar cache
  uint32_t num_submaps;
  uint32_t sub_nch[16];    // channels muxed to the submap (0: the submap is skipped, hpp:1191-1199)
  uint32_t sub_rid[16];    // residue of the submap
  uint32_t sub_chan4[16];  // its first four channel numbers, one per byte
  uint8_t mux[VSYN_MAX_CHANNELS];
};
struct VqHeader {
  uint32_t num_books, num_residues, num_maps, pad;
  uint32_t off_books, off_residues, off_maps, off_pool;  // byte offsets from the block base
  uint32_t pool_floats, total_bytes;
  uint32_t max_slots, max_classes;  // largest slot count of any submap vector at blocksize1 / class count of any residue: LDS sizing
  uint32_t off_img, img_floats;     // the value tables in the form the LDS-table kernel keeps per workgroup (0 floats: setup not eligible)
};

// Dynamic LDS of one workgroup (= one wavefront), shared between the host (launch) and the kernel (carving):
//   cp    [max_classes * 8] x 8 B   per (class, pass): value table + entry count / vector length facts
//   pp    [max_classes * 8] x u16   per (class, pass): vectors per partition (0: no codebook) — a row is the scan's four packed counts
//   start [npj_max] x 16 B          per (partition, vector): first entry of its slot in each pass, passes 2k / 2k+1 packed 16+16
//   ent   [VQ_ENT_CAP + 8] x u16    the packet's entry numbers
//   vm    [max_classes] x u8        passes of the class that have a codebook
//   cls   [npj_max] x u8            classification per (partition, vector), 0xFF: takes no part
//   chan  [VSYN_MAX_CHANNELS] x u8  channels of the submap
static inline __host__ __device__ uint32_t vq_align16(uint32_t x) { return (x + 15u) & ~15u; }
static inline __host__ __device__ uint32_t vq_lds_layout(uint32_t max_slots, uint32_t max_classes, uint32_t* off /* [7] */) {
  const uint32_t npj = max_slots / 8u;
  uint32_t o = 0;
  off[0] = o; o += max_classes * 64u;
  off[1] = o; o += max_classes * 16u;
  off[2] = o; o += npj * 16u;
  off[3] = o; o += vq_align16((VQ_ENT_CAP + 8u) * 2u);
  off[4] = o; o += vq_align16(max_classes);
  off[5] = o; o += vq_align16(npj);
  off[6] = o; o += VSYN_MAX_CHANNELS;
  return o;
}

#ifdef __HIPCC__
#ifdef VQ_STAMPS  /* diagnostic build: cycles per phase of the packet walk, summed per wave (printed by vsyn_destroy) */
#define VQ_NSTAMPS 8
__device__ unsigned long long g_vq_stamps[8192][VQ_NSTAMPS];
#define VQ_STAMP(i)                                                                \
  do {                                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
    vq_st[i] += now_ - vq_t;                                                       \
    vq_t = now_;                                                                   \
  } while (0)
#else
#define VQ_STAMP(i) do { } while (0)
#endif
struct VqCp {  // per (class, pass)
  uint32_t tab;  // float index of the value table: in the pool, or (LDS-table kernel) in the workgroup's table image
  uint32_t inf;  // bits 0..15 entry count - 1; 16..19 log2(vector length) if VQ_CP_FAST; VQ_CP_* flags
};
#define VQ_CP_FAST 0x01000000u  /* 8.6.4 layout (format 1 / 2) and a power-of-two vector length */
typedef uint32_t vq_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t vq_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t vq_u32x2 __attribute__((ext_vector_type(2)));
#define VQ_K __attribute__((address_space(4)))  /* wave-uniform read-only data: scalar loads */

// inclusive prefix sum across the 64 lanes with DPP row shifts / broadcasts (no LDS traffic, 6 VALU steps)
__device__ __forceinline__ uint32_t vq_wave_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2, 3
  return v;
}

// The general element -> (vector, component) mapping, out of line (cold): format 0 (8.6.3, hpp:738-746: element w <- vector
// w % step, component w / step), vector lengths that are not a power of two, partition tails. Returns the float index of the
// element's value in the pool: 0 (the zero vector) beyond the partition, 0xFFFFFFFF for an entry number out of range.
__device__ __noinline__ uint32_t vq_general_index(uint32_t type, uint32_t psize, uint32_t ww, uint32_t step, uint32_t nent1, uint32_t tab,
                                                  uint32_t e0, bool ent_lds, const uint16_t* s_ent, const uint16_t* __restrict__ eg) {
  if (ww >= psize) return 0u;
  const uint32_t dims = psize / step;
  uint32_t ei, li;
  if (type == 0u) {
    li = ww / step;
    ei = ww - li * step;
  } else {
    ei = ww / dims;
    li = ww - ei * dims;
  }
  const uint32_t en = ent_lds ? (uint32_t)s_ent[e0 + ei] : (uint32_t)eg[e0 + ei];
  return en <= nent1 ? tab + en * dims + li : 0xFFFFFFFFu;
}

// Everything one submap vector needs in the accumulate phase (wave-uniform)
struct VqSubCtx {
  uint32_t type, psize, vch, nch, npj, parts, lim_begin, len, n2, gpp, pass_mask, kinds;
  uint32_t chan0, chan1;
  uint32_t ent_base;        // first entry of the submap relative to the packet's first entry
  uint32_t poff_lanes;      // lane k < 8: first entry of pass k relative to ent_base (read with readlane)
  uint32_t ent_shift;       // ENT_LDS: index of the packet's first entry inside s_ent
};

// The accumulate phase: a lane owns VQ_GROUP consecutive elements of one partition of one vector and adds their up-to-8
// contributions in pass order (hpp:711) in registers. ENT_LDS: the packet's entries sit in s_ent, otherwise in eg[].
// A lane works on VQ_BATCH groups (64 groups apart) before it stores any of them: vector-memory operations complete in order, so a
// table load issued behind a store returns only after that store has been acknowledged — batching leaves one such boundary per
// 256 groups instead of one per 64.
// What the kernel pays for here is the scalar unit (one per CU, ~1400 scalar instructions per packet before this form: exec-mask
// bookkeeping of the per-lane vector-length variants), so: which variants a residue can need at all is a setup fact
// (VqResidue::kinds, wave-uniform branches), lengths >= 4 share one route, and "entry out of range" is tracked as a sign bit in a
// vector register instead of a lane mask.
#ifndef VQ_BATCH
#define VQ_BATCH 1
#endif
#define VQ_KIND_V4 1u   /* 8.6.4 layout, vector length a power of two >= 4 */
#define VQ_KIND_V2 2u   /* ... length 2 */
#define VQ_KIND_V1 4u   /* ... length 1 */
#define VQ_KIND_GEN 8u  /* format 0, other lengths, or a partition size that is not a multiple of VQ_GROUP */
template <bool ENT_LDS, bool TLDS>
__device__ __forceinline__ bool vq_accumulate(const VqSubCtx& X, const uint32_t lane, const bool bad, const VqCp* s_cp,
                                               const uint16_t* s_pp, const vq_u32x4* s_start, const uint16_t* s_ent, const uint8_t* s_vm,
                                               const uint8_t* s_cls, const uint8_t* s_chan, const uint16_t* __restrict__ eg,
                                               const __amdgpu_buffer_rsrc_t pool, const float* s_tab, float* __restrict__ out) {
  auto entry = [&](uint32_t i) -> uint32_t { return ENT_LDS ? (uint32_t)s_ent[X.ent_shift + i] : (uint32_t)eg[i]; };
  typedef float vq_f4 __attribute__((ext_vector_type(4)));
  typedef float vq_f2 __attribute__((ext_vector_type(2)));
  const uint32_t groups = X.npj * X.gpp;
  const bool full8 = (X.psize & (VQ_GROUP - 1u)) == 0u;
  const uint32_t q64 = VQ_THREADS / X.gpp, r64 = VQ_THREADS - q64 * X.gpp;
  uint32_t pj_run = lane / X.gpp, wq_run = lane - pj_run * X.gpp;
  uint32_t over = 0;  // sign bit: some entry number was beyond its book
  bool bad_general = false;
  for (uint32_t base = 0; base < groups; base += VQ_THREADS * VQ_BATCH) {
    float acc[VQ_BATCH][VQ_GROUP];
    uint32_t pjs[VQ_BATCH], w0s[VQ_BATCH];
#pragma unroll
    for (int u = 0; u < VQ_BATCH; ++u) {
      pjs[u] = pj_run;
      w0s[u] = wq_run * VQ_GROUP;
      pj_run += q64;
      wq_run += r64;
      if (wq_run >= X.gpp) {
        wq_run -= X.gpp;
        ++pj_run;
      }
    }
    uint32_t cs[VQ_BATCH], vms[VQ_BATCH];
#pragma unroll
    for (int u = 0; u < VQ_BATCH; ++u) {
#pragma unroll
      for (int k = 0; k < VQ_GROUP; ++k) acc[u][k] = 0.f;
      const bool in_range = base + (uint32_t)u * VQ_THREADS + lane < groups;
      const uint32_t c = in_range && !bad ? (uint32_t)s_cls[pjs[u]] : 0xFFu;
      cs[u] = c;
      vms[u] = c != 0xFFu ? (uint32_t)s_vm[c] : 0u;
    }
    // pass order = order of the additions (hpp:711); only passes in which some class of this residue has a codebook (wave-uniform).
    // Within a pass the lane's VQ_BATCH groups are independent: all their table loads are issued before the first one is consumed.
#pragma unroll 1
    for (uint32_t pm = X.pass_mask; pm; pm &= pm - 1u) {
      const uint32_t ps = (uint32_t)__builtin_ctz(pm);
      const uint32_t pass_base = X.ent_base + (uint32_t)__builtin_amdgcn_readlane((int)X.poff_lanes, (int)ps);
      float add[VQ_BATCH][VQ_GROUP];
#pragma unroll
      for (int u = 0; u < VQ_BATCH; ++u) {
#pragma unroll
        for (int k = 0; k < VQ_GROUP; ++k) add[u][k] = 0.f;
        if (base + (uint32_t)u * VQ_THREADS >= groups) continue;  // wave-uniform
        if (!((vms[u] >> ps) & 1u)) continue;
        const uint32_t pj = pjs[u], w0 = w0s[u], c = cs[u];
        const uint32_t cnt_el = full8 ? (uint32_t)VQ_GROUP : min((uint32_t)VQ_GROUP, X.psize - w0);
        const VqCp cp = s_cp[c * 8u + ps];
        const uint32_t stw = ((const uint32_t*)&s_start[pj])[ps >> 1];
        const uint32_t e0 = pass_base + ((stw >> (16u * (ps & 1u))) & 0xFFFFu);
        const uint32_t nent1 = cp.inf & 0xFFFFu, sh = (cp.inf >> 16) & 15u;
        const bool fast = (X.kinds & VQ_KIND_GEN) == 0u || ((cp.inf & VQ_CP_FAST) && cnt_el == VQ_GROUP);
        // The value tables are read through a buffer descriptor over the pool: 32-bit offsets (no 64-bit address arithmetic) and
        // hardware bounds checking — an out-of-range entry number can only read other table data or, past the pool, zeros; it is
        // flagged (the packet's values are unspecified then), never dereferenced outside the pool.
        if ((X.kinds & (VQ_KIND_V4 | VQ_KIND_V2)) && fast && sh >= 1u) {
          // every length >= 2 the same way: four 8-byte reads (a pair of consecutive elements never straddles a vector)
          const uint32_t dmask = (1u << sh) - 1u;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint32_t w = w0 + 2u * (uint32_t)m;
            const uint32_t en = entry(e0 + (w >> sh));
            over |= nent1 - en;
            const uint32_t idx = cp.tab + (en << sh) + (w & dmask);
            vq_f2 a;
            if (TLDS) a = *(const vq_f2*)&s_tab[idx];
            else a = __builtin_bit_cast(vq_f2, __builtin_amdgcn_raw_buffer_load_b64(pool, idx * 4u, 0, 0));
            add[u][2 * m] = a.x;
            add[u][2 * m + 1] = a.y;
          }
        }
        if ((X.kinds & VQ_KIND_V1) && fast && sh == 0u) {  // eight scalars
          uint32_t en[VQ_GROUP];
#pragma unroll
          for (int k = 0; k < VQ_GROUP; ++k) en[k] = entry(e0 + w0 + k);
#pragma unroll
          for (int k = 0; k < VQ_GROUP; ++k) {
            over |= nent1 - en[k];
            if (TLDS) add[u][k] = s_tab[cp.tab + en[k]];
            else add[u][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pool, (cp.tab + en[k]) * 4u, 0, 0));  // (the builtin returns the bits)
          }
        }
        if ((X.kinds & VQ_KIND_GEN) && !fast) {
          const uint32_t step = s_pp[c * 8u + ps];  // vectors per partition
#pragma unroll
          for (int k = 0; k < VQ_GROUP; ++k) {
            const uint32_t idx = vq_general_index(X.type, X.psize, w0 + k, step, nent1, cp.tab, e0, ENT_LDS, s_ent + X.ent_shift, eg);
            bad_general |= idx == 0xFFFFFFFFu;
            add[u][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pool, (idx == 0xFFFFFFFFu ? 0u : idx) * 4u, 0, 0));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < VQ_BATCH; ++u)
#pragma unroll
        for (int k = 0; k < VQ_GROUP; ++k) acc[u][k] += add[u][k];
    }
    // store (de-interleaving format 2: element e of the interleaved vector is bin e / nch of channel e % nch, hpp:690-692)
#pragma unroll
    for (int u = 0; u < VQ_BATCH; ++u) {
      if (base + (uint32_t)u * VQ_THREADS + lane >= groups) continue;
      const uint32_t pj = pjs[u], w0 = w0s[u];
      uint32_t pc = pj, j = 0;
      if (X.vch != 1u) {
        pc = pj / X.vch;
        j = pj - pc * X.vch;
      }
      const uint32_t cnt_el = full8 ? (uint32_t)VQ_GROUP : min((uint32_t)VQ_GROUP, X.psize - w0);
      const uint32_t e_first = X.lim_begin + pc * X.psize + w0;
      const bool fmt2 = X.type == 2u;
      if (fmt2 && X.nch == 2u && cnt_el == VQ_GROUP && !(e_first & 1u)) {  // stereo: 4 consecutive bins per channel
        float* o0 = out + (size_t)X.chan0 * X.n2 + (e_first >> 1);
        float* o1 = out + (size_t)X.chan1 * X.n2 + (e_first >> 1);
        if (((uintptr_t)o0 & 15u) == 0 && ((uintptr_t)o1 & 15u) == 0) {
          *(float4*)o0 = make_float4(acc[u][0], acc[u][2], acc[u][4], acc[u][6]);
          *(float4*)o1 = make_float4(acc[u][1], acc[u][3], acc[u][5], acc[u][7]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o0[k] = acc[u][2 * k];
            o1[k] = acc[u][2 * k + 1];
          }
        }
      } else if (!fmt2 && cnt_el == VQ_GROUP && (((uintptr_t)(out + (size_t)s_chan[j] * X.n2 + e_first)) & 15u) == 0) {
        float4* o = (float4*)(out + (size_t)s_chan[j] * X.n2 + e_first);
        o[0] = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
        o[1] = make_float4(acc[u][4], acc[u][5], acc[u][6], acc[u][7]);
      } else {
#pragma unroll
        for (int k = 0; k < VQ_GROUP; ++k)
          if ((uint32_t)k < cnt_el) {
            const uint32_t e = e_first + k;
            if (fmt2) out[(size_t)s_chan[e % X.nch] * X.n2 + e / X.nch] = acc[u][k];
            else out[(size_t)s_chan[j] * X.n2 + e] = acc[u][k];
          }
      }
    }
  }
  return bad_general || (int32_t)over < 0;
}

// TLDS = false: grid of as many single-wave workgroups as are resident at once, each walking packets p = blockIdx.x, + gridDim.x,
// ...; value tables read from the pool in global memory (L1/L2 hits).
// TLDS = true (setups whose tables fit, see vq_build_block): workgroups of several waves that share one copy of the value tables in
// LDS, staged once at the start; after that barrier each wave walks its packets on its own exactly as above. The accumulate phase
// then has no vector-memory LOAD left in it — which matters twice: a table row costs an LDS round trip instead of an L1/L2 one, and
// (vector-memory operations complete in order) no load ever queues behind the previous group's stores.
// The staged per-residue records survive from packet to packet while the residue in use stays the same.
template <bool TLDS>
__global__ void __launch_bounds__(TLDS ? 1024 : VQ_THREADS) vsyn_residue_vq_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ vqb, uint32_t P,
                                                                     const PktInfo* __restrict__ info, const vsyn_vq_packet* __restrict__ vqp,
                                                                     const uint8_t* __restrict__ cls_all, uint64_t num_cls,
                                                                     const uint16_t* __restrict__ ent_all, uint64_t num_ent,
                                                                     float* __restrict__ residue, DevStatus* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_raw0[];
  const ConstHeader* H = hdr_of(cb);
  const VQ_K VqHeader* VH = (const VQ_K VqHeader*)(uintptr_t)vqb;
  const uint32_t C = H->channels, lane = threadIdx.x & (VQ_THREADS - 1u);
  const uint32_t wave = TLDS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / VQ_THREADS)) : 0u;
  const uint32_t waves = TLDS ? blockDim.x / VQ_THREADS : 1u;
  uint32_t lo[7];
  const uint32_t wave_bytes = vq_lds_layout(VH->max_slots, VH->max_classes, lo);
  const uint32_t tab_bytes = TLDS ? vq_align16(VH->img_floats * 4u) : 0u;
  const float* s_tab = (const float*)s_raw0;
  if (TLDS) {  // the table image, once per workgroup
    const vq_u32x4* src = (const vq_u32x4*)(vqb + VH->off_img);
    for (uint32_t i = threadIdx.x; i < tab_bytes / 16u; i += blockDim.x) ((vq_u32x4*)s_raw0)[i] = src[i];
    __syncthreads();
  }
  uint8_t* const s_raw = s_raw0 + tab_bytes + wave * wave_bytes;
  VqCp* s_cp = (VqCp*)(s_raw + lo[0]);
  uint16_t* s_pp = (uint16_t*)(s_raw + lo[1]);
  vq_u32x4* s_start = (vq_u32x4*)(s_raw + lo[2]);
  uint16_t* s_ent = (uint16_t*)(s_raw + lo[3]);
  uint8_t* s_vm = s_raw + lo[4];
  uint8_t* s_cls = s_raw + lo[5];
  uint8_t* s_chan = s_raw + lo[6];
  const VqBook* books = (const VqBook*)(vqb + VH->off_books);
  // the pool of value tables behind a buffer descriptor (raw buffer, no stride: offsets are bounds-checked against its byte size)
  const __amdgpu_buffer_rsrc_t pool = __builtin_amdgcn_make_buffer_rsrc((void*)(vqb + VH->off_pool), 0, VH->pool_floats * 4u, 0x00020000);
  const uint32_t max_slots = VH->max_slots;
  uint32_t staged_residue = 0xFFFFFFFFu, staged_map = 0xFFFFFFFFu, staged_sub = 0xFFFFFFFFu;
#ifdef VQ_STAMPS
  unsigned long long vq_st[VQ_NSTAMPS] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long vq_t = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t p = blockIdx.x * waves + wave; p < P; p += gridDim.x * waves) {
    // wave-uniform descriptors through the scalar cache (the layout kernel / the copies that wrote them finished before this launch).
    // (Requesting them a packet ahead does not pay: scalar loads return out of order, so the next LDS wait waits for them too, and
    // a vector load behind the previous packet's stores completes only after those stores — measured slower.)
    const vq_u32x8 iw = *(const VQ_K vq_u32x8*)(uintptr_t)(info + p);  // PktInfo
    const uint32_t pn = iw[6] & 0xFFFFu, pmapping = iw[7] & 0xFFu, pbad = (iw[7] >> 8) & 0xFFu, pused = iw[4];
    if (pbad || pn == 0) continue;  // flagged by the layout kernel; nothing downstream reads this packet's residue
    VQ_STAMP(0);  // PktInfo arrived
    const uint32_t n2 = pn / 2u;
    const vq_u32x4 vw = *(const VQ_K vq_u32x4*)(uintptr_t)(vqp + p);   // vsyn_vq_packet
    const uint64_t entry_off = (uint64_t)vw[0] | ((uint64_t)vw[1] << 32);
    const uint32_t pkt_entries = vw[2];
    const VQ_K VqMap* mp = (const VQ_K VqMap*)(uintptr_t)(vqb + VH->off_maps) + pmapping;
    bool bad = entry_off + pkt_entries > num_ent;
    const uint16_t* ent = ent_all + entry_off;
    uint32_t cls_cur = vw[3], ent_cur = 0;  // cursors over the packet's classification bytes / entries, submap after submap
    float* const out = residue + ((uint64_t)iw[0] | ((uint64_t)iw[1] << 32));

    // entries -> LDS, 16 bytes per lane and instruction straight into s_ent (no registers; waited for before the accumulate phase).
    // The 16-byte pieces are aligned, so the first and the last one stay inside the pages that hold the packet's entries.
    const bool ent_lds = !bad && pkt_entries != 0u && pkt_entries <= (uint32_t)VQ_ENT_CAP;
    uint32_t ent_shift = 0;
    if (ent_lds) {
      const uintptr_t a0 = (uintptr_t)ent, a16 = a0 & ~(uintptr_t)15;
      ent_shift = (uint32_t)(a0 - a16) / 2u;
      const uint32_t pieces = (ent_shift + pkt_entries + 7u) / 8u;
      for (uint32_t c0 = 0; c0 < pieces; c0 += VQ_THREADS)
        if (c0 + lane < pieces)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a16 + (uintptr_t)(c0 + lane) * 16u),
                                           (__attribute__((address_space(3))) void*)(s_ent + c0 * 8u), 16, 0, 0);
    }

    VQ_STAMP(1);  // vsyn_vq_packet arrived, entry staging issued
    const uint32_t num_submaps = mp->num_submaps;
    for (uint32_t s = 0; s < num_submaps; ++s) {
      const uint32_t nch = mp->sub_nch[s];
      if (nch == 0) continue;
      const uint32_t rid = mp->sub_rid[s], chan4 = mp->sub_chan4[s];
      const VqResidue* r = (const VqResidue*)(vqb + VH->off_residues) + rid;
      const vq_u32x8 rw = *(const VQ_K vq_u32x8*)(uintptr_t)r;  // type, begin, end, psize, nclass, classwords, pass_mask
      VqSubCtx X;
      X.type = rw[0];
      X.psize = rw[3];
      X.pass_mask = rw[6];
      X.kinds = rw[7];
      const uint32_t nclass = rw[4];
      const bool fmt2 = X.type == 2u;
      X.nch = nch;
      X.n2 = n2;
      X.vch = fmt2 ? 1u : nch;             // vectors decoded side by side (format 2: one interleaved vector)
      X.len = fmt2 ? nch * n2 : n2;        // hpp:687-688
      const uint32_t lim_begin = min(rw[1], X.len), lim_end = min(rw[2], X.len);  // hpp:696-698
      X.lim_begin = lim_begin;
      X.parts = lim_end > lim_begin ? (lim_end - lim_begin) / X.psize : 0u;
      X.npj = X.parts * X.vch;             // (partition, vector) pairs; slot = (pass, pj), pj = pc * vch + j
      X.gpp = (X.psize + VQ_GROUP - 1) / VQ_GROUP;  // element groups per partition
      X.chan0 = chan4 & 0xFFu;
      X.chan1 = (chan4 >> 8) & 0xFFu;
      X.ent_base = ent_cur;
      X.ent_shift = ent_shift;
      asm volatile("" ::: "memory");  // (single wave: LDS operations execute in order; this only pins the compiler)
      if (staged_map != pmapping || staged_sub != s) {
        if (lane == 0) {
          uint32_t k = 0;
          for (uint32_t ch = 0; ch < C; ++ch)
            if (mp->mux[ch] == s) s_chan[k++] = (uint8_t)ch;
        }
        staged_map = pmapping;
        staged_sub = s;
      }
      if (staged_residue != rid) {
        for (uint32_t i = lane; i < nclass * 8u; i += VQ_THREADS) {
          const int book = r->books[i];
          VqCp e;
          e.tab = 0;
          e.inf = 0;
          uint32_t per_part = 0;
          if (book >= 0) {
            const VqBook bk = books[book];
            per_part = X.psize / bk.dims;
            e.tab = TLDS ? bk.lds_off : bk.table_off;
            e.inf = (bk.entries - 1u) & 0xFFFFu;
            if (X.type != 0u && (bk.dims & (bk.dims - 1u)) == 0u) e.inf |= VQ_CP_FAST | ((31u - (uint32_t)__clz((int)bk.dims)) << 16);
          }
          s_cp[i] = e;
          s_pp[i] = (uint16_t)per_part;
        }
        for (uint32_t c = lane; c < nclass; c += VQ_THREADS) {
          uint32_t m = 0;
          for (uint32_t ps = 0; ps < 8; ++ps) m |= (r->books[c * 8u + ps] >= 0 ? 1u : 0u) << ps;
          s_vm[c] = (uint8_t)m;
        }
        staged_residue = rid;
      }
      VQ_STAMP(2);  // submap / residue facts
      const uint8_t* cls = cls_all + cls_cur;
      if ((uint64_t)cls_cur + (uint64_t)X.npj > num_cls || 8u * X.npj > max_slots) bad = true;
      asm volatile("" ::: "memory");
      uint32_t vused = 0;  // bit j: vector j takes part (format 2: always, hpp:685-694; else floor_output_used, hpp:729)
      if (fmt2) vused = 1u;
      else
        for (uint32_t j = 0; j < X.vch; ++j)
          if ((pused >> s_chan[j]) & 1u) vused |= 1u << j;

      // ---- entries per slot + exclusive scan in decode order (pass, partition, vector) ----
      // A lane takes one (partition, vector) pair and its 8 passes; two passes share a 32-bit word (a pass holds at most
      // len <= 65535 entries: 16 bits each), so four DPP scans cover the eight per-pass prefix sums. A row of s_pp is
      // exactly the lane's four packed counts.
      uint32_t carry[4] = {0, 0, 0, 0};
      bool bad_cls = false;
      for (uint32_t base = 0; base < X.npj && !bad; base += VQ_THREADS) {
        const uint32_t pj = base + lane;
        vq_u32x4 cnt = {0, 0, 0, 0};
        if (pj < X.npj) {
          uint32_t pc = pj, j = 0;
          if (X.vch != 1u) {
            pc = pj / X.vch;
            j = pj - pc * X.vch;
          }
          uint32_t c = 0xFFu;
          if ((vused >> j) & 1u) {
            c = cls[(size_t)j * X.parts + pc];
            if (c >= nclass) {
              bad_cls = true;
              c = 0xFFu;
            }
          }
          s_cls[pj] = (uint8_t)c;
          if (c != 0xFFu) cnt = *(const vq_u32x4*)&s_pp[c * 8u];
        }
        vq_u32x4 st;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t inc = vq_wave_scan(cnt[k]);
          st[k] = carry[k] + inc - cnt[k];
          carry[k] += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (pj < X.npj) s_start[pj] = st;
      }
      if (__any(bad_cls)) {
        if (lane == 0) raise_status(status, VSYN_ST_BAD_VQ, p);
      }
      uint32_t run = 0;
      X.poff_lanes = 0;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        if (lane == (uint32_t)ps) X.poff_lanes = run;
        run += (carry[ps >> 1] >> (16 * (ps & 1))) & 0xFFFFu;
      }
      VQ_STAMP(3);  // classifications read, scan
      const uint32_t sub_entries = bad ? 0u : run;
      if ((uint64_t)ent_cur + sub_entries > pkt_entries) bad = true;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the entries have landed in s_ent
      VQ_STAMP(4);  // wait for the entries (and for the previous packet's stores)

      // ---- accumulate + store ----
      bool bad_entry;
      if (ent_lds) bad_entry = vq_accumulate<true, TLDS>(X, lane, bad, s_cp, s_pp, s_start, s_ent, s_vm, s_cls, s_chan, ent, pool, s_tab, out);
      else bad_entry = vq_accumulate<false, TLDS>(X, lane, bad, s_cp, s_pp, s_start, s_ent, s_vm, s_cls, s_chan, ent, pool, s_tab, out);
      if (bad_entry) raise_status(status, VSYN_ST_BAD_VQ, p);
      VQ_STAMP(5);  // accumulate + stores issued

      // elements outside the partitions stay zero (hpp:1186-1190): [0, lim_begin) and [lim_begin + parts * psize, len) of every
      // vector; per channel these are two runs of consecutive bins (format 2: element e is bin e / nch of channel e % nch)
      const uint32_t tail0 = lim_begin + X.parts * X.psize;
      for (uint32_t k = 0; k < nch; ++k) {
        float* o = out + (size_t)s_chan[k] * n2;
        const uint32_t h1 = fmt2 ? (lim_begin + nch - 1u - k) / nch : lim_begin;  // bins below: b * nch + k < lim_begin
        const uint32_t t0 = fmt2 ? (tail0 + nch - 1u - k) / nch : tail0;          // bins from: b * nch + k >= tail0
        for (uint32_t b = lane; b < h1; b += VQ_THREADS) o[b] = 0.f;
        for (uint32_t b = t0 + lane; b < n2; b += VQ_THREADS) o[b] = 0.f;
      }
      cls_cur += X.npj;
      ent_cur += sub_entries;
      VQ_STAMP(6);  // zero fill issued
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a packet without submaps: the entry loads must not outlive s_ent's reuse)
    if (lane == 0 && (bad || ent_cur != pkt_entries)) raise_status(status, VSYN_ST_BAD_VQ, p);  // count must match the classifications
#ifdef VQ_STAMPS
    vq_st[VQ_NSTAMPS - 1] += 1;
#endif
  }
#ifdef VQ_STAMPS
  if (lane == 0 && blockIdx.x * waves + wave < 8192)
    for (int i = 0; i < VQ_NSTAMPS; ++i) g_vq_stamps[blockIdx.x * waves + wave][i] = vq_st[i];
#endif
}
#endif  // __HIPCC__

// ------------------------------------------------------------------------------------------------
// host side: validation + device block
// ------------------------------------------------------------------------------------------------
#include <string>
#include <vector>

// Builds the device block image; returns an empty string on success, the reason otherwise.
static inline std::string vq_build_block(const vsyn_vq_setup* vq, const ConstHeader& H, std::vector<uint8_t>& block) {
  if (!vq) return "vq setup is NULL";
  if (vq->num_mappings != H.num_mappings) return "vq setup: mapping count differs from the stream setup";
  if (vq->num_codebooks == 0 || vq->num_codebooks > 256 || !vq->codebooks) return "vq setup: codebook count out of range";
  if (vq->num_residues == 0 || vq->num_residues > VSYN_MAX_TABLES || !vq->residues) return "vq setup: residue count out of range";
  if (!vq->mappings) return "vq setup: mappings is NULL";
  std::vector<VqBook> books(vq->num_codebooks);
  std::vector<float> pool(8, 0.f);  // a zero vector first: what an out-of-range entry number reads instead of a table row
  for (uint32_t i = 0; i < vq->num_codebooks; ++i) {
    const vsyn_codebook& b = vq->codebooks[i];
    books[i].dims = b.dimensions;
    books[i].entries = b.num_entries;
    books[i].table_off = 0xFFFFFFFFu;
    books[i].lds_off = 0xFFFFFFFFu;
    if (b.lookup) {
      if (b.dimensions == 0 || b.dimensions > 65535u) return "vq setup: codebook " + std::to_string(i) + " has a bad vector length";
      if (b.num_entries == 0 || b.num_entries > 65536u) return "vq setup: codebook " + std::to_string(i) + " has more than 65536 entries";
      while (pool.size() & 7u) pool.push_back(0.f);  // 32-byte aligned tables: vectors of 4 / 8 floats are read with one / two 16-byte loads
      books[i].table_off = (uint32_t)pool.size();
      pool.insert(pool.end(), b.lookup, b.lookup + (size_t)b.dimensions * b.num_entries);
      if (pool.size() > (1u << 28)) return "vq setup: codebook tables too large";
    }
  }
  std::vector<VqResidue> residues(vq->num_residues);
  for (uint32_t i = 0; i < vq->num_residues; ++i) {
    const vsyn_residue& r = vq->residues[i];
    VqResidue& d = residues[i];
    memset(&d, 0, sizeof(d));
    if (r.type > 2) return "vq setup: residue type > 2";
    if (r.partition_size == 0 || r.begin > r.end) return "vq setup: residue " + std::to_string(i) + " has a bad range";
    if (r.num_classifications == 0 || r.num_classifications > 64 || !r.books) return "vq setup: residue classification count out of range";
    d.type = r.type;
    d.begin = r.begin;
    d.end = r.end;
    d.psize = r.partition_size;
    d.nclass = r.num_classifications;
    d.classwords = r.classwords;
    d.pass_mask = 0;
    for (uint32_t k = 0; k < 64 * 8; ++k) d.books[k] = -1;
    for (uint32_t k = 0; k < r.num_classifications * 8; ++k) {
      const int b = r.books[k];
      if (b < 0) continue;
      if ((uint32_t)b >= vq->num_codebooks) return "vq setup: residue names a codebook that does not exist";
      if (books[b].table_off == 0xFFFFFFFFu) return "vq setup: residue uses codebook " + std::to_string(b) + " which has no value table";
      if (r.partition_size % books[b].dims) return "vq setup: vector length of codebook " + std::to_string(b) + " does not divide the partition size";
      d.books[k] = (int16_t)b;
      d.pass_mask |= 1u << (k & 7u);
      const uint32_t dm = books[b].dims;
      if (r.type == 0 || (dm & (dm - 1u)) || (r.partition_size & (VQ_GROUP - 1u))) d.kinds |= 8u;  // VQ_KIND_GEN (partition tails too)
      if (r.type != 0 && !(dm & (dm - 1u))) d.kinds |= dm >= 4u ? 1u : (dm == 2u ? 2u : 4u);  // VQ_KIND_V4 / V2 / V1
    }
  }
  // The LDS table image: every table some residue uses, as it is. A setup is eligible for the LDS-table kernel if the image fits
  // VQ_IMG_MAX_BYTES and no residue needs the general route. (Measured and dropped: storing the big lattice books libvorbis writes —
  // value k a function of digit k of the entry number, e.g. the stereo fixture's 8 x 6561 table of 205 KB — as half vectors indexed
  // by entry % cq and entry / cq makes every real setup fit, bit-exactly, but the index arithmetic costs more than the L2 gathers it
  // replaces: 0.207 ms against 0.193 ms with the pool in global memory.)
  std::vector<float> img;
  bool img_ok = true;
  {
    std::vector<uint8_t> used(vq->num_codebooks, 0);
    for (const VqResidue& r : residues) {
      if (r.kinds & 8u) img_ok = false;  // VQ_KIND_GEN
      for (uint32_t k = 0; k < r.nclass * 8u; ++k)
        if (r.books[k] >= 0) used[r.books[k]] = 1;
    }
    for (uint32_t i = 0; i < vq->num_codebooks && img_ok; ++i) {
      if (!used[i]) continue;
      const vsyn_codebook& b = vq->codebooks[i];
      while (img.size() & 3u) img.push_back(0.f);
      books[i].lds_off = (uint32_t)img.size();
      if (((size_t)b.dimensions * b.num_entries + img.size()) * 4u > VQ_IMG_MAX_BYTES) img_ok = false;
      else img.insert(img.end(), b.lookup, b.lookup + (size_t)b.dimensions * b.num_entries);
    }
    if (!img_ok) img.clear();
    while (img.size() & 3u) img.push_back(0.f);
  }
  std::vector<VqMap> maps(vq->num_mappings);
  const uint32_t n2max = H.bs[1] / 2u;
  uint32_t max_slots = 64, max_classes = 1;
  for (const VqResidue& r : residues) max_classes = std::max(max_classes, r.nclass);
  for (uint32_t m = 0; m < vq->num_mappings; ++m) {
    const vsyn_vq_mapping& s = vq->mappings[m];
    VqMap& d = maps[m];
    memset(&d, 0, sizeof(d));
    if (s.num_submaps == 0 || s.num_submaps > 16 || !s.mux || !s.submap_residue) return "vq setup: submap count out of range";
    d.num_submaps = s.num_submaps;
    for (uint32_t ch = 0; ch < H.channels; ++ch) {
      if (s.mux[ch] >= s.num_submaps) return "vq setup: channel mux out of range";
      d.mux[ch] = s.mux[ch];
    }
    for (uint32_t k = 0; k < s.num_submaps; ++k) {
      if (s.submap_residue[k] >= vq->num_residues) return "vq setup: submap names a residue that does not exist";
      d.sub_rid[k] = s.submap_residue[k];
      uint32_t nch = 0;
      for (uint32_t ch = 0; ch < H.channels; ++ch)
        if (s.mux[ch] == k) {
          if (nch < 4) d.sub_chan4[k] |= ch << (8 * nch);
          ++nch;
        }
      d.sub_nch[k] = nch;
      if (!nch) continue;
      const VqResidue& r = residues[s.submap_residue[k]];
      const uint32_t len = r.type == 2 ? nch * n2max : n2max, vch = r.type == 2 ? 1u : nch;
      const uint32_t parts = (std::min(r.end, len) - std::min(r.begin, len)) / r.psize;
      if ((uint64_t)8 * parts * vch > VQ_MAX_SLOTS) return "vq setup: more than 8192 (pass, partition, channel) slots per packet";
      if (len > 65535u) return "vq setup: residue vector longer than 65535";  // per-pass entry counts are scanned as 16-bit halves
      max_slots = std::max(max_slots, 8u * parts * vch);
    }
  }
  VqHeader vh;
  memset(&vh, 0, sizeof(vh));
  vh.num_books = vq->num_codebooks;
  vh.num_residues = vq->num_residues;
  vh.num_maps = vq->num_mappings;
  vh.max_slots = max_slots;
  vh.max_classes = max_classes;
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t off = align16(sizeof(VqHeader));
  vh.off_books = (uint32_t)off;
  off = align16(off + books.size() * sizeof(VqBook));
  vh.off_residues = (uint32_t)off;
  off = align16(off + residues.size() * sizeof(VqResidue));
  vh.off_maps = (uint32_t)off;
  off = align16(off + maps.size() * sizeof(VqMap));
  vh.off_pool = (uint32_t)off;
  vh.pool_floats = (uint32_t)pool.size();
  off = align16(off + pool.size() * sizeof(float) + 16);
  vh.off_img = (uint32_t)off;
  vh.img_floats = (uint32_t)img.size();
  off = align16(off + img.size() * sizeof(float));
  vh.total_bytes = (uint32_t)off;
  block.assign(off, 0);
  memcpy(block.data(), &vh, sizeof(vh));
  memcpy(block.data() + vh.off_books, books.data(), books.size() * sizeof(VqBook));
  memcpy(block.data() + vh.off_residues, residues.data(), residues.size() * sizeof(VqResidue));
  memcpy(block.data() + vh.off_maps, maps.data(), maps.size() * sizeof(VqMap));
  if (!pool.empty()) memcpy(block.data() + vh.off_pool, pool.data(), pool.size() * sizeof(float));
  if (!img.empty()) memcpy(block.data() + vh.off_img, img.data(), img.size() * sizeof(float));
  return std::string();
}
