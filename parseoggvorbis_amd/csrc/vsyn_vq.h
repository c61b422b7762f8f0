// vsyn_vq.h — residue VQ stage: the data-parallel half of VorbisResidue::decode on the device.
//
// Reference: src/ParseOggVorbis.hpp:670-762.  The host keeps the bit-serial half (one classification word per partition
// group, one codebook entry number per vector) and ships those numbers; this kernel does what the reference does with
// them — look the entry's value vector up (lookup_table_, hpp:367-374) and add it into the residue vector, pass after
// pass (hpp:737-753), then de-interleave format 2 (hpp:687-693) — and writes "after_residue" in the packing the
// synthesis kernels read.
//
// One wavefront per packet (a fixed grid of single-wave workgroups walks the packets; the staged tables survive from packet
// to packet while the residue in use stays the same), two phases per submap vector:
//   scan        how many entries each (pass, partition, vector) slot consumes, and the exclusive scan of that in decode order
//               -> where each slot's first entry sits.  Lane = one (partition, vector) pair with its 8 passes; two passes share
//               a 32-bit word, so four DPP scans give the eight per-pass prefix sums.
//   accumulate  every ELEMENT belongs to exactly one partition and receives at most one value per pass, so a lane that owns
//               VQ_GROUP consecutive elements of a partition adds their up-to-8 contributions in pass order in registers —
//               the same sequence of f32 additions as the reference —, then stores each element once (zero outside the
//               partitions, format 2 de-interleaved, 16-byte stores).  No atomics, no zero-fill pass.  Vector lengths 1, 2, 4
//               and 8+ (powers of two) read their entries and value vectors with whole-vector loads; a pass nobody in the wave
//               has a book for is skipped wave-uniformly; format 0, odd lengths and partition tails go through one out-of-line
//               per-element routine.
// Measured alternatives (bench.py --workload config3_vq, 65 536 stereo long packets, kernel time): this design 0.54 ms;
// workgroup of 128/256 threads per packet 0.67-0.88 ms; separate scan + accumulate kernels (thread = element group, no
// per-packet loop) 0.69 ms; thread = one entry of a pass with f32 accumulators in LDS and the entry stream staged in LDS
// 0.69-0.77 ms (1.0 ms with 4 entries per thread in flight).  All of them are bound by the chain of dependent phases of one
// packet times the number of packets a CU keeps in flight, not by bytes or instructions.
#pragma once
#include "vsyn_device.h"

#define VQ_THREADS 64      /* one wavefront per packet: every synchronisation is wave-local */
#define VQ_GROUP 8
#define VQ_MAX_SLOTS 8192  /* (pass, partition, channel) slots of one submap vector; vsyn_attach_vq enforces it. The kernel's
                              dynamic LDS is sized to the largest count the attached setup can produce (fixtures: 400) */

struct VqBook {       // 16 bytes
  uint32_t dims, entries;
  uint32_t table_off;  // float index into the pool; 0xFFFFFFFF: no value table
  uint32_t pad;
};
struct VqResidue {
  uint32_t type, begin, end, psize, nclass, classwords, pad0, pad1;
  int16_t books[64 * 8];
};
struct VqMap {
  uint32_t num_submaps;
  uint8_t mux[VSYN_MAX_CHANNELS];
  uint8_t submap_residue[16];
};
struct VqHeader {
  uint32_t num_books, num_residues, num_maps, pad;
  uint32_t off_books, off_residues, off_maps, off_pool;  // byte offsets from the block base
  uint32_t pool_floats, total_bytes;
  uint32_t max_slots, pad2;  // largest slot count of any submap vector at blocksize1: the kernel's dynamic LDS holds max_slots / 2 words
};

#ifdef __HIPCC__
struct VqBookLds {  // per-book facts the accumulate loop needs, staged in LDS when the residue in use changes
  uint32_t table_off;
  uint16_t dims, per_part;   // vector length; vectors per partition = psize / dims
  uint32_t entries;          // bits 0..23 entry count; bit 31: dims is a power of two, bits 24..28: log2(dims) then
};
#define VQ_MAX_BOOKS 256

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

// The general element -> (vector, component) mapping, out of line: format 0 (8.6.3, hpp:738-746: element w <- vector w % step,
// component w / step), vector lengths that are not a power of two, partition tails. Returns true if an entry number is out
// of range; add[k] = 0 for such elements and for k beyond the partition.
__device__ __noinline__ bool vq_gather_general(float (&add)[VQ_GROUP], uint32_t type, uint32_t psize, uint32_t w0, uint32_t step, uint32_t dims,
                                               uint32_t nent, const uint16_t* __restrict__ e0, const float* __restrict__ tab) {
  bool bad = false;
  for (int k = 0; k < VQ_GROUP; ++k) {
    const uint32_t ww = w0 + k;
    float v = 0.f;
    if (ww < psize) {
      uint32_t ei, li;
      if (type == 0) {
        li = ww / step;
        ei = ww - li * step;
      } else {
        ei = ww / dims;
        li = ww - ei * dims;
      }
      const uint32_t en = e0[ei];
      if (en < nent) v = tab[(size_t)en * dims + li];
      else bad = true;
    }
    add[k] = v;
  }
  return bad;
}

// grid: a fixed number of single-wave workgroups, each walking packets p = blockIdx.x, + gridDim.x, ... (the staged tables
// survive from packet to packet while the residue in use stays the same)
__global__ void __launch_bounds__(VQ_THREADS) vsyn_residue_vq_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ vqb, uint32_t P,
                                                                     const PktInfo* __restrict__ info, const vsyn_vq_packet* __restrict__ vqp,
                                                                     const uint8_t* __restrict__ cls_all, uint64_t num_cls,
                                                                     const uint16_t* __restrict__ ent_all, uint64_t num_ent,
                                                                     float* __restrict__ residue, DevStatus* __restrict__ status) {
  extern __shared__ uint32_t s_start[];  // [4][npj]: first entry of slot (pass, pj) relative to the pass's first entry; passes 2k, 2k+1 packed 16+16
  __shared__ VqBookLds s_book[VQ_MAX_BOOKS];
  __shared__ __attribute__((aligned(16))) int16_t s_books[64 * 8];
  __shared__ uint32_t s_pass_off[9];
  __shared__ uint8_t s_chan[VSYN_MAX_CHANNELS];
  const ConstHeader* H = hdr_of(cb);
  const VqHeader* VH = (const VqHeader*)vqb;
  const uint32_t C = H->channels, lane = threadIdx.x;
  const VqBook* books = (const VqBook*)(vqb + VH->off_books);
  const float* pool = (const float*)(vqb + VH->off_pool);
  const uint32_t nbooks = min(VH->num_books, (uint32_t)VQ_MAX_BOOKS);
  uint32_t staged_residue = 0xFFFFFFFFu, staged_map = 0xFFFFFFFFu, staged_sub = 0xFFFFFFFFu;

  for (uint32_t p = blockIdx.x; p < P; p += gridDim.x) {
    const PktInfo pi = info[p];
    if (pi.bad || pi.n == 0) continue;  // flagged by the layout kernel; nothing downstream reads this packet's residue
    const uint32_t n2 = pi.n / 2u;
    const VqMap* mp = (const VqMap*)(vqb + VH->off_maps) + pi.mapping;
    const vsyn_vq_packet vp = vqp[p];
    bool bad = vp.entry_off + vp.num_entries > num_ent;
    const uint16_t* ent = ent_all + vp.entry_off;
    uint32_t cls_cur = vp.cls_off, ent_cur = 0;  // cursors over the packet's classification bytes / entries, submap after submap
    float* const out = residue + pi.res_off;

    for (uint32_t s = 0; s < mp->num_submaps; ++s) {
      // channels of this submap, in channel order (hpp:1191-1199)
      uint32_t nch = 0;
      for (uint32_t ch = 0; ch < C; ++ch) nch += mp->mux[ch] == s;
      if (nch == 0) continue;
      const uint32_t rid = mp->submap_residue[s];
      const VqResidue* r = (const VqResidue*)(vqb + VH->off_residues) + rid;
      const bool fmt2 = r->type == 2;
      const uint32_t vch = fmt2 ? 1u : nch;            // vectors decoded side by side (format 2: one interleaved vector)
      const uint32_t len = fmt2 ? nch * n2 : n2;       // hpp:687-688
      const uint32_t psize = r->psize;
      const uint32_t lim_begin = min(r->begin, len), lim_end = min(r->end, len);  // hpp:696-698
      const uint32_t parts = lim_end > lim_begin ? (lim_end - lim_begin) / psize : 0u;
      const uint32_t npj = parts * vch;                // (partition, vector) pairs; slot = (pass, pj), pj = pc * vch + j
      __syncthreads();  // (single wave: orders this wave's LDS traffic around the table updates)
      if (staged_map != pi.mapping || staged_sub != s) {
        if (lane == 0) {
          uint32_t k = 0;
          for (uint32_t ch = 0; ch < C; ++ch)
            if (mp->mux[ch] == s) s_chan[k++] = (uint8_t)ch;
        }
        staged_map = pi.mapping;
        staged_sub = s;
      }
      if (staged_residue != rid) {
        for (uint32_t i = lane; i < 64 * 8; i += VQ_THREADS) s_books[i] = r->books[i];
        for (uint32_t i = lane; i < nbooks; i += VQ_THREADS) {
          const VqBook bk = books[i];
          VqBookLds e;
          e.table_off = bk.table_off;
          e.dims = (uint16_t)bk.dims;
          e.per_part = (uint16_t)(bk.dims ? psize / bk.dims : 0u);
          e.entries = bk.entries & 0x00FFFFFFu;
          if (bk.dims && (bk.dims & (bk.dims - 1u)) == 0u) e.entries |= 0x80000000u | ((31u - (uint32_t)__clz((int)bk.dims)) << 24);
          s_book[i] = e;
        }
        staged_residue = rid;
      }
      const uint8_t* chan = s_chan;
      const uint8_t* cls = cls_all + cls_cur;
      if ((uint64_t)cls_cur + (uint64_t)npj > num_cls || 8u * npj > VH->max_slots) bad = true;
      __syncthreads();
      uint32_t vused = 0;  // bit j: vector j takes part (format 2: always, hpp:685-694; else floor_output_used, hpp:729)
      for (uint32_t j = 0; j < vch; ++j)
        if (fmt2 || ((pi.used >> chan[j]) & 1u)) vused |= 1u << j;

      // ---- entries per slot + exclusive scan in decode order (pass, partition, vector) ----
      // A lane takes one (partition, vector) pair and its 8 passes; two passes share a 32-bit word (a pass holds at most
      // len <= 65535 entries... enforced: 16 bits each), so four DPP scans cover the eight per-pass prefix sums.
      uint32_t carry[4] = {0, 0, 0, 0};
      for (uint32_t base = 0; base < npj && !bad; base += VQ_THREADS) {
        const uint32_t pj = base + lane;
        uint32_t cnt[4] = {0, 0, 0, 0};
        if (pj < npj) {
          const uint32_t pc = pj / vch, j = pj - pc * vch;
          const uint32_t c = cls[(size_t)j * parts + pc];
          if (((vused >> j) & 1u) && c < r->nclass) {
            const uint4 bw = *(const uint4*)&s_books[c * 8];  // the class's 8 books, one LDS read
            const uint32_t w[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
              const int book = (int16_t)(w[ps >> 1] >> (16 * (ps & 1)));
              if (book >= 0) cnt[ps >> 1] += (uint32_t)s_book[book].per_part << (16 * (ps & 1));
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t inc = vq_wave_scan(cnt[k]);
          if (pj < npj) s_start[k * npj + pj] = carry[k] + inc - cnt[k];
          carry[k] += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
      }
      if (lane == 0) {
        uint32_t run = 0;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          s_pass_off[ps] = run;
          run += (carry[ps >> 1] >> (16 * (ps & 1))) & 0xFFFFu;
        }
        s_pass_off[8] = run;
      }
      __syncthreads();
      const uint32_t sub_entries = bad ? 0u : s_pass_off[8];
      if ((uint64_t)ent_cur + sub_entries > vp.num_entries) bad = true;

      // ---- accumulate: a lane owns VQ_GROUP consecutive elements of one partition of one vector ----
      const uint32_t gpp = (psize + VQ_GROUP - 1) / VQ_GROUP;      // groups per partition
      const uint32_t body = parts * psize;                         // elements that can receive values: [lim_begin, lim_begin + body)
      const uint32_t groups = npj * gpp;
      const uint16_t* esub = ent + ent_cur;
      for (uint32_t gi = lane; gi < groups; gi += VQ_THREADS) {
        const uint32_t pj = gi / gpp, w0 = (gi - pj * gpp) * VQ_GROUP;
        const uint32_t pc = pj / vch, j = pj - pc * vch;
        const uint32_t cnt_el = min((uint32_t)VQ_GROUP, psize - w0);
        float acc[VQ_GROUP];
#pragma unroll
        for (int k = 0; k < VQ_GROUP; ++k) acc[k] = 0.f;
        if (!bad && ((vused >> j) & 1u)) {
          const uint32_t c = cls[(size_t)j * parts + pc];
          if (c >= r->nclass) {
            raise_status(status, VSYN_ST_BAD_VQ, p);
          } else {
            const uint4 bw = *(const uint4*)&s_books[c * 8];
            const uint32_t w[4] = {bw.x, bw.y, bw.z, bw.w};
            uint32_t st4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) st4[k] = s_start[k * npj + pj];
            bool bad_entry = false;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {  // pass order = order of the additions (hpp:711)
              const int book = (int16_t)(w[ps >> 1] >> (16 * (ps & 1)));
              if (!__any(book >= 0)) continue;  // wave-uniform: nobody has a codebook in this pass (typical for passes 3..7)
              if (book < 0) continue;
              const VqBookLds bk = s_book[book];
              const uint16_t* e0 = esub + s_pass_off[ps] + ((st4[ps >> 1] >> (16 * (ps & 1))) & 0xFFFFu);
              const float* tab = pool + bk.table_off;  // 32-byte aligned (vq_build_block)
              const uint32_t dims = bk.dims, nent = bk.entries & 0x00FFFFFFu;
              const uint32_t sh = (bk.entries >> 24) & 31u;
              const bool fast = r->type != 0 && (bk.entries >> 31) != 0u && cnt_el == VQ_GROUP;  // 8.6.4, power-of-two vector length
              float add[VQ_GROUP];
              if (fast && sh >= 3) {         // one vector covers the group's 8 elements: 8 consecutive components
                const uint32_t en = e0[w0 >> sh];
                const bool ok = en < nent;
                bad_entry |= !ok;
                const float4* v4 = (const float4*)(tab + (size_t)(ok ? en : 0u) * dims + (w0 & (dims - 1u)));
                const float4 a = v4[0], b = v4[1];
                add[0] = a.x; add[1] = a.y; add[2] = a.z; add[3] = a.w;
                add[4] = b.x; add[5] = b.y; add[6] = b.z; add[7] = b.w;
                if (!ok) {
#pragma unroll
                  for (int k = 0; k < VQ_GROUP; ++k) add[k] = 0.f;
                }
              } else if (fast && sh == 2) {  // two vectors of 4
                const uint32_t i0 = w0 >> 2;
                const uint32_t en0 = e0[i0], en1 = e0[i0 + 1];
                const bool ok0 = en0 < nent, ok1 = en1 < nent;
                bad_entry |= !(ok0 && ok1);
                const float4 a = *(const float4*)(tab + (size_t)(ok0 ? en0 : 0u) * 4u);
                const float4 b = *(const float4*)(tab + (size_t)(ok1 ? en1 : 0u) * 4u);
                add[0] = ok0 ? a.x : 0.f; add[1] = ok0 ? a.y : 0.f; add[2] = ok0 ? a.z : 0.f; add[3] = ok0 ? a.w : 0.f;
                add[4] = ok1 ? b.x : 0.f; add[5] = ok1 ? b.y : 0.f; add[6] = ok1 ? b.z : 0.f; add[7] = ok1 ? b.w : 0.f;
              } else if (fast && sh == 1) {  // four vectors of 2
                const uint32_t i0 = w0 >> 1;
                uint32_t en[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) en[k] = e0[i0 + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const bool ok = en[k] < nent;
                  bad_entry |= !ok;
                  const float2 a = *(const float2*)(tab + (size_t)(ok ? en[k] : 0u) * 2u);
                  add[2 * k] = ok ? a.x : 0.f;
                  add[2 * k + 1] = ok ? a.y : 0.f;
                }
              } else if (fast) {             // eight scalars
                uint32_t en[VQ_GROUP];
#pragma unroll
                for (int k = 0; k < VQ_GROUP; ++k) en[k] = e0[w0 + k];
#pragma unroll
                for (int k = 0; k < VQ_GROUP; ++k) {
                  const bool ok = en[k] < nent;
                  bad_entry |= !ok;
                  const float v = tab[ok ? en[k] : 0u];
                  add[k] = ok ? v : 0.f;
                }
              } else {
                bad_entry |= vq_gather_general(add, r->type, psize, w0, bk.per_part, dims, nent, e0, tab);
              }
#pragma unroll
              for (int k = 0; k < VQ_GROUP; ++k) acc[k] += add[k];
            }
            if (bad_entry) raise_status(status, VSYN_ST_BAD_VQ, p);
          }
        }
        // store (de-interleaving format 2: element e of the interleaved vector is bin e / nch of channel e % nch, hpp:690-692)
        const uint32_t e_first = lim_begin + pc * psize + w0;
        if (fmt2 && nch == 2 && cnt_el == VQ_GROUP && !(e_first & 1u)) {  // stereo: 4 consecutive bins per channel
          float* o0 = out + (size_t)chan[0] * n2 + (e_first >> 1);
          float* o1 = out + (size_t)chan[1] * n2 + (e_first >> 1);
          if (((uintptr_t)o0 & 15u) == 0 && ((uintptr_t)o1 & 15u) == 0) {
            *(float4*)o0 = make_float4(acc[0], acc[2], acc[4], acc[6]);
            *(float4*)o1 = make_float4(acc[1], acc[3], acc[5], acc[7]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o0[k] = acc[2 * k];
              o1[k] = acc[2 * k + 1];
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < VQ_GROUP; ++k)
            if ((uint32_t)k < cnt_el) {
              const uint32_t e = e_first + k;
              if (fmt2) out[(size_t)chan[e % nch] * n2 + e / nch] = acc[k];
              else out[(size_t)chan[j] * n2 + e] = acc[k];
            }
        }
      }
      // elements outside the partitions stay zero (hpp:1186-1190): [0, lim_begin) and [lim_begin + body, len)
      const uint32_t tail0 = lim_begin + body;
      for (uint32_t j = 0; j < vch; ++j) {
        for (uint32_t e = lane; e < lim_begin; e += VQ_THREADS) {
          if (fmt2) out[(size_t)chan[e % nch] * n2 + e / nch] = 0.f;
          else out[(size_t)chan[j] * n2 + e] = 0.f;
        }
        for (uint32_t e = tail0 + lane; e < len; e += VQ_THREADS) {
          if (fmt2) out[(size_t)chan[e % nch] * n2 + e / nch] = 0.f;
          else out[(size_t)chan[j] * n2 + e] = 0.f;
        }
      }
      cls_cur += npj;
      ent_cur += sub_entries;
    }
    if (lane == 0 && (bad || ent_cur != vp.num_entries)) raise_status(status, VSYN_ST_BAD_VQ, p);  // count must match the classifications
  }
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
  std::vector<float> pool;
  for (uint32_t i = 0; i < vq->num_codebooks; ++i) {
    const vsyn_codebook& b = vq->codebooks[i];
    books[i].dims = b.dimensions;
    books[i].entries = b.num_entries;
    books[i].table_off = 0xFFFFFFFFu;
    books[i].pad = 0;
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
    for (uint32_t k = 0; k < 64 * 8; ++k) d.books[k] = -1;
    for (uint32_t k = 0; k < r.num_classifications * 8; ++k) {
      const int b = r.books[k];
      if (b < 0) continue;
      if ((uint32_t)b >= vq->num_codebooks) return "vq setup: residue names a codebook that does not exist";
      if (books[b].table_off == 0xFFFFFFFFu) return "vq setup: residue uses codebook " + std::to_string(b) + " which has no value table";
      if (r.partition_size % books[b].dims) return "vq setup: vector length of codebook " + std::to_string(b) + " does not divide the partition size";
      d.books[k] = (int16_t)b;
    }
  }
  std::vector<VqMap> maps(vq->num_mappings);
  const uint32_t n2max = H.bs[1] / 2u;
  uint32_t max_slots = 64;
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
      d.submap_residue[k] = s.submap_residue[k];
      uint32_t nch = 0;
      for (uint32_t ch = 0; ch < H.channels; ++ch) nch += s.mux[ch] == k;
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
  vh.total_bytes = (uint32_t)off;
  block.assign(off, 0);
  memcpy(block.data(), &vh, sizeof(vh));
  memcpy(block.data() + vh.off_books, books.data(), books.size() * sizeof(VqBook));
  memcpy(block.data() + vh.off_residues, residues.data(), residues.size() * sizeof(VqResidue));
  memcpy(block.data() + vh.off_maps, maps.data(), maps.size() * sizeof(VqMap));
  if (!pool.empty()) memcpy(block.data() + vh.off_pool, pool.data(), pool.size() * sizeof(float));
  return std::string();
}
