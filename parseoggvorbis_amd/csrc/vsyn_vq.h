// vsyn_vq.h — residue VQ stage: the data-parallel half of VorbisResidue::decode on the device.
//
// Reference: src/ParseOggVorbis.hpp:670-762.  The host keeps the bit-serial half (one classification word per partition
// group, one codebook entry number per vector) and ships those numbers; this kernel does what the reference does with
// them — look the entry's value vector up (lookup_table_, hpp:367-374) and add it into the residue vector, pass after
// pass (hpp:737-753), then de-interleave format 2 (hpp:687-693) — and writes "after_residue" in the packing the
// synthesis kernels read.
//
// Work decomposition: one workgroup per packet.  Every ELEMENT of the residue vector belongs to exactly one partition
// and receives at most one value per pass, so a thread that owns an element adds its up-to-8 contributions in pass
// order in a register: the same sequence of f32 additions as the reference, no atomics, no zero-fill pass, one store
// per element.  A thread owns VQ_GROUP consecutive elements of one partition: the partition-level look-ups
// (classification, book of the pass, first entry of the (pass, partition, channel) slot) are shared by the group, and
// consecutive elements share their entry number while they sit in the same vector.
// The first entry of each slot comes from an exclusive scan over the slots in decode order (pass, partition, channel).
#pragma once
#include "vsyn_device.h"

#define VQ_THREADS 128     /* two waves per packet: a stereo long block has ~200 element groups, a short one ~26 */
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
  uint32_t max_slots, pad2;  // largest slot count of any submap vector at blocksize1: the kernel's dynamic LDS (x 4 bytes)
};

#ifdef __HIPCC__
// grid: one workgroup per packet
__global__ void __launch_bounds__(VQ_THREADS) vsyn_residue_vq_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ vqb, uint32_t P,
                                                                     const PktInfo* __restrict__ info, const vsyn_vq_packet* __restrict__ vqp,
                                                                     const uint8_t* __restrict__ cls_all, uint64_t num_cls,
                                                                     const uint16_t* __restrict__ ent_all, uint64_t num_ent,
                                                                     float* __restrict__ residue, DevStatus* __restrict__ status) {
  extern __shared__ uint32_t s_start[];  // [max_slots] exclusive scan: first entry of each slot, relative to the submap's first entry
  __shared__ uint32_t s_part[VQ_THREADS];
  __shared__ int16_t s_books[64 * 8];
  __shared__ uint8_t s_chan[VSYN_MAX_CHANNELS];
  __shared__ uint32_t s_total;
  const uint32_t p = blockIdx.x;
  if (p >= P) return;
  const ConstHeader* H = hdr_of(cb);
  const VqHeader* VH = (const VqHeader*)vqb;
  const PktInfo pi = info[p];
  if (pi.bad || pi.n == 0) return;  // flagged by the layout kernel; nothing downstream reads this packet's residue
  const uint32_t C = H->channels, n2 = pi.n / 2u, tid = threadIdx.x;
  const VqBook* books = (const VqBook*)(vqb + VH->off_books);
  const float* pool = (const float*)(vqb + VH->off_pool);
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
    __syncthreads();  // previous submap done with the shared tables
    if (tid == 0) {
      uint32_t k = 0;
      for (uint32_t ch = 0; ch < C; ++ch)
        if (mp->mux[ch] == s) s_chan[k++] = (uint8_t)ch;
    }
    const uint8_t* chan = s_chan;
    const VqResidue* r = (const VqResidue*)(vqb + VH->off_residues) + mp->submap_residue[s];
    const bool fmt2 = r->type == 2;
    const uint32_t vch = fmt2 ? 1u : nch;            // vectors decoded side by side (format 2: one interleaved vector)
    const uint32_t len = fmt2 ? nch * n2 : n2;       // hpp:687-688
    const uint32_t psize = r->psize;
    const uint32_t lim_begin = min(r->begin, len), lim_end = min(r->end, len);  // hpp:696-698
    const uint32_t parts = lim_end > lim_begin ? (lim_end - lim_begin) / psize : 0u;
    const uint32_t slots = 8u * parts * vch;
    for (uint32_t i = tid; i < 64 * 8; i += VQ_THREADS) s_books[i] = r->books[i];
    const uint8_t* cls = cls_all + cls_cur;
    if ((uint64_t)cls_cur + (uint64_t)vch * parts > num_cls || slots > VH->max_slots) bad = true;
    __syncthreads();
    uint32_t vused = 0;  // bit j: vector j takes part (format 2: always, hpp:685-694; else floor_output_used, hpp:729)
    for (uint32_t j = 0; j < vch; ++j)
      if (fmt2 || ((pi.used >> chan[j]) & 1u)) vused |= 1u << j;

    // ---- entries per slot, exclusive scan in decode order: slot = (pass * parts + pc) * vch + j ----
    const uint32_t per = (slots + VQ_THREADS - 1) / VQ_THREADS;  // consecutive slots per thread
    uint32_t mine = 0;
    if (!bad)
      for (uint32_t k = 0; k < per; ++k) {
        const uint32_t sl = tid * per + k;
        if (sl >= slots) break;
        const uint32_t j = sl % vch, pc = (sl / vch) % parts, pass = sl / (vch * parts);
        uint32_t cnt = 0;
        if ((vused >> j) & 1u) {
          const uint32_t c = cls[(size_t)j * parts + pc];
          const int book = c < r->nclass ? (int)s_books[c * 8 + pass] : -1;
          if (book >= 0) cnt = psize / books[book].dims;
        }
        s_start[sl] = cnt;
        mine += cnt;
      }
    s_part[tid] = mine;
    __syncthreads();
    if (tid < 64) {  // scan the per-thread sums with one wave
      constexpr int PER_LANE = VQ_THREADS / 64;
      uint32_t v[PER_LANE], sum = 0;
#pragma unroll
      for (int k = 0; k < PER_LANE; ++k) {
        v[k] = s_part[tid * PER_LANE + k];
        sum += v[k];
      }
      uint32_t inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if ((int)tid >= d) inc += o;
      }
      uint32_t run = inc - sum;
#pragma unroll
      for (int k = 0; k < PER_LANE; ++k) {
        s_part[tid * PER_LANE + k] = run;
        run += v[k];
      }
      if (tid == 63) s_total = inc;
    }
    __syncthreads();
    if (!bad) {
      uint32_t run = s_part[tid];
      for (uint32_t k = 0; k < per; ++k) {
        const uint32_t sl = tid * per + k;
        if (sl >= slots) break;
        const uint32_t cnt = s_start[sl];
        s_start[sl] = run;
        run += cnt;
      }
    }
    __syncthreads();
    const uint32_t sub_entries = bad ? 0u : s_total;
    if ((uint64_t)ent_cur + sub_entries > vp.num_entries) bad = true;

    // ---- accumulate: a thread owns VQ_GROUP consecutive elements of one partition of one vector ----
    const uint32_t gpp = (psize + VQ_GROUP - 1) / VQ_GROUP;      // groups per partition
    const uint32_t body = parts * psize;                         // elements that can receive values: [lim_begin, lim_begin + body)
    const uint32_t groups = vch * parts * gpp;
    for (uint32_t gi = tid; gi < groups; gi += VQ_THREADS) {
      const uint32_t j = gi / (parts * gpp), rem = gi % (parts * gpp), pc = rem / gpp, w0 = (rem % gpp) * VQ_GROUP;
      const uint32_t cnt_el = min((uint32_t)VQ_GROUP, psize - w0);
      float acc[VQ_GROUP];
#pragma unroll
      for (int k = 0; k < VQ_GROUP; ++k) acc[k] = 0.f;
      if (!bad && ((vused >> j) & 1u)) {
        const uint32_t c = cls[(size_t)j * parts + pc];
        if (c >= r->nclass) {
          raise_status(status, VSYN_ST_BAD_VQ, p);
        } else {
          for (uint32_t pass = 0; pass < 8; ++pass) {  // pass order = order of the additions (hpp:711)
            const int book = s_books[c * 8 + pass];
            if (book < 0) continue;
            const VqBook bk = books[book];
            const uint16_t* e0 = ent + ent_cur + s_start[(pass * parts + pc) * vch + j];
            const float* tab = pool + bk.table_off;
            if (r->type == 0) {  // 8.6.3 (hpp:738-746): element w <- vector w % step, component w / step
              const uint32_t step = psize / bk.dims;
#pragma unroll
              for (int k = 0; k < VQ_GROUP; ++k)
                if ((uint32_t)k < cnt_el) {
                  const uint32_t w = w0 + k, en = e0[w % step];
                  if (en >= bk.entries) raise_status(status, VSYN_ST_BAD_VQ, p);
                  else acc[k] += tab[(size_t)en * bk.dims + w / step];
                }
            } else {             // 8.6.4 (hpp:747-754): element w <- vector w / dims, component w % dims
              uint32_t i = w0 / bk.dims, l = w0 - i * bk.dims;
              uint32_t en = e0[i];
              bool ok = en < bk.entries;
              if (!ok) raise_status(status, VSYN_ST_BAD_VQ, p);
              const float* vec = tab + (size_t)(ok ? en : 0u) * bk.dims;
#pragma unroll
              for (int k = 0; k < VQ_GROUP; ++k)
                if ((uint32_t)k < cnt_el) {
                  if (ok) acc[k] += vec[l];
                  if (++l == bk.dims && (uint32_t)k + 1u < cnt_el) {
                    l = 0;
                    en = e0[++i];
                    ok = en < bk.entries;
                    if (!ok) raise_status(status, VSYN_ST_BAD_VQ, p);
                    vec = tab + (size_t)(ok ? en : 0u) * bk.dims;
                  }
                }
            }
          }
        }
      }
      // store (de-interleaving format 2: element e of the interleaved vector is bin e / nch of channel e % nch, hpp:690-692)
      const uint32_t e_first = lim_begin + pc * psize + w0;
#pragma unroll
      for (int k = 0; k < VQ_GROUP; ++k)
        if ((uint32_t)k < cnt_el) {
          const uint32_t e = e_first + k;
          if (fmt2) out[(size_t)chan[e % nch] * n2 + e / nch] = acc[k];
          else out[(size_t)chan[j] * n2 + e] = acc[k];
        }
    }
    // elements outside the partitions stay zero (hpp:1186-1190): [0, lim_begin) and [lim_begin + body, len)
    const uint32_t tail0 = lim_begin + body;
    for (uint32_t j = 0; j < vch; ++j)
      for (uint32_t e = tid; e < len; e += VQ_THREADS)
        if (e < lim_begin || e >= tail0) {
          if (fmt2) out[(size_t)chan[e % nch] * n2 + e / nch] = 0.f;
          else out[(size_t)chan[j] * n2 + e] = 0.f;
        }
    cls_cur += vch * parts;
    ent_cur += sub_entries;
  }
  if (tid == 0 && (bad || ent_cur != vp.num_entries)) raise_status(status, VSYN_ST_BAD_VQ, p);  // count must match the classifications
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
