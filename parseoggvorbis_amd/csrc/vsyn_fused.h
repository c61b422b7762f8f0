// vsyn_fused.h — the speed path: ONE kernel from coded floor posts + residue to PCM for runs of long blocks
// (n = 2048, <= 2 channels), one WAVEFRONT per run of consecutive packets of a stream.
//
//   * residue is read once (coalesced 8-byte loads, 512 B per wave-instruction) and PCM is written once
//     (coalesced 8-byte stores): no intermediate ever touches HBM.
//   * per lane: 8 complex points per channel.  Point k = lane + 64 t  (t = 0..7) is built from bins 2k and
//     1023-2k; the second comes from the mirror lane (63 - lane) by one ds_bpermute.
//   * FFT-512 = three in-lane radix-8 DIF passes; the two index exchanges between them go through a
//     wave-private, bank-conflict-free padded LDS image (no barriers: a wave's LDS ops execute in order).
//     Twiddles (pre/post rotation, W512) are staged once per workgroup in LDS.  No MFMA: this is a butterfly
//     network on f32, not a contraction.
//   * after the post rotation lane `l` holds points m = kappa + 64 c' (kappa = swap3(l)), and
//     point m of consecutive packets produces the SAME output samples (s, 1023-s): the overlap carry is
//     kept in 8 registers per channel for the whole run; the window is applied on the fly.
//   * the first packet of a run whose predecessor belongs to another wave is recomputed (one-packet halo,
//     1/R extra work) instead of communicated.
//
// Everything the reference rounds separately is rounded separately here (file is built -ffp-contract=off;
// FMAs are explicit): coupling and floor product are bit-exact, `pcm += block*window` is mul-then-add.
#pragma once
#include <hip/hip_runtime.h>

#include "vsyn_device.h"

#ifndef FUSED_WAVES
#define FUSED_WAVES 8     // waves per workgroup (tables are shared per workgroup)
#endif
#ifndef FUSED_MIN_WAVES_PER_SIMD
#define FUSED_MIN_WAVES_PER_SIMD 4
#endif
#define FUSED_XSLOTS 576  // float2 slots of the per-wave exchange image: 8 rows x 72 (>= 8 x 65)
// Diagnostic builds only (tools/ab_variants.sh): -DVSYN_KNOCKOUT=<bits> removes one phase of the long-block loop so that its
// marginal cost can be timed (the results are then wrong by construction): 1 channel hand-off + coupling, 2 floor product,
// 4 FFT, 8 PCM stores, 16 floor set-up; of the packed short pass: 32 PCM stores, 64 floor (tables, look-ups, product), 128 FFT-64,
// 256 residue loads, 512 coupling hand-off.
#ifndef VSYN_KNOCKOUT
#define VSYN_KNOCKOUT 0
#endif
// -DVSYN_STAMPS: per-phase cycle counts of the steady long-block loop (s_memtime at phase boundaries, which also drains the
// wave's LDS queue there: phases no longer overlap each other, so the sum exceeds the unstamped iteration). Read back and
// printed by vsyn_destroy. Diagnostic builds only.
#ifdef VSYN_STAMPS
#define VSYN_NSTAMPS 16
__device__ unsigned long long g_vsyn_stamps[8192][VSYN_NSTAMPS];
#define STAMP(i)                                                  \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    const unsigned long long t_ = __builtin_readcyclecounter();   \
    st_acc[i] += t_ - st_last;                                    \
    st_last = t_;                                                 \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// Read-only LDS image of the fused kernel, built once per handle on the host in exactly the order the lanes read
// it (every table is lane-major: lane l of a wave-instruction reads element [..][l], so all reads are conflict-free).
struct FusedLdsImage {
  float2 pre[8][64];      // pre-rotation   exp(-i pi (4k+1)/(4M)),  k = lane + 64 t
  float2 post[8][64];     // post-rotation  exp(-i pi m / M),        m = swap3(lane) + 64 k
  float2 tw1[8][64];      // W512^(lane * t)        (row 0 unused)
  float2 tw2[8][8];       // W64^(c * a), c = lane & 7 (row 0 unused)
  float win[2][2][8][64]; // [prev/next flag][0: at s, 1: at 1023-s][k][lane]: left half of the long window at the
                          // output sample s of point m; the right half for next flag f is its mirror (hpp:850-859)
  float invdb[260];       // Vorbis I 10.1 (hpp:588); [255] == 1.0f, extra [256] == 0.0f (see floor product)
  // short blocks (blocksize0 == 256: 64 complex points, one per lane; used by mixed-block runs only)
  float2 pre_s[64];       // pre-rotation of point k = lane
  float2 post_s[64];      // post-rotation of bin m = bitrev6(lane)
  float2 tws[6][64];      // DIF stage twiddles: stage i pairs lane l with l ^ (32 >> i); W_(64>>i)^(l & ((32>>i)-1))
  float wsl[2][64];       // short window at sample s(lane) and at 127 - s(lane)
  // packed short pass (8 blocks of n = 256 at once, fused_short_pass): bin f = a' + 8 c' in register c' of the lanes with lane & 7 == a'
  float2 post8[8][8];     // [c'][a'] post-rotation of bin f
  float wsl8[2][8][8];    // [0: at s, 1: at 127 - s][c'][a'], s = 2f - 64 (c' >= 4) resp. 63 - 2f
};

struct FusedTables {
  uint8_t* d_binseg = nullptr;  // [num_floors][bs1/2]: sorted-post interval containing bin x
  FusedLdsImage* d_lds = nullptr;
  int waves_per_cu = 12;
  int coupling_mode = 0;
};

struct FusedArgs {
  const uint8_t* cb;
  const uint8_t* binseg;
  const FusedLdsImage* lds_image;
  const vsyn_packet* packets;
  const vsyn_segment* segs;
  const PktInfo* info;
  const SegInfo* sinfo;
  uint16_t* curve;         // feature tap "floor1 floor" (vsyn_taps.floor_curve) or nullptr; written by the tap variant of the kernel only
  const uint8_t* run_cls;  // [S][runs_per_seg] from the layout kernel: 1 steady long run, 2 mixed-block run, 0 staged, 0xFF none
  const float* residue;
  const uint16_t* fy;      // unwrapped floor rows (vsyn_prep_kernel / vsyn_floor_unwrap_kernel)
  float* pcm;
  float* carry;
  DevStatus* status;
  uint64_t plane_stride;
  uint32_t S, R, fused_ok, coupling_mode, runs_per_seg;  // fused_ok: bit 0 steady long runs, bit 1 mixed-block runs
};

__device__ __forceinline__ float2 f2(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return f2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return f2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return f2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return f2(a.y, -a.x); }  // * (-i)

// forward DFT-8, natural-order output, in place (decimation in frequency)
__device__ __forceinline__ void dft8(float2 (&x)[8]) {
  const float h = 0.70710678118654752440f;
  float2 a0 = cadd(x[0], x[4]), a1 = cadd(x[1], x[5]), a2 = cadd(x[2], x[6]), a3 = cadd(x[3], x[7]);
  float2 b0 = csub(x[0], x[4]), b1 = csub(x[1], x[5]), b2 = csub(x[2], x[6]), b3 = csub(x[3], x[7]);
  b1 = f2((b1.x + b1.y) * h, (b1.y - b1.x) * h);   // * (1 - i)/sqrt2
  b2 = mul_mi(b2);                                 // * (-i)
  b3 = f2((b3.y - b3.x) * h, -(b3.x + b3.y) * h);  // * (-1 - i)/sqrt2
  float2 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), c3 = mul_mi(csub(a1, a3));
  x[0] = cadd(c0, c2);
  x[4] = csub(c0, c2);
  x[2] = cadd(c1, c3);
  x[6] = csub(c1, c3);
  float2 d0 = cadd(b0, b2), d1 = csub(b0, b2), d2 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
  x[1] = cadd(d0, d2);
  x[5] = csub(d0, d2);
  x[3] = cadd(d1, d3);
  x[7] = csub(d1, d3);
}

// Register-to-register forms of the two index exchanges (no LDS): a transposition of three register-index bits with three
// lane bits is three swap stages; lane bits 5 and 4 have swap instructions (v_permlane32_swap / v_permlane16_swap), lane bits
// 3 and 2 are DPP row shifts under a bank mask, lane bits 1 and 0 a DPP quad permutation plus a select.
typedef unsigned xch_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xch_swap32(float& a, float& b) {  // a[32..63] <-> b[0..31]
  const xch_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xch_swap16(float& a, float& b) {  // odd rows of a <-> even rows of b
  const xch_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
template <int SH>  // lane bit 3 (SH = 8) or 2 (SH = 4)
__device__ __forceinline__ void xch_row(float& lo, float& hi) {
  constexpr int shr = 0x110 + SH, shl = 0x100 + SH;
  constexpr int up = SH == 8 ? 0xC : 0xA, dn = SH == 8 ? 0x3 : 0x5;
  const unsigned l = __float_as_uint(lo), h = __float_as_uint(hi);
  const unsigned nl = __builtin_amdgcn_update_dpp(l, h, shr, 0xF, up, false);  // lanes with the bit set: hi of lane - SH
  const unsigned nh = __builtin_amdgcn_update_dpp(h, l, shl, 0xF, dn, false);  // lanes with the bit clear: lo of lane + SH
  lo = __uint_as_float(nl);
  hi = __uint_as_float(nh);
}
template <int BIT>  // lane bit 1 or 0
__device__ __forceinline__ void xch_quad(float& lo, float& hi, bool bitset) {
  constexpr int qp = BIT == 1 ? 0x4E : 0xB1;
  const unsigned l = __float_as_uint(lo), h = __float_as_uint(hi);
  const unsigned th = __builtin_amdgcn_mov_dpp(h, qp, 0xF, 0xF, false);
  const unsigned tl = __builtin_amdgcn_mov_dpp(l, qp, 0xF, 0xF, false);
  lo = __uint_as_float(bitset ? th : l);
  hi = __uint_as_float(bitset ? h : tl);
}

// FFT-512 across one wave: in: lane l holds z[t] = point l + 64 t; out: lane l holds Z[c'] = bin swap3(l) + 64 c'.
// xb = this wave's exchange image (FUSED_XSLOTS float2), w = W512^j table (LDS).
__device__ __forceinline__ void fft512_wave(float2 (&z)[8], float2* __restrict__ xb, const FusedLdsImage* __restrict__ T, uint32_t lane) {
  const uint32_t c = lane & 7u, hi = lane >> 3;
  dft8(z);  // over t -> t'
#pragma unroll
  for (int t = 1; t < 8; ++t) z[t] = cmulf(z[t], T->tw1[t][lane]);
  // exchange 1: element (t', a, c): lane 8a+c reg t'  ->  lane 8t'+c reg a.   row stride 72: conflict-free both ways
#if defined(VSYN_FFT_REGS) && (VSYN_FFT_REGS & 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) { xch_swap32(z[i].x, z[i + 4].x); xch_swap32(z[i].y, z[i + 4].y); }
#pragma unroll
  for (int i = 0; i < 8; ++i) if (!(i & 2)) { xch_swap16(z[i].x, z[i + 2].x); xch_swap16(z[i].y, z[i + 2].y); }
#pragma unroll
  for (int i = 0; i < 8; i += 2) { xch_row<8>(z[i].x, z[i + 1].x); xch_row<8>(z[i].y, z[i + 1].y); }
#else
#pragma unroll
  for (int t = 0; t < 8; ++t) xb[t * 72 + lane] = z[t];
#pragma unroll
  for (int a = 0; a < 8; ++a) z[a] = xb[hi * 72 + a * 8 + c];
#endif
  dft8(z);  // over a -> a'
#pragma unroll
  for (int a = 1; a < 8; ++a) z[a] = cmulf(z[a], T->tw2[a][c]);  // W64^(c a')
  // exchange 2: element (h, a', c): lane 8h+c reg a'  ->  lane 8h+a' reg c.   row stride 65
#if defined(VSYN_FFT_REGS) && (VSYN_FFT_REGS & 2)
  {
    const bool b1 = (lane & 2u) != 0, b0 = (lane & 1u) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { xch_row<4>(z[i].x, z[i + 4].x); xch_row<4>(z[i].y, z[i + 4].y); }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (!(i & 2)) { xch_quad<1>(z[i].x, z[i + 2].x, b1); xch_quad<1>(z[i].y, z[i + 2].y, b1); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { xch_quad<0>(z[i].x, z[i + 1].x, b0); xch_quad<0>(z[i].y, z[i + 1].y, b0); }
  }
#else
#pragma unroll
  for (int a = 0; a < 8; ++a) xb[a * 65 + lane] = z[a];
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = xb[c * 65 + hi * 8 + k];
#endif
  dft8(z);  // over c -> c'
}

// The compiler's wait-count insertion merges conservatively where control flow joins: a vector load in a RARE branch (floor
// change, re-read of a coded row) right before a join makes every later use of ANY loaded register wait for vmcnt(0) — which,
// inside the packet loop, means waiting for the residue prefetch that was issued a few instructions earlier and for the PCM
// stores of the previous packet (measured with -DVSYN_STAMPS: a third of the loop's cycles). Rare branches therefore finish
// their own loads before they rejoin, and so does the loop's prologue: the joins then only carry the steady state's order
// (row of packet q+1, PCM stores of q, residue of q+1) and the waits inside the loop are the counted ones.
__device__ __forceinline__ void vmem_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // gfx9 encoding: vmcnt(0), expcnt/lgkmcnt untouched

// hpp:1220-1239, branch-free (selects only; same comparisons, same single add/sub per output, so bit-identical):
//   d = m > 0 ? a : -a;   a > 0 ? (M, A) = (m, m - d) : (M, A) = (m + d, m)
// Workgroup barrier for LDS hand-offs between waves that must NOT drain the vector-memory queue (the look-ahead residue
// loads stay in flight across it): wait for this wave's LDS traffic only, then s_barrier. (__syncthreads() would add
// vmcnt(0).)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pairwise hand-off between the two waves of a coupled channel pair, without involving the other waves of the workgroup
// (an s_barrier puts all 8 waves in lockstep: they then stall on the same LDS / HBM phases together and the SIMDs idle).
// Monotonic per-wave counters in LDS. LDS executes one wave's instructions in order, so `post` after a wave's data
// accesses needs no wait, and a partner that has seen the counter sees (or no longer disturbs) that data.
// (Explicit LDS address space: a volatile access through a generic pointer compiles to flat_load/flat_store, whose wait
// also drains the vector-memory queue and with it the look-ahead residue loads.)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ void pair_post(lds_u32* flag, uint32_t v) {
  asm volatile("" ::: "memory");
  *(volatile lds_u32*)flag = v;
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pair_wait(const lds_u32* flag, uint32_t v) {
  asm volatile("" ::: "memory");
  while (*(const volatile lds_u32*)flag < v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// v_min_f32 / v_max_f32 against 0 without the canonicalising v_max x,x the compiler adds in IEEE mode
__device__ __forceinline__ float min0(float a) {
  float r;
  asm("v_min_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ float max0(float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
// hpp:1220-1239 with one comparison per output: the a > 0 test folds into min(a,0) / max(a,0) (adding the resulting
// +-0 returns the other operand unchanged), the m > 0 test picks the sign. Same single add per output, same value.
__device__ __forceinline__ float couple_mag(float m, float a) {  // new magnitude-channel value
  const float na = min0(a);
  return m + (m > 0.f ? na : -na);
}
__device__ __forceinline__ float couple_ang(float m, float a) {  // new angle-channel value
  const float pa = max0(a);
  return m + (m > 0.f ? -pa : pa);
}

// Per-packet metadata through the scalar unit. Left to itself the compiler fetches PktInfo per lane and moves every
// field to an SGPR with v_readfirstlane: it cannot prove that the PCM stores of the loop leave the array alone, so it
// will not use s_load. Reading through the CONSTANT address space states exactly that (the layout kernel of an earlier
// launch wrote the array; nothing writes it while this kernel runs) and turns a wave-uniform access into one
// s_load_dwordx8 that the compiler schedules and waits for itself. Sub-dword fields are unpacked with scalar shifts
// (there are no sub-dword scalar loads).
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
static_assert(sizeof(PktInfo) == 32, "PktInfo is fetched as one s_load_dwordx8");
struct PktScalars {
  uint64_t res_off;
  uint32_t out_pos, emit, used, own, widx, mapping, lng, bad;
};
__device__ __forceinline__ PktScalars pkt_fields(const u32x8 r) {
  PktScalars k;
  k.res_off = (uint64_t)r[0] | ((uint64_t)r[1] << 32);
  k.out_pos = r[2];
  k.emit = r[3];
  k.used = r[4];
  k.own = r[5];
  k.widx = r[6] >> 24;
  k.mapping = r[7] & 0xFFu;
  k.lng = (r[6] >> 16) & 0xFFu;
  k.bad = (r[7] >> 8) & 0xFFu;
  return k;
}
__device__ __forceinline__ PktScalars pkt_load(const PktInfo* p) {  // p wave-uniform
  typedef const __attribute__((address_space(4))) u32x8* const_words;
  return pkt_fields(*(const_words)(uintptr_t)p);
}

enum { K_REG = 0, K_LDS = 1, K_CARRY = 2 };
__device__ __forceinline__ uint32_t bitrev6(uint32_t l) { return __brev(l) >> 26; }

// Packed short pass of the mixed-block path (blocksize0 = 256): up to 8 consecutive short blocks of one mapping at once. A short
// block has 64 complex points, so element (register t, lane l) is point l of packet t: every load is one packet's 512 contiguous
// bytes, the floor look-ups use the same lane -> bin map for every t, and the FFT-64 of all 8 packets is the last two passes of
// the FFT-512 network (DFT-8 over lane bits 5..3, twiddle W64, DFT-8 over lane bits 2..0). Afterwards packet j sits in lanes
// [8j, 8j+8), bin f = (lane & 7) + 8 c' in register c'; the overlap term of packet j is packet j-1's value 8 lanes below.
// One block per iteration (six cross-lane radix-2 stages for one point per lane) spent the same chain of steps on 1/8 of the data.
// Returns the number of packets taken (>= 1). P[] = unwindowed right-half values of the last packet, in the lanes of group 0.
template <int ROLE, bool TAPC>
__device__ __forceinline__ uint32_t fused_short_pass(const FusedArgs& A, const FusedLdsImage& T, float2* __restrict__ xb, const float2* __restrict__ pxb,
                                                     float2* seg2, lds_f32* cbuf, lds_u32* my_flags, const lds_u32* partner_flags, const uint32_t lane,
                                                     const vsyn_segment sg, const SegInfo si, const uint32_t q, const uint32_t qa, const uint32_t qb,
                                                     const uint32_t C, const uint32_t c, float* plane, const PktInfo* ip, const uint32_t epoch,
                                                     const uint32_t prev_kind, const uint32_t prev_half, float (&P)[8], const uint32_t bseg0,
                                                     const uint32_t sidx, const uint32_t xsl, const uint32_t posts, const uint32_t ys_stride, bool& hand_over_out
#ifdef VSYN_STAMPS
                                                     , unsigned long long (&st_acc)[VSYN_NSTAMPS], unsigned long long& st_last
#endif
                                                     ) {
  constexpr uint32_t MS = 128u, ML = 1024u;
  const ConstHeader* H = hdr_of(A.cb);
  // ---- the pass: packets q .. q+Jp-1 (lane j < 8 holds the descriptor of packet q + j) ----------------------------------------------
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t idx = min(q + (lane & 7u), qb - 1u);
  const u32x4 da = ((const u32x4*)(ip + idx))[0], db = ((const u32x4*)(ip + idx))[1];
  // (also the descriptor behind the pass: what follows its last packet)
  const uint32_t map0 = __builtin_amdgcn_readfirstlane(db[3] & 0xFFu);
  const bool cand = lane < 8u && q + lane < qb;
  const uint64_t okm = __ballot(cand && !((db[3] >> 8) & 0xFFu) && !((db[2] >> 16) & 0xFFu) && (db[3] & 0xFFu) == map0);
  const uint32_t Jp = (uint32_t)__builtin_ctzll(~okm);  // >= 1: the caller saw a valid short block at q
  STAMP(9);   // short pass: descriptors arrived
  const uint32_t qn = q + Jp, p0 = sg.first_packet + q;
  const uint32_t emit_l = q + (lane & 7u) < qa ? 0u : da[3];  // the halo emits nothing
  // ---- loads -------------------------------------------------------------------------------------------------------------------------
  float2 raw[8];
  uint32_t vrow[8];
#pragma unroll
  for (uint32_t t = 0; t < 8; ++t) {
    const uint32_t tc = min(t, Jp - 1u);
    const uint64_t off = ((uint64_t)__builtin_amdgcn_readlane(da[1], tc) << 32) | (uint32_t)__builtin_amdgcn_readlane(da[0], tc);
    raw[t] = (VSYN_KNOCKOUT & 256) ? f2(0.001f * (float)lane, 0.002f) : ((const float2*)(A.residue + off + (size_t)c * MS))[lane];
    vrow[t] = (A.fy + ((size_t)(p0 + tc) * C + c) * ys_stride)[sidx];
  }
#pragma unroll
  for (uint32_t t = 0; t < 8; ++t)
    if (t >= Jp) raw[t] = f2(0.f, 0.f);
  // ---- inverse coupling through the partner's image ----------------------------------------------------------------------------------
  float2 r[8];
  if (ROLE != 0 && !(VSYN_KNOCKOUT & 512)) {
#pragma unroll
    for (int t = 0; t < 8; ++t) xb[t * 64 + lane] = raw[t];
    STAMP(10);  // short pass: rows arrived, written to the exchange image
    pair_post(&my_flags[0], epoch);
    pair_wait(&partner_flags[0], epoch);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float2 oth = pxb[t * 64 + lane];
      r[t] = ROLE == 1 ? f2(couple_mag(raw[t].x, oth.x), couple_mag(raw[t].y, oth.y)) : f2(couple_ang(oth.x, raw[t].x), couple_ang(oth.y, raw[t].y));
    }
    pair_post(&my_flags[1], epoch);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = raw[t];
  }
  // ---- floor curve + product (the lane -> bin map is the same for every packet: bins 2 lane, 2 lane + 1) --------------------------------
  // Two phases over the pass's packets, so that the eight set-ups (ballot, two bpermutes, table write) and then the eight look-up
  // chains (entry, inverse-dB value) overlap each other instead of running back to back: the eight tables live in the exchange
  // image, which is idle between the hand-off and the FFT (one table per packet, 512 bytes each).
  if (ROLE != 0 && !(VSYN_KNOCKOUT & 512)) pair_wait(&partner_flags[1], epoch);  // the partner has read this wave's image: it may be reused
  bool floor_bad = false;
  uint32_t floor_bad_pkt = 0;
  uint32_t nocurve_mask = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; ++t) {
    if (t >= Jp || (VSYN_KNOCKOUT & 64)) break;
    float2* const tab = xb + 64u * t;
    const uint32_t own_t = __builtin_amdgcn_readlane(db[1], t), used_t = __builtin_amdgcn_readlane(db[0], t);
    if (!((own_t >> c) & 1u)) {
      nocurve_mask |= 1u << t;
      tab[lane] = f2(0.f, ((used_t >> c) & 1u) ? 256.5f : 255.5f);
    } else {
      uint32_t v = vrow[t];
      if (lane >= posts) v = 0;
      const uint64_t mask = __ballot((v >> 15) != 0) | 1ull;
      const uint64_t below = mask & ((2ull << lane) - 1ull);
      const uint32_t lo = 63u - (uint32_t)__clzll((long long)below);
      const uint64_t above = lane < 63u ? (mask >> (lane + 1u)) : 0ull;
      const bool has_hi = above != 0ull;
      const uint32_t hi = lane + (uint32_t)__ffsll((long long)above);
      const uint32_t packed = (xsl << 16) | (v & 0x7FFFu);
      const uint32_t plo = (uint32_t)__shfl((int)packed, (int)lo);
      const uint32_t phi = (uint32_t)__shfl((int)packed, (int)(has_hi ? hi : lo));
      if ((v & 0x7FFFu) > 255u) {
        floor_bad = true;
        floor_bad_pkt = p0 + t;
      }
      const float x0 = (float)(plo >> 16), y0 = fminf((float)(plo & 0xFFFFu), 255.f);
      const float x1 = (float)(phi >> 16), y1 = fminf((float)(phi & 0xFFFFu), 255.f);
      const float inv = has_hi ? __builtin_amdgcn_rcpf(x1 - x0) : 0.f;
      const float ady = fabsf(y1 - y0);
      const float a = ady * inv, b = __builtin_fmaf(-ady, x0, 0.5f) * inv;
      tab[lane] = y1 >= y0 ? f2(a, b + y0) : f2(-a, (y0 + 1.f) - b);
    }
  }
  if (!(VSYN_KNOCKOUT & 64)) {
    typedef float lds_vf2 __attribute__((ext_vector_type(2)));
    const uint32_t seg_base = (uint32_t)(uintptr_t)(lds_u32*)seg2, xb_base = (uint32_t)(uintptr_t)(lds_u32*)xb;
    // bseg0 holds the entry addresses inside the wave's own floor table: rebase them onto table t of the image
    const uint32_t o0 = (bseg0 & 0xFFFFu) - seg_base + xb_base, o1 = (bseg0 >> 16) - seg_base + xb_base;
    const float xf = (float)(2u * lane);
    uint32_t i0[8], i1[8];
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) {
      if (t >= Jp) break;
      const lds_vf2 e0 = *(const __attribute__((address_space(3))) lds_vf2*)(uintptr_t)(o0 + 512u * t);
      const lds_vf2 e1 = *(const __attribute__((address_space(3))) lds_vf2*)(uintptr_t)(o1 + 512u * t);
      i0[t] = (uint32_t)__builtin_fmaf(xf, e0.x, e0.y);
      i1[t] = (uint32_t)__builtin_fmaf(xf + 1.f, e1.x, e1.y);
    }
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) {
      if (t >= Jp) break;
      r[t] = f2(r[t].x * T.invdb[i0[t]], r[t].y * T.invdb[i1[t]]);
      if (TAPC && !((nocurve_mask >> t) & 1u)) {  // feature tap "floor1 floor" (hpp:585): the table indices are the rendered curve
        const uint64_t off = ((uint64_t)__builtin_amdgcn_readlane(da[1], t) << 32) | (uint32_t)__builtin_amdgcn_readlane(da[0], t);
        ((uint32_t*)(A.curve + off + (size_t)c * MS))[lane] = i0[t] | (i1[t] << 16);
      }
    }
  }
  STAMP(11);  // short pass: coupling (partner waits), floor set-up and product
  // ---- IMDCT x 8 ------------------------------------------------------------------------------------------------------------------------
  float2 z[8];
  {
    const float2 pre = T.pre_s[lane];
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = cmulf(f2(r[t].x, __shfl(r[t].y, 63 - (int)lane)), pre);  // X[127 - 2k] lives in lane 63-k
  }
  if (!(VSYN_KNOCKOUT & 128)) {
    const uint32_t cl = lane & 7u, hi = lane >> 3;
#pragma unroll
    for (int t = 0; t < 8; ++t) xb[t * 72 + lane] = z[t];
#pragma unroll
    for (int a = 0; a < 8; ++a) z[a] = xb[hi * 72 + a * 8 + cl];
    dft8(z);
#pragma unroll
    for (int a = 1; a < 8; ++a) z[a] = cmulf(z[a], T.tw2[a][cl]);  // W64^(c a')
#pragma unroll
    for (int a = 0; a < 8; ++a) xb[a * 65 + lane] = z[a];
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = xb[cl * 65 + hi * 8 + k];
    dft8(z);
  }
  const uint32_t kap = lane & 7u, jo = lane >> 3;
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = cmulf(z[k], T.post8[k][kap]);
  // ---- window + overlap-add + PCM --------------------------------------------------------------------------------------------------------
  const bool valid_o = jo < Jp, grp0 = jo == 0u;
  const uint32_t emit = valid_o ? (uint32_t)__shfl((int)emit_l, (int)jo) : 0u;
  float* const out = plane + (uint32_t)__shfl((int)da[2], (int)min(jo, Jp - 1u));
  const uint32_t shift = (grp0 && prev_half == ML) ? 448u : 0u;  // after a long block the chunk starts with its 448 frames
  float wl0[8], wl1[8], wr0[8], wr1[8], Pin[8], Pn[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wl0[k] = T.wsl8[0][k][kap];
    wl1[k] = T.wsl8[1][k][kap];
    wr0[k] = wl0[k];
    wr1[k] = wl1[k];
    Pn[k] = k >= 4 ? z[k].y : -z[k].x;
  }
  {
    const int below = (int)((lane - 8u) & 63u);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float sh = __shfl(Pn[k], below);
      Pin[k] = grp0 ? P[k] : sh;
    }
  }
  if (prev_kind == K_LDS && grp0) {  // after a long block / a carry-in: the 128 windowed overlap frames are in the carry image
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t f = kap + 8u * k;
      const uint32_t s = k >= 4 ? 2u * f - 64u : 63u - 2u * f;
      Pin[k] = 1.f;
      wr1[k] = cbuf[s];
      wr0[k] = cbuf[MS - 1u - s];
    }
  }
  // what follows the pass (descriptor behind it, if it was among the eight fetched; else a scalar fetch)
  const bool last_of_segment = qn == sg.num_packets;
  bool hand_over = false;
  uint32_t next_emit = 0, next_out = 0;
  if (qn < qb) {
    uint32_t nlng, nbad;
    if (Jp < 8u) {
      nlng = __builtin_amdgcn_readlane((db[2] >> 16) & 0xFFu, Jp);
      nbad = __builtin_amdgcn_readlane((db[3] >> 8) & 0xFFu, Jp);
      next_emit = __builtin_amdgcn_readlane(da[3], Jp);
      next_out = __builtin_amdgcn_readlane(da[2], Jp);
    } else {
      const PktScalars nx = pkt_load(ip + qn);
      nlng = nx.lng;
      nbad = nx.bad;
      next_emit = nx.emit;
      next_out = nx.out_pos;
    }
    hand_over = !nbad && nlng;
  }
  STAMP(12);  // short pass: FFT, post-rotation, window reads
  vmem_drain();  // loads only are in flight: finish them before the PCM stores (see vmem_drain)
  float oh_s[4], oh_m[4], n_s[4], n_m[4];
  const int mirror = (int)(lane ^ 7u);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kh = 4 + j, kl = 3 - j;
    const float cch = z[kh].x, ccl = -z[kl].y;
    const float ah_s = Pin[kh] * wr1[kh], ah_m = Pin[kh] * wr0[kh], al_s = Pin[kl] * wr1[kl], al_m = Pin[kl] * wr0[kl];
    oh_s[j] = ah_s + cch * wl0[kh];
    oh_m[j] = ah_m + (-cch) * wl1[kh];
    const float ol_s = al_s + ccl * wl0[kl];
    const float ol_m = al_m + (-ccl) * wl1[kl];
    n_s[j] = __shfl(ol_s, mirror);
    n_m[j] = __shfl(ol_m, mirror);
  }
  if (!(VSYN_KNOCKOUT & 32) || oh_s[0] == 1234.567f) {
    const bool whole = emit == MS + shift && (((uintptr_t)out & 7u) == 0);
    if (__all(whole || !valid_o)) {
      if (valid_o) {
        float* up = out + shift + 2u * kap;
        float* dn = out + shift + MS - 2u - 2u * kap;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (plain stores: a packet gets 64 contiguous bytes per instruction here, half a cache line — as streaming stores those
          // partial lines cost config 4 9 %; the long blocks' 512-byte rows are what the non-temporal hint is for)
          *(float2*)(up + 16 * j) = f2(oh_s[j], n_s[j]);
          *(float2*)(dn - 16 * j) = f2(n_m[j], oh_m[j]);
        }
      }
    } else if (emit) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t s = 2u * kap + 16u * j;
        const uint32_t f0 = s + shift, f1 = s + 1u + shift, f2m = MS - 2u - s + shift, f3m = MS - 1u - s + shift;
        if (f0 < emit) out[f0] = oh_s[j];
        if (f1 < emit) out[f1] = n_s[j];
        if (f2m < emit) out[f2m] = n_m[j];
        if (f3m < emit) out[f3m] = oh_m[j];
      }
    }
  }
  (void)next_emit;
  (void)next_out;
  if ((last_of_segment || hand_over) && jo == Jp - 1u) {
    // windowed right half of the last packet in natural order: into the stream's carry buffer (segment end) or the carry image
    // (a long block follows: its samples 448..575 meet these 128)
    const size_t carry_half = (size_t)H->max_streams * C * ML;
    float* cout = A.carry + (si.parity_in ^ 1u) * carry_half + ((size_t)sg.stream * C + c) * ML;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t f = kap + 8u * k;
      const uint32_t s = k >= 4 ? 2u * f - 64u : 63u - 2u * f, sm = MS - 1u - s;
      const float v_s = Pn[k] * wl1[k], v_m = Pn[k] * wl0[k];
      if (last_of_segment) {
        cout[s] = v_s;
        cout[sm] = v_m;
      } else {
        cbuf[s] = v_s;
        cbuf[sm] = v_m;
      }
    }
  }
  {
    const int from = (int)((Jp - 1u) * 8u + kap);
#pragma unroll
    for (int k = 0; k < 8; ++k) P[k] = __shfl(Pn[k], from);
  }
  if (__any(floor_bad)) {
    if (floor_bad) raise_status(A.status, VSYN_ST_FLOOR_VALUE, floor_bad_pkt);
    vmem_drain();
  }
  hand_over_out = hand_over;
  return Jp;
}

// The whole per-wave job: the run [qa, qb) of segment g, output channel c.
// ROLE: 0 channel c is not coupled; 1 c is the magnitude channel of the (single) coupling step; 2 c is the angle channel.
// A coupled wave gets the partner channel's residue from the partner wave's exchange image (`pxb`) and keeps only its own
// side of hpp:1219-1240 (5 VALU per bin instead of 7 for both).
//
// MIXED = false: every block of the run (halo included) is long and there is no carry-in — the steady state, nothing below
// about short blocks is compiled in.  MIXED = true: short blocks (n = 256: one complex point per lane, FFT-64 as six
// cross-lane radix-2 stages), window switches and a carry-in from an earlier submit.  The overlap term of a block comes
//   K_REG     from registers when the previous block had the same size (P[8] / Ps, unwindowed, same lane layout),
//   K_LDS     from the wave's 128-float carry image when the size changed: a long block followed by a short one writes the
//             448 frames that only it contributes to straight into the next chunk and its 128 windowed overlap frames into the
//             image; a short block followed by a long one writes its 128 windowed right-half samples there; the next block
//             reads the image in its own lane layout.  Every chunk sample is stored exactly once (no read-modify-write
//             through memory, no fences),
//   K_CARRY   from the stream's carry buffer (windowed right half in natural order) for a long block after a long carry-in.
// `buf = 0; buf += prev*w; buf += cur*w` (hpp:1008-1017) with the same two roundings in every case.

template <int ROLE, bool MIXED, bool TAPC>
__device__ __forceinline__ void fused_run(const FusedArgs& A, const FusedLdsImage& T, float2* __restrict__ xb, const float2* __restrict__ pxb, float4* __restrict__ seg,
                                          lds_f32* cbuf, lds_u32* my_flags, const lds_u32* partner_flags, const uint32_t lane0, const uint32_t g,
                                          const vsyn_segment sg, const SegInfo si, const uint32_t qa, const uint32_t qb, const uint32_t C, const uint32_t c) {
  constexpr uint32_t ML = 1024;
  const uint8_t* __restrict__ cb = A.cb;
  const ConstHeader* H = hdr_of(cb);
  const uint32_t num = sg.num_packets;

  float P[8];  // overlap carry of the run: -u_prev[511 - s] per point, unwindowed
#pragma unroll
  for (int k = 0; k < 8; ++k) P[k] = 0.f;
  uint32_t prev_next_long = 1;
  uint32_t prev_kind = K_REG;   // MIXED: where the overlap term comes from (wave-uniform)
  uint32_t prev_half = 0;       // MIXED: samples the previous block contributes to the current chunk's frame count: 1024 / 128 / 0
  float* const plane = A.pcm + ((size_t)g * C + c) * A.plane_stride;
  const float* cin = nullptr;   // MIXED: carry-in of this (stream, channel)

  // lane constants of the floor in use (reloaded only when the floor changes, wave-uniform)
  const uint32_t seg_base = (uint32_t)(uintptr_t)(lds_u32*)seg;  // this wave's entry table inside LDS (entries are 8 bytes)
  uint32_t bseg[8];  // LDS address of the segment entry of each of this lane's 16 bins, 16 bits each; until a floor is seen:
                     // entry 0, which is where a channel without a curve finds its constant entry
#pragma unroll
#ifdef VSYN_BSEG_PACK
  for (int t = 0; t < 8; ++t) bseg[t] = 0u;
#else
  for (int t = 0; t < 8; ++t) bseg[t] = seg_base | (seg_base << 16);
#endif
  uint32_t sidx = 0, xsl = 0;       // header index / x of sorted post `lane`
  int cur_floor = -1;
  uint32_t vrow = 0;
  bool vrow_ok = false;

  // loop-invariant header fields, read once (inside the loop they would be re-fetched per packet: the stores in between
  // keep the compiler from hoisting them)
  const uint32_t ys_stride = __builtin_amdgcn_readfirstlane(H->ys_stride);
  const MapConst* const maps = (const MapConst*)(cb + H->off_map);
  const FloorConst* const floors = (const FloorConst*)(cb + H->off_floor);
  uint32_t cur_map = 0xFFFFFFFFu, map_floor = 0, posts = 0;

  const uint32_t q0 = qa ? qa - 1 : 0;
  const PktInfo* const ip = A.info + __builtin_amdgcn_readfirstlane(sg.first_packet);
  PktScalars pi = pkt_load(ip + q0);
  float2 raw[8];  // own channel's residue, requested one packet ahead (mixed runs: when the next block is a long one, too)
  bool raw_ahead = false;  // MIXED: raw[] already holds (or will hold) this packet's residue
  // MIXED, long blocks: the residue rows come into the hand-off image by LDS-DMA (global_load_lds: no registers), and the NEXT long
  // block's rows are requested as soon as this block's FFT has released the image — the 128-VGPR mixed path has no registers for a
  // look-ahead, and without one every long block of a mixed run waited a full memory latency.
  bool dma_ahead = false;     // this packet's rows are already in flight towards (or in) the image
  uint32_t dma_younger = 0;   // vector-memory operations issued behind that request: 8 = exactly the fast path's PCM stores
  auto issue_dma = [&](uint64_t off, const uint32_t ln) {  // ln: the laundered lane number of the iteration (a pointer built from
                                                           // lane0 is hoisted out of the packet loop and ends up spilled)
    const char* gsrc = (const char*)(A.residue + off) + 16u * ln;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + 1024 * i),
                                       (__attribute__((address_space(3))) void*)((lds_u32*)xb + 256 * i), 16, 0, 0);
  };
  if (!MIXED) {
    const float2* src = (const float2*)(A.residue + pi.res_off + (size_t)c * ML) + lane0;
#pragma unroll
    for (int t = 0; t < 8; ++t) raw[t] = stream_load2(src + 64 * t);
  }
  if (MIXED && qa == 0 && si.has_carry && !pi.bad) {
    // the previous submit left this stream's windowed right half in natural order (carry_n / 2 samples)
    const size_t carry_half = (size_t)H->max_streams * C * ML;
    cin = A.carry + si.parity_in * carry_half + ((size_t)sg.stream * C + c) * ML;
    const bool carry_long = si.carry_n == 2u * ML;
    prev_half = si.carry_n / 2u;
    if (carry_long && pi.lng) {
      prev_kind = K_CARRY;
    } else {
      // a size change (or two short blocks): the 128 overlap frames go through the carry image; a long carry before a
      // short block also owns the first 448 frames of the chunk outright
      const uint32_t base = carry_long ? 448u : 0u;
      if (carry_long)
        for (uint32_t i = lane0; i < min(448u, pi.emit); i += 64) plane[pi.out_pos + i] = cin[i];
      cbuf[lane0] = cin[base + lane0];
      cbuf[lane0 + 64u] = cin[base + lane0 + 64u];
      prev_kind = K_LDS;
    }
  }
  uint32_t lane_v = lane0;
  vmem_drain();  // see vmem_drain(): the loop must not inherit "the residue registers were loaded last" from here
#ifdef VSYN_STAMPS
  unsigned long long st_acc[VSYN_NSTAMPS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_readcyclecounter();
#endif
  // The two waves of a coupled channel pair (adjacent waves, same run, same packets) each load ONLY their own channel
  // from HBM and hand it to the partner through their exchange image, which is idle at that point: loading both channels
  // in both waves costs a second HBM fetch of the whole input (measured: concurrent misses on a line are not merged;
  // FETCH_SIZE x2 = 1.10 GB vs 0.55 GB per launch). my_flags[0] = "my image holds packet #n's residue",
  // my_flags[1] = "I have read the partner's image of packet #n".
  uint32_t q = q0;
  for (uint32_t it = 0; q < qb; ++it, ++q) {  // (a packed short pass advances q by more than one: `it` counts hand-off epochs)
    STAMP(0);  // loop overhead / previous iteration's tail
    // launder the lane id once per packet: keeps the lane-derived LDS/global addresses from being hoisted out of
    // the loop and pinned in VGPRs for its whole duration (recomputing them costs a few VALU ops)
    asm volatile("" : "+v"(lane_v));
    const uint32_t lane = lane_v;
    const uint32_t kappa = ((lane & 7u) << 3) | (lane >> 3);
    const uint32_t p = sg.first_packet + q;
    const bool halo = q < qa, has_next = q + 1 < qb;
    const PktScalars pin = pkt_load(ip + (has_next ? q + 1 : q));  // one packet ahead
    const bool lng = !MIXED || pi.lng != 0u;
    const bool nlng = !MIXED || pin.lng != 0u;
    const uint32_t M = lng ? ML : 128u;
    if (MIXED && pi.bad) {
      // invalid mode number: the layout kernel flagged it; nothing to synthesise (outputs after it are unspecified). Both
      // waves of a pair skip it, so the hand-off counters stay in step.
#pragma unroll
      for (int k = 0; k < 8; ++k) P[k] = 0.f;
      prev_kind = K_REG;
      prev_half = 0;
      vrow_ok = false;
      pi = pin;
      continue;
    }
    if (pi.mapping != cur_map) {  // wave-uniform, rare
      cur_map = pi.mapping;
      {  // (through the scalar unit: one dword of the channel -> floor bytes; a per-lane byte load keeps its address pair in VGPRs)
        typedef const __attribute__((address_space(4))) uint32_t* kptr;
        const uint32_t w = *(kptr)(uintptr_t)((const uint8_t*)maps[cur_map].chfloor + (c & ~3u));
        map_floor = (w >> (8u * (c & 3u))) & 0xFFu;
      }
    }
    if (MIXED && !lng) {
      // ---- short blocks: up to eight consecutive ones as one pass (fused_short_pass) ---------------------------------------------
      if (cur_floor != (int)map_floor) {  // wave-uniform: the short floor's lane constants
        const FloorConst* fc = floors + map_floor;
        posts = __builtin_amdgcn_readfirstlane(fc->posts);
        const uint8_t* bs = A.binseg + (size_t)map_floor * ML;
        const uint32_t seg_base0 = (uint32_t)(uintptr_t)(lds_u32*)seg;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const uint32_t two = *(const uint16_t*)(bs + 2u * (lane + 64u * t));
          bseg[t] = (seg_base0 + 8u * (two & 0xFFu)) | ((seg_base0 + 8u * (two >> 8)) << 16);
        }
        const bool in = lane < posts;
        sidx = in ? fc->sorted_idx[lane] : 0u;
        xsl = in ? fc->xs_sorted[lane] : 0u;
        cur_floor = (int)map_floor;
        vmem_drain();
      }
      bool ho = false;
      const uint32_t Jp = fused_short_pass<ROLE, TAPC>(A, T, xb, pxb, (float2*)seg, cbuf, my_flags, partner_flags, lane, sg, si, q, qa, qb, C, c, plane, ip, it + 1u,
                                                 prev_kind, prev_half, P, bseg[0], sidx, xsl, posts, ys_stride, ho
#ifdef VSYN_STAMPS
                                                 , st_acc, st_last
#endif
                                                 );
      prev_kind = ho ? K_LDS : K_REG;
      prev_half = 128u;
      prev_next_long = 0u;
      vrow_ok = false;
      q += Jp - 1u;
      pi = pkt_load(ip + min(q + 1u, qb - 1u));
      STAMP(8);  // (mixed runs: a packed short pass, whole)
      continue;
    }

    // ---- residue: bins (2k, 2k+1), k = lane + 64 t (requested one packet ahead, see below); inverse coupling keeps
    //      this wave's side only (hpp:1213-1241) ----------------------------------------------------------------
    float2 r[8];
    // L = rows of the block: all 8 (long) or the first (short); called with a literal so that each copy is straight-line code
    auto residue_rows = [&](const bool L) {
#ifndef VSYN_NO_MIXED_DMA
      if (MIXED && L) {
        if (!dma_ahead) {
          issue_dma(pi.res_off + (uint64_t)c * ML, lane);
          dma_younger = 0;
        }
        if (dma_younger == 8u) __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8): the rows have landed; the 8 PCM stores behind them may still fly
        else vmem_drain();
        dma_ahead = false;
        if (ROLE != 0) {
          pair_post(&my_flags[0], it + 1);
          pair_wait(&partner_flags[0], it + 1);  // both channels of the pair are in LDS
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float2 own = xb[t * 64 + lane];
          if (ROLE == 0) {
            r[t] = own;
          } else {
            const float2 oth = pxb[t * 64 + lane];
            r[t] = ROLE == 1 ? f2(couple_mag(own.x, oth.x), couple_mag(own.y, oth.y)) : f2(couple_ang(oth.x, own.x), couple_ang(oth.y, own.y));
          }
        }
        return;
      }
#endif
      if (MIXED && !raw_ahead) {
        const float2* src = (const float2*)(A.residue + pi.res_off + (size_t)c * (L ? ML : 128u)) + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t == 0 || L) raw[t] = src[64 * t];
      }
      if (ROLE != 0 && !(VSYN_KNOCKOUT & 1)) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t == 0 || L) xb[t * 64 + lane] = raw[t];
        pair_post(&my_flags[0], it + 1);
        pair_wait(&partner_flags[0], it + 1);  // both channels of the pair are in LDS
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t != 0 && !L) continue;
        if (ROLE == 0 || (VSYN_KNOCKOUT & 1)) {
          r[t] = raw[t];
        } else {
          const float2 oth = pxb[t * 64 + lane];
          r[t] = ROLE == 1 ? f2(couple_mag(raw[t].x, oth.x), couple_mag(raw[t].y, oth.y))
                           : f2(couple_ang(oth.x, raw[t].x), couple_ang(oth.y, raw[t].y));
        }
      }
    };
    if (lng) residue_rows(true);
    else residue_rows(false);
    STAMP(1);  // residue arrival (vmcnt), hand-off through LDS, partner wait, coupling
    if (ROLE != 0 && !(VSYN_KNOCKOUT & 1)) pair_post(&my_flags[1], it + 1);
    // `raw` is dead: request packet q+1 now, so that its 4 KiB stay in flight behind this packet's floor product, FFT and
    // overlap (memory-level parallelism bounded this kernel, not occupancy). Unconditional on purpose: on a run's last
    // packet the current block is re-read (cache-resident, 1/R of the loads) — a `has_next` guard lets the compiler fold
    // these loads back into the loop header.
    if (!MIXED) {
      const float2* src = (const float2*)(A.residue + pin.res_off + (size_t)c * ML) + lane;  // one 64-bit add, immediate offsets
#pragma unroll
      for (int t = 0; t < 8; ++t) raw[t] = stream_load2(src + 64 * t);
    } else {
#ifdef VSYN_MIXED_PREFETCH
      // Measured (config 4): the 16 registers this keeps live across the floor product, FFT and overlap do not exist in the 128-VGPR
      // mixed path — 224 B/lane of scratch, kernel 0.068 -> 0.121 ms. Off; it needs a launch of its own at 3 waves per SIMD.
      raw_ahead = has_next && nlng && !pin.bad;  // (this point is only reached by long blocks: short ones take fused_short_pass)
      if (raw_ahead) {
        const float2* src = (const float2*)(A.residue + pin.res_off + (size_t)c * ML) + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t) raw[t] = src[64 * t];
      }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- floor-1 step 2 set-up: one table entry per sorted-post interval (hpp:563-584) ---------------------
    bool floor_bad = false;
    float2* const seg2 = (float2*)seg;  // this kernel's entries are 8 bytes: table index = floor(x * e.x + e.y)
    if (VSYN_KNOCKOUT & 16) {
      if (it == 0) seg2[lane] = f2(0.f, 200.5f);
    } else if (!((pi.own >> c) & 1u)) {
      // no curve of its own: one constant entry. Not used at all -> index 255 (table value exactly 1.0f, x*1 == x);
      // used through the coupling propagate -> index 256 (0.0f: floor_outputs stays zero, hpp:1159,1176-1179)
      seg2[lane] = f2(0.f, ((pi.used >> c) & 1u) ? 256.5f : 255.5f);
    } else {
      const uint32_t f = map_floor;
      const uint16_t* row = A.fy + ((size_t)p * C + c) * ys_stride;
      uint32_t v = vrow;
      if (cur_floor != (int)f) {  // wave-uniform, changes only when the mapping changes
        const FloorConst* fc = floors + f;
        posts = __builtin_amdgcn_readfirstlane(fc->posts);
        const uint8_t* bs = A.binseg + (size_t)f * ML;
#pragma unroll
#ifdef VSYN_BSEG_PACK
        for (int t = 0; t < 8; t += 2) {
          const uint32_t two0 = *(const uint16_t*)(bs + 2u * (lane + 64u * t)), two1 = *(const uint16_t*)(bs + 2u * (lane + 64u * (t + 1)));
          bseg[t >> 1] = two0 | (two1 << 16);
        }
#else
        for (int t = 0; t < 8; ++t) {
          const uint32_t two = *(const uint16_t*)(bs + 2u * (lane + 64u * t));  // intervals of bins 2k, 2k+1 (k = lane + 64 t)
          bseg[t] = (seg_base + 8u * (two & 0xFFu)) | ((seg_base + 8u * (two >> 8)) << 16);
        }
#endif
        const bool in = lane < posts;
        sidx = in ? fc->sorted_idx[lane] : 0u;
        xsl = in ? fc->xs_sorted[lane] : 0u;
        cur_floor = (int)f;
        v = row[sidx];
        vmem_drain();
      } else if (!vrow_ok) {
        v = row[sidx];
        vmem_drain();
      }
      if (lane >= posts) v = 0;
      const uint64_t mask = __ballot((v >> 15) != 0) | 1ull;
      const uint64_t below = mask & ((2ull << lane) - 1ull);  // flagged positions <= lane (bit 0 always set)
      const uint32_t lo = 63u - (uint32_t)__clzll((long long)below);
      const uint64_t above = lane < 63u ? (mask >> (lane + 1u)) : 0ull;
      const bool has_hi = above != 0ull;
      const uint32_t hi = lane + (uint32_t)__ffsll((long long)above);
      const uint32_t packed = (xsl << 16) | (v & 0x7FFFu);
      const uint32_t plo = (uint32_t)__shfl((int)packed, (int)lo);
      const uint32_t phi = (uint32_t)__shfl((int)packed, (int)(has_hi ? hi : lo));
      // segment (x0,y0)-(x1,y1) in slope/intercept form: curve(x) = y0 + sgn * floor(u), u = x*A + B with
      // A = |dy|/adx, B = (0.5 - |dy| x0)/adx  ==  y0 +- (|dy| (x - x0)) / adx  in integers (Utils.hpp:122-137).
      // u is an odd multiple of 1/(2 adx), never an integer, so -floor(u) = floor(-u) + 1 and both directions fold into
      //   curve(x) = floor(x * A' + B'),  rising: A' = A, B' = B + y0;  falling: A' = -A, B' = y0 + 1 - B
      // (one fma + one conversion per bin, 8-byte entries). The 0.5/adx guard band dwarfs the f32 rounding, DESIGN.md
      floor_bad = (v & 0x7FFFu) > 255u;  // a post above 255 can only render >= 256 (hpp:587)
      const float x0 = (float)(plo >> 16), y0 = fminf((float)(plo & 0xFFFFu), 255.f);
      const float x1 = (float)(phi >> 16), y1 = fminf((float)(phi & 0xFFFFu), 255.f);
      // 1/adx: v_rcp_f32 (1 ulp) is enough — the guard band above leaves 8x headroom over the total rounding
      const float inv = has_hi ? __builtin_amdgcn_rcpf(x1 - x0) : 0.f;
      const float ady = fabsf(y1 - y0);
      const float a = ady * inv, b = __builtin_fmaf(-ady, x0, 0.5f) * inv;
      seg2[lane] = y1 >= y0 ? f2(a, b + y0) : f2(-a, (y0 + 1.f) - b);
    }
    STAMP(2);  // next packet's loads issued, floor set-up (coded row arrival, ballot, two bpermutes, entry write)
    // coded posts of packet q+1, one packet ahead (valid if the floor does not change)
    vrow_ok = has_next && pin.mapping == pi.mapping && cur_floor >= 0;
    vrow = (A.fy + ((size_t)(has_next ? p + 1 : p) * C + c) * ys_stride)[sidx];

    // ---- floor curve at this lane's 16 bins + product (hpp:585-589, 1243-1255) -----------------------------
    // (a channel without a curve was given a constant x1.0 / x0.0 entry above: no branch here)
    // Two dependent LDS look-ups per bin (segment entry, then inverse-dB value). Done four bins at a time so that a
    // group's look-ups are in flight together: one exposed LDS round trip per group instead of one per bin.
    auto floor_product = [&](const bool L) {
      const float xf0 = (float)(2u * lane);
      float fl[16];
      uint32_t ci[16];  // TAPC: the table indices = the rendered curve
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        if (grp != 0 && !L) continue;  // a short block: bins 2 lane, 2 lane + 1 only (the others of group 0 are computed and unused)
        float2 sgm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = 4 * grp + i;
#ifdef VSYN_BSEG_PACK
          // four interval numbers per register (half the registers, one more VALU op per bin): the steady path alone then needs 122
          // VGPRs instead of 126 — still above the 120 at which a 32-VGPR pre-kernel wave could co-reside; off
          const uint32_t w4 = (b & 2) ? (bseg[b >> 2] >> 16) : bseg[b >> 2];
          const uint32_t addr = seg_base + 8u * ((b & 1) ? ((w4 >> 8) & 0xFFu) : (w4 & 0xFFu));
#else
          const uint32_t addr = (b & 1) ? (bseg[b >> 1] >> 16) : (bseg[b >> 1] & 0xFFFFu);  // one VALU op per bin
#endif
          typedef float lds_vf2 __attribute__((ext_vector_type(2)));
          const lds_vf2 ev = *(const __attribute__((address_space(3))) lds_vf2*)(uintptr_t)addr;
          sgm[i] = f2(ev.x, ev.y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = 4 * grp + i;
          // the argument is positive (it exceeds a table index >= 0), so the conversion's truncation is the floor
          const uint32_t ix = (uint32_t)__builtin_fmaf(xf0 + (float)(128 * (b >> 1) + (b & 1)), sgm[i].x, sgm[i].y);
          fl[b] = T.invdb[ix];
          if (TAPC) ci[b] = ix;
        }
      }
      if (TAPC && ((pi.own >> c) & 1u)) {
        // feature tap "floor1 floor" (hpp:585): the integer curve of this channel's bins, u16, packed like the residue — a lane's
        // bins 2k, 2k+1 are one 32-bit word, a wave-instruction writes 256 contiguous bytes. Channels without a curve of their
        // own are left alone, as in the staged kernels.
        uint32_t* dst = (uint32_t*)(A.curve + pi.res_off + (size_t)c * (L ? ML : 128u)) + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t == 0 || L) dst[64 * t] = ci[2 * t] | (ci[2 * t + 1] << 16);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t == 0 || L) r[t] = f2(r[t].x * fl[2 * t], r[t].y * fl[2 * t + 1]);
    };

    const uint32_t emit = halo ? 0u : pi.emit;
    float* out = plane + pi.out_pos;
    const uint32_t cur_next_long = (pi.widx >> 1) & 1u;
    const bool last_of_segment = q == num - 1;
    // MIXED: does this wave go on to the next packet with a block of the other size? (then the right half changes hands
    // through the carry image; at the end of the run the next run's wave recomputes this block as its halo)
    const bool hand_over = MIXED && has_next && !pin.bad && (nlng != lng);
    if (lng) {
      if (!(VSYN_KNOCKOUT & 2)) floor_product(true);
      STAMP(3);  // floor curve look-ups + product
      // ---- IMDCT: mirror exchange, pre-rotation, FFT-512, post-rotation -------------------------------------
      float2 z[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float im = __shfl(r[7 - t].y, 63 - (int)lane);  // X[1023 - 2k] lives in the mirror lane, slot 7-t
        z[t] = cmulf(f2(r[t].x, im), T.pre[t][lane]);
      }
      STAMP(4);  // mirror exchange + pre-rotation
      if (ROLE != 0 && !(VSYN_KNOCKOUT & 1)) pair_wait(&partner_flags[1], it + 1);  // the partner has read this wave's image: the FFT may reuse it
      STAMP(5);  // second partner wait
#ifdef VSYN_EXP_SETPRIO
      __builtin_amdgcn_s_setprio(VSYN_EXP_SETPRIO);
#endif
      if (!(VSYN_KNOCKOUT & 4)) fft512_wave(z, xb, &T, lane);
#ifdef VSYN_EXP_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      STAMP(6);  // FFT-512
#ifndef VSYN_NO_MIXED_DMA
      if (MIXED) {
        // the image is free until the next packet's hand-off (the partner finished with it before the FFT): request the next long
        // block's rows into it now
        dma_ahead = has_next && nlng && !pin.bad;
        dma_younger = 0xFFu;
        if (dma_ahead) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own reads of the image are done
          issue_dma(pin.res_off + (uint64_t)c * ML, lane);
          dma_younger = 0;
        }
      }
#endif
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = cmulf(z[k], T.post[k][lane]);

      // ---- window + overlap-add + PCM store (hpp:1008-1059) --------------------------------------------------
      // point m = kappa + 64k gives samples s and 1023-s of this packet's output:
      //   out[s]      = fl(P*wr(s))      + fl( cc*wl(s))        cc =  u_cur[512+s]
      //   out[1023-s] = fl(P*wr(1023-s)) + fl(-cc*wl(1023-s))   P  = -u_prev[511-s]
      // Handled as four point pairs (k, 7-k): each pair yields the contiguous samples (s, s+1) and (1022-s, 1023-s)
      // after one exchange with the mirror lane. Computed for halo / first packets too (emit == 0): only the stores are
      // guarded. P is zero until the run has seen a block; 0*w + x == x exactly as in `buf = 0; buf += x`.
      const float (*TL)[8][64] = T.win[pi.widx & 1u];    // this block's left-half window
      const float (*TR)[8][64] = T.win[prev_next_long];  // previous block's right-half window, mirrored
      // chunk of this block = [centre of the previous block, centre of this one): left-half sample s sits at frame s + shift
      const uint32_t shift = (!MIXED || prev_half != 128u) ? 0u : 0u - 448u;
      const bool fast_store = emit == ML && (!MIXED || shift == 0u) && (((uintptr_t)out & 7u) == 0);
      float oh_s[4], oh_m[4], n_s[4], n_m[4];
      // window values of this lane's 8 points: left half of this block and mirrored right half of the previous one. Between
      // two long blocks that is the same table (wave-uniform test): read it once.
      float wl0[8], wl1[8], wr0[8], wr1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wl0[k] = TL[0][k][lane];
        wl1[k] = TL[1][k][lane];
      }
      if (MIXED && prev_kind != K_REG) {
        // The overlap term arrives already windowed: as P = 1, "window" = the value itself (1 * x == x exactly), so that
        // the arithmetic below is the same straight-line code in every case.
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          P[k] = 1.f;
          wr0[k] = 0.f;
          wr1[k] = 0.f;
        }
        if (prev_kind == K_LDS) {
          // after a short block: its 128 windowed samples meet this block's samples 448..575 — point 7 of the lanes with
          // kappa >= 32 (s = 2 kappa + 384) and point 0 of the others (s = 511 - 2 kappa), both s and 1023-s
          const bool up = kappa >= 32u;
          const float c0 = cbuf[up ? 2u * kappa - 64u : 63u - 2u * kappa];   // at s
          const float c1 = cbuf[up ? 191u - 2u * kappa : 64u + 2u * kappa];  // at 1023-s
          wr1[7] = up ? c0 : 0.f;
          wr0[7] = up ? c1 : 0.f;
          wr1[0] = up ? 0.f : c0;
          wr0[0] = up ? 0.f : c1;
        } else {  // K_CARRY: a long carry-in, natural order
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t m = kappa + 64u * k;
            const uint32_t sk = k >= 4 ? 2u * m - 512u : 511u - 2u * m;
            wr1[k] = cin[sk];
            wr0[k] = cin[1023u - sk];
          }
        }
      } else if (TR == TL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          wr0[k] = wl0[k];
          wr1[k] = wl1[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          wr0[k] = TR[0][k][lane];
          wr1[k] = TR[1][k][lane];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kh = 4 + j, kl = 3 - j;  // kh: even sample s = 2m-512 (own), kl: odd sample 511-2m (goes to the mirror lane)
        const float cch = z[kh].x, ccl = -z[kl].y;
        const float ah_s = P[kh] * wr1[kh], ah_m = P[kh] * wr0[kh], al_s = P[kl] * wr1[kl], al_m = P[kl] * wr0[kl];
        oh_s[j] = ah_s + cch * wl0[kh];
        oh_m[j] = ah_m + (-cch) * wl1[kh];
        const float ol_s = al_s + ccl * wl0[kl];
        const float ol_m = al_m + (-ccl) * wl1[kl];
        P[kh] = z[kh].y;
        P[kl] = -z[kl].x;
        // partner point 511 - m of (this lane, kh) is (mirror lane, slot kl): it yields samples s+1 and 1022-s
        n_s[j] = __shfl(ol_s, 63 - (int)lane);
        n_m[j] = __shfl(ol_m, 63 - (int)lane);
      }
      // Everything this wave has in flight here is LOADS issued long ago (residue and coded row of packet q+1): finish them before
      // the PCM stores go out, so that no later wait in the loop (the compiler places the loop-carried copies of those registers
      // behind the stores, where only vmcnt(0) is safe on every path) ever waits for a store. The stores are never waited for.
#ifndef VSYN_NO_MIXED_DMA
      if (!MIXED) vmem_drain();  // (mixed runs: the next block's rows are in flight on purpose; they are waited for by count)
#else
      vmem_drain();
#endif
      STAMP(7);  // post-rotation, window reads, overlap arithmetic, mirror exchange of the outputs
      // sample s = 2*kappa + 128*j of (lane, kh = 4 + j): two lane pointers, every store at an immediate offset
      if (VSYN_KNOCKOUT & 8) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += oh_s[j] + n_s[j] + n_m[j] + oh_m[j];
        if (acc == 12345.678f) out[lane] = acc;  // keeps the arithmetic alive without the stores
      } else if (fast_store) {
        if (MIXED && dma_younger == 0u) dma_younger = 8u;
        float* up = out + 2u * kappa;            // samples s, s+1
        float* dn = out + 1022u - 2u * kappa;    // samples 1022-s, 1023-s
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!MIXED) {  // steady runs: whole 512-byte rows, written once, never read back -> streamed (non-temporal)
            stream_store2(up + 128 * j, oh_s[j], n_s[j]);
            stream_store2(dn - 128 * j, n_m[j], oh_m[j]);
          } else {  // (mixed runs: chunks of different store shapes share cache lines; streaming them cost config 4 9 %)
            *(float2*)(up + 128 * j) = f2(oh_s[j], n_s[j]);
            *(float2*)(dn - 128 * j) = f2(n_m[j], oh_m[j]);
          }
        }
      } else if (emit) {
        if (MIXED) dma_younger = 0xFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // frames of samples s, s+1, 1022-s, 1023-s (a frame before the chunk wraps around and is never < emit)
          const uint32_t s = 2u * kappa + 128u * j;
          const uint32_t f0 = s + shift, f1 = s + 1u + shift, f2m = 1022u - s + shift, f3m = 1023u - s + shift;
          if (f0 < emit) out[f0] = oh_s[j];
          if (f1 < emit) out[f1] = n_s[j];
          if (f2m < emit) out[f2m] = n_m[j];
          if (f3m < emit) out[f3m] = oh_m[j];
        }
      }
      STAMP(8);  // PCM stores issued
      if (MIXED && (last_of_segment || hand_over)) dma_younger = 0xFFu;
      if (last_of_segment || hand_over) {
        // the windowed right half in natural order: sample s of point k at position s.  Last block of the segment: all of it
        // into the stream's carry buffer for the next submit.  Before a short block: frames 0..447 are final (nothing else
        // lands there) and go into the next chunk, frames 448..575 into the carry image, the rest is windowed to zero.
        const size_t carry_half = (size_t)H->max_streams * C * ML;
        float* cout = A.carry + (si.parity_in ^ 1u) * carry_half + ((size_t)sg.stream * C + c) * ML;
        float* nxt = plane + pin.out_pos;
        const float (*TN)[8][64] = T.win[cur_next_long];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t m = kappa + 64u * k;
          const uint32_t s = k >= 4 ? 2u * m - 512u : 511u - 2u * m;
          const float v_s = P[k] * TN[1][k][lane], v_m = P[k] * TN[0][k][lane];
          if (last_of_segment) {
            cout[s] = v_s;
            cout[1023u - s] = v_m;
          } else {
            const uint32_t sm = 1023u - s;
            if (s < 448u) {
              if (s < pin.emit) nxt[s] = v_s;
            } else if (s < 576u) {
              cbuf[s - 448u] = v_s;
            }
            if (sm < 448u) {
              if (sm < pin.emit) nxt[sm] = v_m;
            } else if (sm < 576u) {
              cbuf[sm - 448u] = v_m;
            }
          }
        }
        // (stores only: nothing here loads a register, so the join needs no drain — one here waited for the block's PCM stores at every
        // long -> short switch of a mixed run; the steady path reaches this block once per segment)
        if (!MIXED) vmem_drain();
      }
    } else {
      // (short blocks of the mixed-block path are taken by fused_short_pass above)
    }
    if (MIXED) {
      prev_kind = hand_over ? K_LDS : K_REG;
      prev_half = M;  // (registers of the other size are never consulted: a size change always goes through K_LDS, and a
                      //  skipped packet clears both)
    }
    if (__any(floor_bad)) {
      if (lane == 0) raise_status(A.status, VSYN_ST_FLOOR_VALUE, p);
      vmem_drain();
    }
    prev_next_long = cur_next_long;
    pi = pin;
  }
#ifdef VSYN_STAMPS
  if (lane0 == 0) {
    const uint32_t unit = blockIdx.x * FUSED_WAVES + (threadIdx.x >> 6);
    if (unit < 8192)
      for (int i = 0; i < VSYN_NSTAMPS; ++i) g_vsyn_stamps[unit][i] = i == VSYN_NSTAMPS - 1 ? (unsigned long long)(qb - q0) : st_acc[i];
  }
#endif
}

static_assert(FUSED_WAVES % 2 == 0, "the two channel waves of a run must share a workgroup");
// ONE launch for both kinds of run. grid: groups of FUSED_WAVES (segment, run, channel) units — the channels of a run sit in
// adjacent waves of one workgroup, which is what the pairwise LDS hand-off needs. Every wave looks up the
// class the layout kernel gave its run (1: all long blocks, steady windows, no carry-in -> fused_run; 2: anything else the
// fused paths cover -> fused_run<.., MIXED = true>) and takes that path; waves of one workgroup may take different ones (they only ever
// meet their coupling partner, which shares the run and therefore the class).
template <bool TAPC>
__device__ __forceinline__ void fused_kernel_body(const FusedArgs& A) {
  // one LDS block with a fixed member order: the floor entry tables come first so that their addresses fit the 16 bits
  // fused_run packs them into (the whole block is 72 KB)
  struct Lds {
    float4 seg[FUSED_WAVES][64];
    uint32_t flag[FUSED_WAVES][2];
    FusedLdsImage t;
    float2 x[FUSED_WAVES][FUSED_XSLOTS];
  };
  __shared__ Lds s_lds;
  FusedLdsImage& s_t = s_lds.t;
  float2 (&s_x)[FUSED_WAVES][FUSED_XSLOTS] = s_lds.x;
  float4 (&s_seg)[FUSED_WAVES][64] = s_lds.seg;
  uint32_t (&s_flag)[FUSED_WAVES][2] = s_lds.flag;
  static_assert(sizeof(s_lds.seg) + sizeof(s_lds.flag) < 65536, "floor entry addresses are packed into 16 bits");
  const ConstHeader* H = hdr_of(A.cb);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  // unit = (segment, run, channel), flattened over the whole batch so that a workgroup's 8 waves are busy whatever the
  // segment lengths are (a batch of many short streams has one or two units per segment). The channels of a run are adjacent
  // units; with 2 channels an even unit is channel 0, so a coupled pair always shares a workgroup.
  const uint32_t C = H->channels;
  const uint32_t per_seg = A.runs_per_seg * C;
  const uint32_t unit = blockIdx.x * FUSED_WAVES + wave;
  const uint32_t g = __builtin_amdgcn_readfirstlane(unit / per_seg);  // (the divisions run on the vector unit)
  const uint32_t rem = unit - g * per_seg;
  const uint32_t run = __builtin_amdgcn_readfirstlane(rem / C), c = __builtin_amdgcn_readfirstlane(rem % C);
  vsyn_segment sg = {};
  SegInfo si = {};
  uint32_t cls = 0xFFu;
  if (g < A.S) {
    sg = A.segs[g];
    if (sg.stream < H->max_streams && !(sg.residue_off & 3)) {  // (otherwise the layout kernel flagged the segment)
      si = A.sinfo[g];
      cls = __builtin_amdgcn_readfirstlane((uint32_t)A.run_cls[(size_t)g * A.runs_per_seg + run]);
    }
  }
  const uint32_t qa = run * A.R;
  const uint32_t qb = min(sg.num_packets, qa + A.R);
  const bool active = (cls == 1u && (A.fused_ok & 1u)) || (cls == 2u && (A.fused_ok & 2u));
  if (!__syncthreads_or(active ? 1 : 0)) return;  // nothing in this workgroup: leave before staging the tables
  {
    const uint4* src = (const uint4*)A.lds_image;
    uint4* dst = (uint4*)&s_t;
    for (uint32_t i = threadIdx.x; i < sizeof(FusedLdsImage) / 16; i += FUSED_WAVES * 64) dst[i] = src[i];
    if (threadIdx.x < FUSED_WAVES * 2) (&s_flag[0][0])[threadIdx.x] = 0u;
  }
  __syncthreads();  // the only workgroup-wide barrier: from here on a wave meets nobody but its coupling partner (pair_post/pair_wait)
  if (!active) return;  // a run is active for all its channels or for none: no partner is left waiting
  const uint32_t mag = A.coupling_mode == 1 ? 0u : 1u;  // the magnitude channel of the (single) coupling step
  const int role = (A.coupling_mode == 0 || C < 2) ? 0 : (c == mag ? 1 : 2);
  const uint32_t pw = role ? (wave ^ 1u) : wave;  // partner wave
  lds_u32* mf = (lds_u32*)s_flag[wave];
  const lds_u32* pf = (const lds_u32*)s_flag[pw];
  // mixed runs: the 128 overlap frames that change hands when the block size changes live in the second half of the wave's
  // floor-entry block (the entries are 8 bytes: 64 of them fill the first 512 bytes)
  lds_f32* cbw = (lds_f32*)((float*)s_seg[wave] + 128);
  if (cls == 1u) {
    if (role == 0) fused_run<0, false, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
    else if (role == 1) fused_run<1, false, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
    else fused_run<2, false, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
  } else {
    if (role == 0) fused_run<0, true, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
    else if (role == 1) fused_run<1, true, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
    else fused_run<2, true, TAPC>(A, s_t, s_x[wave], s_x[pw], s_seg[wave], cbw, mf, pf, lane, g, sg, si, qa, qb, C, c);
  }
}

__global__ void __launch_bounds__(FUSED_WAVES * 64, FUSED_MIN_WAVES_PER_SIMD) vsyn_fused_kernel(const FusedArgs A) { fused_kernel_body<false>(A); }
// the same kernel with the "floor1 floor" feature tap written on the way: its own launch, so that the 16 extra
// registers of the tap never weigh on the plain kernel
__global__ void __launch_bounds__(FUSED_WAVES * 64, FUSED_MIN_WAVES_PER_SIMD) vsyn_fused_tap_kernel(const FusedArgs A) { fused_kernel_body<true>(A); }

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline bool fused_supported(const ConstHeader& H) {
  return H.bs[1] == 2048 && H.bs[0] <= 2048 && H.channels <= 2;
}

// coupling structure shared by every mapping that a long-block mode can select: 0 none, 1 (mag 0, ang 1), 2 (mag 1, ang 0);
// -1 if the long-mode mappings disagree or chain several steps (those streams take the staged kernels)
static inline int fused_coupling_mode(const ConstHeader& H, const uint8_t* host_const) {
  const MapConst* mp = (const MapConst*)(host_const + H.off_map);
  int mode = -2;
  for (uint32_t k = 0; k < H.num_modes; ++k) {
    if (!H.mode_blockflag[k]) continue;
    const MapConst& m = mp[H.mode_mapping[k]];
    if (m.ncoup > 1) return -1;
    const int cur = m.ncoup == 0 ? 0 : (m.coup[0] == 0 ? 1 : 2);
    if (mode == -2) mode = cur;
    else if (mode != cur) return -1;
  }
  return mode == -2 ? 0 : mode;
}

// bit 0: steady long runs usable, bit 1: mixed-block runs usable (needs blocksize0 == 256 and the same coupling
// structure in every mapping, short-block modes included)
static inline uint32_t fused_ok_mask(const ConstHeader& H, const uint8_t* host_const);

static inline bool fused_setup_ok(const ConstHeader& H, const uint8_t* host_const) {
  if (!fused_supported(H)) return false;
  const FloorConst* fl = (const FloorConst*)(host_const + H.off_floor);
  for (uint32_t f = 0; f < H.num_floors; ++f)
    if (fl[f].posts > 64) return false;  // one ballot covers the sorted posts
  return fused_coupling_mode(H, host_const) >= 0;
}

static inline uint32_t fused_ok_mask(const ConstHeader& H, const uint8_t* host_const) {
  if (!fused_setup_ok(H, host_const)) return 0;
  uint32_t mask = 1;
  if (H.bs[0] == 256) {
    const MapConst* mp = (const MapConst*)(host_const + H.off_map);
    const int want = fused_coupling_mode(H, host_const);
    bool same = true;
    for (uint32_t k = 0; k < H.num_modes; ++k) {
      const MapConst& m = mp[H.mode_mapping[k]];
      const int cur = m.ncoup == 0 ? 0 : (m.ncoup == 1 ? (m.coup[0] == 0 ? 1 : 2) : -1);
      same = same && cur == want;
    }
    if (same) mask |= 2;
  }
  return mask;
}

static inline hipError_t fused_tables_create(const ConstHeader& H, const uint8_t* host_const, FusedTables* ft) {
  const uint32_t half = H.bs[1] / 2;
  std::vector<uint8_t> tab((size_t)H.num_floors * half);
  const FloorConst* fl = (const FloorConst*)(host_const + H.off_floor);
  for (uint32_t f = 0; f < H.num_floors; ++f) {
    uint32_t s = 0;
    for (uint32_t x = 0; x < half; ++x) {
      while (s + 1 < fl[f].posts && fl[f].xs_sorted[s + 1] <= x) ++s;
      tab[(size_t)f * half + x] = (uint8_t)s;
    }
  }
  hipError_t e = hipMalloc((void**)&ft->d_binseg, tab.size());
  if (e != hipSuccess) return e;
  e = hipMemcpy(ft->d_binseg, tab.data(), tab.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return e;
  if (fused_supported(H)) {
    std::vector<FusedLdsImage> imgv(1);
    FusedLdsImage& im = imgv[0];
    memset(&im, 0, sizeof(im));
    const float2* pre = (const float2*)(host_const + H.off_pre[1]);
    const float2* post = (const float2*)(host_const + H.off_post[1]);
    const float2* tw = (const float2*)(host_const + H.off_fft[1]);
    const float* win = (const float*)(host_const + H.off_win[1]);
    for (uint32_t l = 0; l < 64; ++l) {
      const uint32_t kappa = ((l & 7u) << 3) | (l >> 3);
      for (uint32_t k = 0; k < 8; ++k) {
        im.pre[k][l] = pre[l + 64 * k];
        im.post[k][l] = post[kappa + 64 * k];
        im.tw1[k][l] = tw[(l * k) & 511u];
        const uint32_t m = kappa + 64 * k;
        const uint32_t s = k >= 4 ? 2 * m - 512 : 511 - 2 * m;
        for (uint32_t f = 0; f < 2; ++f) {  // window table index = prev + 2*next: left half depends on prev only
          im.win[f][0][k][l] = win[(size_t)f * H.bs[1] + s];
          im.win[f][1][k][l] = win[(size_t)f * H.bs[1] + 1023 - s];
        }
      }
    }
    for (uint32_t a = 0; a < 8; ++a)
      for (uint32_t c = 0; c < 8; ++c) im.tw2[a][c] = tw[(8 * c * a) & 511u];
    memcpy(im.invdb, host_const + H.off_invdb, 256 * sizeof(float));
    if (H.bs[0] == 256) {
      const float2* pre0 = (const float2*)(host_const + H.off_pre[0]);
      const float2* post0 = (const float2*)(host_const + H.off_post[0]);
      const float2* tw0 = (const float2*)(host_const + H.off_fft[0]);  // W64^j
      const float* win0 = (const float*)(host_const + H.off_win[0]);
      for (uint32_t l = 0; l < 64; ++l) {
        uint32_t m = 0;
        for (int b = 0; b < 6; ++b) m |= ((l >> b) & 1u) << (5 - b);
        im.pre_s[l] = pre0[l];
        im.post_s[l] = post0[m];
        for (int i = 0; i < 6; ++i) {
          const uint32_t d = 32u >> i;
          im.tws[i][l] = tw0[((l & (d - 1u)) * (32u / d)) & 63u];
        }
        const uint32_t sidx = m >= 32 ? 2 * m - 64 : 63 - 2 * m;
        im.wsl[0][l] = win0[sidx];
        im.wsl[1][l] = win0[127 - sidx];
      }
      for (uint32_t cc = 0; cc < 8; ++cc)
        for (uint32_t a = 0; a < 8; ++a) {
          const uint32_t f = a + 8 * cc, s8 = cc >= 4 ? 2 * f - 64 : 63 - 2 * f;
          im.post8[cc][a] = post0[f];
          im.wsl8[0][cc][a] = win0[s8];
          im.wsl8[1][cc][a] = win0[127 - s8];
        }
    }
    e = hipMalloc((void**)&ft->d_lds, sizeof(FusedLdsImage));
    if (e != hipSuccess) return e;
    e = hipMemcpy(ft->d_lds, &im, sizeof(FusedLdsImage), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
  }
  ft->coupling_mode = fused_coupling_mode(H, host_const);
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, vsyn_fused_kernel, FUSED_WAVES * 64, 0) == hipSuccess && blocks > 0)
    ft->waves_per_cu = blocks * FUSED_WAVES;
  if (getenv("VSYN_DEBUG")) fprintf(stderr, "vsyn: fused kernel %d workgroups/CU -> %d waves/CU\n", blocks, ft->waves_per_cu);
  return hipSuccess;
}

static inline void fused_tables_destroy(FusedTables* ft) {
  if (ft->d_binseg) (void)hipFree(ft->d_binseg);
  if (ft->d_lds) (void)hipFree(ft->d_lds);
  ft->d_binseg = nullptr;
  ft->d_lds = nullptr;
}

static inline const char* fused_kernel_name(const ConstHeader&) { return "vsyn_fused_kernel"; }
static inline const char* fused_imdct_kernel_name(uint32_t) { return "vsyn_imdct_wave_kernel"; }

// run length: as few runs as fill the chip once (halo overhead is 1/R), never below 4
static inline uint32_t fused_pick_run_len(int waves_per_cu, uint32_t S, uint32_t channels, uint32_t max_seg_packets, int num_cus) {
  const char* env = getenv("VSYN_RUN_LEN");
  if (env && atoi(env) > 0) return (uint32_t)atoi(env);
  const uint64_t slots = (uint64_t)num_cus * (uint64_t)waves_per_cu;
  uint32_t R = 4;
  while (R < max_seg_packets && (uint64_t)S * channels * ((max_seg_packets + R - 1) / R) > slots) ++R;
  return R;
}

static inline hipError_t fused_launch(const ConstHeader& H, const FusedTables& ft, FusedArgs a, uint32_t max_seg_packets, hipStream_t s) {
  const uint64_t units = (uint64_t)a.S * a.runs_per_seg * H.channels;  // a.runs_per_seg = ceil(max_seg_packets / R)
  (void)max_seg_packets;
  if (units == 0 || units > 0x7FFFFFF8ull) return hipErrorInvalidValue;
  dim3 grid((uint32_t)((units + FUSED_WAVES - 1) / FUSED_WAVES));
  const char* xl = getenv("VSYN_EXTRA_LDS");  // experiment knob: extra dynamic LDS lowers the occupancy
  const size_t dyn = xl ? (size_t)atoi(xl) : 0;
  a.coupling_mode = (uint32_t)ft.coupling_mode;
  if (a.curve) vsyn_fused_tap_kernel<<<grid, FUSED_WAVES * 64, dyn, s>>>(a);
  else vsyn_fused_kernel<<<grid, FUSED_WAVES * 64, dyn, s>>>(a);
  return hipGetLastError();
}

// runs with short blocks / window switches / a carry-in: same grid, every wave of a run the long kernel took idles out

// ------------------------------------------------------------------------------------------------
// IMDCT only (BASELINE config 2; vsyn_imdct_device): in [count][n/2] -> out [count][n], one WAVEFRONT per block for the two
// block sizes the fused paths know (n = 2048: FFT-512 in registers + LDS exchanges as in fused_run; n = 256: FFT-64 across the
// lanes as in the mixed path). What mdct_backward (mdct.cpp:433-527) computes, by the DCT-IV route:
//   u[2m] = Re d[m], u[M-1-2m] = -Im d[m] (M = n/2, d = post-rotated FFT output), then
//   y[i] = u[i + M/2] (i < M/2),  -u[3M/2 - 1 - i] (M/2 <= i < 3M/2),  -u[i - 3M/2] (i >= 3M/2)
// u goes through the wave's LDS image once so that the n outputs leave as contiguous 16-byte stores. Twiddles are read from
// the (L2-resident) table image in global memory: a batch may be far too small to pay for staging 27 KB per workgroup.
// ------------------------------------------------------------------------------------------------
#define IMDCT_WAVES 4
template <int LONG>
__global__ void __launch_bounds__(IMDCT_WAVES * 64) vsyn_imdct_wave_kernel(const FusedLdsImage* __restrict__ T, uint32_t count,
                                                                           const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float2 s_x[IMDCT_WAVES][FUSED_XSLOTS];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  float2* xb = s_x[wave];
  float* u = (float*)xb;
  constexpr uint32_t M = LONG ? 1024u : 128u;
  for (uint32_t blk = blockIdx.x * IMDCT_WAVES + wave; blk < count; blk += gridDim.x * IMDCT_WAVES) {
    const float2* src = (const float2*)(in + (size_t)blk * M) + lane;
    float* y = out + (size_t)blk * (2u * M);
    if (LONG) {
      float2 r[8], z[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) r[t] = src[64 * t];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float im = __shfl(r[7 - t].y, 63 - (int)lane);  // X[1023 - 2k] lives in the mirror lane, slot 7-t
        z[t] = cmulf(f2(r[t].x, im), T->pre[t][lane]);
      }
      fft512_wave(z, xb, T, lane);
      const uint32_t kappa = ((lane & 7u) << 3) | (lane >> 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the exchange image is about to be reused for u
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float2 d = cmulf(z[k], T->post[k][lane]);
        const uint32_t m = kappa + 64u * k;
        u[2u * m] = d.x;
        u[M - 1u - 2u * m] = -d.y;
      }
    } else {
      const float2 r = src[0];
      const float im = __shfl(r.y, 63 - (int)lane);  // X[127 - 2k] lives in lane 63-k
      float2 z = cmulf(f2(r.x, im), T->pre_s[lane]);
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // radix-2 decimation in frequency across lanes: partner l ^ d, d = 32 .. 1
        const int d = 32 >> i;
        const float ox = __shfl_xor(z.x, d), oy = __shfl_xor(z.y, d);
        const bool upper = (lane & (uint32_t)d) != 0;
        const float2 sum = f2(z.x + ox, z.y + oy);
        const float2 dif = cmulf(f2(ox - z.x, oy - z.y), T->tws[i][lane]);  // (lower - upper) * W, evaluated in the upper lane
        z = upper ? dif : sum;
      }
      const float2 d = cmulf(z, T->post_s[lane]);
      const uint32_t m = bitrev6(lane);
      u[2u * m] = d.x;
      u[M - 1u - 2u * m] = -d.y;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: u is complete once its own LDS writes have landed
    // y in runs of 4: the three ranges of the definition are multiples of 4 long, so a run never straddles two of them
    for (uint32_t i4 = lane; i4 < 2u * M / 4u; i4 += 64) {
      const uint32_t i = 4u * i4;
      float4 v;
      if (i < M / 2u) {
        v = *(const float4*)&u[i + M / 2u];
      } else if (i < 3u * M / 2u) {
        const float4 w = *(const float4*)&u[3u * M / 2u - 4u - i];  // u[3M/2-1-i-3 .. 3M/2-1-i], reversed below
        v = make_float4(-w.w, -w.z, -w.y, -w.x);
      } else {
        const float4 w = *(const float4*)&u[i - 3u * M / 2u];
        v = make_float4(-w.x, -w.y, -w.z, -w.w);
      }
      *(float4*)&y[i] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // u has been read before the next block reuses the image
  }
}

static inline hipError_t fused_imdct_launch(const ConstHeader& H, const uint8_t*, const FusedTables& ft, int, uint32_t n, uint32_t count,
                                            const float* d_in, float* d_out, hipStream_t s, bool* done) {
  *done = false;
  if (!ft.d_lds) return hipSuccess;  // no table image for this setup (it exists iff blocksize1 is 2048): the generic kernel takes it
  if ((((uintptr_t)d_in) & 7u) || (((uintptr_t)d_out) & 15u)) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)count + IMDCT_WAVES - 1) / IMDCT_WAVES, 256u * 8u);
  if (n == 2048 && H.bs[1] == 2048) vsyn_imdct_wave_kernel<1><<<grid, IMDCT_WAVES * 64, 0, s>>>(ft.d_lds, count, d_in, d_out);
  else if (n == 256 && H.bs[0] == 256) vsyn_imdct_wave_kernel<0><<<grid, IMDCT_WAVES * 64, 0, s>>>(ft.d_lds, count, d_in, d_out);
  else return hipSuccess;
  *done = true;
  return hipGetLastError();
}
