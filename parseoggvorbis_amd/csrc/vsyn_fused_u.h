// vsyn_fused_u.h — the size-generic fused synthesis kernel ("U"): every block-size pair 64..8192 with blocksize0 <= 2048 or equal to
// blocksize1 (the reference runs every size through the same code, hpp:1294-1298, mdct.cpp:353-373; blocks above 2048 as 2 / 4 register
// sets, UBig), any mix of short and long blocks, carry-ins, up to 16 coupled channels with any list of coupling steps (ROLE 3: replayed in
// place by the channel waves of a run, which share a workgroup) or any number of uncoupled ones, floors of up to the spec's 65 posts. One
// wavefront per (segment, run, channel) as in vsyn_fused.h; what is new:
//
//   * a wave always works on 512 complex points = 8 per lane. A block of n samples has Np = n/4 points, so a PASS takes
//     J = 512/Np consecutive packets of the same size and mapping at once (n = 256: 8 packets; n = 1024: 2; n = 2048: 1; at most 8):
//     element (register t, lane l), e = 64 t + l, is point k = e mod Np of packet j = e div Np of the pass.
//   * the FFT-Np of all J packets is the SAME three-pass network as the FFT-512 of vsyn_fused.h with the leading radix-2 stages
//     of pass 1 (register index) resp. pass 2 switched off: DFT-R1 over the low register bits (R1 = Np/64), twiddle, exchange,
//     DFT-8 (or 4 / 2 when Np < 64) over lane bits 5..3, twiddle, exchange, DFT-8 over lane bits 2..0. Afterwards packet j sits in
//     the G = Np/8 consecutive lanes [jG, (j+1)G), bin f = kappa(lane) + G c' in register c'. All size dependence is in host-built
//     lane-major tables (ULdsSize) and a few wave-uniform scalars: the device code is one instance.
//   * the overlap term of packet j comes from packet j-1 of the same pass (G lanes below, one ds_bpermute per register), from the
//     previous pass (registers), or — when the block size changes — through the wave's carry image, exactly as in vsyn_fused.h.
//   * the floor: one table of per-interval line records per packet of the pass, side by side in the idle exchange image; every element
//     reads the records of its two bins from the table of its own packet (round 2 rendered each packet in bin order and staged the
//     factors through LDS); everything else follows vsyn_fused.h (same roundings: coupling and floor product bit-exact, window
//     product and overlap sum rounded separately, hpp:1008-1017).
#pragma once
#include <hip/hip_runtime.h>

#include "vsyn_device.h"
#include "vsyn_staged.h"
#include "vsyn_fused.h"

struct ULdsSize {           // one block size; every table in exactly the order the lanes read it
  float2 pre[8][64];        // element (t, lane): pre-rotation exp(-i pi (4k+1)/(4M)) of its point k
  float2 post[8][64];       // (c', lane): post-rotation exp(-i pi f / M) of bin f
  float2 tw1[8][64];        // pass-1 twiddle W_Np^(lane * t') (1 where the pass is absent)
  float2 tw2[8][8];         // pass-2 twiddle [a][c]
  float win[2][2][8][64];   // [window flag][0: at s, 1: at M-1-s][c'][lane]: left half of the window (hpp:850-859); the right half
                            // for next-flag f is its mirror. Short blocks: both flags hold the one short window
};
struct ULdsImage {
  ULdsSize sz[2];           // [0] blocksize0, [1] blocksize1 (blocksize1 > 2048: the tables of a 512-point FFT, see UBig)
  float invdb[260];         // Vorbis I 10.1; [255] = 1.0f, [256] = 0.0f as in FusedLdsImage
};
// Long blocks of 4096 / 8192 samples (NS = 2 / 4): Np = 512 NS points, NS register sets of 8 per lane; point k = 512 u + 64 t + lane.
// One radix-NS stage across the sets (twiddle W_Np^((64 t + lane) u')), then NS FFT-512 through the usual network; bin
// f = u' + NS (kappa + 64 c') ends up in set u', register c'. Follows ULdsImage in LDS.
template <int NS>
struct UBig {
  float2 pre[NS][8][64];
  float2 post[NS][8][64];
  float2 tw0[NS][8][64];        // row 0 unused
  float win[2][2][NS][8][64];   // [window flag][0: at s, 1: at M-1-s][u'][c'][lane]
};

// per-wave LDS block (dynamic): exchange image | floor entries | packet info of the pass | next descriptors | hand-off flags | carry image
#define U_XB_BYTES 4608u
#define U_SEG_OFF U_XB_BYTES
#define U_PINF_OFF (U_SEG_OFF + 528u)   /* 65 interval records: a 65-post floor's last post owns the flat stretch behind it */
#define U_DNEXT_OFF (U_PINF_OFF + 256u)   /* descriptors of the next pass's candidates, 8 x 32 B, filled by LDS-DMA */
#define U_FLAG_OFF (U_DNEXT_OFF + 256u)
#define U_CBUF_OFF (U_FLAG_OFF + 16u)
#define U_MAX_J 8u
#define U_TAB_STRIDE 65u  /* line records per packet table of a packed pass (a 65-post floor's last post owns the stretch behind it) */
#ifndef U_MAX_THREADS
#define U_MAX_THREADS 1024  // 16 waves per workgroup, one workgroup per CU: 128 VGPRs per wave (the kernel needs 153: 20 B of scratch, a few
                            // hoisted addresses reloaded once per pass; 12 waves at 168 VGPRs measured 7.5 % slower once the pass body had slimmed to this)
#endif

struct UArgs {
  FusedArgs f;
  const ULdsImage* img;
  uint32_t wave_bytes;      // per-wave LDS block size (multiple of 16)
  uint32_t table_bytes;     // sizeof(ULdsImage) rounded up
  uint32_t role_mode;       // 0 no coupling anywhere, 1 / 2 one step (mag 0, ang 1) / (mag 1, ang 0) in every mapping of a stereo stream:
                            // the two channel waves swap rows pairwise; 3 anything else (<= U_MAX_CH channels, any coupling list per
                            // mapping): the channel waves of a run replay the steps in place, in order
};
#define U_MAX_CH 16u  /* coupled channels: the channel waves of a run share one workgroup (16 waves; 8 for blocks above 2048) */

__device__ __forceinline__ void u_dft4(float2& x0, float2& x1, float2& x2, float2& x3) {  // natural order, forward
  const float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_mi(csub(x1, x3));
  x0 = cadd(a, c);
  x1 = cadd(b, d);
  x2 = csub(a, c);
  x3 = csub(b, d);
}
// DFT of size 2^stages over the low `stages` bits of the register index, batched over the high bits (stages wave-uniform)
__device__ __forceinline__ void u_dft_regs(float2 (&z)[8], uint32_t stages) {
  if (stages == 3u) {
    dft8(z);
  } else if (stages == 2u) {
    u_dft4(z[0], z[1], z[2], z[3]);
    u_dft4(z[4], z[5], z[6], z[7]);
  } else if (stages == 1u) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const float2 s = cadd(z[i], z[i + 1]), d = csub(z[i], z[i + 1]);
      z[i] = s;
      z[i + 1] = d;
    }
  }
}

typedef uint32_t u_u32x4 __attribute__((ext_vector_type(4)));
typedef float u_f32x4 __attribute__((ext_vector_type(4)));
typedef float u_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u_u32x4 u_lds_u32x4;
typedef __attribute__((address_space(3))) u_f32x4 u_lds_f32x4;
typedef __attribute__((address_space(3))) u_f32x2 u_lds_f32x2;

// Wait on a counter of the channel group, bounded: the waves of a group always reach the same counts (same packets, same mappings),
// so the bound is never met — but a wave that waited for ever would take the device down with it, so after ~2 s it flags the batch
// (VSYN_ST_BAD_SEGMENT) and goes on.
__device__ __forceinline__ void group_wait(const lds_u32* flag, uint32_t v, DevStatus* status) {
  asm volatile("" ::: "memory");
  uint32_t spins = 0;
  while (*(const volatile lds_u32*)flag < v) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 24)) {
      if ((threadIdx.x & 63u) == 0) atomicOr(&status->flags, VSYN_ST_BAD_SEGMENT);
      break;
    }
  }
  asm volatile("" ::: "memory");
}

// what a wave knows about one pass (all wave-uniform)
struct UPass {
  uint32_t q, Jp, LG, lng, map, last_widx, qn, buf;
  bool valid, after_bad;  // after_bad: an invalid packet was skipped in front of this pass (the overlap chain restarts)
};

template <int ROLE, int UNS, bool TAPC>
__device__ __forceinline__ void u_run(const FusedArgs& A, const ULdsImage& T, uint8_t* wmem, const uint8_t* pmem, const uint32_t wave_bytes, const uint32_t lane_in,
                                      const uint32_t g, const vsyn_segment sg, const SegInfo si, const uint32_t qa, const uint32_t qb, const uint32_t C,
                                      const uint32_t c) {
  uint32_t lane = lane_in;  // laundered at the top of every pass: nothing derived from it is hoisted out of the pass loop (and then spilled)
  const uint8_t* __restrict__ cb = A.cb;
  const ConstHeader* H = hdr_of(cb);
  float2* const xb = (float2*)wmem;
  const float2* const pxb = (const float2*)pmem;
  float2* const seg2 = (float2*)(wmem + U_SEG_OFF);
  lds_u32* const pinf_base = (lds_u32*)(wmem + U_PINF_OFF);  // packet table of the pass in work
  lds_u32* const my_flags = (lds_u32*)(wmem + U_FLAG_OFF);
  const lds_u32* const partner_flags = (const lds_u32*)(pmem + U_FLAG_OFF);
  lds_f32* const cbuf = (lds_f32*)(wmem + U_CBUF_OFF);

  const uint32_t num = sg.num_packets;
  const uint32_t lgp[2] = {__builtin_amdgcn_readfirstlane(H->lg[0]) - 2u, __builtin_amdgcn_readfirstlane(H->lg[1]) - 2u};  // log2(points)
  const uint32_t ys_stride = __builtin_amdgcn_readfirstlane(H->ys_stride);
  const uint32_t half1 = __builtin_amdgcn_readfirstlane(H->bs[1]) / 2u;
  const MapConst* const maps = (const MapConst*)(cb + H->off_map);
  const FloorConst* const floors = (const FloorConst*)(cb + H->off_floor);
  const size_t carry_half = (size_t)H->max_streams * C * half1;
  float* const plane = A.pcm + ((size_t)g * C + c) * A.plane_stride;
  const PktInfo* const ip = A.info + sg.first_packet;

  float P[8];  // unwindowed right-half values of the previous block, in the lanes of group 0 (see the carry step below)
#pragma unroll
  for (int k = 0; k < 8; ++k) P[k] = 0.f;
  uint32_t prev_M = 0;          // half size of the previous block, 0: none (0 * w + x == x: the arithmetic needs no special case)
  uint32_t prev_kind = K_REG;
  uint32_t prev_next_long = 1;
  const float* cin = nullptr;
  const uint32_t q0 = qa ? qa - 1u : 0u;
  if (qa == 0 && si.has_carry) {
    cin = A.carry + si.parity_in * carry_half + ((size_t)sg.stream * C + c) * half1;
    prev_M = si.carry_n / 2u;
    prev_kind = K_CARRY;
  }
  // floor in use (wave-uniform; reloaded when it changes)
  int cur_floor = -1;
  uint32_t cur_floor_M = 0, posts = 0, sidx = 0, xsl = 0;
  // A 65-post floor (the spec's maximum) has one sorted post more than the wave has lanes: the last one — always header post 1, the
  // largest x, always in use (hpp:533-534) — is kept wave-uniform: its header index and x here, its value read per packet.
  uint32_t sidx64 = 0, xs64 = 0;
  uint32_t bsege[4] = {0, 0, 0, 0};  // sorted-post intervals of the two bins of this lane's element t (point k = (64 t + lane) mod Np: bins 2k, 2k + 1), 16 bits per t

  // packet descriptors of the candidates of the pass to be formed next: lane j < 8 holds packet (start) + j
  // They come by LDS-DMA (one 4-byte piece per lane: 256 contiguous bytes = descriptors qq .. qq+7; the workspace has slack behind
  // its last packet, and candidates beyond the run are masked) — in registers they were eight VGPRs live across the whole pass.
  lds_u32* const dnext = (lds_u32*)(wmem + U_DNEXT_OFF);
  auto fetch_info = [&](uint32_t qq) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)(ip + qq) + 4u * lane),
                                     (__attribute__((address_space(3))) void*)dnext, 4, 0, 0);
  };
  // Form the pass that starts at packet qs from the descriptors in nfa/nfb (already landed): consecutive valid packets of one size
  // and mapping, at most J; packet table into pinf[buf]; then request the descriptors behind it.
  auto form = [&](uint32_t qs, uint32_t buf) -> UPass {
    u_u32x4 nfa = *(const u_lds_u32x4*)(dnext + 8u * (lane & 7u)), nfb = *(const u_lds_u32x4*)(dnext + 8u * (lane & 7u) + 4u);
    UPass ps;
    ps.after_bad = false;
    ps.valid = false;
    ps.buf = buf;
    while (qs < qb) {
      const uint32_t bad0 = __builtin_amdgcn_readfirstlane((nfb[3] >> 8) & 0xFFu);
      if (!bad0) break;
      ps.after_bad = true;  // invalid mode number (flagged by the layout kernel): nothing to synthesise
      ++qs;
      if (qs < qb) {
        fetch_info(qs);
        vmem_drain();
        nfa = *(const u_lds_u32x4*)(dnext + 8u * (lane & 7u));
        nfb = *(const u_lds_u32x4*)(dnext + 8u * (lane & 7u) + 4u);
      }
    }
    ps.q = qs;
    if (qs >= qb) {
      ps.Jp = ps.LG = ps.lng = ps.map = ps.last_widx = 0;
      ps.qn = qb;
      return ps;
    }
    const bool cand = lane < U_MAX_J && qs + lane < qb;
    const uint32_t bad_l = (nfb[3] >> 8) & 0xFFu, lng_l = (nfb[2] >> 16) & 0xFFu, map_l = nfb[3] & 0xFFu;
    ps.lng = __builtin_amdgcn_readfirstlane(lng_l);
    ps.map = __builtin_amdgcn_readfirstlane(map_l);
    ps.LG = ps.lng ? lgp[1] : lgp[0];
    const uint32_t J = max(1u, min(U_MAX_J, 512u >> ps.LG));
    const uint64_t okm = __ballot(cand && !bad_l && lng_l == ps.lng && map_l == ps.map && lane < J);
    ps.Jp = (uint32_t)__builtin_ctzll(~okm);  // >= 1
    // {res_off lo, hi, out_pos, emit (0 for the halo), used, own, widx, -}
    if (lane < ps.Jp) {
      lds_u32* pinf = pinf_base + 64u * buf;
      u_u32x4 a = nfa;
      if (qs + lane < qa) a[3] = 0u;
      u_u32x4 b = {nfb[0], nfb[1], nfb[2] >> 24, 0u};
      *(u_lds_u32x4*)(pinf + 8u * lane) = a;
      *(u_lds_u32x4*)(pinf + 8u * lane + 4u) = b;
    }
    ps.qn = qs + ps.Jp;
    ps.last_widx = __builtin_amdgcn_readlane(nfb[2] >> 24, ps.Jp - 1u);
    if (ps.qn < qb) fetch_info(ps.qn);
    ps.valid = true;
    return ps;
  };
  float2 raw[8];
  uint32_t vrow[U_MAX_J];
  auto load_residue = [&](const UPass& ps) {
    const lds_u32* pinf = pinf_base + 64u * ps.buf;
    const uint32_t Mp = 2u << ps.LG, kmask = (1u << ps.LG) - 1u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint32_t e = 64u * t + lane, jt = e >> ps.LG, k = e & kmask;
      const uint32_t jc = min(jt, ps.Jp - 1u);
      typedef uint32_t u_u32x2 __attribute__((ext_vector_type(2)));
      const u_u32x2 ro = *(const __attribute__((address_space(3))) u_u32x2*)(pinf + 8u * jc);
      const uint64_t off = ((uint64_t)ro[1] << 32) | ro[0];
      raw[t] = ((const float2*)(A.residue + off + (size_t)c * Mp))[k];
    }
  };
  auto load_rows = [&](const UPass& ps) {
    const uint32_t p0 = sg.first_packet + ps.q;
#pragma unroll
    for (uint32_t j = 0; j < U_MAX_J; ++j) {
      const uint32_t pj = p0 + min(j, ps.Jp - 1u);
      vrow[j] = (A.fy + ((size_t)pj * C + c) * ys_stride)[sidx];
    }
  };
  auto floor_of_pass = [&](const UPass& ps) -> uint32_t { return __builtin_amdgcn_readfirstlane((uint32_t)maps[ps.map].chfloor[c]); };
  auto floor_update = [&](uint32_t f, uint32_t Mp) {  // wave-uniform, rare
    const FloorConst* fc = floors + f;
    posts = __builtin_amdgcn_readfirstlane(fc->posts);
    const uint8_t* bs = A.binseg + (size_t)f * half1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t k0 = (64u * (2u * i) + lane) & (Mp / 2u - 1u), k1 = (64u * (2u * i + 1u) + lane) & (Mp / 2u - 1u);
      bsege[i] = (uint32_t) * (const uint16_t*)(bs + 2u * k0) | ((uint32_t) * (const uint16_t*)(bs + 2u * k1) << 16);
    }
    const bool in = lane < posts;
    sidx = in ? fc->sorted_idx[lane] : 0u;
    xsl = in ? fc->xs_sorted[lane] : 0u;
    if (posts > 64u) {
      sidx64 = __builtin_amdgcn_readfirstlane((uint32_t)fc->sorted_idx[64]);
      xs64 = __builtin_amdgcn_readfirstlane((uint32_t)fc->xs_sorted[64]);
    }
    cur_floor = (int)f;
    cur_floor_M = Mp;
    vmem_drain();
  };

  fetch_info(q0);
  vmem_drain();
  uint32_t it = 0, ep = 0, qnext = q0, steps_done = 0, coupled_passes = 0;  // ep: hand-off rounds so far (the pairwise counters)
  float PB[UNS > 1 ? UNS : 1][8];  // blocks above 2048: unwindowed right-half values of the previous such block
#pragma unroll
  for (int u = 0; u < (UNS > 1 ? UNS : 1); ++u)
#pragma unroll
    for (int k = 0; k < 8; ++k) PB[u][k] = 0.f;
  for (;;) {
    asm volatile("" : "+v"(lane));
    // (Forming the NEXT pass early and requesting its residue a pass ahead was measured: 24 more live registers, no gain at the 8-12
    // waves per CU this kernel runs at — 0.221 vs 0.218 ms per 65 536 n = 1024 packets without spills, slower with them.)
    const UPass cur = form(qnext, 0u);
    if (!cur.valid) break;
    if (cur.after_bad) {  // the chain restarts behind an invalid packet
#pragma unroll
      for (int k = 0; k < 8; ++k) P[k] = 0.f;
      prev_M = 0;
      prev_kind = K_REG;
    }
    if ((int)floor_of_pass(cur) != cur_floor || (2u << cur.LG) != cur_floor_M) floor_update(floor_of_pass(cur), 2u << cur.LG);
    if (UNS > 1 && cur.LG > 9u) {
      // ============ one block of 4096 / 8192 samples: NS register sets ============================================================
      constexpr int NS = UNS > 1 ? UNS : 2;
      const UBig<NS>& B = *(const UBig<NS>*)((const uint8_t*)&T + ((sizeof(ULdsImage) + 15u) & ~15u));
      const ULdsSize& F = T.sz[1];  // tw1 / tw2 of the 512-point FFT
      ++it;
      const uint32_t Npb = 512u * NS, Mb = 2u * Npb, qn = cur.qn;
      const uint32_t p = sg.first_packet + cur.q;
      const lds_u32* const pinf = pinf_base;
      const uint64_t roff = ((uint64_t)pinf[1] << 32) | pinf[0];
      // (from a laundered lane number: everything derived from kappa — the carry-in addresses of the rare K_CARRY branch alone are 16 NS
      // pointers — is otherwise hoisted out of the block loop and spilled)
      uint32_t lane_k = lane;
      asm volatile("" : "+v"(lane_k));
      const uint32_t kap = ((lane_k & 7u) << 3) | (lane_k >> 3);
      // ---- loads ----
      float2 rb[NS][8];
      {
        const float2* src = (const float2*)(A.residue + roff + (size_t)c * Mb) + lane;
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
          for (int t = 0; t < 8; ++t) rb[u][t] = src[512 * u + 64 * t];
      }
      uint32_t v = (A.fy + ((size_t)p * C + c) * ys_stride)[sidx];
      // ---- inverse coupling, one register set per hand-off round ----
      if (ROLE == 3) {
        uint8_t* const g0 = wmem - (size_t)c * wave_bytes;
        lds_u32* const gfl = (lds_u32*)(g0 + U_FLAG_OFF);
        const MapConst* mc = maps + cur.map;
        const uint32_t ncoup = __builtin_amdgcn_readfirstlane(mc->ncoup);
        if (ncoup) {
#pragma unroll
          for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int t = 0; t < 8; ++t) xb[t * 64 + lane] = rb[u][t];
            ++coupled_passes;
            if (lane == 0) __atomic_fetch_add(gfl + 2, 1u, __ATOMIC_RELAXED);
            for (uint32_t i = 0; i < ncoup; ++i) {
              const uint32_t k = ncoup - 1u - i;
              const uint32_t cm = __builtin_amdgcn_readfirstlane((uint32_t)mc->coup[2 * k]), ca = __builtin_amdgcn_readfirstlane((uint32_t)mc->coup[2 * k + 1]);
              if (c == cm) {
                group_wait(gfl + 2, C * coupled_passes, A.status);
                group_wait(gfl + 3, steps_done + i, A.status);
                float2* im = (float2*)(g0 + (size_t)cm * wave_bytes);
                float2* ia = (float2*)(g0 + (size_t)ca * wave_bytes);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                  float2 vm = im[t * 64 + lane], va = ia[t * 64 + lane];
                  inverse_couple(vm.x, va.x);
                  inverse_couple(vm.y, va.y);
                  im[t * 64 + lane] = vm;
                  ia[t * 64 + lane] = va;
                }
                pair_post(gfl + 3, steps_done + i + 1u);
              }
            }
            group_wait(gfl + 3, steps_done + ncoup, A.status);
            steps_done += ncoup;
#pragma unroll
            for (int t = 0; t < 8; ++t) rb[u][t] = xb[t * 64 + lane];
          }
        }
      } else if (ROLE != 0) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
          for (int t = 0; t < 8; ++t) xb[t * 64 + lane] = rb[u][t];
          ++ep;
          pair_post(&my_flags[0], ep);
          pair_wait(&partner_flags[0], ep);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float2 oth = pxb[t * 64 + lane];
            rb[u][t] = ROLE == 1 ? f2(couple_mag(rb[u][t].x, oth.x), couple_mag(rb[u][t].y, oth.y))
                                 : f2(couple_ang(oth.x, rb[u][t].x), couple_ang(oth.y, rb[u][t].y));
          }
          pair_post(&my_flags[1], ep);
          pair_wait(&partner_flags[1], ep);
        }
      }
      // ---- floor curve + product (hpp:563-589, 1243-1255): table per sorted-post interval, bins of this lane looked up directly ----
      bool floor_bad = false;
      {
        const uint32_t own_b = pinf[5], used_b = pinf[4];
        const bool nocurve = !((own_b >> c) & 1u);
        if (nocurve) {
          seg2[lane] = f2(0.f, ((used_b >> c) & 1u) ? 256.5f : 255.5f);
        } else {
          if (lane >= posts) v = 0;
          const bool p65 = posts > 64u;
          uint32_t packed64 = 0;
          if (p65) {  // (rare: the extra load is exposed)
            const uint32_t v64 = __builtin_amdgcn_readfirstlane((uint32_t)(A.fy + ((size_t)p * C + c) * ys_stride)[sidx64]) & 0x7FFFu;
            packed64 = (xs64 << 16) | v64;
            if (lane == 0) seg2[64] = f2(0.f, fminf((float)v64, 255.f));
            floor_bad = v64 > 255u;
          }
          const uint64_t mask = __ballot((v >> 15) != 0) | 1ull;
          const uint64_t below = mask & ((2ull << lane) - 1ull);
          const uint32_t lo = 63u - (uint32_t)__clzll((long long)below);
          const uint64_t above = lane < 63u ? (mask >> (lane + 1u)) : 0ull;
          const bool has_hi = above != 0ull || p65;
          const uint32_t hi = lane + (uint32_t)__ffsll((long long)above);
          const uint32_t packed = (xsl << 16) | (v & 0x7FFFu);
          const uint32_t plo = (uint32_t)__shfl((int)packed, (int)lo);
          uint32_t phi = (uint32_t)__shfl((int)packed, (int)(above != 0ull ? hi : lo));
          if (p65 && above == 0ull) phi = packed64;
          floor_bad = floor_bad || (v & 0x7FFFu) > 255u;
          const float x0 = (float)(plo >> 16), y0 = fminf((float)(plo & 0xFFFFu), 255.f);
          const float x1 = (float)(phi >> 16), y1 = fminf((float)(phi & 0xFFFFu), 255.f);
          const float inv = has_hi ? __builtin_amdgcn_rcpf(x1 - x0) : 0.f;
          const float ady = fabsf(y1 - y0);
          const float a = ady * inv, b = __builtin_fmaf(-ady, x0, 0.5f) * inv;
          seg2[lane] = y1 >= y0 ? f2(a, b + y0) : f2(-a, (y0 + 1.f) - b);
        }
        const uint8_t* bs = A.binseg + (size_t)cur_floor * half1;
        const uint32_t seg_base = (uint32_t)(uintptr_t)(lds_u32*)seg2;
        // (the bins' x as floats from a laundered lane number, so that the 16 NS values are not hoisted out of the block loop and
        // spilled; 2 lane + an even literal is exact in f32)
        uint32_t lane_x = lane;
        asm volatile("" : "+v"(lane_x));
        const float xf2 = (float)(2u * lane_x);
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const uint32_t k = 512u * u + 64u * t + lane;
            const uint32_t two = nocurve ? 0u : (uint32_t) * (const uint16_t*)(bs + 2u * k);
            const u_f32x2 e0 = *(const u_lds_f32x2*)(uintptr_t)(seg_base + 8u * (two & 0xFFu));
            const u_f32x2 e1 = *(const u_lds_f32x2*)(uintptr_t)(seg_base + 8u * (two >> 8));
            const uint32_t i0 = (uint32_t)__builtin_fmaf(xf2 + (float)(1024 * u + 128 * t), e0.x, e0.y),
                           i1 = (uint32_t)__builtin_fmaf(xf2 + (float)(1024 * u + 128 * t + 1), e1.x, e1.y);
            rb[u][t] = f2(rb[u][t].x * T.invdb[i0], rb[u][t].y * T.invdb[i1]);
            if (TAPC && !nocurve) {  // feature tap "floor1 floor" (hpp:585): the table indices are the rendered curve; packed like the residue
              ((uint32_t*)(A.curve + roff + (size_t)c * Mb))[k] = i0 | (i1 << 16);
            }
          }
      }
      // ---- IMDCT: mirror element (set NS-1-u, register 7-t, lane 63-l), pre-rotation, radix-NS across the sets, NS x FFT-512 ----
      float2 zb[NS][8];
#pragma unroll
      for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) zb[u][t] = cmulf(f2(rb[u][t].x, __shfl(rb[NS - 1 - u][7 - t].y, 63 - (int)lane)), B.pre[u][t][lane]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (NS == 2) {
          const float2 a = cadd(zb[0][t], zb[1][t]), d = csub(zb[0][t], zb[1][t]);
          zb[0][t] = a;
          zb[1][t] = d;
        } else {
          u_dft4(zb[0][t], zb[1][t], zb[2 % NS][t], zb[3 % NS][t]);
        }
#pragma unroll
        for (int u = 1; u < NS; ++u) zb[u][t] = cmulf(zb[u][t], B.tw0[u][t][lane]);
      }
      {
        const uint32_t cl = lane & 7u, hi = lane >> 3;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          float2(&z)[8] = zb[u];
          dft8(z);
#pragma unroll
          for (int t = 1; t < 8; ++t) z[t] = cmulf(z[t], F.tw1[t][lane]);
#pragma unroll
          for (int t = 0; t < 8; ++t) xb[t * 72 + lane] = z[t];
#pragma unroll
          for (int a = 0; a < 8; ++a) z[a] = xb[hi * 72 + a * 8 + cl];
          dft8(z);
#pragma unroll
          for (int a = 1; a < 8; ++a) z[a] = cmulf(z[a], F.tw2[a][cl]);
#pragma unroll
          for (int a = 0; a < 8; ++a) xb[a * 65 + lane] = z[a];
#pragma unroll
          for (int k = 0; k < 8; ++k) z[k] = xb[cl * 65 + hi * 8 + k];
          dft8(z);
#pragma unroll
          for (int k = 0; k < 8; ++k) z[k] = cmulf(z[k], B.post[u][k][lane]);
        }
      }
      // ---- window + overlap-add + PCM: bin f = u' + NS (kappa + 64 c'); s = 2f - Np (c' >= 4) resp. Np - 1 - 2f ----
      const uint32_t emit = pinf[3], widx_b = pinf[6];
      float* const out = plane + pinf[2];
      const uint32_t shift = (prev_M && prev_M != Mb) ? (uint32_t)(((int32_t)prev_M - (int32_t)Mb) / 2) : 0u;
      const uint32_t fL = widx_b & 1u, fR = prev_next_long;
      if (prev_kind == K_CARRY && prev_M != Mb) {  // carry-in of a smaller block (a larger one does not exist): into the carry image
        for (uint32_t i = lane; i < prev_M; i += 64) cbuf[i] = cin[i];
        prev_kind = K_LDS;
      }
      const bool last_of_segment = qn == num;
      uint32_t next_M = 0, next_emit = 0, next_out = 0;
      vmem_drain();
      if (qn < qb) {
        const uint32_t nbad = __builtin_amdgcn_readfirstlane((dnext[7] >> 8) & 0xFFu), nlng = __builtin_amdgcn_readfirstlane((dnext[6] >> 16) & 0xFFu);
        next_M = nbad ? 0u : (2u << (nlng ? lgp[1] : lgp[0]));
        next_emit = __builtin_amdgcn_readfirstlane(qn < qa ? 0u : dnext[3]);
        next_out = __builtin_amdgcn_readfirstlane(dnext[2]);
      }
      const bool hand_over = next_M && next_M != Mb;
      const uint32_t cur_next_long = (widx_b >> 1) & 1u;
      const bool whole = emit == Mb + shift && (int)shift >= 0 && (((uintptr_t)out & 15u) == 0);
      float os[NS][4], om[NS][4], ns_[NS][4], nm_[NS][4];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        float wl0[8], wl1[8], wr0[8], wr1[8], Pin[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          wl0[k] = B.win[fL][0][u][k][lane];
          wl1[k] = B.win[fL][1][u][k][lane];
          wr0[k] = B.win[fR][0][u][k][lane];
          wr1[k] = B.win[fR][1][u][k][lane];
          Pin[k] = PB[u][k];
        }
        if (prev_kind != K_REG) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t f = u + NS * (kap + 64u * k);
            const uint32_t s0 = k >= 4 ? 2u * f - Npb : Npb - 1u - 2u * f;
            float vs, vm;
            if (prev_kind == K_CARRY) {
              vs = cin[s0];
              vm = cin[Mb - 1u - s0];
            } else {  // after a smaller block: its prev_M windowed samples meet this block's samples [D, D + prev_M)
              const uint32_t D = (Mb - prev_M) / 2u;
              const bool in = s0 >= D && s0 < D + prev_M;
              vs = in ? cbuf[s0 - D] : 0.f;
              vm = in ? cbuf[Mb - 1u - s0 - D] : 0.f;
            }
            Pin[k] = 1.f;
            wr1[k] = vs;
            wr0[k] = vm;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kh = 4 + j, kl = 3 - j;
          const float cch = zb[u][kh].x, ccl = -zb[u][kl].y;
          const float ah_s = Pin[kh] * wr1[kh], ah_m = Pin[kh] * wr0[kh], al_s = Pin[kl] * wr1[kl], al_m = Pin[kl] * wr0[kl];
          os[u][j] = ah_s + cch * wl0[kh];
          om[u][j] = ah_m + (-cch) * wl1[kh];
          // the odd neighbours of set NS-1-u's even samples: exchanged with the mirror lane below
          ns_[u][j] = al_s + ccl * wl0[kl];
          nm_[u][j] = al_m + (-ccl) * wl1[kl];
        }
        if ((last_of_segment || hand_over)) {
          float* cout = A.carry + (si.parity_in ^ 1u) * carry_half + ((size_t)sg.stream * C + c) * half1;
          float* nxtp = plane + next_out;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t f = u + NS * (kap + 64u * k);
            const uint32_t s0 = k >= 4 ? 2u * f - Npb : Npb - 1u - 2u * f, sm = Mb - 1u - s0;
            const float pn = k >= 4 ? zb[u][k].y : -zb[u][k].x;
            const float v_s = pn * B.win[cur_next_long][1][u][k][lane], v_m = pn * B.win[cur_next_long][0][u][k][lane];
            if (last_of_segment) {
              cout[s0] = v_s;
              cout[sm] = v_m;
            } else {  // a smaller block follows
              const uint32_t D = (Mb - next_M) / 2u;
              if (s0 < D) {
                if (s0 < next_emit) nxtp[s0] = v_s;
              } else if (s0 < D + next_M) {
                cbuf[s0 - D] = v_s;
              }
              if (sm < D) {
                if (sm < next_emit) nxtp[sm] = v_m;
              } else if (sm < D + next_M) {
                cbuf[sm - D] = v_m;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) PB[u][k] = k >= 4 ? zb[u][k].y : -zb[u][k].x;
      }
      // the sample next to (set u, c' = 4 + j)'s even sample s comes from (set NS-1-u, c' = 3 - j) of the mirror lane
#pragma unroll
      for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = __shfl(ns_[u][j], 63 - (int)lane), b = __shfl(nm_[u][j], 63 - (int)lane);
          ns_[u][j] = a;
          nm_[u][j] = b;
        }
      // set u, c' = 4 + j: s = 2u + 2 NS kappa + 128 NS j; a lane's NS sets give 2 NS consecutive samples (and 2 NS descending ones)
      if (whole) {
        float* up = out + shift + 2u * NS * kap;
        float* dn = out + shift + Mb - 2u * NS - 2u * NS * kap;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int h = 0; h < NS / 2; ++h) {
            *(float4*)(up + 128 * NS * j + 4 * h) = make_float4(os[2 * h][j], ns_[NS - 1 - 2 * h][j], os[2 * h + 1][j], ns_[NS - 2 - 2 * h][j]);
            *(float4*)(dn - 128 * NS * j + 4 * h) =
                make_float4(nm_[2 * h][j], om[NS - 1 - 2 * h][j], nm_[2 * h + 1][j], om[NS - 2 - 2 * h][j]);
          }
        }
      } else if (emit) {
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t s0 = 2u * u + 2u * NS * kap + 128u * NS * j;
            const uint32_t f0 = s0 + shift, f1 = s0 + 1u + shift, f2m = Mb - 2u - s0 + shift, f3m = Mb - 1u - s0 + shift;
            if (f0 < emit) out[f0] = os[u][j];
            if (f1 < emit) out[f1] = ns_[NS - 1 - u][j];
            if (f2m < emit) out[f2m] = nm_[NS - 1 - u][j];
            if (f3m < emit) out[f3m] = om[u][j];
          }
      }
      if (__any(floor_bad)) {
        if (lane == 0) raise_status(A.status, VSYN_ST_FLOOR_VALUE, p);
        vmem_drain();
      }
      prev_kind = hand_over ? K_LDS : K_REG;
      prev_M = Mb;
      prev_next_long = cur_next_long;
      qnext = qn;
      if (qnext >= qb) break;
      continue;
    }
    load_residue(cur);
    load_rows(cur);
    ++it;
    const uint32_t LG = cur.LG, lng0 = cur.lng, Jp = cur.Jp, qn = cur.qn;
    const uint32_t Np = 1u << LG, M = 2u * Np, G = Np >> 3, lgG = LG - 3u, kmask = Np - 1u;
    const uint32_t p0 = sg.first_packet + cur.q;
    const lds_u32* const pinf = pinf_base + 64u * cur.buf;

    // ---- inverse coupling through the partner's image (hpp:1213-1241), as in vsyn_fused.h ------------------------------------------
    float2 r[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (((64u * t + lane) >> LG) >= Jp) raw[t] = f2(0.f, 0.f);  // elements beyond the pass's last packet
    if (ROLE == 3) {
      // General case: the C channel waves of this run (adjacent waves, channel 0 first) put their rows into their images; the coupling
      // steps of the pass's mapping are then replayed IN PLACE, last step first (hpp:1213-1241), each by the wave of its magnitude
      // channel, one after the other; finally every wave reads its own rows back. Two monotonic counters in channel 0's flag block:
      // [2] rows in place (one count per wave and pass), [3] steps done.
      uint8_t* const g0 = wmem - (size_t)c * wave_bytes;
      lds_u32* const gfl = (lds_u32*)(g0 + U_FLAG_OFF);
      const MapConst* mc = maps + cur.map;
      const uint32_t ncoup = __builtin_amdgcn_readfirstlane(mc->ncoup);
#pragma unroll
      for (int t = 0; t < 8; ++t) xb[t * 64 + lane] = raw[t];
      if (ncoup) {
        ++coupled_passes;  // (passes of a mapping without coupling steps take no part in the counters)
        if (lane == 0) __atomic_fetch_add(gfl + 2, 1u, __ATOMIC_RELAXED);
        for (uint32_t i = 0; i < ncoup; ++i) {
          const uint32_t k = ncoup - 1u - i;
          const uint32_t cm = __builtin_amdgcn_readfirstlane((uint32_t)mc->coup[2 * k]), ca = __builtin_amdgcn_readfirstlane((uint32_t)mc->coup[2 * k + 1]);
          if (c == cm) {
            group_wait(gfl + 2, C * coupled_passes, A.status);
            group_wait(gfl + 3, steps_done + i, A.status);
            float2* im = (float2*)(g0 + (size_t)cm * wave_bytes);
            float2* ia = (float2*)(g0 + (size_t)ca * wave_bytes);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              float2 vm = im[t * 64 + lane], va = ia[t * 64 + lane];
              inverse_couple(vm.x, va.x);
              inverse_couple(vm.y, va.y);
              im[t * 64 + lane] = vm;
              ia[t * 64 + lane] = va;
            }
            pair_post(gfl + 3, steps_done + i + 1u);
          }
        }
        group_wait(gfl + 3, steps_done + ncoup, A.status);
        steps_done += ncoup;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) r[t] = xb[t * 64 + lane];
    } else if (ROLE != 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) xb[t * 64 + lane] = raw[t];
      ++ep;
      pair_post(&my_flags[0], ep);
      pair_wait(&partner_flags[0], ep);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float2 oth = pxb[t * 64 + lane];
        r[t] = ROLE == 1 ? f2(couple_mag(raw[t].x, oth.x), couple_mag(raw[t].y, oth.y)) : f2(couple_ang(oth.x, raw[t].x), couple_ang(oth.y, raw[t].y));
      }
      pair_post(&my_flags[1], ep);
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) r[t] = raw[t];
    }
    (void)steps_done;
    (void)coupled_passes;
    // coded rows of this pass, two per register (the row registers are about to be reused)
    uint32_t vcur[U_MAX_J / 2];
#pragma unroll
    for (uint32_t j = 0; j < U_MAX_J / 2; ++j) vcur[j] = (vrow[2 * j] & 0xFFFFu) | (vrow[2 * j + 1] << 16);

    if (ROLE == 1 || ROLE == 2) pair_wait(&partner_flags[1], ep);  // the partner has read this wave's image: it may be reused (floor factors, FFT)

    // ---- floor curve + product (hpp:563-589, 1243-1255) --------------------------------------------------------------------------------
    // One table of per-interval line records per packet of the pass (65 records each, side by side in the exchange image, which is
    // idle between the hand-off and the FFT), then every ELEMENT looks its two bins up in the table of its own packet: no detour of
    // the factors through LDS in bin order (round 2's form: a 16-byte write and an 8-byte read per four / two bins, ~100 more vector
    // instructions per pass).
    bool floor_bad = false;
    uint32_t floor_bad_pkt = 0;
    uint32_t nocurve_mask = 0;
    float2* const tabs = xb;
#pragma unroll
    for (uint32_t j = 0; j < U_MAX_J; ++j) {
      if (j >= Jp) break;
      float2* const tab = tabs + U_TAB_STRIDE * j;
      const uint32_t own_j = pinf[8u * j + 5u], used_j = pinf[8u * j + 4u];
      if (!((own_j >> c) & 1u)) {
        nocurve_mask |= 1u << j;
        tab[lane] = f2(0.f, ((used_j >> c) & 1u) ? 256.5f : 255.5f);  // x1.0 resp. x0.0 (hpp:1159,1176-1179), as in vsyn_fused.h
        if (lane == 0) tab[64] = f2(0.f, ((used_j >> c) & 1u) ? 256.5f : 255.5f);
      } else {
        uint32_t v = (vcur[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
        if (lane >= posts) v = 0;
        const bool p65 = posts > 64u;
        uint32_t packed64 = 0;
        if (p65) {  // (rare: the extra load is exposed)
          const uint32_t v64 = __builtin_amdgcn_readfirstlane((uint32_t)(A.fy + ((size_t)(p0 + j) * C + c) * ys_stride)[sidx64]) & 0x7FFFu;
          packed64 = (xs64 << 16) | v64;
          if (lane == 0) tab[64] = f2(0.f, fminf((float)v64, 255.f));
          if (v64 > 255u) {
            floor_bad = true;
            floor_bad_pkt = p0 + j;
          }
        }
        const uint64_t mask = __ballot((v >> 15) != 0) | 1ull;
        const uint64_t below = mask & ((2ull << lane) - 1ull);
        const uint32_t lo = 63u - (uint32_t)__clzll((long long)below);
        const uint64_t above = lane < 63u ? (mask >> (lane + 1u)) : 0ull;
        const bool has_hi = above != 0ull || p65;
        const uint32_t hi = lane + (uint32_t)__ffsll((long long)above);
        const uint32_t packed = (xsl << 16) | (v & 0x7FFFu);
        const uint32_t plo = (uint32_t)__shfl((int)packed, (int)lo);
        uint32_t phi = (uint32_t)__shfl((int)packed, (int)(above != 0ull ? hi : lo));
        if (p65 && above == 0ull) phi = packed64;
        if ((v & 0x7FFFu) > 255u) {
          floor_bad = true;
          floor_bad_pkt = p0 + j;
        }
        const float x0 = (float)(plo >> 16), y0 = fminf((float)(plo & 0xFFFFu), 255.f);
        const float x1 = (float)(phi >> 16), y1 = fminf((float)(phi & 0xFFFFu), 255.f);
        const float inv = has_hi ? __builtin_amdgcn_rcpf(x1 - x0) : 0.f;
        const float ady = fabsf(y1 - y0);
        const float a = ady * inv, b = __builtin_fmaf(-ady, x0, 0.5f) * inv;
        tab[lane] = y1 >= y0 ? f2(a, b + y0) : f2(-a, (y0 + 1.f) - b);  // index = floor(x a + b), see vsyn_fused.h
      }
    }
    {
      // element (t, lane) = point k of packet jt: bins 2k, 2k + 1. Their x as floats from a laundered lane number (left to itself the
      // compiler hoists the per-register values out of the pass loop and spills them); 2 (lane & kmask) + an even literal is exact.
      uint32_t lane_x = lane;
      asm volatile("" : "+v"(lane_x));
      const float xfl = (float)(2u * (lane_x & kmask));
      const uint32_t tabs_base = (uint32_t)(uintptr_t)(lds_u32*)tabs;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t e = 64u * t + lane, je = e >> LG, jt = min(je, Jp - 1u);
        const uint32_t two = (bsege[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;  // interval of bin 2k | of bin 2k + 1 << 8
        const uint32_t tb = tabs_base + (U_TAB_STRIDE * 8u) * jt;
        const u_f32x2 e0 = *(const u_lds_f32x2*)(uintptr_t)(tb + 8u * (two & 0xFFu));
        const u_f32x2 e1 = *(const u_lds_f32x2*)(uintptr_t)(tb + 8u * (two >> 8));
        const float xk = xfl + (float)(2u * ((64u * (uint32_t)t) & kmask));
        const uint32_t i0 = (uint32_t)__builtin_fmaf(xk, e0.x, e0.y), i1 = (uint32_t)__builtin_fmaf(xk + 1.f, e1.x, e1.y);
        r[t] = f2(r[t].x * T.invdb[i0], r[t].y * T.invdb[i1]);
        if (TAPC && je < Jp && !((nocurve_mask >> jt) & 1u)) {  // feature tap "floor1 floor" (hpp:585): packed like the residue
          const uint64_t roff = ((uint64_t)pinf[8u * jt + 1u] << 32) | pinf[8u * jt];
          ((uint32_t*)(A.curve + roff + (size_t)c * M))[e & kmask] = i0 | (i1 << 16);
        }
      }
    }

    // ---- IMDCT: mirror element, pre-rotation, FFT-Np x J, post-rotation (mdct.cpp:433-527 by the DCT-IV route) ---------------------
    const ULdsSize& TS = T.sz[lng0 ? 1 : 0];
    float2 z[8];
    {
      const int ml = (int)(lane ^ min(63u, kmask));  // lane of the mirror point Np-1-k; its register is t ^ ((Np-1) >> 6)
      float im[8];
      const uint32_t tm = kmask >> 6;
      if (tm == 7u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) im[t] = __shfl(r[7 - t].y, ml);
      } else if (tm == 3u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) im[t] = __shfl(r[t ^ 3].y, ml);
      } else if (tm == 1u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) im[t] = __shfl(r[t ^ 1].y, ml);
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) im[t] = __shfl(r[t].y, ml);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) z[t] = cmulf(f2(r[t].x, im[t]), TS.pre[t][lane]);
    }
    {
      const uint32_t cl = lane & 7u, hi = lane >> 3;
      const uint32_t p1 = LG > 6u ? LG - 6u : 0u, p2 = LG >= 6u ? 3u : LG - 3u;
      if (p1) {
        u_dft_regs(z, p1);
#pragma unroll
        for (int t = 1; t < 8; ++t) z[t] = cmulf(z[t], TS.tw1[t][lane]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) xb[t * 72 + lane] = z[t];
#pragma unroll
      for (int a = 0; a < 8; ++a) z[a] = xb[hi * 72 + a * 8 + cl];
      u_dft_regs(z, p2);
#pragma unroll
      for (int a = 1; a < 8; ++a) z[a] = cmulf(z[a], TS.tw2[a][cl]);
#pragma unroll
      for (int a = 0; a < 8; ++a) xb[a * 65 + lane] = z[a];
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = xb[cl * 65 + hi * 8 + k];
      dft8(z);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = cmulf(z[k], TS.post[k][lane]);

    // ---- window + overlap-add + PCM (hpp:1008-1059) --------------------------------------------------------------------------------
    // bin f = kappa + G c' of packet jo = lane >> lgG; left-half samples s = 2f - Np (c' >= 4) resp. Np - 1 - 2f, and M-1-s
    const uint32_t jo = lane >> lgG, gl = lane & (G - 1u);
    const uint32_t kappa = LG >= 6u ? (((lane >> 3) & ((1u << (LG - 6u)) - 1u)) | ((lane & 7u) << (LG - 6u))) : gl;
    const bool valid_o = jo < Jp;
    const uint32_t jq = min(jo, Jp - 1u);
    const u_u32x4 oi = *(const u_lds_u32x4*)(pinf + 8u * jq);
    const uint32_t widx_o = pinf[8u * jq + 6u];
    const uint32_t emit = valid_o ? oi[3] : 0u;
    float* const out = plane + oi[2];
    const bool grp0 = jo == 0u;
    // chunk of a packet = [centre of the previous block, centre of this one): left-half sample s sits at frame s + shift
    const uint32_t shift = (grp0 && prev_M && prev_M != M) ? (uint32_t)(((int32_t)prev_M - (int32_t)M) / 2) : 0u;  // negative (wrapped) after a smaller block
    const uint32_t fL = lng0 ? (widx_o & 1u) : 0u;
    uint32_t fR = lng0 ? prev_next_long : 0u;
    if (lng0 && !grp0) fR = (pinf[8u * (jq ? jq - 1u : 0u) + 6u] >> 1) & 1u;
    float wl0[8], wl1[8], wr0[8], wr1[8], Pin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      wl0[k] = TS.win[fL][0][k][lane];
      wl1[k] = TS.win[fL][1][k][lane];
    }
    if (__all(fL == fR)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wr0[k] = wl0[k];
        wr1[k] = wl1[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wr0[k] = TS.win[fR][0][k][lane];
        wr1[k] = TS.win[fR][1][k][lane];
      }
    }
    // this pass's own right-half values (unwindowed), then the overlap term of every packet: from the packet G lanes below, or
    // (first packet of the pass) from the previous pass
    float Pn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) Pn[k] = k >= 4 ? z[k].y : -z[k].x;
    if (G < 64u) {
      const int below = (int)((lane - G) & 63u);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float sh = __shfl(Pn[k], below);
        Pin[k] = grp0 ? P[k] : sh;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) Pin[k] = P[k];
    }
    if (prev_kind == K_CARRY && prev_M != M) {
      // carry-in of another block size (wave-uniform, first pass only): into the carry image, as a block of this submit would have
      // left it; a larger block also owns the first D frames of the chunk outright
      if (prev_M > M) {
        const uint32_t D = (prev_M - M) / 2u;
        const uint32_t e0 = __builtin_amdgcn_readfirstlane(pinf[3]), o0 = __builtin_amdgcn_readfirstlane(pinf[2]);  // the pass's first packet
        for (uint32_t i = lane; i < min(D, e0); i += 64) plane[o0 + i] = cin[i];
        for (uint32_t i = lane; i < M; i += 64) cbuf[i] = cin[D + i];
      } else {
        for (uint32_t i = lane; i < prev_M; i += 64) cbuf[i] = cin[i];
      }
      prev_kind = K_LDS;
    }
    if (prev_kind != K_REG && grp0) {
      // the overlap term arrives already windowed (carry-in of an earlier submit, or the previous block had another size):
      // as P = 1 and "window" = the value itself (1 * x == x), so that the arithmetic below stays one straight line
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t f = kappa + G * k;
        const uint32_t s = k >= 4 ? 2u * f - Np : Np - 1u - 2u * f;
        float vs, vm;
        if (prev_kind == K_CARRY) {  // same size: natural order
          vs = cin[s];
          vm = cin[M - 1u - s];
        } else if (prev_M < M) {  // after a smaller block: its prev_M windowed samples meet this block's samples [D, D + prev_M)
          const uint32_t D = (M - prev_M) / 2u;
          const bool in = s >= D && s < D + prev_M;
          vs = in ? cbuf[s - D] : 0.f;
          vm = in ? cbuf[M - 1u - s - D] : 0.f;
        } else {  // after a larger block: the image holds the M frames of its right half that overlap this block
          vs = cbuf[s];
          vm = cbuf[M - 1u - s];
        }
        Pin[k] = 1.f;
        wr1[k] = vs;
        wr0[k] = vm;
      }
    }
    // Everything in flight here is loads (the next pass's residue, rows and descriptors; carry-in reads): finish them before the PCM
    // stores go out, so that nothing later in the loop ever waits for a store (see vmem_drain)
    vmem_drain();
    const bool last_of_segment = qn == num;
    uint32_t next_M = 0, next_emit = 0, next_out = 0;
    if (qn < qb) {  // the descriptors behind this pass have landed by now (drained above)
      const uint32_t nbad = __builtin_amdgcn_readfirstlane((dnext[7] >> 8) & 0xFFu), nlng = __builtin_amdgcn_readfirstlane((dnext[6] >> 16) & 0xFFu);
      next_M = nbad ? 0u : (2u << (nlng ? lgp[1] : lgp[0]));
      next_emit = __builtin_amdgcn_readfirstlane(qn < qa ? 0u : dnext[3]);
      next_out = __builtin_amdgcn_readfirstlane(dnext[2]);
    }
    const bool hand_over = next_M && next_M != M;
    float oh_s[4], oh_m[4], n_s[4], n_m[4];
    const int mirror = (int)(lane ^ (G - 1u));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kh = 4 + j, kl = 3 - j;
      const float cch = z[kh].x, ccl = -z[kl].y;
      const float ah_s = Pin[kh] * wr1[kh], ah_m = Pin[kh] * wr0[kh], al_s = Pin[kl] * wr1[kl], al_m = Pin[kl] * wr0[kl];
      oh_s[j] = ah_s + cch * wl0[kh];
      oh_m[j] = ah_m + (-cch) * wl1[kh];
      const float ol_s = al_s + ccl * wl0[kl];
      const float ol_m = al_m + (-ccl) * wl1[kl];
      n_s[j] = __shfl(ol_s, mirror);  // bin Np-1-f (mirror lane of the group, register 7-c') yields the neighbouring samples
      n_m[j] = __shfl(ol_m, mirror);
    }
    {
      // sample s = 2 kappa + 2 G j of (lane, c' = 4 + j): s, s+1 and M-2-s, M-1-s
      const bool whole = emit == M + shift && (int)shift >= 0 && (((uintptr_t)out & 7u) == 0);  // every left-half sample is inside the chunk
      if (__all(whole || !valid_o)) {
        if (valid_o) {
          float* up = out + shift + 2u * kappa;
          float* dn = out + shift + M - 2u - 2u * kappa;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            *(float2*)(up + 2u * G * j) = f2(oh_s[j], n_s[j]);
            *(float2*)(dn - 2u * G * j) = f2(n_m[j], oh_m[j]);
          }
        }
      } else if (emit) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t s = 2u * kappa + 2u * G * j;
          const uint32_t f0 = s + shift, f1 = s + 1u + shift, f2m = M - 2u - s + shift, f3m = M - 1u - s + shift;
          if (f0 < emit) out[f0] = oh_s[j];
          if (f1 < emit) out[f1] = n_s[j];
          if (f2m < emit) out[f2m] = n_m[j];
          if (f3m < emit) out[f3m] = oh_m[j];
        }
      }
    }

    // ---- what the last packet of the pass leaves behind -----------------------------------------------------------------------------
    const uint32_t cur_next_long = lng0 ? (cur.last_widx >> 1) & 1u : 0u;
    if ((last_of_segment || hand_over) && jo == Jp - 1u) {
      // windowed right half in natural order: sample s of bin f at position s. Segment end: all of it into the stream's carry buffer.
      // Before a smaller block: frames [0, D) are final and go straight into the next chunk, [D, D + next_M) into the carry image,
      // the rest is windowed to zero. Before a larger block: all M frames into the carry image.
      float* cout = A.carry + (si.parity_in ^ 1u) * carry_half + ((size_t)sg.stream * C + c) * half1;
      float* nxtp = plane + next_out;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t f = kappa + G * k;
        const uint32_t s = k >= 4 ? 2u * f - Np : Np - 1u - 2u * f, sm = M - 1u - s;
        const float v_s = Pn[k] * TS.win[cur_next_long][1][k][lane], v_m = Pn[k] * TS.win[cur_next_long][0][k][lane];
        if (last_of_segment) {
          cout[s] = v_s;
          cout[sm] = v_m;
        } else if (next_M < M) {
          const uint32_t D = (M - next_M) / 2u;
          if (s < D) {
            if (s < next_emit) nxtp[s] = v_s;
          } else if (s < D + next_M) {
            cbuf[s - D] = v_s;
          }
          if (sm < D) {
            if (sm < next_emit) nxtp[sm] = v_m;
          } else if (sm < D + next_M) {
            cbuf[sm - D] = v_m;
          }
        } else {
          cbuf[s] = v_s;
          cbuf[sm] = v_m;
        }
      }
    }
    // carry registers for the next pass: the last packet's values, moved to the lanes of group 0
    if (G < 64u) {
      const int from = (int)((Jp - 1u) * G + gl);
#pragma unroll
      for (int k = 0; k < 8; ++k) P[k] = __shfl(Pn[k], from);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) P[k] = Pn[k];
    }
    if (__any(floor_bad)) {
      if (floor_bad) raise_status(A.status, VSYN_ST_FLOOR_VALUE, floor_bad_pkt);
      vmem_drain();
    }
    prev_kind = hand_over ? K_LDS : K_REG;
    prev_M = M;
    prev_next_long = cur_next_long;
    qnext = qn;
    if (qnext >= qb) break;
  }
}

// grid: groups of WPB = blockDim.x / 64 units (segment, run, channel), flattened as in vsyn_fused_kernel; the channels of a run are
// adjacent waves of one workgroup. Takes every run of class 2 (with the tuned long-run kernel absent: every run).
template <int UNS, bool TAPC>
__device__ __forceinline__ void u_kernel_body(const UArgs& U) {
  extern __shared__ __attribute__((aligned(16))) uint8_t u_lds[];
  const FusedArgs& A = U.f;
  const ConstHeader* H = hdr_of(A.cb);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t WPB = blockDim.x >> 6;
  const uint32_t C = H->channels;
  const uint32_t per_seg = A.runs_per_seg * C;
  const uint32_t unit = blockIdx.x * WPB + wave;
  const uint32_t g = __builtin_amdgcn_readfirstlane(unit / per_seg);
  const uint32_t rem = unit - g * per_seg;
  const uint32_t run = __builtin_amdgcn_readfirstlane(rem / C), c = __builtin_amdgcn_readfirstlane(rem % C);
  vsyn_segment sg = {};
  SegInfo si = {};
  uint32_t cls = 0xFFu;
  if (g < A.S) {
    sg = A.segs[g];
    if (sg.stream < H->max_streams && !(sg.residue_off & 3)) {
      si = A.sinfo[g];
      cls = __builtin_amdgcn_readfirstlane((uint32_t)A.run_cls[(size_t)g * A.runs_per_seg + run]);
    }
  }
  const uint32_t qa = run * A.R;
  const uint32_t qb = min(sg.num_packets, qa + A.R);
  const bool active = cls == 2u && (A.fused_ok & 2u);
  if (!__syncthreads_or(active ? 1 : 0)) return;
  {
    const uint4* src = (const uint4*)U.img;
    uint4* dst = (uint4*)u_lds;
    for (uint32_t i = threadIdx.x; i < U.table_bytes / 16; i += blockDim.x) dst[i] = src[i];
  }
  uint8_t* wmem = u_lds + U.table_bytes + wave * U.wave_bytes;
  if (lane < 4) ((uint32_t*)(wmem + U_FLAG_OFF))[lane] = 0u;
  __syncthreads();
  if (!active) return;
  const ULdsImage& T = *(const ULdsImage*)u_lds;
  const uint32_t mag = U.role_mode == 1 ? 0u : 1u;
  const int role = U.role_mode == 3 ? 3 : ((U.role_mode == 0 || C < 2) ? 0 : (c == mag ? 1 : 2));
  const uint32_t pw = (role == 1 || role == 2) ? (wave ^ 1u) : wave;
  const uint8_t* pmem = u_lds + U.table_bytes + pw * U.wave_bytes;
  if (role == 0) u_run<0, UNS, TAPC>(A, T, wmem, pmem, U.wave_bytes, lane, g, sg, si, qa, qb, C, c);
  else if (role == 1) u_run<1, UNS, TAPC>(A, T, wmem, pmem, U.wave_bytes, lane, g, sg, si, qa, qb, C, c);
  else if (role == 2) u_run<2, UNS, TAPC>(A, T, wmem, pmem, U.wave_bytes, lane, g, sg, si, qa, qb, C, c);
  else u_run<3, UNS, TAPC>(A, T, wmem, pmem, U.wave_bytes, lane, g, sg, si, qa, qb, C, c);
}

template <int UNS>
__global__ void __launch_bounds__(UNS == 1 ? U_MAX_THREADS : 512) vsyn_fused_u_kernel(const UArgs U) { u_kernel_body<UNS, false>(U); }
// the same kernel with the "floor1 floor" feature tap (returnn_import.py:74-115 builds its features from that curve for
// ANY file, so every block-size pair has the tap on its fast path) written on the way — its own launch, as in vsyn_fused.h
template <int UNS>
__global__ void __launch_bounds__(UNS == 1 ? U_MAX_THREADS : 512) vsyn_fused_u_tap_kernel(const UArgs U) { u_kernel_body<UNS, true>(U); }

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct UTables {
  ULdsImage* d_img = nullptr;
  uint32_t wave_bytes = 0, table_bytes = 0, waves_per_block = 0, waves_per_cu = 0, role_mode = 0, ns = 1;
};

static inline uint32_t u_role_mode(const ConstHeader& H, const uint8_t* host_const);
static inline bool u_supported(const ConstHeader& H, const uint8_t* host_const) {
  // blocksize1 up to 8192 (register sets, UBig); blocksize0 must fit the packed passes. The channel waves of a run share one workgroup
  // (16 waves, 8 with blocks above 2048: u_tables_create turns a setup down that does not fit) when any mapping couples channels; a
  // setup without coupling steps has no such tie: any channel count.
  // (a blocksize0 above 2048 only when it equals blocksize1: every block then takes the register-set path, whatever its mode says)
  if ((H.bs[0] > 2048 && H.bs[0] != H.bs[1]) || (H.channels > U_MAX_CH && u_role_mode(H, host_const) != 0u)) return false;
  const FloorConst* fl = (const FloorConst*)(host_const + H.off_floor);
  for (uint32_t f = 0; f < H.num_floors; ++f)
    if (fl[f].posts > VSYN_MAX_POSTS) return false;  // (65 posts: the last sorted post is kept wave-uniform, see u_run)
  return true;
}

// 0 / 1 / 2: stereo or mono with the same (at most one) coupling step in every mapping — the pairwise row swap (0: no coupling at all,
// any channel count); 3: the general replay
static inline uint32_t u_role_mode(const ConstHeader& H, const uint8_t* host_const) {
  const MapConst* mp = (const MapConst*)(host_const + H.off_map);
  if (H.channels > 2) {  // no coupling step in any mapping in use: every channel wave on its own (role 0), whatever the channel count
    bool any = false;
    for (uint32_t k = 0; k < H.num_modes; ++k) any = any || mp[H.mode_mapping[k]].ncoup != 0;
    return any ? 3u : 0u;
  }
  int want = -2;
  for (uint32_t k = 0; k < H.num_modes; ++k) {
    const MapConst& m = mp[H.mode_mapping[k]];
    const int cur = m.ncoup == 0 ? 0 : (m.ncoup == 1 ? (m.coup[0] == 0 ? 1 : 2) : -1);
    if (cur < 0) return 3;
    if (want == -2) want = cur;
    else if (want != cur) return 3;
  }
  return want < 0 ? 0u : (uint32_t)want;
}

static inline void u_fill_size(const ConstHeader& H, const uint8_t* host_const, int b, ULdsSize& S) {
  const uint32_t n = H.bs[b], M = n / 2, Np = n / 4;
  uint32_t LG = 0;
  while ((1u << LG) < Np) ++LG;
  const uint32_t p1 = LG > 6 ? LG - 6 : 0, R1 = 1u << p1, G = Np / 8;
  const float2* pre = (const float2*)(host_const + H.off_pre[b]);
  const float2* post = (const float2*)(host_const + H.off_post[b]);
  const float2* tw = (const float2*)(host_const + H.off_fft[b]);  // W_Np^j
  const float* win = (const float*)(host_const + H.off_win[b]);
  for (uint32_t l = 0; l < 64; ++l) {
    const uint32_t gl = l & (G - 1), kappa = LG >= 6 ? (((l >> 3) & (R1 - 1)) | ((l & 7) << p1)) : gl;
    for (uint32_t t = 0; t < 8; ++t) {
      const uint32_t e = 64 * t + l, k = e & (Np - 1);
      S.pre[t][l] = pre[k];
      const uint32_t tl = t & (R1 - 1);
      S.tw1[t][l] = LG >= 6 ? tw[(l * tl) & (Np - 1)] : make_float2(1.f, 0.f);
      const uint32_t f = kappa + G * t;  // t plays c' here
      S.post[t][l] = post[f & (Np - 1)];
      const uint32_t s = t >= 4 ? 2 * f - Np : Np - 1 - 2 * f;
      for (uint32_t fl = 0; fl < 2; ++fl) {
        const uint32_t widx = b ? fl : 0;  // long: left half depends on the prev flag only (widx = prev + 2 next); short: one window
        S.win[fl][0][t][l] = win[(size_t)widx * n + (s & (M - 1))];
        S.win[fl][1][t][l] = win[(size_t)widx * n + ((M - 1 - s) & (M - 1))];
      }
    }
  }
  for (uint32_t a = 0; a < 8; ++a)
    for (uint32_t c = 0; c < 8; ++c) {
      if (LG >= 6) {
        S.tw2[a][c] = tw[(c * a * (Np / 64)) & (Np - 1)];
      } else {
        const uint32_t R2 = Np / 8;
        S.tw2[a][c] = tw[(c * (a & (R2 - 1))) & (Np - 1)];
      }
    }
}

template <int NS>
static inline void u_fill_big(const ConstHeader& H, const uint8_t* host_const, ULdsSize& F, UBig<NS>& B) {
  const uint32_t n = H.bs[1], M = n / 2, Np = n / 4;
  const float2* pre = (const float2*)(host_const + H.off_pre[1]);
  const float2* post = (const float2*)(host_const + H.off_post[1]);
  const float2* tw = (const float2*)(host_const + H.off_fft[1]);  // W_Np^j
  const float* win = (const float*)(host_const + H.off_win[1]);
  memset(&F, 0, sizeof(F));
  for (uint32_t l = 0; l < 64; ++l) {
    const uint32_t kappa = ((l & 7) << 3) | (l >> 3);
    for (uint32_t t = 0; t < 8; ++t) {
      F.tw1[t][l] = tw[((l * t) & 511u) * NS];  // W512^(l t)
      for (uint32_t u = 0; u < (uint32_t)NS; ++u) {
        B.pre[u][t][l] = pre[512 * u + 64 * t + l];
        B.tw0[u][t][l] = tw[((64 * t + l) * u) & (Np - 1)];
        const uint32_t f = u + NS * (kappa + 64 * t);  // t plays c' here
        B.post[u][t][l] = post[f];
        const uint32_t s = t >= 4 ? 2 * f - Np : Np - 1 - 2 * f;
        for (uint32_t fl = 0; fl < 2; ++fl) {
          B.win[fl][0][u][t][l] = win[(size_t)fl * n + s];
          B.win[fl][1][u][t][l] = win[(size_t)fl * n + (M - 1 - s)];
        }
      }
    }
  }
  for (uint32_t a = 0; a < 8; ++a)
    for (uint32_t c = 0; c < 8; ++c) F.tw2[a][c] = tw[((8 * c * a) & 511u) * NS];  // W64^(c a)
}

static inline hipError_t u_tables_create(const ConstHeader& H, const uint8_t* host_const, UTables* ut) {
  const uint32_t ns = H.bs[1] > 2048 ? H.bs[1] / 2048 : 1;
  const uint32_t img_bytes = (uint32_t)((sizeof(ULdsImage) + 15u) & ~15u);
  const uint32_t big_bytes = ns == 2 ? (uint32_t)sizeof(UBig<2>) : (ns == 4 ? (uint32_t)sizeof(UBig<4>) : 0u);
  std::vector<uint8_t> blob(img_bytes + big_bytes, 0);
  ULdsImage& im = *(ULdsImage*)blob.data();
  if (H.bs[0] <= 2048) u_fill_size(H, host_const, 0, im.sz[0]);  // (the packed passes' tables; unused when every block is above 2048)
  if (ns == 1) u_fill_size(H, host_const, 1, im.sz[1]);
  else if (ns == 2) u_fill_big<2>(H, host_const, im.sz[1], *(UBig<2>*)(blob.data() + img_bytes));
  else u_fill_big<4>(H, host_const, im.sz[1], *(UBig<4>*)(blob.data() + img_bytes));
  memcpy(im.invdb, host_const + H.off_invdb, 256 * sizeof(float));
  hipError_t e = hipMalloc((void**)&ut->d_img, blob.size());
  if (e != hipSuccess) return e;
  e = hipMemcpy(ut->d_img, blob.data(), blob.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return e;
  ut->ns = ns;
  ut->table_bytes = (uint32_t)blob.size();
  const uint32_t cbuf = H.bs[0] != H.bs[1] ? (H.bs[0] / 2) * 4u : 16u;
  ut->wave_bytes = (U_CBUF_OFF + cbuf + 15u) & ~15u;
  const uint32_t budget = 156u * 1024u;
  uint32_t w = (budget - ut->table_bytes) / ut->wave_bytes;
  w = std::min<uint32_t>(ns == 1 ? U_MAX_THREADS / 64u : 8u, w);
  ut->role_mode = u_role_mode(H, host_const);
  // the channel waves of a run share a workgroup: a multiple of the channel count (of 2 for the pairwise swap)
  const uint32_t grp = ut->role_mode == 3 ? H.channels : 2u;
  if (w < grp) return hipErrorInvalidValue;
  ut->waves_per_block = w / grp * grp;
  ut->waves_per_cu = ut->waves_per_block;  // one workgroup per CU (the tables take a quarter of the LDS)
  const void* kfn = ns == 1 ? (const void*)vsyn_fused_u_kernel<1> : (ns == 2 ? (const void*)vsyn_fused_u_kernel<2> : (const void*)vsyn_fused_u_kernel<4>);
  // (the attribute belongs to the kernel, not to the handle: one fixed value — the planning budget above — for every handle, so that a
  // later handle with a smaller LDS block never lowers the limit under an earlier one's launches)
  e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
  if (e == hipSuccess) {
    const void* kft = ns == 1 ? (const void*)vsyn_fused_u_tap_kernel<1> : (ns == 2 ? (const void*)vsyn_fused_u_tap_kernel<2> : (const void*)vsyn_fused_u_tap_kernel<4>);
    e = hipFuncSetAttribute(kft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
  }
  if (e != hipSuccess && getenv("VSYN_DEBUG"))
    fprintf(stderr, "vsyn: hipFuncSetAttribute(%u B dynamic LDS) failed: %s\n", budget, hipGetErrorString(e));
  return e == hipSuccess ? hipSuccess : hipErrorInvalidValue;
}

static inline void u_tables_destroy(UTables* ut) {
  if (ut->d_img) (void)hipFree(ut->d_img);
  ut->d_img = nullptr;
}

static inline hipError_t u_launch(const ConstHeader& H, const UTables& ut, const FusedArgs& a, hipStream_t s) {
  const uint64_t units = (uint64_t)a.S * a.runs_per_seg * H.channels;
  if (units == 0 || units > 0x7FFFFFF0ull) return hipErrorInvalidValue;
  UArgs u;
  u.f = a;
  u.img = ut.d_img;
  u.wave_bytes = ut.wave_bytes;
  u.table_bytes = ut.table_bytes;
  u.role_mode = ut.role_mode;
  const uint32_t wpb = ut.waves_per_block;
  dim3 grid((uint32_t)((units + wpb - 1) / wpb));
  const size_t lds = ut.table_bytes + (size_t)wpb * ut.wave_bytes;
  if (a.curve) {
    if (ut.ns == 1) vsyn_fused_u_tap_kernel<1><<<grid, wpb * 64, lds, s>>>(u);
    else if (ut.ns == 2) vsyn_fused_u_tap_kernel<2><<<grid, wpb * 64, lds, s>>>(u);
    else vsyn_fused_u_tap_kernel<4><<<grid, wpb * 64, lds, s>>>(u);
  } else if (ut.ns == 1) vsyn_fused_u_kernel<1><<<grid, wpb * 64, lds, s>>>(u);
  else if (ut.ns == 2) vsyn_fused_u_kernel<2><<<grid, wpb * 64, lds, s>>>(u);
  else vsyn_fused_u_kernel<4><<<grid, wpb * 64, lds, s>>>(u);
  return hipGetLastError();
}
