// vorbis_synth_hip.hip — host side of the C-ABI in include/vorbis_synth_hip.h + kernel launches.
// gfx950 (MI355X) only; built by __graft_entry__.build() with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// No CPU compute path exists in this library: without a HIP device every entry point returns
// VSYN_ERR_NO_DEVICE.  The only host arithmetic is the once-per-stream constant block (twiddles, windows,
// floor neighbour tables), which the reference also builds once per stream (mdct.cpp:88-127, hpp:837-862).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include <stdlib.h>

#include "vsyn_device.h"
#include "vsyn_staged.h"
#include "vsyn_prep.h"
#include "vsyn_fused.h"
#include "vsyn_fused_u.h"
#include "vsyn_vq.h"
#include "vsyn_pcm.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

static const uint32_t k_inverse_db_bits[256] = {
#include "vorbis_floor1_inverse_db.inc"
};

namespace {

thread_local char g_err[512];

int fail(const char** err, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  if (err) *err = g_err;
  return code;
}

#define HIPCHK(call)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(err, VSYN_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
struct DevBuf {  // grow-only device buffer
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  ~DevBuf() { release(); }  // (vsyn_destroy selects the device before the handle goes away)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }
uint32_t ilog2(uint32_t v) {
  uint32_t r = 0;
  while ((1u << r) < v) ++r;
  return r;
}
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct vsyn_handle {
  int device = 0;
  ConstHeader H{};
  std::vector<uint8_t> host_const;
  uint8_t* d_const = nullptr;
  uint8_t* d_vq = nullptr;             // residue VQ stage: VqHeader, books, residues, maps, value pool (vsyn_attach_vq)
  uint32_t vq_lds_bytes = 0;           // dynamic LDS of the residue VQ kernel
  uint32_t vq_grid = 0;                // workgroups of the VQ kernel that are resident at once (its grid: every wave walks packets)
  uint32_t vq_waves = 1;               // waves per workgroup: 1, or (value tables shared in LDS) several
  bool vq_tables_in_lds = false;
  uint32_t last_S = 0, last_wb = 0;    // segments / workspace half of the most recent submit (vsyn_pcm_interleave_device)
  DevBuf<vsyn_vq_packet> st_vqpk;      // vsyn_submit_host_vq staging
  DevBuf<uint8_t> st_cls;
  DevBuf<uint16_t> st_ent;
  StreamState* d_state = nullptr;
  float* d_carry = nullptr;
  DevStatus* d_status = nullptr;
  FusedTables fused{};
  UTables utab{};
  bool fused_ok = false;
  uint32_t fused_mask = 0;             // what the layout kernel classifies by: bit 0 long-run kernel usable, bit 1 mixed-block runs fused too
  uint32_t tuned_mask = 0;             // the same without the size-generic kernel (used when a tap it cannot write is requested)
  bool u_mixed = false;                // class-2 runs go to the size-generic kernel (vsyn_fused_u.h) instead of fused_run<.., MIXED>
  int num_cus = 256;
  hipStream_t host_stream = nullptr;   // vsyn_submit_host: copies in, kernels, copies out
  hipStream_t side = nullptr;          // the (usually empty) staged work list runs beside the fused kernel
  hipStream_t pre = nullptr;           // layout + floor unwrap of submit i+1 run beside the fused kernel of submit i
  hipEvent_t ev_join = nullptr, ev_self = nullptr;
  // Workspace ring: submit i uses slot i % WS_RING. When its preparation kernels run on the internal stream `pre` (beside the previous
  // submit's synthesis kernel) the slot's previous user, submit i - WS_RING, must be done: known from an event recorded on the caller's
  // stream behind every EV_EVERY-th submit whose preparation ran there. WS_RING = 2, EV_EVERY = 1 on purpose: a deeper ring lets the
  // preparation run further ahead, but then its workgroups land in the MIDDLE of an exact-fit synthesis grid instead of at its start
  // (measured with 8 / 4: config 3's synthesis kernel 0.258 instead of 0.242 ms).
  static constexpr uint32_t WS_RING = 2, EV_EVERY = 1, EV_RING = 2, CNT_RING = 4;
  hipEvent_t ev_pre_done[WS_RING] = {}, ev_ring[EV_RING] = {};
  uint64_t ev_ring_submit[EV_RING] = {~0ull, ~0ull};  // submit index each ring event was recorded behind
  bool pre_done_valid[WS_RING] = {};
  hipStream_t last_pre_stream = nullptr;  // stream the previous submit's preparation ran on (valid iff pre_done_valid[its slot])
  bool last_prep_on_main = false;      // the previous submit's preparation ran on the caller's stream
  bool last_ran_layout = true;         // the previous submit ran vsyn_layout_kernel (it keeps the staged list counters one slot ahead)
  uint64_t long_modes = 0;             // bit m: mode m selects a long block
  uint64_t nsub = 0;                   // submits so far
  uint32_t prep_lds_bytes = 0;         // dynamic LDS of vsyn_prep_kernel: one 32-bit column of the longest floor's posts per thread
  uint32_t submit_count = 0;
  // workspace
  // per-batch workspace, a ring indexed by the submit number so that the preparation of later submits can run ahead
  DevBuf<uint32_t> ws_list[WS_RING];  // staged work list
  DevBuf<uint32_t> ws_count;    // its counters: a ring of CNT_RING (the layout kernel of submit i clears the slot of submit i+1)
  DevBuf<PktInfo> ws_info[WS_RING];
  DevBuf<SegInfo> ws_seg[WS_RING];
  DevBuf<uint32_t> ws_segmap[WS_RING];
  DevBuf<uint16_t> ws_fy[WS_RING];
  DevBuf<uint8_t> ws_runcls[WS_RING];
  DevBuf<LayoutChunk> ws_chunks;   // look-back records of the chunked layout scan (segments beyond LAYOUT_CHUNK_PACKETS)
  DevBuf<float> ws_env, ws_blk;
  // host-submit staging
  DevBuf<vsyn_packet> st_pk;
  DevBuf<vsyn_segment> st_seg;
  DevBuf<uint16_t> st_ys, st_fy;
  DevBuf<float> st_res, st_pcm, st_env, st_blk;
  DevBuf<uint16_t> st_curve;
  DevBuf<uint32_t> st_emit;
  DevBuf<double> st_sum;               // vsyn_pcm_abs_sum_host
  DevBuf<uint8_t> st_conv;             // vsyn_pcm_fetch_host: interleaved output
  DevBuf<uint32_t> st_frames;
  uint64_t last_host_plane = 0;        // plane_stride of the most recent vsyn_submit_host* (0: none yet)
  // profiling
  bool profile = false;
  int profile_which = 1;  // 1 / 2: the fused kernel (steady / mixed workloads: same kernel), 3: residue VQ kernel
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  const char* profile_kernel = "";
  std::mutex mu;
};

// ------------------------------------------------------------------------------------------------
// constant block
// ------------------------------------------------------------------------------------------------
static void host_window(uint32_t bs0, uint32_t bs1, int lng, int prev, int next, float* w) {
  // VorbisModeNumber::precalc, hpp:837-862 (same float/double mix: sinf of a double-computed argument)
  const uint32_t n = lng ? bs1 : bs0;
  if (!lng) prev = next = 0;
  const uint32_t left = (prev ? bs1 : bs0) / 2, right = (next ? bs1 : bs0) / 2;
  const uint32_t left_begin = n / 4 - left / 2, right_begin = n - n / 4 - right / 2;
  for (uint32_t i = 0; i < n; ++i) w[i] = 0.f;
  for (uint32_t i = 0; i < left; ++i) {
    float x = sinf((float)(M_PI_2 * (i + 0.5) / left));
    w[left_begin + i] = sinf((float)(M_PI_2 * x * x));
  }
  for (uint32_t i = left_begin + left; i < right_begin; ++i) w[i] = 1.f;
  for (uint32_t i = 0; i < right; ++i) {
    float x = sinf((float)(M_PI_2 * (right - i - .5) / right));
    w[right_begin + i] = sinf((float)(M_PI_2 * x * x));
  }
}

static int build_const(const vsyn_setup* su, uint32_t max_streams, vsyn_handle* h, const char** err) {
  if (!su) return fail(err, VSYN_ERR_INVALID, "setup is NULL");
  if (su->channels < 1 || su->channels > VSYN_MAX_CHANNELS) return fail(err, VSYN_ERR_INVALID, "channels %u not in 1..%d", su->channels, VSYN_MAX_CHANNELS);
  if (!is_pow2(su->blocksize0) || !is_pow2(su->blocksize1) || su->blocksize0 < VSYN_MIN_BLOCKSIZE ||
      su->blocksize1 > VSYN_MAX_BLOCKSIZE || su->blocksize0 > su->blocksize1)
    return fail(err, VSYN_ERR_INVALID, "blocksizes %u/%u invalid (hpp:1294-1298)", su->blocksize0, su->blocksize1);
  if (su->num_floors < 1 || su->num_floors > VSYN_MAX_TABLES || su->num_mappings < 1 || su->num_mappings > VSYN_MAX_TABLES ||
      su->num_modes < 1 || su->num_modes > VSYN_MAX_TABLES || !su->floors || !su->mappings || !su->modes)
    return fail(err, VSYN_ERR_INVALID, "floor/mapping/mode counts out of range");
  if (max_streams < 1) return fail(err, VSYN_ERR_INVALID, "max_streams must be >= 1");

  ConstHeader& H = h->H;
  memset(&H, 0, sizeof(H));
  H.channels = su->channels;
  H.bs[0] = su->blocksize0;
  H.bs[1] = su->blocksize1;
  H.lg[0] = ilog2(su->blocksize0);
  H.lg[1] = ilog2(su->blocksize1);
  H.num_floors = su->num_floors;
  H.num_mappings = su->num_mappings;
  H.num_modes = su->num_modes;
  H.max_streams = max_streams;

  std::vector<FloorConst> floors(su->num_floors);
  uint32_t maxp = 2;
  for (uint32_t f = 0; f < su->num_floors; ++f) {
    const vsyn_floor1& sf = su->floors[f];
    FloorConst& fc = floors[f];
    memset(&fc, 0, sizeof(fc));
    if (sf.multiplier < 1 || sf.multiplier > 4) return fail(err, VSYN_ERR_INVALID, "floor %u: multiplier %u (hpp:486-492)", f, sf.multiplier);
    if (sf.num_posts < 2 || sf.num_posts > VSYN_MAX_POSTS || !sf.xs) return fail(err, VSYN_ERR_INVALID, "floor %u: %u posts", f, sf.num_posts);
    static const uint32_t range_of[5] = {0, 256, 128, 86, 64};
    fc.mult = sf.multiplier;
    fc.posts = sf.num_posts;
    fc.range = range_of[sf.multiplier];
    maxp = std::max(maxp, sf.num_posts);
    for (uint32_t i = 0; i < sf.num_posts; ++i) {
      if (sf.xs[i] > 0xFFFFu) return fail(err, VSYN_ERR_INVALID, "floor %u: x[%u]=%u too large", f, i, sf.xs[i]);
      fc.xs[i] = (uint16_t)sf.xs[i];
    }
    if (sf.xs[0] != 0) return fail(err, VSYN_ERR_INVALID, "floor %u: xs[0] must be 0 (hpp:449)", f);
    std::vector<uint32_t> order(sf.num_posts);
    for (uint32_t i = 0; i < sf.num_posts; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sf.xs[a] < sf.xs[b]; });
    for (uint32_t s = 0; s < sf.num_posts; ++s) {
      if (s && sf.xs[order[s]] == sf.xs[order[s - 1]]) return fail(err, VSYN_ERR_INVALID, "floor %u: duplicate x %u (render_line needs x0<x1, Utils.hpp:145)", f, sf.xs[order[s]]);
      fc.sorted_idx[s] = (uint8_t)order[s];
      fc.xs_sorted[s] = (uint16_t)sf.xs[order[s]];
    }
    for (uint32_t i = 2; i < sf.num_posts; ++i) {  // Utils.hpp:60-118
      int lo = -1, hi = -1;
      for (uint32_t j = 0; j < i; ++j) {
        if (sf.xs[j] < sf.xs[i] && (lo < 0 || sf.xs[j] > sf.xs[lo])) lo = (int)j;
        if (sf.xs[j] > sf.xs[i] && (hi < 0 || sf.xs[j] < sf.xs[hi])) hi = (int)j;
      }
      if (lo < 0 || hi < 0) return fail(err, VSYN_ERR_INVALID, "floor %u: post %u has no low/high neighbour (xs[1] must be the maximum)", f, i);
      fc.lo[i] = (uint8_t)lo;
      fc.hi[i] = (uint8_t)hi;
      fc.pk[i].lo = (uint16_t)lo;
      fc.pk[i].hi = (uint16_t)hi;
      fc.pk[i].dxi = (uint16_t)(sf.xs[i] - sf.xs[lo]);
      fc.pk[i].adx = (uint16_t)(sf.xs[hi] - sf.xs[lo]);
      fc.pk[i].inv_adx = 1.0f / (float)(sf.xs[hi] - sf.xs[lo]);
      fc.pk[i].idx = i;
    }
    {
      // posts by depth in the neighbour tree (posts 0 and 1 carry their coded values: depth 0), four of one depth to a group
      std::vector<uint32_t> depth(sf.num_posts, 0);
      uint32_t maxd = 0;
      for (uint32_t i = 2; i < sf.num_posts; ++i) {
        depth[i] = 1u + std::max(depth[fc.lo[i]], depth[fc.hi[i]]);
        maxd = std::max(maxd, depth[i]);
      }
      uint32_t ng = 0;
      for (uint32_t d = 1; d <= maxd; ++d) {
        std::vector<uint32_t> at;
        for (uint32_t i = 2; i < sf.num_posts; ++i)
          if (depth[i] == d) at.push_back(i);
        for (size_t k = 0; k < at.size(); k += 4, ++ng)
          for (uint32_t e = 0; e < 4; ++e) fc.sched[ng][e] = fc.pk[at[std::min(k + e, at.size() - 1)]];
      }
      fc.ngroups = ng;  // <= 63 (one group per post at worst)
    }
    // floors of <= 32 posts take the register chain of vsyn_prep.h, which reads the flag bits a coded post touches — itself and its two
    // neighbours — from the 4th dword of pk[] (sched[] above keeps the post's own index there)
    if (sf.num_posts <= 32)
      for (uint32_t i = 2; i < sf.num_posts; ++i) fc.pk[i].idx = (1u << fc.lo[i]) | (1u << fc.hi[i]) | (1u << i);
  }
  H.ys_stride = (maxp + 3u) & ~3u;
  h->prep_lds_bytes = maxp > 32 ? H.ys_stride * PREP_THREADS * (uint32_t)sizeof(uint32_t) : 16u;  // (floors of <= 32 posts stay in registers)

  std::vector<MapConst> maps(su->num_mappings);
  for (uint32_t m = 0; m < su->num_mappings; ++m) {
    const vsyn_mapping& sm = su->mappings[m];
    MapConst& mc = maps[m];
    memset(&mc, 0, sizeof(mc));
    if (sm.num_couplings > 256 || (sm.num_couplings && !sm.couplings) || !sm.channel_floor) return fail(err, VSYN_ERR_INVALID, "mapping %u invalid", m);
    mc.ncoup = sm.num_couplings;
    for (uint32_t k = 0; k < sm.num_couplings; ++k) {
      const vsyn_coupling& c = sm.couplings[k];
      if (c.magnitude == c.angle || c.magnitude >= su->channels || c.angle >= su->channels)
        return fail(err, VSYN_ERR_INVALID, "mapping %u coupling %u invalid (hpp:788-790)", m, k);
      mc.coup[2 * k] = c.magnitude;
      mc.coup[2 * k + 1] = c.angle;
    }
    for (uint32_t c = 0; c < su->channels; ++c) {
      if (sm.channel_floor[c] >= su->num_floors) return fail(err, VSYN_ERR_INVALID, "mapping %u: floor index out of range (hpp:807)", m);
      mc.chfloor[c] = sm.channel_floor[c];
    }
  }
  for (uint32_t k = 0; k < su->num_modes; ++k) {
    if (su->modes[k].mapping >= su->num_mappings) return fail(err, VSYN_ERR_INVALID, "mode %u: mapping out of range (hpp:832)", k);
    H.mode_blockflag[k] = su->modes[k].block_flag ? 1 : 0;
    H.mode_mapping[k] = su->modes[k].mapping;
  }

  // lay the block out
  size_t off = align_up(sizeof(ConstHeader), 256);
  H.off_floor = (uint32_t)off;
  off = align_up(off + sizeof(FloorConst) * floors.size(), 256);
  H.off_map = (uint32_t)off;
  off = align_up(off + sizeof(MapConst) * maps.size(), 256);
  H.off_invdb = (uint32_t)off;
  off = align_up(off + 256 * sizeof(float), 256);
  for (int b = 0; b < 2; ++b) {
    const uint32_t n = H.bs[b];
    H.off_pre[b] = (uint32_t)off;
    off = align_up(off + (n / 4) * sizeof(float2), 256);
    H.off_post[b] = (uint32_t)off;
    off = align_up(off + (n / 4) * sizeof(float2), 256);
    H.off_fft[b] = (uint32_t)off;
    off = align_up(off + (n / 4) * sizeof(float2), 256);
    H.off_win[b] = (uint32_t)off;
    off = align_up(off + 4 * (size_t)n * sizeof(float), 256);
  }
  H.total_bytes = (uint32_t)off;
  h->host_const.assign(off, 0);
  uint8_t* base = h->host_const.data();
  memcpy(base, &H, sizeof(H));
  memcpy(base + H.off_floor, floors.data(), sizeof(FloorConst) * floors.size());
  memcpy(base + H.off_map, maps.data(), sizeof(MapConst) * maps.size());
  memcpy(base + H.off_invdb, k_inverse_db_bits, sizeof(k_inverse_db_bits));
  for (int b = 0; b < 2; ++b) {
    const uint32_t n = H.bs[b], M = n / 2, N4 = n / 4;
    float2* pre = (float2*)(base + H.off_pre[b]);
    float2* post = (float2*)(base + H.off_post[b]);
    float2* tw = (float2*)(base + H.off_fft[b]);
    for (uint32_t k = 0; k < N4; ++k) {  // double precision, stored as f32 (as mdct_init does)
      const double a = -M_PI * (4.0 * k + 1.0) / (4.0 * M);
      pre[k] = make_float2((float)cos(a), (float)sin(a));
      const double p = -M_PI * (double)k / (double)M;
      post[k] = make_float2((float)cos(p), (float)sin(p));
      const double t = -2.0 * M_PI * (double)k / (double)N4;
      tw[k] = make_float2((float)cos(t), (float)sin(t));
    }
    float* win = (float*)(base + H.off_win[b]);
    for (int w = 0; w < 4; ++w) host_window(H.bs[0], H.bs[1], b, w & 1, (w >> 1) & 1, win + (size_t)w * n);
  }
  return VSYN_OK;
}

// ------------------------------------------------------------------------------------------------
// API
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* vsyn_version(void) { return "parseoggvorbis_amd vsyn 0.1 (gfx950)"; }
int vsyn_abi_version(void) { return VSYN_ABI_VERSION; }

int vsyn_create(const vsyn_setup* setup, int device, uint32_t max_streams, vsyn_handle** out, const char** err) {
  if (!out) return fail(err, VSYN_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(err, VSYN_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(err, VSYN_ERR_NO_DEVICE, "device %d not in 0..%d", device, ndev - 1);
  vsyn_handle* h = new vsyn_handle();
  h->device = device;
  int rc = build_const(setup, max_streams, h, err);
  if (rc) {
    delete h;
    return rc;
  }
  auto cleanup = [&](int code) {
    vsyn_destroy(h);
    return code;
  };
  hipError_t e;
#define HC(call)                                                                                                       \
  if ((e = (call)) != hipSuccess) {                                                                                    \
    fail(err, VSYN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e));                                             \
    return cleanup(VSYN_ERR_HIP);                                                                                      \
  }
  HC(hipSetDevice(device));
  hipDeviceProp_t prop;
  HC(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fail(err, VSYN_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return cleanup(VSYN_ERR_NO_DEVICE);
  }
  const ConstHeader& H = h->H;
  HC(hipMalloc((void**)&h->d_const, h->host_const.size()));
  HC(hipMemcpy(h->d_const, h->host_const.data(), h->host_const.size(), hipMemcpyHostToDevice));
  HC(hipMalloc((void**)&h->d_state, sizeof(StreamState) * 2 * max_streams));  // two tagged records per slot (vsyn_device.h)
  HC(hipMemset(h->d_state, 0, sizeof(StreamState) * 2 * max_streams));
  const size_t carry_floats = 2ull * max_streams * H.channels * (H.bs[1] / 2);
  HC(hipMalloc((void**)&h->d_carry, carry_floats * sizeof(float)));
  HC(hipMemset(h->d_carry, 0, carry_floats * sizeof(float)));
  HC(hipMalloc((void**)&h->d_status, sizeof(DevStatus)));
  DevStatus init = {0u, 0xFFFFFFFFu};
  HC(hipMemcpy(h->d_status, &init, sizeof(init), hipMemcpyHostToDevice));
  h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HC(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  HC(hipStreamCreateWithFlags(&h->pre, hipStreamNonBlocking));
  HC(hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
  // ordering events between this handle's own streams: device-scope release is enough (the one host read,
  // vsyn_sync_status, synchronises its stream); the system-scope fence of a default event costs ~3 us per submit
  const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
  HC(hipEventCreateWithFlags(&h->ev_join, evf));
  HC(hipEventCreateWithFlags(&h->ev_self, evf));
  for (uint32_t k = 0; k < H.num_modes && k < 64; ++k)
    if (H.mode_blockflag[k]) h->long_modes |= 1ull << k;
  for (uint32_t b = 0; b < vsyn_handle::WS_RING; ++b) HC(hipEventCreateWithFlags(&h->ev_pre_done[b], evf));
  for (uint32_t b = 0; b < vsyn_handle::EV_RING; ++b) HC(hipEventCreateWithFlags(&h->ev_ring[b], evf));
  // (the attribute is per kernel, not per handle: only ever raised, to the largest any setup can need — 68 posts x 256 threads x 4 B)
  HC(hipFuncSetAttribute((const void*)vsyn_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * PREP_THREADS * 4));
  HC(h->ws_count.ensure(vsyn_handle::CNT_RING));
  HC(hipMemset(h->ws_count.p, 0, sizeof(uint32_t) * vsyn_handle::CNT_RING));
  h->fused_mask = h->tuned_mask = fused_ok_mask(h->H, h->host_const.data());
  if ((e = fused_tables_create(h->H, h->host_const.data(), &h->fused)) != hipSuccess) {
    fail(err, VSYN_ERR_HIP, "fused table upload failed: %s", hipGetErrorString(e));
    return cleanup(VSYN_ERR_HIP);
  }
  // The size-generic kernel takes the mixed-block runs of every setup it covers: all of them where the 256/2048 kernel has no
  // mixed path, and (VSYN_U_MIXED=1) in its place where it has one.
  if (u_supported(h->H, h->host_const.data()) && !getenv("VSYN_NO_U")) {
    const bool want = !(h->fused_mask & 2u) || (getenv("VSYN_U_MIXED") && atoi(getenv("VSYN_U_MIXED")));
    if (want) {
      e = u_tables_create(h->H, h->host_const.data(), &h->utab);
      if (e == hipSuccess) {
        h->u_mixed = true;
        h->fused_mask |= 2u;
      } else if (e == hipErrorInvalidValue) {
        // the setup does not fit (its channel waves plus the tables of an 8192-sample block exceed one CU's LDS): staged kernels
        u_tables_destroy(&h->utab);
        (void)hipGetLastError();
      } else {
        fail(err, VSYN_ERR_HIP, "generic fused table upload failed: %s", hipGetErrorString(e));
        return cleanup(VSYN_ERR_HIP);
      }
    }
  }
  h->fused_ok = h->fused_mask != 0;
#undef HC
  *out = h;
  return VSYN_OK;
}

void vsyn_destroy(vsyn_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
#ifdef VQ_STAMPS
  {  // diagnostic build: the residue VQ kernel's cycles per phase and packet (last launch), averaged over its waves
    static unsigned long long host[8192][VQ_NSTAMPS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vq_stamps), sizeof(host)) == hipSuccess) {
      double sum[VQ_NSTAMPS] = {0};
      unsigned long long pk = 0, waves = 0;
      for (int u = 0; u < 8192; ++u) {
        if (!host[u][VQ_NSTAMPS - 1]) continue;
        ++waves;
        pk += host[u][VQ_NSTAMPS - 1];
        for (int i = 0; i + 1 < VQ_NSTAMPS; ++i) sum[i] += (double)host[u][i];
      }
      if (pk) {
        fprintf(stderr, "[vq stamps] %llu waves, %.1f packets each; s_memtime ticks per packet:", waves, (double)pk / waves);
        for (int i = 0; i + 1 < VQ_NSTAMPS; ++i) fprintf(stderr, " %d:%.0f", i, sum[i] / pk);
        fprintf(stderr, "\n");
      }
    }
  }
#endif
#ifdef PREP_STAMPS
  {  // diagnostic build: cycles per phase of vsyn_prep_kernel (last launch), averaged over the waves of each role, and when the waves
     // of each role started / ended relative to the first wave of the launch (100 MHz clock)
    static unsigned long long host[8192][PREP_NSTAMPS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prep_stamps), sizeof(host)) == hipSuccess) {
      unsigned long long t_first = ~0ull;
      for (int u = 0; u < 8192; ++u)
        if (host[u][7] && host[u][5] < t_first) t_first = host[u][5];
      for (unsigned role = 0; role < 2; ++role) {
        double sum[5] = {0}, st = 0, en = 0, en_max = 0, st_max = 0;
        unsigned long long waves = 0;
        for (int u = 0; u < 8192; ++u) {
          if (host[u][7] != 1ull + role) continue;
          ++waves;
          for (int i = 0; i < 5; ++i) sum[i] += (double)host[u][i];
          const double a = (double)(host[u][5] - t_first) / 100.0, b = (double)(host[u][6] - t_first) / 100.0;
          st += a;
          en += b;
          if (a > st_max) st_max = a;
          if (b > en_max) en_max = b;
        }
        if (!waves) continue;
        static const char* nm[5] = {"header + stream state", "scan in front of the chunk", "descriptors, scan, PktInfo", "floor role: descriptors, floor ids", "floor role: chains"};
        fprintf(stderr, "prep stamps, %s role: %llu waves; start %.2f us (latest %.2f), end %.2f us (latest %.2f) after the launch's first wave\n",
                role ? "floor" : "layout", waves, st / waves, st_max, en / waves, en_max);
        for (int i = 0; i < 5; ++i)
          if (sum[i] > 0) fprintf(stderr, "  %-36s %8.0f cycles\n", nm[i], sum[i] / waves);
      }
    }
  }
#endif
#ifdef VSYN_STAMPS
  {  // diagnostic build: per-phase cycles of the LAST launch's steady runs, averaged over the waves that ran one
    static unsigned long long host[8192][VSYN_NSTAMPS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vsyn_stamps), sizeof(host)) == hipSuccess) {
      double sum[VSYN_NSTAMPS] = {0};
      unsigned long long waves = 0, pk = 0;
      for (int u = 0; u < 8192; ++u) {
        if (!host[u][VSYN_NSTAMPS - 1]) continue;
        ++waves;
        pk += host[u][VSYN_NSTAMPS - 1];
        for (int i = 0; i + 1 < VSYN_NSTAMPS; ++i) sum[i] += (double)host[u][i];
      }
      if (pk) {
        static const char* nm[VSYN_NSTAMPS - 1] = {"loop", "residue+handoff+couple", "loads+floor setup", "floor product", "mirror+pre-rot", "partner wait 2",
                                                   "fft512", "post+window+overlap", "stores / short pass: stores", "short: descriptors", "short: rows", "short: couple+floor", "short: fft+window", "-", "-"};
        double tot = 0;
        for (int i = 0; i + 1 < VSYN_NSTAMPS; ++i) tot += sum[i];
        fprintf(stderr, "vsyn stamps: %llu waves, %llu wave-packets, %.0f cycles per wave-packet\n", waves, pk, tot / pk);
        for (int i = 0; i + 1 < VSYN_NSTAMPS; ++i) fprintf(stderr, "  %-26s %8.0f cycles  %5.1f %%\n", nm[i], sum[i] / pk, 100.0 * sum[i] / tot);
      }
    }
  }
#endif
  fused_tables_destroy(&h->fused);
  u_tables_destroy(&h->utab);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->pre) (void)hipStreamDestroy(h->pre);
  if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_self) (void)hipEventDestroy(h->ev_self);
  for (uint32_t b = 0; b < vsyn_handle::WS_RING; ++b)
    if (h->ev_pre_done[b]) (void)hipEventDestroy(h->ev_pre_done[b]);
  for (uint32_t b = 0; b < vsyn_handle::EV_RING; ++b)
    if (h->ev_ring[b]) (void)hipEventDestroy(h->ev_ring[b]);
  if (h->d_const) (void)hipFree(h->d_const);
  if (h->d_vq) (void)hipFree(h->d_vq);
  h->st_curve.release(); h->st_vqpk.release(); h->st_cls.release(); h->st_ent.release();
  if (h->d_state) (void)hipFree(h->d_state);
  if (h->d_carry) (void)hipFree(h->d_carry);
  if (h->d_status) (void)hipFree(h->d_status);
  h->ws_count.release();
  h->ws_chunks.release();
  for (uint32_t b = 0; b < vsyn_handle::WS_RING; ++b) {
    h->ws_list[b].release(); h->ws_info[b].release(); h->ws_seg[b].release(); h->ws_segmap[b].release(); h->ws_fy[b].release(); h->ws_runcls[b].release();
  }
  h->ws_env.release(); h->ws_blk.release();
  h->st_pk.release(); h->st_seg.release(); h->st_ys.release(); h->st_fy.release(); h->st_res.release(); h->st_pcm.release();
  h->st_env.release(); h->st_blk.release(); h->st_emit.release();
  h->st_sum.release(); h->st_conv.release(); h->st_frames.release();
  for (auto& ev : h->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  delete h;
}

uint32_t vsyn_ys_stride(const vsyn_handle* h) { return h ? h->H.ys_stride : 0; }
uint32_t vsyn_channels(const vsyn_handle* h) { return h ? h->H.channels : 0; }
uint32_t vsyn_fused_paths(const vsyn_handle* h) { return h ? (h->fused_mask | (h->vq_tables_in_lds ? 0x100u : 0u)) : 0; }
size_t vsyn_const_block_bytes(const vsyn_handle* h) { return h ? h->host_const.size() : 0; }

int vsyn_profile_enable(vsyn_handle* h, int on) {
  if (!h) return VSYN_ERR_INVALID;
  h->profile = on != 0;
  if (on >= 1 && on <= 3) h->profile_which = on;
  return VSYN_OK;
}

int vsyn_profile_read(vsyn_handle* h, double* mean_ms, uint32_t* launches, const char** kernel_name) {
  if (!h) return VSYN_ERR_INVALID;
  std::lock_guard<std::mutex> lk(h->mu);
  (void)hipSetDevice(h->device);
  double total = 0;
  for (size_t i = 0; i < h->events_used; ++i) {
    (void)hipEventSynchronize(h->events[i].second);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second);
    total += ms;
  }
  if (mean_ms) *mean_ms = h->events_used ? total / (double)h->events_used : 0.0;
  if (launches) *launches = (uint32_t)h->events_used;
  if (kernel_name) *kernel_name = h->profile_kernel;
  h->events_used = 0;
  return VSYN_OK;
}

static hipError_t profile_begin(vsyn_handle* h, hipStream_t s, const char* name) {
  if (!h->profile) return hipSuccess;
  if (h->events_used == h->events.size()) {
    hipEvent_t a, b;
    const unsigned pf = hipEventDisableSystemFence;  // timestamps only: no cache flush around the timed kernel
    hipError_t e = hipEventCreateWithFlags(&a, pf);
    if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&b, pf);
    if (e != hipSuccess) return e;
    h->events.emplace_back(a, b);
  }
  h->profile_kernel = name;
  return hipEventRecord(h->events[h->events_used].first, s);
}
static hipError_t profile_end(vsyn_handle* h, hipStream_t s) {
  if (!h->profile) return hipSuccess;
  hipError_t e = hipEventRecord(h->events[h->events_used].second, s);
  ++h->events_used;
  return e;
}

int vsyn_reset_streams(vsyn_handle* h, void* hip_stream, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemsetAsync(h->d_state, 0, sizeof(StreamState) * 2 * h->H.max_streams, (hipStream_t)hip_stream));
  return VSYN_OK;
}

int vsyn_sync_status(vsyn_handle* h, void* hip_stream, vsyn_status* status, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
  DevStatus ds;
  HIPCHK(hipMemcpy(&ds, h->d_status, sizeof(ds), hipMemcpyDeviceToHost));
  if (ds.flags) {
    DevStatus init = {0u, 0xFFFFFFFFu};
    HIPCHK(hipMemcpy(h->d_status, &init, sizeof(init), hipMemcpyHostToDevice));
  }
  if (status) {
    status->flags = ds.flags;
    status->first_bad_packet = ds.first_bad_packet;
  }
  if (ds.flags) return fail(err, VSYN_ERR_STREAM, "device flagged the batch: flags=0x%x first_bad_packet=%u", ds.flags, ds.first_bad_packet);
  return VSYN_OK;
}

// d_vq == nullptr: d_residue is the input ("after_residue"). Otherwise d_residue is scratch that the residue VQ kernel
// fills from the entry numbers (after the layout kernel, which provides each packet's offset, beside the floor unwrap).
static int submit_device_impl(vsyn_handle* h, uint32_t P, const vsyn_packet* d_packets, uint32_t S, const vsyn_segment* d_segments,
                              uint32_t max_seg_packets, const uint16_t* d_ys, const vsyn_vq_batch* d_vq, float* d_residue, float* d_pcm,
                              uint64_t plane_stride, uint32_t* d_emit_len, const vsyn_taps* taps, uint32_t flags, void* hip_stream,
                              const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (P == 0 || S == 0) return VSYN_OK;
  if (!d_packets || !d_segments || !d_ys || !d_residue || !d_pcm) return fail(err, VSYN_ERR_INVALID, "NULL batch pointer");
  if (d_vq) {
    if (!h->d_vq) return fail(err, VSYN_ERR_INVALID, "vsyn_attach_vq has not been called on this handle");
    if (!d_vq->packets || (d_vq->num_cls && !d_vq->cls) || (d_vq->num_entries && !d_vq->entries)) return fail(err, VSYN_ERR_INVALID, "NULL vq batch pointer");
  }
  if (max_seg_packets == 0 || max_seg_packets > P) max_seg_packets = P;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const ConstHeader& H = h->H;
  const uint32_t C = H.channels;
  // The intermediate-signal taps (after_envelope, pcm_after_mdct) exist only in the staged kernels. The feature taps — the rendered
  // floor curve and the unwrapped posts — do not force them: the posts come from the unwrap kernel either way and
  // the curve from the tap variant of the fused kernel.
  const bool want_taps = taps && (taps->after_envelope || taps->pcm_after_mdct);
  const bool use_u = h->u_mixed;  // (both fused kernels have a floor-curve tap variant)
  const uint32_t fmask = use_u ? h->fused_mask : h->tuned_mask;
  const bool force_staged = want_taps || (flags & VSYN_SUBMIT_STAGED) || !fmask;
  const uint32_t R = force_staged ? std::min<uint32_t>(max_seg_packets, 1024u)
                                  : fused_pick_run_len((fmask & 1u) || !use_u ? h->fused.waves_per_cu : (int)h->utab.waves_per_cu, S, C,
                                                       max_seg_packets, h->num_cus);

  // Workspace of this submit: slot i % WS_RING of a ring, so that the preparation of later submits can run ahead of the synthesis
  // kernels (see vsyn_handle).
  const uint64_t isub = h->nsub++;
  const uint32_t wb = (uint32_t)(isub % vsyn_handle::WS_RING), wb_prev = (uint32_t)((isub + vsyn_handle::WS_RING - 1u) % vsyn_handle::WS_RING);
  uint32_t* cnt = h->ws_count.p + (isub % vsyn_handle::CNT_RING);
  uint32_t* cnt_next = h->ws_count.p + ((isub + 1u) % vsyn_handle::CNT_RING);
  ++h->submit_count;
  // the submit's number tags the stream-state records and the look-back records of the chunked scan; 0 means "never written"
  const uint32_t epoch = (h->submit_count & 0x3FFFFFFFu) ? h->submit_count : ++h->submit_count;
  HIPCHK(h->ws_info[wb].ensure((size_t)P + 8));  // (slack: the generic kernel fetches descriptors eight at a time)
  HIPCHK(h->ws_seg[wb].ensure(S));
  HIPCHK(h->ws_segmap[wb].ensure(P));
  HIPCHK(h->ws_list[wb].ensure(2 * (size_t)P + 64));
  uint16_t* fy = taps && taps->floor_final ? taps->floor_final : nullptr;
  if (!fy) {
    HIPCHK(h->ws_fy[wb].ensure((size_t)P * C * H.ys_stride));
    fy = h->ws_fy[wb].p;
  }
  const uint32_t runs_per_seg = (max_seg_packets + R - 1) / R;
  HIPCHK(h->ws_runcls[wb].ensure((size_t)S * runs_per_seg + 16));
  PktInfo* info = h->ws_info[wb].p;
  SegInfo* sinfo = h->ws_seg[wb].p;
  uint32_t* segmap = h->ws_segmap[wb].p;
  uint32_t* list = h->ws_list[wb].p;

  // Preparation of the batch (layout scan, floor-1 step 1): two ways.
  //   (a) vsyn_prep_kernel (vsyn_prep.h): ONE dependency-free kernel — layout workgroups and floor workgroups side by side, ~17 us — in
  //       front of the synthesis kernel on the caller's stream; no second queue, no events. The default whenever every run is taken by
  //       a fused kernel and no segment is longer than PREP_MAX_SEG_PACKETS.
  //   (b) The chained layout and unwrap kernels (vsyn_staged.h): for staged work lists, intermediate-signal taps, the residue VQ stage
  //       (its kernel needs the packets' offsets first), very long segments, VSYN_SUBMIT_PRE_KERNELS. With VSYN_SUBMIT_INPUTS_READY
  //       they run on the internal stream `pre`, beside the previous submit's synthesis kernel: their ~35 us of dependent latency are
  //       hidden, at the price of one event record and one cross-queue wait per submit (~15 us between two synthesis kernels) and ~6 us
  //       of interference with an exact-fit synthesis grid.
  //   Until late in round 3 (b)-hidden was the default for VSYN_SUBMIT_INPUTS_READY: against a 20 us preparation kernel it tied on
  //   config 3 (0.256-0.259 ms per step either way) and won config 4 by 3 %. With the preparation's two halves running side by side
  //   (a) wins everywhere measured: config 3 0.2540 vs 0.2572 ms, config 4 0.0796 vs 0.0823, 128/1024 0.187 vs 0.195 (same box,
  //   profiles/r03_experiments/batch_preparation_ab.txt). VSYN_PREP_SERIAL=0 brings (b)-hidden back for such submits, for A/B runs; the
  //   preparation kernel on the internal stream (VSYN_PREP_OVERLAP=1) lost to both (0.266 ms).
  static const bool env_no_prep_kernel = getenv("VSYN_NO_PREP_KERNEL") && atoi(getenv("VSYN_NO_PREP_KERNEL"));
  static const int env_prep_serial_mode = getenv("VSYN_PREP_SERIAL") ? (atoi(getenv("VSYN_PREP_SERIAL")) ? 1 : 0) : -1;
  const bool env_prep_serial = env_prep_serial_mode != 0;
  static const bool env_prep_overlap = getenv("VSYN_PREP_OVERLAP") && atoi(getenv("VSYN_PREP_OVERLAP"));
  const bool prep_ok = !force_staged && (fmask & 2u) && !d_vq && max_seg_packets <= PREP_MAX_SEG_PACKETS && !(flags & VSYN_SUBMIT_PRE_KERNELS) && !env_no_prep_kernel;
  const bool overlap_pre = (flags & VSYN_SUBMIT_INPUTS_READY) && !force_staged && !(env_prep_serial && prep_ok);
  const bool prep_kernel = prep_ok && (!overlap_pre || env_prep_overlap);
  hipStream_t ps = overlap_pre ? h->pre : s;
  if (ps != s) {
    // the slot's previous user, submit isub - WS_RING, has to be done: the first ring event recorded at or behind it says so
    if (isub >= vsyn_handle::WS_RING) {
      const uint64_t need = isub - vsyn_handle::WS_RING;
      const uint64_t jstar = need + ((vsyn_handle::EV_EVERY - 1u) - need % vsyn_handle::EV_EVERY);  // first j >= need with j % EV_EVERY == EV_EVERY - 1
      const uint32_t slot = (uint32_t)((jstar / vsyn_handle::EV_EVERY) % vsyn_handle::EV_RING);
      if (h->ev_ring_submit[slot] == jstar) {
        HIPCHK(hipStreamWaitEvent(ps, h->ev_ring[slot], 0));
      } else {  // that submit's preparation ran on the caller's stream (no record): order behind everything queued there so far
        HIPCHK(hipEventRecord(h->ev_self, s));
        HIPCHK(hipStreamWaitEvent(ps, h->ev_self, 0));
      }
    }
    if (h->last_prep_on_main && isub > 0) {
      // the previous submit left the stream state from a kernel on the caller's stream
      HIPCHK(hipEventRecord(h->ev_self, s));
      HIPCHK(hipStreamWaitEvent(ps, h->ev_self, 0));
    }
  }
  // Consecutive preparations chain through the stream state (abs position, carry parity) and the list-counter ring: when this one
  // runs on another stream than the previous one did (flags differ between submits), that order has to be stated.
  if (h->pre_done_valid[wb_prev] && h->last_pre_stream != ps) HIPCHK(hipStreamWaitEvent(ps, h->ev_pre_done[wb_prev], 0));
  const bool staged_may_work_pre = force_staged || !(fmask & 2u);
  if (!prep_kernel) {
    // the list-counter ring is cleared one submit ahead by the layout kernel; submits that ran none in between break that chain
    if (!h->last_ran_layout) HIPCHK(hipMemsetAsync(h->ws_count.p, 0, sizeof(uint32_t) * vsyn_handle::CNT_RING, ps));
    {
      // segments longer than LAYOUT_CHUNK_PACKETS are scanned in chunks (a multiple of R each) chained by a look-back; the usual batch
      // has one chunk per segment
      const uint32_t chunk_packets = max_seg_packets <= LAYOUT_CHUNK_PACKETS ? runs_per_seg * R : (LAYOUT_CHUNK_PACKETS + R - 1u) / R * R;
      const uint32_t chunks_per_seg = (max_seg_packets + chunk_packets - 1u) / chunk_packets;
      const uint32_t lt = std::min(chunk_packets, max_seg_packets) <= LAYOUT_SHORT_PACKETS ? LAYOUT_THREADS_SHORT : LAYOUT_THREADS;
      if (chunks_per_seg > 1) {
        const size_t need = (size_t)S * chunks_per_seg;
        if (need > h->ws_chunks.cap) {
          HIPCHK(h->ws_chunks.ensure(need));
          HIPCHK(hipMemsetAsync(h->ws_chunks.p, 0, h->ws_chunks.cap * sizeof(LayoutChunk), ps));  // flags are epoch-tagged: cleared once
        }
      }
      if ((uint64_t)S * chunks_per_seg > 0x7FFFFFFFull) return fail(err, VSYN_ERR_INVALID, "too many layout chunks");
      vsyn_layout_kernel<<<S * chunks_per_seg, lt, layout_lds_bytes(lt, chunk_packets), ps>>>(
          h->d_const, P, d_packets, S, d_segments, plane_stride, info, sinfo, h->d_state, d_emit_len, h->d_status, R, force_staged ? 0u : fmask, list, cnt,
          cnt_next, segmap, h->ws_runcls[wb].p, runs_per_seg, chunk_packets, chunks_per_seg, h->ws_chunks.p, epoch);
    }
    {
      const uint32_t rows = P * C;
      vsyn_floor_unwrap_kernel<<<(rows + UNWRAP_THREADS - 1) / UNWRAP_THREADS, UNWRAP_THREADS, 0, ps>>>(h->d_const, P, nullptr, nullptr, info,
                                                                                                         d_ys, fy, h->d_status);
    }
    if (d_vq) {
      if (h->profile_which == 3) HIPCHK(profile_begin(h, ps, "vsyn_residue_vq_kernel"));
      if (h->vq_tables_in_lds)
        vsyn_residue_vq_kernel<true><<<std::min<uint32_t>((P + h->vq_waves - 1u) / h->vq_waves, h->vq_grid), VQ_THREADS * h->vq_waves, h->vq_lds_bytes, ps>>>(
            h->d_const, h->d_vq, P, info, d_vq->packets, d_vq->cls, d_vq->num_cls, d_vq->entries, d_vq->num_entries, d_residue, h->d_status);
      else
        vsyn_residue_vq_kernel<false><<<std::min<uint32_t>(P, h->vq_grid), VQ_THREADS, h->vq_lds_bytes, ps>>>(
            h->d_const, h->d_vq, P, info, d_vq->packets, d_vq->cls, d_vq->num_cls, d_vq->entries, d_vq->num_entries, d_residue, h->d_status);
    }
    if (d_vq && h->profile_which == 3) HIPCHK(profile_end(h, ps));
  } else {
    PrepCtx pc;
    pc.cb = h->d_const;
    pc.packets = d_packets;
    pc.segs = d_segments;
    pc.ys = d_ys;
    pc.fy = fy;
    pc.info = info;
    pc.sinfo = sinfo;
    pc.state = h->d_state;
    pc.status = h->d_status;
    pc.emit_len = d_emit_len;
    pc.run_cls = h->ws_runcls[wb].p;
    pc.plane_stride = plane_stride;
    pc.long_modes = h->long_modes;
    pc.S = S;
    pc.R = R;
    pc.runs_per_seg = runs_per_seg;
    pc.fused_ok = fmask;
    pc.P = P;
    pc.epoch = epoch;
    // a workgroup takes whole runs, as many as give about one (packet, channel) row per thread; a batch of short segments (thousands of
    // streams with a few packets each) gets smaller workgroups — whole waves — instead of 256 threads with a handful of rows
    const uint32_t nt = std::min<uint32_t>(PREP_THREADS, std::max<uint32_t>(64u, ((max_seg_packets * C + 63u) / 64u) * 64u));
    const uint32_t ppp = std::max<uint32_t>(1u, nt / C);
    pc.chunk_runs = std::max<uint32_t>(1u, ppp / R);
    pc.chunks_per_seg = (runs_per_seg + pc.chunk_runs - 1u) / pc.chunk_runs;
    // a layout workgroup and a floor workgroup per (segment, chunk), dealt in alternating groups of eight (vsyn_prep.h)
    const uint64_t wgs = (((uint64_t)S * pc.chunks_per_seg + 7u) / 8u) * 16u;
    if (wgs > 0x7FFFFFF0ull) return fail(err, VSYN_ERR_INVALID, "too many runs");
    vsyn_prep_kernel<<<(uint32_t)wgs, nt, h->prep_lds_bytes, ps>>>(pc);
  }
  h->last_ran_layout = !prep_kernel;
  h->last_prep_on_main = ps == s;
  if (ps != s) {
    HIPCHK(hipEventRecord(h->ev_pre_done[wb], ps));
    h->pre_done_valid[wb] = true;
    HIPCHK(hipStreamWaitEvent(s, h->ev_pre_done[wb], 0));
  } else {
    h->pre_done_valid[wb] = false;  // (ordered by the caller's stream itself)
  }
  h->last_pre_stream = ps;
  (void)staged_may_work_pre;

  // staged kernels walk the work list the layout kernel built: everything when forced, otherwise only the runs the
  // fused kernel declines (short / mixed blocks, carry-in). In fused mode they run on a forked side stream beside the
  // fused kernel (disjoint outputs) and exit at once when the list is empty.
  // With the mixed-block kernel available every run is taken by one of the two fused kernels (run_class() never answers
  // 0 then; packets with an invalid mode are skipped by both paths): the staged kernels are not launched at all.
  const bool staged_may_work = force_staged || !(fmask & 2u);
  hipStream_t ss = force_staged ? s : h->side;
  if (staged_may_work) {
    if (!force_staged) {  // the side stream starts behind the preparation
      if (ps != s) {
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_pre_done[wb], 0));
      } else {
        HIPCHK(hipEventRecord(h->ev_self, s));
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_self, 0));
      }
    }
    // residue floats upper bound (the descriptors are device resident, so the exact sum is not known here)
    const size_t bound = (size_t)P * C * (H.bs[1] / 2);
    float* env = taps && taps->after_envelope ? taps->after_envelope : nullptr;
    float* blk = taps && taps->pcm_after_mdct ? taps->pcm_after_mdct : nullptr;
    if (!env) {
      HIPCHK(h->ws_env.ensure(bound));
      env = h->ws_env.p;
    }
    if (!blk) {
      HIPCHK(h->ws_blk.ensure(2 * bound));
      blk = h->ws_blk.p;
    }
    const uint32_t grid = force_staged ? std::min<uint32_t>(P * C, 256u * 32u) : 512u;
    vsyn_spectrum_kernel<<<std::min<uint32_t>(grid, P), 256, 0, ss>>>(h->d_const, list, cnt, info, d_residue, fy, env,
                                                                      taps ? taps->floor_curve : nullptr, h->d_status);
    if (force_staged) HIPCHK(profile_begin(h, s, "vsyn_imdct_staged_kernel"));
    vsyn_imdct_staged_kernel<<<grid, 256, (size_t)H.bs[1] * 4, ss>>>(h->d_const, list, cnt, info, env, blk);
    if (force_staged) HIPCHK(profile_end(h, s));
    vsyn_overlap_kernel<<<grid, 256, 0, ss>>>(h->d_const, list, cnt, info, d_segments, sinfo, segmap, blk, d_pcm, plane_stride, h->d_carry);
  }
  if (!force_staged) {
    FusedArgs a;
    a.cb = h->d_const;
    a.binseg = h->fused.d_binseg;
    a.lds_image = h->fused.d_lds;
    a.packets = d_packets;
    a.segs = d_segments;
    a.info = info;
    a.sinfo = sinfo;
    a.run_cls = h->ws_runcls[wb].p;
    a.runs_per_seg = runs_per_seg;
    a.residue = d_residue;
    a.curve = taps ? taps->floor_curve : nullptr;
    a.fy = fy;
    a.pcm = d_pcm;
    a.carry = h->d_carry;
    a.status = h->d_status;
    a.plane_stride = plane_stride;
    a.S = S;
    a.R = R;
    a.fused_ok = use_u ? (fmask & 1u) : fmask;
    a.coupling_mode = (uint32_t)h->fused.coupling_mode;
    if (staged_may_work) HIPCHK(hipEventRecord(h->ev_join, h->side));
    // one launch covers the long-run and the mixed-block runs of the 256/2048 kernel (each wave takes the path of its run's class);
    // with the size-generic kernel in charge of the class-2 runs that is a second launch behind it (disjoint outputs)
    const bool time_u = use_u && (h->profile_which == 2 || !(fmask & 1u));
    hipError_t e = hipSuccess;
    if (a.fused_ok) {
      if (!time_u && (h->profile_which == 1 || h->profile_which == 2)) HIPCHK(profile_begin(h, s, a.curve ? "vsyn_fused_tap_kernel" : fused_kernel_name(H)));
      e = fused_launch(H, h->fused, a, max_seg_packets, s);
      if (e != hipSuccess) return fail(err, VSYN_ERR_HIP, "fused launch failed: %s", hipGetErrorString(e));
      if (!time_u && (h->profile_which == 1 || h->profile_which == 2)) HIPCHK(profile_end(h, s));
    }
    if (use_u) {
      FusedArgs au = a;
      au.fused_ok = fmask;
      if (time_u && (h->profile_which == 1 || h->profile_which == 2)) HIPCHK(profile_begin(h, s, au.curve ? "vsyn_fused_u_tap_kernel" : "vsyn_fused_u_kernel"));
      e = u_launch(H, h->utab, au, s);
      if (e != hipSuccess) return fail(err, VSYN_ERR_HIP, "generic fused launch failed: %s", hipGetErrorString(e));
      if (time_u && (h->profile_which == 1 || h->profile_which == 2)) HIPCHK(profile_end(h, s));
    }
    if (staged_may_work) HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
  }
  if (ps != s && isub % vsyn_handle::EV_EVERY == vsyn_handle::EV_EVERY - 1u) {  // (a record costs ~4.6 us between two synthesis kernels)
    const uint32_t slot = (uint32_t)((isub / vsyn_handle::EV_EVERY) % vsyn_handle::EV_RING);
    HIPCHK(hipEventRecord(h->ev_ring[slot], s));
    h->ev_ring_submit[slot] = isub;
  }
  h->last_S = S;
  h->last_wb = wb;
  HIPCHK(hipGetLastError());
  return VSYN_OK;
}

int vsyn_submit_device(vsyn_handle* h, uint32_t P, const vsyn_packet* d_packets, uint32_t S, const vsyn_segment* d_segments,
                       uint32_t max_seg_packets, const uint16_t* d_ys, const float* d_residue, float* d_pcm,
                       uint64_t plane_stride, uint32_t* d_emit_len, const vsyn_taps* taps, uint32_t flags, void* hip_stream,
                       const char** err) {
  return submit_device_impl(h, P, d_packets, S, d_segments, max_seg_packets, d_ys, nullptr, const_cast<float*>(d_residue), d_pcm, plane_stride,
                            d_emit_len, taps, flags, hip_stream, err);
}

int vsyn_submit_device_vq(vsyn_handle* h, uint32_t P, const vsyn_packet* d_packets, uint32_t S, const vsyn_segment* d_segments,
                          uint32_t max_seg_packets, const uint16_t* d_ys, const vsyn_vq_batch* d_vq, float* d_residue, float* d_pcm,
                          uint64_t plane_stride, uint32_t* d_emit_len, const vsyn_taps* taps, uint32_t flags, void* hip_stream,
                          const char** err) {
  if (!d_vq) return fail(err, VSYN_ERR_INVALID, "vq batch is NULL");
  return submit_device_impl(h, P, d_packets, S, d_segments, max_seg_packets, d_ys, d_vq, d_residue, d_pcm, plane_stride, d_emit_len, taps, flags,
                            hip_stream, err);
}

int vsyn_attach_vq(vsyn_handle* h, const vsyn_vq_setup* vq, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  std::vector<uint8_t> block;
  const std::string why = vq_build_block(vq, h->H, block);
  if (!why.empty()) return fail(err, VSYN_ERR_INVALID, "%s", why.c_str());
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());  // a previously attached block may still be in use
  if (h->d_vq) (void)hipFree(h->d_vq);
  h->d_vq = nullptr;
  HIPCHK(hipMalloc((void**)&h->d_vq, block.size()));
  HIPCHK(hipMemcpy(h->d_vq, block.data(), block.size(), hipMemcpyHostToDevice));
  {
    const VqHeader* vh = (const VqHeader*)block.data();
    uint32_t lds_off[7];
    const uint32_t wave_bytes = vq_lds_layout(vh->max_slots, vh->max_classes, lds_off);
    h->vq_tables_in_lds = false;
    h->vq_waves = 1;
    if (vh->img_floats && !(getenv("VSYN_VQ_NO_LDS_TABLES") && atoi(getenv("VSYN_VQ_NO_LDS_TABLES")))) {
      // value tables in LDS, one copy per workgroup: k workgroups of w waves per CU, the pair that keeps most waves resident
      const uint32_t tab_bytes = vq_align16(vh->img_floats * 4u);
      hipFuncAttributes fa;
      HIPCHK(hipFuncGetAttributes(&fa, (const void*)vsyn_residue_vq_kernel<true>));
      const uint32_t regs = ((uint32_t)std::max(fa.numRegs, 1) + 7u) / 8u * 8u;  // allocation granule 8, 512 per SIMD lane
      const uint32_t cu_waves = 4u * std::min<uint32_t>(8u, 512u / regs);
      uint32_t best_k = 0, best_w = 0;
      for (uint32_t k = 1; k <= 4; ++k) {
        // (several workgroups per CU: measured co-resident up to 2 x 67 KB, not at 2 x 73 KB — plan those against 128 KB)
        const uint32_t budget = k == 1 ? 156u * 1024u : 128u * 1024u;
        if (budget / k <= tab_bytes + wave_bytes) break;
        const uint32_t w = std::min<uint32_t>({16u, (budget / k - tab_bytes) / wave_bytes, cu_waves / k});
        if (w && k * w > best_k * best_w) best_k = k, best_w = w;
      }
      if (const char* e = getenv("VSYN_VQ_WAVES_PER_WG")) best_w = (uint32_t)std::max(1, std::min(16, atoi(e)));
      if (best_w) {
        const uint32_t lds = tab_bytes + best_w * wave_bytes;
        {
          // the attribute belongs to the kernel, not to the handle: only ever raise it, or a handle with smaller tables would pull the
          // limit below what an older handle still launches with
          static std::mutex mu_attr;
          static uint32_t cur_max = 0;
          std::lock_guard<std::mutex> lk2(mu_attr);
          if (lds > cur_max) {
            HIPCHK(hipFuncSetAttribute((const void*)vsyn_residue_vq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cur_max = lds;
          }
        }
        int per_cu = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vsyn_residue_vq_kernel<true>, (int)(VQ_THREADS * best_w), lds));
        if (per_cu > 0) {
          h->vq_tables_in_lds = true;
          h->vq_waves = best_w;
          h->vq_lds_bytes = lds;
          if (const char* e = getenv("VSYN_VQ_WG_PER_CU")) per_cu = atoi(e);
          h->vq_grid = (uint32_t)h->num_cus * (uint32_t)std::max(per_cu, 1);
          if (getenv("VSYN_DEBUG")) fprintf(stderr, "[vsyn] vq: tables in LDS (%u B), %u waves per workgroup, %d workgroups per CU, %u B LDS\n", tab_bytes, best_w, per_cu, lds);
        }
      }
    }
    if (!h->vq_tables_in_lds) {
      h->vq_lds_bytes = wave_bytes;
      int per_cu = 0;
      HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vsyn_residue_vq_kernel<false>, VQ_THREADS, h->vq_lds_bytes));
      if (const char* e = getenv("VSYN_VQ_WG_PER_CU")) per_cu = atoi(e);
      h->vq_grid = (uint32_t)h->num_cus * (uint32_t)std::max(per_cu, 1);
    }
  }
  return VSYN_OK;
}

// residue != nullptr: floats in. Otherwise vq != nullptr: entry numbers in, floats rebuilt on the device and optionally
// copied back to residue_out.
static int submit_host_impl(vsyn_handle* h, uint32_t P, const vsyn_packet* packets, uint32_t S, const vsyn_segment* segments,
                            const uint16_t* ys, const float* residue, const vsyn_vq_batch* vq, float* residue_out, size_t residue_floats,
                            float* pcm, uint64_t plane_stride, uint32_t* emit_len, const vsyn_taps* taps, uint32_t flags, vsyn_status* status,
                            const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (status) {
    status->flags = 0;
    status->first_bad_packet = 0xFFFFFFFFu;
  }
  if (P == 0 || S == 0) return VSYN_OK;
  const bool keep_pcm = (flags & VSYN_SUBMIT_KEEP_PCM) != 0;
  if (!packets || !segments || !ys || (!residue && !vq) || (!pcm && !keep_pcm)) return fail(err, VSYN_ERR_INVALID, "NULL batch pointer");
  if (vq) {
    if (!h->d_vq) return fail(err, VSYN_ERR_INVALID, "vsyn_attach_vq has not been called on this handle");
    if (!vq->packets || (vq->num_cls && !vq->cls) || (vq->num_entries && !vq->entries)) return fail(err, VSYN_ERR_INVALID, "NULL vq batch pointer");
    for (uint32_t p = 0; p < P; ++p)
      if (vq->packets[p].entry_off + vq->packets[p].num_entries > vq->num_entries || vq->packets[p].cls_off > vq->num_cls)
        return fail(err, VSYN_ERR_INVALID, "vq packet %u points outside the entry / classification arrays", p);
  }
  const ConstHeader& H = h->H;
  const uint32_t C = H.channels;
  // host-visible validation (the device re-checks everything it dereferences)
  uint32_t max_seg = 1;
  for (uint32_t g = 0; g < S; ++g) {
    const vsyn_segment& sg = segments[g];
    if (sg.stream >= H.max_streams || (uint64_t)sg.first_packet + sg.num_packets > P || (sg.residue_off & 3))
      return fail(err, VSYN_ERR_INVALID, "segment %u invalid", g);
    uint64_t need = sg.residue_off;
    for (uint32_t q = 0; q < sg.num_packets; ++q) {
      const uint8_t m = packets[sg.first_packet + q].mode;
      need += (uint64_t)C * ((m < H.num_modes && H.mode_blockflag[m]) ? H.bs[1] : H.bs[0]) / 2;
    }
    if (need > residue_floats) return fail(err, VSYN_ERR_INVALID, "segment %u reads past the residue buffer", g);
    max_seg = std::max(max_seg, sg.num_packets);
  }
  HIPCHK(hipSetDevice(h->device));
  const size_t ys_n = (size_t)P * C * H.ys_stride, pcm_n = (size_t)S * C * plane_stride;
  HIPCHK(h->st_pk.ensure(P));
  HIPCHK(h->st_seg.ensure(S));
  HIPCHK(h->st_ys.ensure(ys_n));
  HIPCHK(h->st_res.ensure(residue_floats + 4));
  HIPCHK(h->st_pcm.ensure(pcm_n));
  HIPCHK(h->st_emit.ensure(P));
  vsyn_taps dt = {nullptr, nullptr, nullptr, nullptr};
  if (taps && taps->after_envelope) {
    HIPCHK(h->st_env.ensure(residue_floats + 4));
    dt.after_envelope = h->st_env.p;
  }
  if (taps && taps->pcm_after_mdct) {
    HIPCHK(h->st_blk.ensure(2 * residue_floats + 8));
    dt.pcm_after_mdct = h->st_blk.p;
  }
  if (taps && taps->floor_final) {
    HIPCHK(h->st_fy.ensure(ys_n));
    dt.floor_final = h->st_fy.p;
  }
  if (taps && taps->floor_curve) {
    HIPCHK(h->st_curve.ensure(residue_floats + 4));
    dt.floor_curve = h->st_curve.p;
  }
  // Everything runs on the handle's own stream, so that several handles driven from several host threads overlap their
  // copies and kernels (the NULL stream would serialise them). With pinned host buffers (vsyn_host_alloc) the copies are
  // direct DMA; pageable buffers work too, staged by the runtime.
  hipStream_t hs = h->host_stream;
  HIPCHK(hipMemcpyAsync(h->st_pk.p, packets, sizeof(vsyn_packet) * P, hipMemcpyHostToDevice, hs));
  HIPCHK(hipMemcpyAsync(h->st_seg.p, segments, sizeof(vsyn_segment) * S, hipMemcpyHostToDevice, hs));
  HIPCHK(hipMemcpyAsync(h->st_ys.p, ys, sizeof(uint16_t) * ys_n, hipMemcpyHostToDevice, hs));
  vsyn_vq_batch dvq;
  if (vq) {
    HIPCHK(h->st_vqpk.ensure(P));
    HIPCHK(h->st_cls.ensure((size_t)vq->num_cls + 16));
    HIPCHK(h->st_ent.ensure((size_t)vq->num_entries + 16));
    HIPCHK(hipMemcpyAsync(h->st_vqpk.p, vq->packets, sizeof(vsyn_vq_packet) * P, hipMemcpyHostToDevice, hs));
    if (vq->num_cls) HIPCHK(hipMemcpyAsync(h->st_cls.p, vq->cls, (size_t)vq->num_cls, hipMemcpyHostToDevice, hs));
    if (vq->num_entries) HIPCHK(hipMemcpyAsync(h->st_ent.p, vq->entries, sizeof(uint16_t) * (size_t)vq->num_entries, hipMemcpyHostToDevice, hs));
    dvq.packets = h->st_vqpk.p;
    dvq.cls = h->st_cls.p;
    dvq.entries = h->st_ent.p;
    dvq.num_cls = vq->num_cls;
    dvq.num_entries = vq->num_entries;
  } else {
    HIPCHK(hipMemcpyAsync(h->st_res.p, residue, sizeof(float) * residue_floats, hipMemcpyHostToDevice, hs));
  }
  HIPCHK(hipMemsetAsync(h->st_pcm.p, 0, sizeof(float) * pcm_n, hs));
  if (dt.floor_final) HIPCHK(hipMemsetAsync(h->st_fy.p, 0, ys_n * sizeof(uint16_t), hs));
  if (dt.floor_curve) HIPCHK(hipMemsetAsync(dt.floor_curve, 0, sizeof(uint16_t) * residue_floats, hs));
  if (dt.after_envelope) HIPCHK(hipMemsetAsync(dt.after_envelope, 0, sizeof(float) * residue_floats, hs));
  if (dt.pcm_after_mdct) HIPCHK(hipMemsetAsync(dt.pcm_after_mdct, 0, sizeof(float) * 2 * residue_floats, hs));
  const bool any_tap = dt.after_envelope || dt.pcm_after_mdct || dt.floor_final || dt.floor_curve;
  int rc = submit_device_impl(h, P, h->st_pk.p, S, h->st_seg.p, max_seg, h->st_ys.p, vq ? &dvq : nullptr, h->st_res.p, h->st_pcm.p, plane_stride,
                              h->st_emit.p, any_tap ? &dt : nullptr, flags & ~(VSYN_SUBMIT_INPUTS_READY | VSYN_SUBMIT_KEEP_PCM), hs, err);
  if (rc) return rc;
  h->last_host_plane = plane_stride;
  if (vq && residue_out) HIPCHK(hipMemcpyAsync(residue_out, h->st_res.p, sizeof(float) * residue_floats, hipMemcpyDeviceToHost, hs));
  // results are queued behind the kernels before the one host wait
  if (!keep_pcm) HIPCHK(hipMemcpyAsync(pcm, h->st_pcm.p, sizeof(float) * pcm_n, hipMemcpyDeviceToHost, hs));
  if (emit_len) HIPCHK(hipMemcpyAsync(emit_len, h->st_emit.p, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, hs));
  if (dt.after_envelope) HIPCHK(hipMemcpyAsync(taps->after_envelope, dt.after_envelope, sizeof(float) * residue_floats, hipMemcpyDeviceToHost, hs));
  if (dt.pcm_after_mdct) HIPCHK(hipMemcpyAsync(taps->pcm_after_mdct, dt.pcm_after_mdct, sizeof(float) * 2 * residue_floats, hipMemcpyDeviceToHost, hs));
  if (dt.floor_final) HIPCHK(hipMemcpyAsync(taps->floor_final, dt.floor_final, sizeof(uint16_t) * ys_n, hipMemcpyDeviceToHost, hs));
  if (dt.floor_curve) HIPCHK(hipMemcpyAsync(taps->floor_curve, dt.floor_curve, sizeof(uint16_t) * residue_floats, hipMemcpyDeviceToHost, hs));
  vsyn_status st;
  rc = vsyn_sync_status(h, hs, &st, err);
  if (status) *status = st;
  return rc;
}

int vsyn_submit_host(vsyn_handle* h, uint32_t P, const vsyn_packet* packets, uint32_t S, const vsyn_segment* segments,
                     const uint16_t* ys, const float* residue, size_t residue_floats, float* pcm, uint64_t plane_stride,
                     uint32_t* emit_len, const vsyn_taps* taps, uint32_t flags, vsyn_status* status, const char** err) {
  if (!residue && P && S) return fail(err, VSYN_ERR_INVALID, "NULL batch pointer");
  return submit_host_impl(h, P, packets, S, segments, ys, residue, nullptr, nullptr, residue_floats, pcm, plane_stride, emit_len, taps, flags, status,
                          err);
}

int vsyn_submit_host_vq(vsyn_handle* h, uint32_t P, const vsyn_packet* packets, uint32_t S, const vsyn_segment* segments,
                        const uint16_t* ys, const vsyn_vq_batch* vq, float* residue_out, size_t residue_floats, float* pcm,
                        uint64_t plane_stride, uint32_t* emit_len, const vsyn_taps* taps, uint32_t flags, vsyn_status* status,
                        const char** err) {
  if (!vq && P && S) return fail(err, VSYN_ERR_INVALID, "vq batch is NULL");
  return submit_host_impl(h, P, packets, S, segments, ys, nullptr, vq, residue_out, residue_floats, pcm, plane_stride, emit_len, taps, flags, status,
                          err);
}

int vsyn_pcm_interleave_device(vsyn_handle* h, int format, const float* d_pcm, uint64_t plane_stride, void* d_out, uint64_t out_stride_frames,
                               uint32_t* d_frames, void* hip_stream, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (format != VSYN_PCM_S16 && format != VSYN_PCM_F32) return fail(err, VSYN_ERR_INVALID, "unknown PCM format %d", format);
  if (!d_pcm || !d_out || plane_stride == 0 || out_stride_frames == 0) return fail(err, VSYN_ERR_INVALID, "NULL pointer / zero stride");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->last_S == 0) return fail(err, VSYN_ERR_INVALID, "no submit on this handle yet");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)hip_stream;
  const uint64_t cap = std::min<uint64_t>(std::min(plane_stride, out_stride_frames), 0xFFFFFFFFull);
  const dim3 grid((uint32_t)((cap + 1023) / 1024), h->last_S);
  const SegInfo* si = h->ws_seg[h->last_wb].p;
  if (format == VSYN_PCM_S16)
    vsyn_pcm_interleave_kernel<VSYN_PCM_S16><<<grid, 256, 0, s>>>(h->d_const, si, h->last_S, d_pcm, plane_stride, d_out, out_stride_frames, d_frames);
  else
    vsyn_pcm_interleave_kernel<VSYN_PCM_F32><<<grid, 256, 0, s>>>(h->d_const, si, h->last_S, d_pcm, plane_stride, d_out, out_stride_frames, d_frames);
  HIPCHK(hipGetLastError());
  return VSYN_OK;
}

int vsyn_pcm_fetch_host(vsyn_handle* h, int format, void* out, uint64_t out_stride_frames, uint32_t* frames_out, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (format != VSYN_PCM_S16 && format != VSYN_PCM_F32) return fail(err, VSYN_ERR_INVALID, "unknown PCM format %d", format);
  if (!out || out_stride_frames == 0) return fail(err, VSYN_ERR_INVALID, "NULL pointer / zero stride");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->last_S == 0 || h->last_host_plane == 0) return fail(err, VSYN_ERR_INVALID, "no vsyn_submit_host on this handle yet");
  HIPCHK(hipSetDevice(h->device));
  const size_t elem = format == VSYN_PCM_S16 ? 2 : 4;
  const size_t bytes = (size_t)h->last_S * out_stride_frames * h->H.channels * elem;
  HIPCHK(h->st_conv.ensure(bytes + 16));
  HIPCHK(h->st_frames.ensure(h->last_S));
  hipStream_t s = h->host_stream;
  HIPCHK(hipMemsetAsync(h->st_conv.p, 0, bytes, s));  // frames past a segment's end come back as silence, not as stale staging memory
  const uint64_t cap = std::min<uint64_t>(std::min<uint64_t>(h->last_host_plane, out_stride_frames), 0xFFFFFFFFull);
  const dim3 grid((uint32_t)((cap + 1023) / 1024), h->last_S);
  const SegInfo* si = h->ws_seg[h->last_wb].p;
  if (format == VSYN_PCM_S16)
    vsyn_pcm_interleave_kernel<VSYN_PCM_S16><<<grid, 256, 0, s>>>(h->d_const, si, h->last_S, h->st_pcm.p, h->last_host_plane, h->st_conv.p, out_stride_frames,
                                                                 h->st_frames.p);
  else
    vsyn_pcm_interleave_kernel<VSYN_PCM_F32><<<grid, 256, 0, s>>>(h->d_const, si, h->last_S, h->st_pcm.p, h->last_host_plane, h->st_conv.p, out_stride_frames,
                                                                 h->st_frames.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, h->st_conv.p, bytes, hipMemcpyDeviceToHost, s));
  if (frames_out) HIPCHK(hipMemcpyAsync(frames_out, h->st_frames.p, sizeof(uint32_t) * h->last_S, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return VSYN_OK;
}

int vsyn_pcm_abs_sum_host(vsyn_handle* h, double* out, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (!out) return fail(err, VSYN_ERR_INVALID, "out is NULL");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->last_S == 0 || h->last_host_plane == 0) return fail(err, VSYN_ERR_INVALID, "no vsyn_submit_host on this handle yet");
  HIPCHK(hipSetDevice(h->device));
  const uint32_t units = h->last_S * h->H.channels;
  HIPCHK(h->st_sum.ensure(units));
  vsyn_pcm_abs_sum_kernel<<<units, 256, 0, h->host_stream>>>(h->d_const, h->ws_seg[h->last_wb].p, h->last_S, h->st_pcm.p, h->last_host_plane,
                                                              h->st_sum.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, h->st_sum.p, sizeof(double) * units, hipMemcpyDeviceToHost, h->host_stream));
  HIPCHK(hipStreamSynchronize(h->host_stream));
  return VSYN_OK;
}

int vsyn_host_alloc(size_t bytes, void** out, const char** err) {
  if (!out) return fail(err, VSYN_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return VSYN_OK;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return VSYN_OK;
}

void vsyn_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int vsyn_imdct_device(vsyn_handle* h, uint32_t n, uint32_t count, const float* d_in, float* d_out, void* hip_stream, const char** err) {
  if (!h) return fail(err, VSYN_ERR_INVALID, "handle is NULL");
  if (count == 0) return VSYN_OK;
  if (!d_in || !d_out) return fail(err, VSYN_ERR_INVALID, "NULL pointer");
  int b;
  if (n == h->H.bs[1]) b = 1;
  else if (n == h->H.bs[0]) b = 0;
  else return fail(err, VSYN_ERR_INVALID, "n=%u is neither blocksize of this handle (%u/%u)", n, h->H.bs[0], h->H.bs[1]);
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)hip_stream;
  hipError_t e = hipSuccess;
  bool done = false;
  HIPCHK(profile_begin(h, s, fused_imdct_kernel_name(n)));
  e = fused_imdct_launch(h->H, h->d_const, h->fused, b, n, count, d_in, d_out, s, &done);
  if (e != hipSuccess) return fail(err, VSYN_ERR_HIP, "imdct launch failed: %s", hipGetErrorString(e));
  if (!done) {
    h->profile_kernel = "vsyn_imdct_plain_kernel";
    const uint32_t grid = std::min<uint32_t>(count, 256u * 16u);
    vsyn_imdct_plain_kernel<<<grid, 256, (size_t)n * 4, s>>>(h->d_const, b, n, count, d_in, d_out);
  }
  HIPCHK(profile_end(h, s));
  HIPCHK(hipGetLastError());
  return VSYN_OK;
}

}  // extern "C"
