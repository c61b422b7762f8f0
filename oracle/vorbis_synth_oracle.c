/*
 * vorbis_synth_oracle.c — CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Restates, in plain C and in index form, the arithmetic of the reference's per-audio-packet synthesis
 * half.  Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build with -ffp-contract=off: the reference (g++ -O2, x86-64, no -mfma) performs every multiply and
 * add as a separately rounded binary32 operation, and so must this file to stay bit-identical to it.
 *
 * Parity status: pinned — see vorbis_synth_oracle.h.
 */
#include "vorbis_synth_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

/* ------------------------------------------------------------------------------------------------
 * Vorbis I spec 10.1 table (reference: src/inverse_db_table.h:13-79), kept as bit patterns.
 * ---------------------------------------------------------------------------------------------- */
static const uint32_t k_inverse_db_bits[256] = {
#include "../parseoggvorbis_amd/csrc/vorbis_floor1_inverse_db.inc"
};
const float* orc_inverse_db_table(void) { return (const float*)(const void*)k_inverse_db_bits; }

/* ------------------------------------------------------------------------------------------------
 * Floor-1 helpers.  src/Utils.hpp:58-183 (Vorbis I spec 9.2.4-9.2.7).
 * ---------------------------------------------------------------------------------------------- */

/* Utils.hpp:60-87: position of the greatest v[j] < v[idx] among j < idx. */
int orc_low_neighbor(const uint32_t* v, int idx) {
  int best = -1;
  for (int j = 0; j < idx; ++j)
    if (v[j] < v[idx] && (best < 0 || v[j] > v[best])) best = j;
  return best;
}

/* Utils.hpp:91-118: position of the smallest v[j] > v[idx] among j < idx. */
int orc_high_neighbor(const uint32_t* v, int idx) {
  int best = -1;
  for (int j = 0; j < idx; ++j)
    if (v[j] > v[idx] && (best < 0 || v[j] < v[best])) best = j;
  return best;
}

/* Utils.hpp:122-137: y at X on the segment (x0,y0)-(x1,y1), unsigned integer arithmetic (wraps like uint32_t). */
uint32_t orc_render_point(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t X) {
  uint32_t adx = x1 - x0;
  int up = y1 >= y0;
  uint32_t ady = up ? (y1 - y0) : (y0 - y1);
  uint32_t off = (ady * (X - x0)) / adx;
  return up ? y0 + off : y0 - off;
}

/* Utils.hpp:143-183: integer DDA; writes vec[x] for x in [x0, min(x1,len)). */
void orc_render_line(size_t x0, uint32_t y0, size_t x1, uint32_t y1, uint32_t* vec, size_t len) {
  if (x0 >= len) return;
  size_t adx = x1 - x0;
  int up = y1 >= y0;
  uint32_t ady = up ? (y1 - y0) : (y0 - y1);
  uint32_t base = (uint32_t)(ady / adx);
  uint32_t step_big = base + 1;
  uint32_t rem = ady - base * (uint32_t)adx;
  uint32_t err = 0;
  uint32_t y = y0;
  vec[x0] = y0;
  for (size_t x = x0 + 1; x < x1 && x < len; ++x) {
    err += rem;
    if (err >= adx) {
      err -= (uint32_t)adx;
      y = up ? y + step_big : y - step_big;
    } else {
      y = up ? y + base : y - base;
    }
    vec[x] = y;
  }
}

/* src/ParseOggVorbis.hpp:484-492 + 521-589: the compute tail of VorbisFloor1::decode. */
int orc_floor1_synth(const uint32_t* xs, int posts, int multiplier, const uint32_t* ys, size_t n,
                     float* out, uint32_t* final_ys_o, uint8_t* flags_o, uint32_t* curve_o) {
  static const uint32_t range_of[5] = {0, 256, 128, 86, 64}; /* hpp:486-492 */
  if (multiplier < 1 || multiplier > 4 || posts < 2 || posts > VSYN_MAX_POSTS) return VSYN_ST_FLOOR_RANGE;
  const uint32_t range = range_of[multiplier];
  uint32_t fy[VSYN_MAX_POSTS];
  uint8_t flag[VSYN_MAX_POSTS];

  /* step 1, amplitude value synthesis — hpp:523-559 */
  flag[0] = flag[1] = 1;
  fy[0] = ys[0];
  fy[1] = ys[1];
  for (int i = 2; i < posts; ++i) {
    int lo = orc_low_neighbor(xs, i), hi = orc_high_neighbor(xs, i);
    if (lo < 0 || hi < 0) return VSYN_ST_FLOOR_RANGE; /* cannot happen for valid xs (xs[0]=0, xs[1]=max) */
    uint32_t predicted = orc_render_point(xs[lo], fy[lo], xs[hi], fy[hi], xs[i]);
    uint32_t val = ys[i];
    if (!(predicted <= range)) return VSYN_ST_FLOOR_RANGE; /* hpp:536 */
    uint32_t high_room = range - predicted, low_room = predicted;
    uint32_t room = (high_room < low_room ? high_room : low_room) * 2;
    if (val == 0) {
      flag[i] = 0;
      fy[i] = predicted;
    } else {
      flag[lo] = flag[hi] = flag[i] = 1;
      if (val >= room)
        fy[i] = (high_room > low_room) ? val - low_room + predicted : predicted - val + high_room - 1;
      else
        fy[i] = (val % 2 == 1) ? predicted - (val + 1) / 2 : predicted + val / 2;
    }
  }
  if (final_ys_o) memcpy(final_ys_o, fy, sizeof(uint32_t) * (size_t)posts);
  if (flags_o) memcpy(flags_o, flag, (size_t)posts);

  /* step 2, curve synthesis — hpp:563-584; posts visited in ascending x (xs_sorted_idx, hpp:459-469) */
  int order[VSYN_MAX_POSTS];
  for (int i = 0; i < posts; ++i) order[i] = i;
  for (int i = 1; i < posts; ++i) { /* stable insertion sort; xs are distinct in a valid stream */
    int k = order[i], j = i;
    while (j > 0 && xs[order[j - 1]] > xs[k]) { order[j] = order[j - 1]; --j; }
    order[j] = k;
  }
  uint32_t* curve = curve_o ? curve_o : (uint32_t*)malloc(sizeof(uint32_t) * n);
  memset(curve, 0, sizeof(uint32_t) * n);
  uint32_t lx = 0, hx = 0, ly = fy[order[0]] * (uint32_t)multiplier, hy = 0;
  for (int i = 1; i < posts; ++i) {
    if (!flag[order[i]]) continue;
    hx = xs[order[i]];
    hy = fy[order[i]] * (uint32_t)multiplier;
    orc_render_line(lx, ly, hx, hy, curve, n);
    lx = hx;
    ly = hy;
  }
  if (hx < n) orc_render_line(hx, hy, n, hy, curve, n);

  /* hpp:586-589 */
  int rc = 0;
  const float* table = orc_inverse_db_table();
  for (size_t i = 0; i < n; ++i) {
    if (curve[i] >= 256) { rc = VSYN_ST_FLOOR_VALUE; break; }
    if (out) out[i] = table[curve[i]];
  }
  if (!curve_o) free(curve);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Inverse channel coupling.  src/ParseOggVorbis.hpp:1219-1240 (one coupling step, all bins).
 * ---------------------------------------------------------------------------------------------- */
void orc_inverse_coupling(float* mag, float* ang, size_t len) {
  for (size_t j = 0; j < len; ++j) {
    float m = mag[j], a = ang[j], m2 = m, a2 = a;
    if (m > 0) {
      if (a > 0) a2 = m - a;
      else { a2 = m; m2 = m + a; }
    } else {
      if (a > 0) a2 = m + a;
      else { a2 = m; m2 = m - a; }
    }
    mag[j] = m2;
    ang[j] = a2;
  }
}

/* ------------------------------------------------------------------------------------------------
 * IMDCT.  src/mdct.cpp:88-127 (tables), 433-527 (mdct_backward) and the helpers it calls:
 * 353-373 (stage driver), 253-351 (radix-2 stages), 130-250 (fixed 32/16/8 tails), 383-431 (bit reverse).
 * Same operations on the same operands; loops are written over complex-pair indices instead of the
 * reference's moving pointers.  out[] doubles as workspace exactly as in the reference.
 * ---------------------------------------------------------------------------------------------- */
struct orc_mdct {
  int n, log2n;
  float* trig; /* n + n/4 */
  int* bitrev; /* n/4 */
};

orc_mdct* orc_mdct_new(int n) {
  orc_mdct* m = (orc_mdct*)calloc(1, sizeof(*m));
  m->n = n;
  m->log2n = (int)rint(log((float)n) / log(2.f)); /* mdct.cpp:94 */
  m->trig = (float*)malloc(sizeof(float) * (size_t)(n + n / 4));
  m->bitrev = (int*)malloc(sizeof(int) * (size_t)(n / 4));
  float* T = m->trig;
  const int n2 = n >> 1;
  for (int i = 0; i < n / 4; ++i) { /* mdct.cpp:101-106 */
    T[2 * i] = (float)cos((M_PI / n) * (4 * i));
    T[2 * i + 1] = (float)-sin((M_PI / n) * (4 * i));
    T[n2 + 2 * i] = (float)cos((M_PI / (2 * n)) * (2 * i + 1));
    T[n2 + 2 * i + 1] = (float)sin((M_PI / (2 * n)) * (2 * i + 1));
  }
  for (int i = 0; i < n / 8; ++i) { /* mdct.cpp:107-110 */
    T[n + 2 * i] = (float)(cos((M_PI / n) * (4 * i + 2)) * .5);
    T[n + 2 * i + 1] = (float)(-sin((M_PI / n) * (4 * i + 2)) * .5);
  }
  { /* mdct.cpp:114-125 */
    const int mask = (1 << (m->log2n - 1)) - 1, msb = 1 << (m->log2n - 2);
    for (int i = 0; i < n / 8; ++i) {
      int acc = 0;
      for (int j = 0; msb >> j; ++j)
        if ((msb >> j) & i) acc |= 1 << j;
      m->bitrev[2 * i] = ((~acc) & mask) - 1;
      m->bitrev[2 * i + 1] = acc;
    }
  }
  return m;
}

void orc_mdct_free(orc_mdct* m) {
  if (!m) return;
  free(m->trig);
  free(m->bitrev);
  free(m);
}
const float* orc_mdct_trig(const orc_mdct* m) { return m->trig; }
const int* orc_mdct_bitrev(const orc_mdct* m) { return m->bitrev; }

static const float C1 = .92387953251128675613F; /* cos(pi/8),  mdct.h:80 */
static const float C2 = .70710678118654752441F; /* cos(2pi/8), mdct.h:79 */
static const float C3 = .38268343236508977175F; /* cos(3pi/8), mdct.h:78 */

/* mdct.cpp:130-151 */
static void tail8(float* x) {
  const float s62 = x[6] + x[2], d62 = x[6] - x[2], s40 = x[4] + x[0], d40 = x[4] - x[0];
  const float d51 = x[5] - x[1], d73 = x[7] - x[3], s51 = x[5] + x[1], s73 = x[7] + x[3];
  x[6] = s62 + s40;
  x[4] = s62 - s40;
  x[0] = d62 + d51;
  x[2] = d62 - d51;
  x[3] = d73 + d40;
  x[1] = d73 - d40;
  x[7] = s73 + s51;
  x[5] = s73 - s51;
}

/* mdct.cpp:154-186 */
static void tail16(float* x) {
  float p, q;
  p = x[1] - x[9];   q = x[0] - x[8];   x[8] += x[0];   x[9] += x[1];
  x[0] = (p + q) * C2;                  x[1] = (p - q) * C2;
  p = x[3] - x[11];  q = x[10] - x[2];  x[10] += x[2];  x[11] += x[3];
  x[2] = p;                             x[3] = q;
  p = x[12] - x[4];  q = x[13] - x[5];  x[12] += x[4];  x[13] += x[5];
  x[4] = (p - q) * C2;                  x[5] = (p + q) * C2;
  p = x[14] - x[6];  q = x[15] - x[7];  x[14] += x[6];  x[15] += x[7];
  x[6] = p;                             x[7] = q;
  tail8(x);
  tail8(x + 8);
}

/* mdct.cpp:189-250 */
static void tail32(float* x) {
  float p, q;
  p = x[30] - x[14]; q = x[31] - x[15]; x[30] += x[14]; x[31] += x[15];
  x[14] = p;                            x[15] = q;
  p = x[28] - x[12]; q = x[29] - x[13]; x[28] += x[12]; x[29] += x[13];
  x[12] = p * C1 - q * C3;              x[13] = p * C3 + q * C1;
  p = x[26] - x[10]; q = x[27] - x[11]; x[26] += x[10]; x[27] += x[11];
  x[10] = (p - q) * C2;                 x[11] = (p + q) * C2;
  p = x[24] - x[8];  q = x[25] - x[9];  x[24] += x[8];  x[25] += x[9];
  x[8] = p * C3 - q * C1;               x[9] = q * C3 + p * C1;
  p = x[22] - x[6];  q = x[7] - x[23];  x[22] += x[6];  x[23] += x[7];
  x[6] = q;                             x[7] = p;
  p = x[4] - x[20];  q = x[5] - x[21];  x[20] += x[4];  x[21] += x[5];
  x[4] = q * C1 + p * C3;               x[5] = q * C3 - p * C1;
  p = x[2] - x[18];  q = x[3] - x[19];  x[18] += x[2];  x[19] += x[3];
  x[2] = (q + p) * C2;                  x[3] = (q - p) * C2;
  p = x[0] - x[16];  q = x[1] - x[17];  x[16] += x[0];  x[17] += x[1];
  x[0] = q * C3 + p * C1;               x[1] = q * C1 - p * C3;
  tail16(x);
  tail16(x + 16);
}

/* One radix-2 stage over a block of `points` floats (= points/2 complex values), mdct.cpp:253-297 (stride 4)
 * and 300-351 (stride trigint).  The reference walks pairs from the top of each half downwards, the twiddle
 * pointer advancing by `tstride` floats per pair; pair q (counted from the bottom) therefore uses
 * T[(points/4 - 1 - q) * tstride]. Pairs are independent, so visiting order does not matter. */
static void radix2_stage(const float* T, float* x, int points, int tstride) {
  const int quarter = points >> 2; /* complex pairs */
  for (int q = 0; q < quarter; ++q) {
    float* hi = x + (points >> 1) + 2 * q;
    float* lo = x + 2 * q;
    const float* t = T + (quarter - 1 - q) * tstride;
    const float dr = hi[0] - lo[0], di = hi[1] - lo[1];
    hi[0] += lo[0];
    hi[1] += lo[1];
    lo[0] = di * t[1] + dr * t[0];
    lo[1] = di * t[0] - dr * t[1];
  }
}

/* mdct.cpp:353-373 */
static void all_stages(const orc_mdct* m, float* x, int points) {
  int stages = m->log2n - 5;
  if (--stages > 0) radix2_stage(m->trig, x, points, 4);
  for (int i = 1; --stages > 0; ++i)
    for (int j = 0; j < (1 << i); ++j) radix2_stage(m->trig, x + (points >> i) * j, points >> i, 4 << i);
  for (int j = 0; j < points; j += 32) tail32(x + j);
}

void orc_mdct_backward(const orc_mdct* m, const float* in, float* out) {
  const int n = m->n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
  const float* T = m->trig;
  float* up = out + n2; /* upper half of out[] = n4 complex work values */

  /* pre-rotation, mdct.cpp:440-466.  Lower n8 complex slots take odd input bins (loop 1, 444-452),
   * upper n8 slots take even bins read from the top down (loop 2, 458-466). */
  for (int c = 0; c < n8; ++c) {
    const float a = in[4 * c + 1], b = in[4 * c + 3];
    const float tc = T[n2 - 2 - 2 * c], ts = T[n2 - 1 - 2 * c];
    up[2 * c] = -b * ts - a * tc;
    up[2 * c + 1] = a * ts - b * tc;
  }
  for (int c = n8; c < n4; ++c) {
    const float a = in[n - 4 - 4 * c], b = in[n - 2 - 4 * c];
    const float tc = T[n2 - 2 - 2 * c], ts = T[n2 - 1 - 2 * c];
    up[2 * c] = a * ts + b * tc;
    up[2 * c + 1] = a * tc - b * ts;
  }

  all_stages(m, up, n2); /* mdct.cpp:468 */

  /* bit-reverse + half twiddle, mdct.cpp:383-431: reads the upper half, writes the lower half */
  for (int h = 0; h < n8; ++h) {
    const float* x0 = up + m->bitrev[2 * h];
    const float* x1 = up + m->bitrev[2 * h + 1];
    const float tc = T[n + 2 * h], ts = T[n + 2 * h + 1];
    const float r0 = x0[1] - x1[1], r1 = x0[0] + x1[0];
    const float r2 = r1 * tc + r0 * ts, r3 = r1 * ts - r0 * tc;
    const float h0 = (x0[1] + x1[1]) * .5f, h1 = (x0[0] - x1[0]) * .5f;
    out[2 * h] = h0 + r2;
    out[n2 - 2 - 2 * h] = h0 - r2;
    out[2 * h + 1] = h1 + r3;
    out[n2 - 1 - 2 * h] = r3 - h1;
  }

  /* post-rotation, mdct.cpp:473-497: lower half (n4 complex) -> upper half */
  for (int k = 0; k < n4; ++k) {
    const float a = out[2 * k], b = out[2 * k + 1];
    const float tc = T[n2 + 2 * k], ts = T[n2 + 2 * k + 1];
    out[n2 + n4 - 1 - k] = a * ts - b * tc;
    out[n2 + n4 + k] = -(a * tc + b * ts);
  }
  /* mirror fills, mdct.cpp:499-525 */
  for (int j = 0; j < n4; ++j) {
    const float v = out[n2 + n4 - 1 - j];
    out[n4 - 1 - j] = v;
    out[n4 + j] = -v;
  }
  for (int j = 0; j < n4; ++j) out[n2 + n4 - 1 - j] = out[n2 + n4 + j];
}

void orc_imdct_closed_form(int n, const float* in, double* out) {
  const int n2 = n / 2;
  for (int i = 0; i < n; ++i) {
    double acc = 0;
    for (int k = 0; k < n2; ++k) acc += (double)in[k] * cos(2.0 * M_PI / n * (i + 0.5 + n / 4.0) * (k + 0.5));
    out[i] = acc;
  }
}

void orc_imdct_batch(int n, uint32_t count, const float* in, float* out) {
  orc_mdct* m = orc_mdct_new(n);
  for (uint32_t i = 0; i < count; ++i) orc_mdct_backward(m, in + (size_t)i * (n / 2), out + (size_t)i * n);
  orc_mdct_free(m);
}

/* ------------------------------------------------------------------------------------------------
 * Window tables.  src/ParseOggVorbis.hpp:837-862 (precalc) for one (block_flag, prev, next) combination;
 * selection rule hpp:874-886 (short blocks ignore prev/next).
 * ---------------------------------------------------------------------------------------------- */
void orc_window(int bs0, int bs1, int block_flag, int prev, int next, float* w) {
  const int n = block_flag ? bs1 : bs0;
  if (!block_flag) prev = next = 0;
  const int left = (prev ? bs1 : bs0) / 2, right = (next ? bs1 : bs0) / 2;
  const int left_begin = n / 4 - left / 2, right_begin = n - n / 4 - right / 2;
  memset(w, 0, sizeof(float) * (size_t)n);
  for (int i = 0; i < left; ++i) {
    float x = sinf((float)(M_PI_2 * (i + 0.5) / left));
    w[left_begin + i] = sinf((float)(M_PI_2 * x * x));
  }
  for (int i = left_begin + left; i < right_begin; ++i) w[i] = 1.f;
  for (int i = 0; i < right; ++i) {
    float x = sinf((float)(M_PI_2 * (right - i - .5) / right));
    w[right_begin + i] = sinf((float)(M_PI_2 * x * x));
  }
}

/* ------------------------------------------------------------------------------------------------
 * Overlap-add decode state.  src/ParseOggVorbis.hpp:975-1115 (VorbisStreamDecodeState), same fields.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float** buf; /* [channels][cap] — hpp:983 pcm_buffer */
  uint32_t cap;
  uint32_t pcm_offset;                  /* hpp:984 */
  int32_t prev_second_half_window_off;  /* hpp:985 (int16_t upstream; the values fit) */
  uint32_t prev_win, cur_win;           /* hpp:986 */
  uint64_t abs_total_pos;               /* hpp:987 */
  int64_t expected_end;                 /* hpp:988 */
} orc_state;

static void state_reset(orc_state* s, int channels) {
  for (int c = 0; c < channels; ++c) memset(s->buf[c], 0, sizeof(float) * s->cap);
  s->pcm_offset = 0;
  s->prev_second_half_window_off = 0;
  s->prev_win = s->cur_win = 0;
  s->abs_total_pos = 0;
  s->expected_end = 0;
}

/* hpp:1069-1109 */
static int state_advance(orc_state* s, int channels, uint32_t next_win) {
  const uint32_t cur = s->cur_win;
  uint32_t second_half = s->pcm_offset + cur / 2;
  int32_t next_off = (int32_t)s->pcm_offset + ((int32_t)cur / 4) * 3 - ((int32_t)next_win / 4);
  if (next_off + (int64_t)next_win >= (int64_t)s->cap) { /* slide left, keep the second half — hpp:1075-1093 */
    int32_t needed = (int32_t)s->pcm_offset + (int32_t)(cur / 2) - next_off;
    second_half = needed < 0 ? 0u : (uint32_t)needed;
    uint32_t wipe_from = second_half + cur / 2;
    for (int c = 0; c < channels; ++c) {
      memmove(s->buf[c] + second_half, s->buf[c] + s->pcm_offset + cur / 2, (cur / 2) * sizeof(float));
      memset(s->buf[c] + wipe_from, 0, (s->cap - wipe_from) * sizeof(float));
    }
    next_off = needed < 0 ? -needed : 0;
  } else if (next_off < 0) { /* short then long: slide right — hpp:1094-1103 */
    uint32_t extra = (uint32_t)(-next_off);
    second_half += extra;
    for (int c = 0; c < channels; ++c) {
      memmove(s->buf[c] + s->pcm_offset + extra, s->buf[c] + s->pcm_offset, cur * sizeof(float));
      memset(s->buf[c], 0, (s->pcm_offset + extra) * sizeof(float));
    }
    next_off = 0;
  }
  if (next_win < cur && !(next_off > 0)) return -1; /* hpp:1104-1105 */
  s->prev_second_half_window_off = (int32_t)second_half - next_off;
  s->pcm_offset = (uint32_t)next_off;
  return 0;
}

/* hpp:1061-1067 */
static int state_begin_packet(orc_state* s, int channels, uint32_t win) {
  if (s->cur_win > 0 && state_advance(s, channels, win)) return -1;
  s->prev_win = s->cur_win;
  s->cur_win = win;
  return 0;
}

/* hpp:1008-1017 */
static void state_add_frame(orc_state* s, int ch, const float* pcm, const float* window, uint32_t n) {
  float* dst = s->buf[ch] + s->pcm_offset;
  for (uint32_t i = 0; i < n; ++i) dst[i] += pcm[i] * window[i];
}

/* hpp:1019-1059; returns number of frames (>=0) copied to dst planes, or -1 on the reference's CHECK failures */
static int64_t state_forward(orc_state* s, int channels, float* const* dst, uint64_t room) {
  uint32_t frames = 0;
  if (s->prev_win > 0) frames = s->prev_win / 4 + s->cur_win / 4; /* hpp:1021-1027 */
  if (s->expected_end >= 0) {
    if (!(s->abs_total_pos <= (uint64_t)s->expected_end)) return -1; /* hpp:1029 */
    if (s->abs_total_pos + frames >= (uint64_t)s->expected_end)
      frames = (uint32_t)((uint64_t)s->expected_end - s->abs_total_pos);
    else
      return -1; /* hpp:1041 */
  }
  if (frames > room) return -2;
  if (frames > 0) {
    for (int c = 0; c < channels; ++c)
      memcpy(dst[c], s->buf[c] + (int64_t)s->pcm_offset + s->prev_second_half_window_off, frames * sizeof(float));
    s->abs_total_pos += frames;
  }
  return frames;
}

/* ------------------------------------------------------------------------------------------------
 * Whole path: one handle = one stream setup + max_streams decode states, mirroring vsyn_*.
 * Follows VorbisStream::parse_audio, src/ParseOggVorbis.hpp:1154-1271, from the point where the
 * entropy decode has produced mode/flags, "floor1 ys" and "after_residue".
 * ---------------------------------------------------------------------------------------------- */
struct orc_handle {
  uint32_t channels, bs[2], ys_stride, max_streams;
  uint32_t num_floors, num_mappings, num_modes;
  struct { uint32_t mult, posts; uint32_t xs[VSYN_MAX_POSTS]; } * floors;
  struct { uint32_t ncoup; vsyn_coupling* coup; uint8_t* chfloor; } * maps;
  vsyn_mode* modes;
  orc_mdct* mdct[2];
  float* win[2][4]; /* [block_flag][prev + 2*next] */
  orc_state* st;
  float *floor_buf, *pcm_buf, *res_buf;
};

static uint32_t round_up4(uint32_t v) { return (v + 3u) & ~3u; }

orc_handle* orc_create(const vsyn_setup* su, uint32_t max_streams) {
  if (!su || su->channels < 1 || su->channels > VSYN_MAX_CHANNELS) return NULL;
  orc_handle* h = (orc_handle*)calloc(1, sizeof(*h));
  h->channels = su->channels;
  h->bs[0] = su->blocksize0;
  h->bs[1] = su->blocksize1;
  h->max_streams = max_streams;
  h->num_floors = su->num_floors;
  h->num_mappings = su->num_mappings;
  h->num_modes = su->num_modes;
  h->floors = calloc(su->num_floors, sizeof(*h->floors));
  uint32_t maxp = 2;
  for (uint32_t f = 0; f < su->num_floors; ++f) {
    h->floors[f].mult = su->floors[f].multiplier;
    h->floors[f].posts = su->floors[f].num_posts;
    memcpy(h->floors[f].xs, su->floors[f].xs, sizeof(uint32_t) * su->floors[f].num_posts);
    if (su->floors[f].num_posts > maxp) maxp = su->floors[f].num_posts;
  }
  h->ys_stride = round_up4(maxp);
  h->maps = calloc(su->num_mappings, sizeof(*h->maps));
  for (uint32_t m = 0; m < su->num_mappings; ++m) {
    h->maps[m].ncoup = su->mappings[m].num_couplings;
    h->maps[m].coup = malloc(sizeof(vsyn_coupling) * (su->mappings[m].num_couplings + 1));
    memcpy(h->maps[m].coup, su->mappings[m].couplings, sizeof(vsyn_coupling) * su->mappings[m].num_couplings);
    h->maps[m].chfloor = malloc(su->channels);
    memcpy(h->maps[m].chfloor, su->mappings[m].channel_floor, su->channels);
  }
  h->modes = malloc(sizeof(vsyn_mode) * su->num_modes);
  memcpy(h->modes, su->modes, sizeof(vsyn_mode) * su->num_modes);
  for (int b = 0; b < 2; ++b) {
    h->mdct[b] = orc_mdct_new((int)h->bs[b]); /* hpp:1352-1353 */
    for (int w = 0; w < 4; ++w) {
      h->win[b][w] = malloc(sizeof(float) * h->bs[b]);
      orc_window((int)h->bs[0], (int)h->bs[1], b, w & 1, (w >> 1) & 1, h->win[b][w]);
    }
  }
  h->st = calloc(max_streams, sizeof(orc_state));
  for (uint32_t s = 0; s < max_streams; ++s) {
    h->st[s].cap = h->bs[0] * 5 + h->bs[1] * 5; /* hpp:1359 */
    h->st[s].buf = malloc(sizeof(float*) * h->channels);
    for (uint32_t c = 0; c < h->channels; ++c) h->st[s].buf[c] = calloc(h->st[s].cap, sizeof(float));
  }
  h->floor_buf = malloc(sizeof(float) * h->bs[1] * h->channels);
  h->res_buf = malloc(sizeof(float) * (h->bs[1] / 2) * h->channels);
  h->pcm_buf = malloc(sizeof(float) * h->bs[1]);
  return h;
}

void orc_destroy(orc_handle* h) {
  if (!h) return;
  for (uint32_t s = 0; s < h->max_streams; ++s) {
    for (uint32_t c = 0; c < h->channels; ++c) free(h->st[s].buf[c]);
    free(h->st[s].buf);
  }
  free(h->st);
  for (int b = 0; b < 2; ++b) {
    orc_mdct_free(h->mdct[b]);
    for (int w = 0; w < 4; ++w) free(h->win[b][w]);
  }
  for (uint32_t m = 0; m < h->num_mappings; ++m) { free(h->maps[m].coup); free(h->maps[m].chfloor); }
  free(h->maps);
  free(h->floors);
  free(h->modes);
  free(h->floor_buf);
  free(h->res_buf);
  free(h->pcm_buf);
  free(h);
}

uint32_t orc_ys_stride(const orc_handle* h) { return h->ys_stride; }

void orc_reset_streams(orc_handle* h) {
  for (uint32_t s = 0; s < h->max_streams; ++s) state_reset(&h->st[s], (int)h->channels);
}

static void flag_status(vsyn_status* st, uint32_t f, uint32_t pkt) {
  if (!st) return;
  st->flags |= f;
  if (pkt < st->first_bad_packet) st->first_bad_packet = pkt;
}

int orc_submit(orc_handle* h, uint32_t num_packets, const vsyn_packet* packets, uint32_t num_segments,
               const vsyn_segment* segments, const uint16_t* ys, const float* residue, float* pcm,
               uint64_t plane_stride, uint32_t* emit_len, const vsyn_taps* taps, vsyn_status* status) {
  const uint32_t C = h->channels;
  vsyn_status local = {0, 0xFFFFFFFFu};
  if (!status) status = &local;
  status->flags = 0;
  status->first_bad_packet = 0xFFFFFFFFu;
  if (emit_len) memset(emit_len, 0, sizeof(uint32_t) * num_packets);

  for (uint32_t g = 0; g < num_segments; ++g) {
    const vsyn_segment* sg = &segments[g];
    if (sg->stream >= h->max_streams || (uint64_t)sg->first_packet + sg->num_packets > num_packets) return VSYN_ERR_INVALID;
    orc_state* st = &h->st[sg->stream];
    if (sg->flags & VSYN_SEG_RESET) state_reset(st, (int)C);
    uint64_t res_off = sg->residue_off;
    uint64_t written = 0;
    float* dst[VSYN_MAX_CHANNELS];

    for (uint32_t q = 0; q < sg->num_packets; ++q) {
      const uint32_t p = sg->first_packet + q;
      const vsyn_packet* pk = &packets[p];
      if (pk->mode >= h->num_modes) { flag_status(status, VSYN_ST_BAD_MODE, p); break; }
      const vsyn_mode* mode = &h->modes[pk->mode];
      const int lng = mode->block_flag ? 1 : 0;
      const uint32_t n = h->bs[lng], n2 = n / 2;
      const int widx = lng ? ((pk->prev_long ? 1 : 0) | (pk->next_long ? 2 : 0)) : 0; /* hpp:874-886 */
      const float* window = h->win[lng][widx];
      if (state_begin_packet(st, (int)C, n)) { flag_status(status, VSYN_ST_GRANULE, p); break; } /* hpp:1156 */

      /* 4.3.2 floor curves, hpp:1159-1172 (floor_outputs is zero-initialised, n entries per channel) */
      uint32_t used = 0;
      int bad = 0;
      memset(h->floor_buf, 0, sizeof(float) * n * C);
      for (uint32_t c = 0; c < C && !bad; ++c) {
        if (!((pk->floor_used >> c) & 1u)) continue;
        const uint32_t f = h->maps[mode->mapping].chfloor[c];
        uint32_t y32[VSYN_MAX_POSTS], fy[VSYN_MAX_POSTS];
        uint8_t fl[VSYN_MAX_POSTS];
        uint32_t curve[VSYN_MAX_BLOCKSIZE];
        const uint16_t* row = ys + ((size_t)p * C + c) * h->ys_stride;
        for (uint32_t i = 0; i < h->floors[f].posts; ++i) y32[i] = row[i];
        int rc = orc_floor1_synth(h->floors[f].xs, (int)h->floors[f].posts, (int)h->floors[f].mult, y32, n,
                                  h->floor_buf + (size_t)n * c, fy, fl, curve);
        if (taps && taps->floor_curve && rc != VSYN_ST_FLOOR_RANGE) { /* "floor1 floor", hpp:585 (first n/2 of the n rendered values) */
          uint16_t* t = taps->floor_curve + res_off + (size_t)c * n2;
          for (uint32_t i = 0; i < n2; ++i) t[i] = (uint16_t)(curve[i] > 65535u ? 65535u : curve[i]);
        }
        if (rc) { flag_status(status, (uint32_t)rc, p); bad = 1; break; }
        used |= 1u << c;
        if (taps && taps->floor_final) {
          uint16_t* t = taps->floor_final + ((size_t)p * C + c) * h->ys_stride;
          for (uint32_t i = 0; i < h->floors[f].posts; ++i)
            t[i] = (uint16_t)((fy[i] * h->floors[f].mult) | ((uint32_t)fl[i] << 15));
        }
      }
      if (bad) break;

      /* 4.3.3 nonzero propagate, hpp:1174-1180 */
      const vsyn_coupling* cp = h->maps[mode->mapping].coup;
      const uint32_t ncoup = h->maps[mode->mapping].ncoup;
      for (uint32_t k = 0; k < ncoup; ++k)
        if (((used >> cp[k].angle) | (used >> cp[k].magnitude)) & 1u) used |= (1u << cp[k].angle) | (1u << cp[k].magnitude);

      /* "after_residue" -> work copy (the reference works in place on residue_outputs, hpp:1183-1211) */
      memcpy(h->res_buf, residue + res_off, sizeof(float) * n2 * C);

      /* 4.3.5 inverse coupling, reverse order, hpp:1214-1241 */
      for (uint32_t k = ncoup; k > 0; --k)
        orc_inverse_coupling(h->res_buf + (size_t)cp[k - 1].magnitude * n2, h->res_buf + (size_t)cp[k - 1].angle * n2, n2);

      /* 4.3.6 dot product, hpp:1245-1255 */
      for (uint32_t c = 0; c < C; ++c) {
        float* r = h->res_buf + (size_t)c * n2;
        if ((used >> c) & 1u) {
          const float* f = h->floor_buf + (size_t)n * c;
          for (uint32_t i = 0; i < n2; ++i) r[i] *= f[i];
        }
        if (taps && taps->after_envelope) memcpy(taps->after_envelope + res_off + (size_t)c * n2, r, sizeof(float) * n2);
      }

      /* 4.3.7 inverse MDCT + overlap/add, hpp:1258-1268 */
      for (uint32_t c = 0; c < C; ++c) {
        orc_mdct_backward(h->mdct[lng], h->res_buf + (size_t)c * n2, h->pcm_buf);
        if (taps && taps->pcm_after_mdct) memcpy(taps->pcm_after_mdct + 2 * res_off + (size_t)c * n, h->pcm_buf, sizeof(float) * n);
        state_add_frame(st, (int)c, h->pcm_buf, window, n);
      }

      /* hpp:1270-1271 with the page granule of hpp:1456-1459 */
      st->expected_end = pk->granule;
      for (uint32_t c = 0; c < C; ++c) dst[c] = pcm + ((size_t)g * C + c) * plane_stride + written;
      int64_t frames = state_forward(st, (int)C, dst, plane_stride - written);
      if (frames == -2) { flag_status(status, VSYN_ST_PLANE_OVERFLOW, p); break; }
      if (frames < 0) { flag_status(status, VSYN_ST_GRANULE, p); break; }
      if (emit_len) emit_len[p] = (uint32_t)frames;
      written += (uint64_t)frames;
      res_off += (uint64_t)n2 * C;
    }
  }
  return status->flags ? VSYN_ERR_STREAM : VSYN_OK;
}


/* ------------------------------------------------------------------------------------------------
 * Residue VQ accumulate — follows VorbisResidue::decode, src/ParseOggVorbis.hpp:670-762, with the two
 * bit-serial reads (class_codebook.decodeScalar hpp:714, vq_codebook.decodeVector hpp:742/750) replaced
 * by the next number from cls[] / entries[]; decodeVector's table look-up is hpp:367-374
 * (lookup_table_[entry * dimensions_ + l]).  The caller loop over submaps is VorbisStream::parse_audio,
 * hpp:1184-1209.  Partition counter: advanced once per classword (Vorbis I 8.6.2), which is what the
 * reference does for the 1-channel vectors of both fixtures; see DESIGN.md §7 for hpp:756.
 * ------------------------------------------------------------------------------------------------ */
static int orc_residue_vq_one(const vsyn_vq_setup* vq, const vsyn_residue* r, int type, uint32_t nch, const uint8_t* ch_used,
                              uint32_t len, const uint8_t** cls_io, const uint8_t* cls_end, const uint16_t** ent_io,
                              const uint16_t* ent_end, float* const* out) {
  uint32_t lim_begin = r->begin < len ? r->begin : len; /* hpp:696-698 */
  uint32_t lim_end = r->end < len ? r->end : len;
  if (lim_begin > lim_end) return VSYN_ST_BAD_VQ;
  uint32_t n_to_read = lim_end - lim_begin;
  if (n_to_read == 0) return 0; /* hpp:705-706 */
  uint32_t parts = n_to_read / r->partition_size; /* hpp:707 */
  const uint8_t* cls = *cls_io;
  if ((size_t)(cls_end - cls) < (size_t)nch * parts) return VSYN_ST_BAD_VQ;
  *cls_io = cls + (size_t)nch * parts;
  const uint16_t* ent = *ent_io;
  for (int pass = 0; pass < 8; ++pass) {       /* hpp:711 */
    for (uint32_t pc = 0; pc < parts; ++pc) {  /* hpp:713-759 without the bit reads */
      for (uint32_t j = 0; j < nch; ++j) {
        if (!ch_used[j]) continue;             /* hpp:729 */
        uint32_t vq_class = cls[(size_t)j * parts + pc];
        if (vq_class >= r->num_classifications) return VSYN_ST_BAD_VQ;
        int book = r->books[vq_class * 8 + pass]; /* hpp:731 */
        if (book < 0) continue;
        if ((uint32_t)book >= vq->num_codebooks) return VSYN_ST_BAD_VQ;
        const vsyn_codebook* cb = &vq->codebooks[book];
        if (!cb->lookup || cb->dimensions == 0) return VSYN_ST_BAD_VQ; /* decodeVector on a scalar-only book fails, hpp:743 */
        float* v = out[j];
        uint32_t offset = lim_begin + pc * r->partition_size; /* hpp:735 */
        if (type == 0) { /* 8.6.3, hpp:738-746 */
          uint32_t step = r->partition_size / cb->dimensions;
          for (uint32_t k = 0; k < step; ++k) {
            if (ent >= ent_end || *ent >= cb->num_entries) return VSYN_ST_BAD_VQ;
            const float* t = cb->lookup + (size_t)(*ent++) * cb->dimensions;
            for (uint32_t l = 0; l < cb->dimensions; ++l) {
              if (offset + k + l * step >= len) return VSYN_ST_BAD_VQ;
              v[offset + k + l * step] += t[l];
            }
          }
        } else { /* 8.6.4, hpp:747-754 */
          for (uint32_t k = 0; k < r->partition_size;) {
            if (ent >= ent_end || *ent >= cb->num_entries) return VSYN_ST_BAD_VQ;
            const float* t = cb->lookup + (size_t)(*ent++) * cb->dimensions;
            for (uint32_t l = 0; l < cb->dimensions; ++l, ++k) {
              if (offset + k >= len) return VSYN_ST_BAD_VQ;
              v[offset + k] += t[l];
            }
          }
        }
      }
    }
  }
  *ent_io = ent;
  return 0;
}

int orc_residue_vq(const vsyn_vq_setup* vq, uint32_t mapping, uint32_t channels, uint32_t n2, uint32_t used_mask,
                   const uint8_t* cls, size_t num_cls, const uint16_t* entries, size_t num_entries, float* out) {
  if (mapping >= vq->num_mappings || channels == 0 || channels > VSYN_MAX_CHANNELS) return VSYN_ST_BAD_VQ;
  const vsyn_vq_mapping* mp = &vq->mappings[mapping];
  const uint8_t* cls_end = cls + num_cls;
  const uint16_t* ent = entries;
  const uint16_t* ent_end = entries + num_entries;
  memset(out, 0, sizeof(float) * (size_t)channels * n2); /* residue_outputs start at zero, hpp:1186-1190 */
  for (uint32_t s = 0; s < mp->num_submaps; ++s) {        /* hpp:1184-1209 */
    float* outs[VSYN_MAX_CHANNELS];
    uint8_t used[VSYN_MAX_CHANNELS];
    uint32_t chan[VSYN_MAX_CHANNELS];
    uint32_t nch = 0;
    for (uint32_t ch = 0; ch < channels; ++ch)
      if (mp->mux[ch] == s) {
        chan[nch] = ch;
        outs[nch] = out + (size_t)ch * n2;
        used[nch] = (uint8_t)((used_mask >> ch) & 1u);
        ++nch;
      }
    if (nch == 0) continue;
    if (mp->submap_residue[s] >= vq->num_residues) return VSYN_ST_BAD_VQ;
    const vsyn_residue* r = &vq->residues[mp->submap_residue[s]];
    int rc;
    if (r->type == 2) { /* hpp:685-694: one interleaved vector decoded as format 1, always, then de-interleaved */
      float* tmp = (float*)calloc((size_t)nch * n2, sizeof(float));
      if (!tmp) return VSYN_ST_BAD_VQ;
      uint8_t one = 1;
      float* tv[1] = {tmp};
      rc = orc_residue_vq_one(vq, r, 1, 1, &one, nch * n2, &cls, cls_end, &ent, ent_end, tv);
      if (rc == 0)
        for (uint32_t j = 0; j < nch; ++j)
          for (uint32_t i = 0; i < n2; ++i) outs[j][i] = tmp[j + (size_t)nch * i];
      free(tmp);
    } else {
      rc = orc_residue_vq_one(vq, r, (int)r->type, nch, used, n2, &cls, cls_end, &ent, ent_end, outs);
    }
    if (rc) return rc;
    (void)chan;
  }
  if (ent != ent_end) return VSYN_ST_BAD_VQ; /* the packet's entry count must match what its classifications call for */
  return 0;
}


/* PCM post-stage: vorbis_vorbisfile.c:2022-2032 (host-endian signed 16-bit branch) / float pass-through */
#include <fenv.h>
void orc_pcm_interleave(int format, uint32_t channels, uint32_t frames, const float* planar, uint64_t plane_stride, void* out) {
  for (uint32_t i = 0; i < channels; ++i) {
    const float* src = planar + (size_t)i * plane_stride;
    for (uint32_t j = 0; j < frames; ++j) {
      if (format == VSYN_PCM_S16) {
        volatile float prod = src[j] * 32768.f;   /* the product is rounded to f32 first (vorbis_ftoi takes a double of it) */
        long val = lrint((double)prod);           /* cvtsd2si: current rounding mode = to nearest even */
        if (val > 32767) val = 32767;
        else if (val < -32768) val = -32768;
        ((int16_t*)out)[(size_t)j * channels + i] = (int16_t)val;
      } else {
        ((float*)out)[(size_t)j * channels + i] = src[j];
      }
    }
  }
}
