/*
 * vorbis_synth_oracle.h — CPU ORACLE for the spectral-synthesis hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm (albertz/ParseOggVorbis, citations per function in
 * the .c file).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (parseoggvorbis_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  oracle/Makefile builds the reference's own sources (where they lie under
 * /root/reference) into oracle/_ref/; tests/test_oracle_vs_ref.py checks this restatement bit-for-bit
 * against that build (mdct_backward, window tables, render_point/render_line/neighbours, the overlap-add
 * decode state) and tests/test_oracle_golden.py checks it against committed dumps of the reference
 * decoder run on the reference's two .ogg fixtures (tests/golden/, made by oracle/make_golden.py).
 */
#ifndef VORBIS_SYNTH_ORACLE_H_
#define VORBIS_SYNTH_ORACLE_H_

#include "../include/vorbis_synth_hip.h" /* POD batch structs only; no product code is linked */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- building blocks (each cites the reference lines it follows in the .c) ---- */
int orc_low_neighbor(const uint32_t* v, int idx);  /* -1 if none */
int orc_high_neighbor(const uint32_t* v, int idx); /* -1 if none */
uint32_t orc_render_point(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t X);
void orc_render_line(size_t x0, uint32_t y0, size_t x1, uint32_t y1, uint32_t* vec, size_t len);
const float* orc_inverse_db_table(void); /* 256 floats */

/* floor-1 synthesis tail: coded ys -> final_ys/step2 flags -> integer curve[n] -> out[n] floats.
 * returns 0, or VSYN_ST_FLOOR_RANGE / VSYN_ST_FLOOR_VALUE. Any of final_ys/flags/curve may be NULL. */
int orc_floor1_synth(const uint32_t* xs, int posts, int multiplier, const uint32_t* ys, size_t n,
                     float* out, uint32_t* final_ys, uint8_t* flags, uint32_t* curve);

void orc_inverse_coupling(float* mag, float* ang, size_t len);

typedef struct orc_mdct orc_mdct;
orc_mdct* orc_mdct_new(int n);
void orc_mdct_free(orc_mdct* m);
void orc_mdct_backward(const orc_mdct* m, const float* in, float* out); /* in[n/2] -> out[n] */
const float* orc_mdct_trig(const orc_mdct* m);   /* n + n/4 floats */
const int* orc_mdct_bitrev(const orc_mdct* m);   /* n/4 ints */
/* analytic IMDCT, double accumulation: out[i] = sum_k in[k] cos(2pi/n (i+1/2+n/4)(k+1/2)) */
void orc_imdct_closed_form(int n, const float* in, double* out);

void orc_window(int blocksize0, int blocksize1, int block_flag, int prev, int next, float* out /* [blocksize] */);

/* ---- whole path, same batch contract as vsyn_submit_host ---- */
typedef struct orc_handle orc_handle;
orc_handle* orc_create(const vsyn_setup* setup, uint32_t max_streams);
void orc_destroy(orc_handle* h);
uint32_t orc_ys_stride(const orc_handle* h);
void orc_reset_streams(orc_handle* h);
int orc_submit(orc_handle* h,
               uint32_t num_packets, const vsyn_packet* packets,
               uint32_t num_segments, const vsyn_segment* segments,
               const uint16_t* ys, const float* residue,
               float* pcm, uint64_t plane_stride,
               uint32_t* emit_len, const vsyn_taps* taps, vsyn_status* status);

/* ---- residue VQ stage: the accumulate half of VorbisResidue::decode ----
 * One packet: rebuilds "after_residue" (out: channels x n2 floats, channel-major) from the classifications and codebook
 * entry numbers the bit-serial half left behind (layout: include/vorbis_synth_hip.h, vsyn_vq_packet).
 * used_mask: floor_output_used after the nonzero propagate (hpp:1174-1180). Returns 0, or VSYN_ST_BAD_VQ when an entry /
 * classification is out of range or the entry count does not match the classifications. */
int orc_residue_vq(const vsyn_vq_setup* vq, uint32_t mapping, uint32_t channels, uint32_t n2, uint32_t used_mask,
                   const uint8_t* cls, size_t num_cls, const uint16_t* entries, size_t num_entries, float* out);

/* ---- PCM post-stage: planar f32 -> interleaved int16 / f32 ----
 * PARITY UNPINNED by execution: the rule is ov_read's (reference tree tests/libvorbis-standalone/vorbis_vorbisfile.c:2026-2029,
 * vorbis_ftoi of os.h:156-158: cvtsd2si of the f32 product x * 32768.f, round to nearest even; clamp to [-32768, 32767]), restated
 * from the text; that file cannot be compiled here (ogg/config_types.h is not vendored) and no fixture holds s16 output. */
void orc_pcm_interleave(int format, uint32_t channels, uint32_t frames, const float* planar, uint64_t plane_stride, void* out);

/* IMDCT only, for BASELINE config 2 and the cpu_baseline leg */
void orc_imdct_batch(int n, uint32_t count, const float* in, float* out);

#ifdef __cplusplus
}
#endif
#endif
