/*
 * vorbis_synth_hip.h — C-ABI of the MI355X-native batched Vorbis spectral-synthesis path.
 *
 * This is the drop-in boundary for ONE hot path of albertz/ParseOggVorbis: everything the reference
 * does per audio packet AFTER the entropy decode, i.e.
 *
 *   floor-1 amplitude unwrap + curve render   src/ParseOggVorbis.hpp:521-590, src/Utils.hpp:58-183
 *   nonzero-vector propagate                  src/ParseOggVorbis.hpp:1174-1180
 *   inverse channel coupling                  src/ParseOggVorbis.hpp:1213-1241
 *   floor x residue ("dot product")           src/ParseOggVorbis.hpp:1243-1255
 *   inverse MDCT                              src/mdct.cpp:433-527 (tables: src/mdct.cpp:88-127)
 *   window + overlap-add + PCM hand-off       src/ParseOggVorbis.hpp:837-886, 1008-1109
 *
 * The reference runs that per packet inside VorbisStream::parse_audio (hpp:1128-1274); a host decoder
 * that keeps the sequential Ogg/Huffman parse on the CPU calls vsyn_submit_* once per BATCH instead
 * (between hpp:1211 "after_residue" and hpp:1213), then replays hooks / ParseCallbacks::gotPcmData in
 * packet order.  Plain C: POD structs, raw pointers and sizes, int status + const char** error, no
 * C++/torch types, no exceptions across the boundary.  All compute is HIP on gfx950; there is no CPU
 * fallback — every entry point fails with VSYN_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef VORBIS_SYNTH_HIP_H_
#define VORBIS_SYNTH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSYN_ABI_VERSION 5 /* 2: + residue VQ stage (vsyn_attach_vq, vsyn_vq_batch), page-locked host buffers; 3: + vsyn_pcm_abs_sum_host,
                              vsyn_pcm_fetch_host, VSYN_SUBMIT_KEEP_PCM (additive; the feature taps no longer force the staged kernels);
                              4: + vsyn_fused_paths (additive); 5: + VSYN_SUBMIT_PRE_KERNELS (additive) */

#define VSYN_MAX_CHANNELS 32 /* floor_used is a 32-bit mask (reference: uint8_t audio_channels) */
#define VSYN_MAX_POSTS 65    /* Vorbis I: 2 + 31 partitions x <=8 dims, capped at 65 by the spec */
#define VSYN_MIN_BLOCKSIZE 64
#define VSYN_MAX_BLOCKSIZE 8192 /* hpp:1294-1296 */

/* status codes (return value of every int function; 0 = ok, like ogg_vorbis_full_read, ParseOggVorbis.cpp:12-42) */
enum {
  VSYN_OK = 0,
  VSYN_ERR_INVALID = 1,   /* bad argument / bad setup (the reference's CHECK(...) on setup fields) */
  VSYN_ERR_NO_DEVICE = 2, /* no HIP device, or not gfx950-compatible code object */
  VSYN_ERR_HIP = 3,       /* a HIP runtime call failed; text in *err */
  VSYN_ERR_STREAM = 4     /* the batch itself is bad (see vsyn_status.flags) */
};

/* vsyn_status.flags — conditions on which the reference fails a CHECK and aborts the read */
enum {
  VSYN_ST_FLOOR_RANGE = 1u << 0,   /* predicted > range, hpp:536 */
  VSYN_ST_FLOOR_VALUE = 1u << 1,   /* rendered floor value >= 256, hpp:587 */
  VSYN_ST_GRANULE = 1u << 2,       /* page granule behind/ahead of what the packets provide, hpp:1029,1041 */
  VSYN_ST_PLANE_OVERFLOW = 1u << 3,/* a segment emits more than plane_stride samples (nothing is written out of bounds) */
  VSYN_ST_BAD_MODE = 1u << 4,      /* mode index >= num_modes */
  VSYN_ST_BAD_SEGMENT = 1u << 5,   /* segment out of range / unknown stream slot / unaligned residue_off */
  VSYN_ST_BAD_VQ = 1u << 6         /* VQ stage: entry or classification number out of range, or entry count inconsistent with the classifications */
};

/* ---- stream setup: the part of VorbisStreamSetup (hpp:889-964) the synthesis half reads ---- */

typedef struct vsyn_floor1 {          /* VorbisFloor1, hpp:416-471 */
  uint32_t multiplier;                /* 1..4 (hpp:446) */
  uint32_t num_posts;                 /* xs.size(), 2..65 */
  const uint32_t* xs;                 /* header order: xs[0]=0, xs[1]=1<<rangebits, then partition posts (hpp:448-456) */
} vsyn_floor1;

typedef struct vsyn_coupling {        /* VorbisMapping::Coupling, hpp:767 */
  uint16_t magnitude, angle;
} vsyn_coupling;

typedef struct vsyn_mapping {         /* VorbisMapping, hpp:765-814 */
  uint32_t num_couplings;
  const vsyn_coupling* couplings;     /* in header order; applied in reverse (hpp:1214) */
  const uint8_t* channel_floor;       /* [channels]: submaps[muxs[ch]].floor (hpp:1162-1163) */
} vsyn_mapping;

typedef struct vsyn_mode {            /* VorbisModeNumber, hpp:816-835 */
  uint8_t block_flag;                 /* 1 = long window (blocksize1) */
  uint8_t mapping;
} vsyn_mode;

typedef struct vsyn_setup {
  uint32_t channels;                  /* VorbisIdHeader.audio_channels, 1..VSYN_MAX_CHANNELS */
  uint32_t blocksize0, blocksize1;    /* powers of two, 64..8192, blocksize0 <= blocksize1 (hpp:1294-1298) */
  uint32_t num_floors;   const vsyn_floor1* floors;   /* all type 1 (type 0 is unimplemented upstream, hpp:402) */
  uint32_t num_mappings; const vsyn_mapping* mappings;
  uint32_t num_modes;    const vsyn_mode* modes;
} vsyn_setup;

/* ---- a batch ---- */

typedef struct vsyn_packet {          /* what hpp:1142-1172 leaves behind for one audio packet; 16 bytes */
  uint8_t mode;                       /* mode_idx, hpp:1146 */
  uint8_t prev_long, next_long;       /* prev/next window flags, hpp:1151-1152 (ignored for short blocks) */
  uint8_t reserved0;
  uint32_t floor_used;                /* bit c = VorbisFloor1::decode set use_output for channel c (hpp:478-482) */
  int64_t granule;                    /* setExpectedEndingPos(): page granule if last packet on its page, else -1 (hpp:1456-1459) */
} vsyn_packet;

#define VSYN_SEG_RESET 1u             /* segment starts a stream: no overlap carry-in, first packet emits nothing (hpp:1021) */

typedef struct vsyn_segment {         /* consecutive packets of one stream inside a batch; 24 bytes */
  uint32_t stream;                    /* stream slot, < max_streams; carries overlap state between submits */
  uint32_t first_packet;              /* index into packets[] / ys rows */
  uint32_t num_packets;
  uint32_t flags;                     /* VSYN_SEG_* */
  uint64_t residue_off;               /* float index of this segment's first residue block (multiple of 4) */
} vsyn_segment;

/*
 * Batch tensors (host or device resident depending on the entry point):
 *   packets  [P]                      vsyn_packet
 *   segments [S]                      vsyn_segment; segments must not overlap, a stream slot at most once per batch
 *   ys       [P][channels][ys_stride] uint16  coded floor-1 Y values ("floor1 ys", hpp:518); row ignored if !floor_used
 *   residue  packed float32: packet p of a segment, channel c, bin i at
 *              seg.residue_off + (sum of channels*n_q/2 over earlier packets q of the segment) + c*n_p/2 + i
 *            = "after_residue" (hpp:1211), length n_p/2 per channel
 *   pcm      [S][channels][plane_stride] float32, planar; segment g writes its emitted samples from offset 0
 *   emit_len [P] uint32 (optional)    samples per channel emitted for packet p = num_frames of forwardReadyPcm (hpp:1019-1059)
 */
typedef struct vsyn_taps {            /* optional debug taps = the reference's push_data_* hooks on this path */
  float* after_envelope;              /* same packing as residue; "after_envelope", hpp:1254 */
  float* pcm_after_mdct;              /* packed like residue with n_p per channel (2x offsets); "pcm_after_mdct", hpp:1265 */
  uint16_t* floor_final;              /* [P][channels][ys_stride]: final_y*multiplier | step2_flag<<15 ("floor1 final_ys"/"step2_flag", hpp:560-561) */
  uint16_t* floor_curve;              /* packed like residue (n_p/2 per channel): the rendered integer floor curve, "floor1 floor"
                                         (hpp:585; the reference's vector has n_p entries: the second half is flat at the last flagged post's y, hpp:583-584);
                                         with after_residue these are the feature tensors of returnn_import.py:74-115.
                                         Rows of channels without a decoded floor are left untouched */
} vsyn_taps;
/* after_envelope / pcm_after_mdct exist only in the staged (any-shape) kernels: asking for either routes the batch through them.
 * The two feature taps do not: floor_final comes from the unwrap kernel, floor_curve from the tap variant of the fused kernel. */

typedef struct vsyn_status {
  uint32_t flags;                     /* VSYN_ST_* OR-ed over the batch */
  uint32_t first_bad_packet;          /* lowest packet index that raised a flag (0xFFFFFFFF if none) */
} vsyn_status;

typedef struct vsyn_handle vsyn_handle;

/* submit flags */
#define VSYN_SUBMIT_STAGED 1u         /* force the staged (tap-capable, any-shape) kernels instead of the fused one */
#define VSYN_SUBMIT_INPUTS_READY 2u   /* vsyn_submit_device: packets/segments/ys are complete already (not produced by work still
                                         pending on hip_stream). Where a submit's preparation consists of the chained layout + floor-unwrap
                                         kernels (staged work, the VQ stage, very long segments, VSYN_SUBMIT_PRE_KERNELS), this lets them
                                         overlap the synthesis kernel of the previous submit; results are identical either way. */

#define VSYN_SUBMIT_PRE_KERNELS 8u    /* diagnostics / A-B: prepare the batch (layout scan, floor-1 step 1) with the two chained kernels also
                                         where the single dependency-free preparation kernel is the default (submits whose runs are all
                                         taken by the fused kernels); with VSYN_SUBMIT_INPUTS_READY they run hidden beside the previous
                                         submit's synthesis kernel (the default of rounds 1-3 for such submits); results are identical. */
#define VSYN_SUBMIT_KEEP_PCM 4u        /* vsyn_submit_host*: leave the PCM on the device (`pcm` may be NULL, nothing is copied back);
                                         fetch it in the form the consumer wants with vsyn_pcm_fetch_host */

const char* vsyn_version(void);
int vsyn_abi_version(void);

/* Builds the per-stream constant block (IMDCT twiddles for both blocksizes, the 1+4 window tables of
 * VorbisModeNumber::precalc hpp:837-862, floor-1 sorted posts + neighbour tables, coupling/mode tables,
 * inverse-dB table) on `device` and allocates overlap state for max_streams stream slots. */
int vsyn_create(const vsyn_setup* setup, int device, uint32_t max_streams, vsyn_handle** out, const char** err);
void vsyn_destroy(vsyn_handle* h);

uint32_t vsyn_ys_stride(const vsyn_handle* h);           /* uint16 elements per (packet,channel) row of ys */
uint32_t vsyn_channels(const vsyn_handle* h);
/* Which synthesis kernels this handle's setup gets (diagnostics, tests): bit 0 = the fused kernel takes runs of long blocks,
 * bit 1 = it also takes mixed-block runs and carry-ins; neither = every batch goes through the staged (any-shape) kernels.
 * Bit 8 (after vsyn_attach_vq) = the residue VQ kernel keeps the attached setup's value tables in LDS (they fit) instead of
 * gathering them from global memory. Results are the same either way; only the speed differs. */
uint32_t vsyn_fused_paths(const vsyn_handle* h);
/* size in bytes of the constant block, and a copy of it (for the one RCCL broadcast of a multi-GPU job) */
size_t vsyn_const_block_bytes(const vsyn_handle* h);

/* All pointers are DEVICE pointers on the handle's device; asynchronous on hip_stream (a hipStream_t, NULL = default stream).
 * max_seg_packets >= every segment's num_packets. Errors found on the device are reported by vsyn_sync_status. */
int vsyn_submit_device(vsyn_handle* h,
                       uint32_t num_packets, const vsyn_packet* d_packets,
                       uint32_t num_segments, const vsyn_segment* d_segments, uint32_t max_seg_packets,
                       const uint16_t* d_ys, const float* d_residue,
                       float* d_pcm, uint64_t plane_stride,
                       uint32_t* d_emit_len, const vsyn_taps* d_taps,
                       uint32_t flags, void* hip_stream, const char** err);

/* Same with HOST pointers: stages to the device, runs, copies pcm / emit_len / taps back, synchronises (on a stream owned
 * by the handle: calls on different handles from different host threads overlap),
 * and returns VSYN_ERR_STREAM (status filled) if the device flagged the batch. residue_floats = total floats in residue. */
int vsyn_submit_host(vsyn_handle* h,
                     uint32_t num_packets, const vsyn_packet* packets,
                     uint32_t num_segments, const vsyn_segment* segments,
                     const uint16_t* ys, const float* residue, size_t residue_floats,
                     float* pcm, uint64_t plane_stride,
                     uint32_t* emit_len, const vsyn_taps* taps,
                     uint32_t flags, vsyn_status* status, const char** err);

/* ---- residue VQ stage: the data-parallel half of the residue decode on the device ----
 *
 * The reference's residue decode (VorbisResidue::decode, hpp:670-762) interleaves a bit-serial part — Huffman decode of
 * one classification word per partition group and one codebook ENTRY NUMBER per vector (VorbisCodebook::decodeScalar,
 * hpp:286-301) — with a data-parallel part: look the entry's value vector up (lookup_table_, hpp:212-245, 367-374) and
 * add it into the residue vector (hpp:737-753), once per cascade pass, then de-interleave format 2 (hpp:687-693).
 * With this stage the host decoder keeps only the bit-serial part and ships per packet the classifications (u8) and
 * entry numbers (u16) instead of the expanded float vectors; the device rebuilds "after_residue" (hpp:1211) in the same
 * order of additions, i.e. bit-identical, and the synthesis kernels continue from there.
 */
typedef struct vsyn_codebook {        /* VQ side of VorbisCodebook, hpp:104-245 */
  uint32_t dimensions;                /* dimensions_ */
  uint32_t num_entries;               /* num_entries_ (<= 65536 for this stage) */
  const float* lookup;                /* lookup_table_: [num_entries][dimensions] value vectors, or NULL (lookup type 0: scalar-only book) */
} vsyn_codebook;

typedef struct vsyn_residue {         /* VorbisResidue, hpp:623-668 */
  uint32_t type;                      /* 0, 1 or 2 */
  uint32_t begin, end, partition_size;
  uint32_t num_classifications;       /* 1..64 */
  uint32_t classwords;                /* dimensions_ of the class codebook (classifications per codeword, hpp:703) */
  const int16_t* books;               /* [num_classifications][8]: codebook per cascade pass, -1 = none (hpp:651-659) */
} vsyn_residue;

typedef struct vsyn_vq_mapping {      /* the residue side of VorbisMapping, hpp:765-814 */
  uint32_t num_submaps;               /* 1..16 */
  const uint8_t* mux;                 /* [channels]: submap of each channel (muxs) */
  const uint8_t* submap_residue;      /* [num_submaps]: residue number of each submap */
} vsyn_vq_mapping;

typedef struct vsyn_vq_setup {
  uint32_t num_codebooks; const vsyn_codebook* codebooks;
  uint32_t num_residues;  const vsyn_residue* residues;
  uint32_t num_mappings;  const vsyn_vq_mapping* mappings;   /* same count and order as vsyn_setup.mappings */
} vsyn_vq_setup;

/* Per packet, in decode order (hpp:708-760). For each submap s = 0.. of the packet's mapping, with the channels whose
 * mux == s in channel order as j = 0..nch-1 (format 2: one virtual channel, always decoded, hpp:685-694):
 *   cls      nch x parts bytes, [j][partition]: the classification numbers (hpp:716-719); parts = (min(end,len) -
 *            min(begin,len)) / partition_size, len = n/2 (format 2: nch*n/2). Rows of unused channels are present, ignored.
 *   entries  for pass 0..7, partition 0..parts-1, j 0..nch-1 (used channels with a codebook in that pass only):
 *            partition_size / dimensions entry numbers (hpp:741, 749) */
typedef struct vsyn_vq_packet {       /* 16 bytes */
  uint64_t entry_off;                 /* index of the packet's first entry in entries[] */
  uint32_t num_entries;
  uint32_t cls_off;                   /* index of the packet's first byte in cls[] */
} vsyn_vq_packet;

typedef struct vsyn_vq_batch {        /* host or device pointers, like the other batch tensors of the call */
  const vsyn_vq_packet* packets;      /* [P] */
  const uint8_t* cls;
  const uint16_t* entries;
  uint64_t num_cls, num_entries;      /* array lengths (bounds for validation / staging) */
} vsyn_vq_batch;

/* Uploads the codebook value tables and residue descriptions. VSYN_ERR_INVALID (with the reason in *err) if the setup
 * is outside what the stage handles — a book with more than 65536 entries, a vector length that does not divide the
 * partition size, more than 8192 (pass, partition, channel) slots per packet; the caller then keeps feeding floats. */
int vsyn_attach_vq(vsyn_handle* h, const vsyn_vq_setup* vq, const char** err);

/* vsyn_submit_device / vsyn_submit_host with the residue given as VQ entries: `residue` becomes an OUTPUT of
 * residue_floats floats (same packing as the input of vsyn_submit_*: it is the "after_residue" tensor; device scratch
 * for vsyn_submit_device_vq, optional — may be NULL — copy-back for vsyn_submit_host_vq). Requires vsyn_attach_vq. */
int vsyn_submit_device_vq(vsyn_handle* h,
                          uint32_t num_packets, const vsyn_packet* d_packets,
                          uint32_t num_segments, const vsyn_segment* d_segments, uint32_t max_seg_packets,
                          const uint16_t* d_ys, const vsyn_vq_batch* d_vq, float* d_residue,
                          float* d_pcm, uint64_t plane_stride,
                          uint32_t* d_emit_len, const vsyn_taps* d_taps,
                          uint32_t flags, void* hip_stream, const char** err);
int vsyn_submit_host_vq(vsyn_handle* h,
                        uint32_t num_packets, const vsyn_packet* packets,
                        uint32_t num_segments, const vsyn_segment* segments,
                        const uint16_t* ys, const vsyn_vq_batch* vq, float* residue_out, size_t residue_floats,
                        float* pcm, uint64_t plane_stride,
                        uint32_t* emit_len, const vsyn_taps* taps,
                        uint32_t flags, vsyn_status* status, const char** err);

/* ---- PCM post-stage: planar f32 -> what the consumer of gotPcmData does next ----
 * Converts the PCM of the MOST RECENT vsyn_submit_device* call on this handle (stream-ordered: pass the same hip_stream)
 * from planar [S][channels][plane_stride] to interleaved frames [S][out_stride_frames][channels]; segment g gets its
 * total emitted frames (the sum of its emit_len), also written to d_frames[g] if d_frames != NULL.
 *   VSYN_PCM_S16  int16, host endian, val = round-to-nearest-even(x * 32768.f) clamped to [-32768, 32767] — ov_read's
 *                 conversion (reference tree: tests/libvorbis-standalone/vorbis_vorbisfile.c:2026-2029 with vorbis_ftoi of
 *                 os.h:156-158); halves the output bytes
 *   VSYN_PCM_F32  float32, values unchanged */
#define VSYN_PCM_S16 1
#define VSYN_PCM_F32 2
int vsyn_pcm_interleave_device(vsyn_handle* h, int format, const float* d_pcm, uint64_t plane_stride,
                               void* d_out, uint64_t out_stride_frames, uint32_t* d_frames,
                               void* hip_stream, const char** err);

/* Per-(segment, channel) digest of the PCM of the MOST RECENT vsyn_submit_host / vsyn_submit_host_vq call on this handle:
 * out[g * channels + c] = sum of |x| over segment g's emitted frames of channel c, accumulated in double in a fixed order (the
 * same PCM always gives the same bits). Computed on the device from the PCM still resident there, so that a corpus decoder
 * can tell that replicas agree — the reference's harness compares decoders sample by sample, compare-debug-out.py:524-542 —
 * without another pass over the PCM on the host. `out` holds S * channels doubles of the last submit. Synchronous. */
int vsyn_pcm_abs_sum_host(vsyn_handle* h, double* out, const char** err);

/* The PCM of the MOST RECENT vsyn_submit_host* call on this handle, converted on the device and copied to the host in the
 * interleaved form of vsyn_pcm_interleave_device (VSYN_PCM_S16 / VSYN_PCM_F32): out[g][frame][channel], out_stride_frames
 * frames per segment (frames past a segment's end are zero); frames_out[g] (optional) = emitted frames of segment g. With VSYN_SUBMIT_KEEP_PCM on the submit, int16
 * output halves the bytes that cross the bus. Synchronous. */
int vsyn_pcm_fetch_host(vsyn_handle* h, int format, void* out, uint64_t out_stride_frames, uint32_t* frames_out, const char** err);

/* Page-locked host memory for the buffers handed to vsyn_submit_host (direct DMA instead of the runtime's staging copies;
 * what a host decoder that batches at corpus scale wants). Pageable memory is accepted by vsyn_submit_host as well. */
int vsyn_host_alloc(size_t bytes, void** out, const char** err);
void vsyn_host_free(void* p);

/* Waits for hip_stream and returns the accumulated device status since the last call (then clears it). */
int vsyn_sync_status(vsyn_handle* h, void* hip_stream, vsyn_status* status, const char** err);

/* Forget all overlap state (every stream slot behaves as VSYN_SEG_RESET on its next segment). */
int vsyn_reset_streams(vsyn_handle* h, void* hip_stream, const char** err);

/* Kernel timing for roofline reporting: when enabled, vsyn_submit_device brackets its dominant kernel with
 * hipEvents on hip_stream; vsyn_profile_read synchronises and returns the mean duration since the last read. */
int vsyn_profile_enable(vsyn_handle* h, int on); /* 0 off, 1 or 2 time the fused synthesis kernel (one kernel serves steady and mixed-block runs), 3 the residue VQ kernel */
int vsyn_profile_read(vsyn_handle* h, double* mean_ms, uint32_t* launches, const char** kernel_name);

/* IMDCT-only entry (BASELINE config 2): in [count][n/2] -> out [count][n], device pointers, n = blocksize0 or blocksize1. */
int vsyn_imdct_device(vsyn_handle* h, uint32_t n, uint32_t count, const float* d_in, float* d_out,
                      void* hip_stream, const char** err);

#ifdef __cplusplus
}
#endif
#endif /* VORBIS_SYNTH_HIP_H_ */
