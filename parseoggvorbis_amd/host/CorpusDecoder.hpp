// CorpusDecoder.hpp — many Ogg Vorbis files, many host threads, one GPU  (BASELINE config 5 per GPU).
//
// The reference decodes one file on one thread, packet by packet (src/main.cpp:53-67 -> OggReader::full_read,
// src/ParseOggVorbis.hpp:1400-1409).  At corpus scale the sequential half (Ogg paging, Huffman / VQ entropy decode,
// hpp:1139-1211) is what bounds throughput, and it is independent per file.  So:
//
//   worker threads (T)   each takes the next file, runs the entropy half of the whole file (OggReader with a SynthSink that
//                        collects the PacketBatch instead of touching the GPU)
//   feeder threads (F)   each groups finished files that share a synthesis setup, packs up to `files_per_submit` of them into
//                        ONE C-ABI batch (one segment + one stream slot per file, VSYN_SEG_RESET each) in page-locked buffers
//                        and runs it on its own vsyn_handle; then delivers the PCM per file.  Handles own their HIP stream,
//                        so one feeder's PCIe copies overlap another's kernels and copies in the other direction.
//
// PCM delivery: CorpusCallbacks::gotFilePcm is called on a feeder thread (never two calls at the same time), once per file,
// in no particular file order, with planar channel views that are valid during the call only (the same lifetime rule as
// ParseCallbacks::gotPcmData).  Debug hooks (Callbacks.h) are not
// replayed on this path; use OggReader for that.
#ifndef PARSEOGGVORBIS_AMD_HOST_CORPUSDECODER_HPP_
#define PARSEOGGVORBIS_AMD_HOST_CORPUSDECODER_HPP_

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "ParseOggVorbis.hpp"

struct CorpusItem {  // one Ogg Vorbis file, in memory (the caller keeps it alive during decode_corpus)
  const uint8_t* data;
  size_t len;
};

struct CorpusFileResult {
  OkOrError status;
  uint32_t channels = 0, sample_rate = 0;
  uint32_t audio_packets = 0;
  uint64_t frames = 0;   // PCM frames (samples per channel) produced
  double abs_sum = 0;    // sum |x| over all channels, in double: a cheap content check that does not need the PCM kept
};

struct CorpusCallbacks {
  virtual ~CorpusCallbacks() {}
  // file_index = index into the items array. Return false to abort the whole run.
  virtual bool gotFilePcm(size_t file_index, const VorbisIdHeader& header, const std::vector<DataRange<const float>>& channelPcms) {
    (void)file_index; (void)header; (void)channelPcms;
    return true;
  }
  // CorpusOptions::pcm_s16: the file's PCM as interleaved host-endian int16 frames (ov_read's conversion, done on the device: half
  // the bytes cross the bus), valid during the call
  virtual bool gotFilePcmS16(size_t file_index, const VorbisIdHeader& header, const int16_t* interleaved, uint64_t frames) {
    (void)file_index; (void)header; (void)interleaved; (void)frames;
    return true;
  }
};

struct CorpusOptions {
  int threads = 0;                  // entropy workers; 0 = std::thread::hardware_concurrency()
  int feeders = 0;                  // GPU feeder threads; 0 = 3
  uint32_t files_per_submit = 64;   // stream slots per GPU submit
  uint32_t max_pending_files = 0;   // entropy-decoded files waiting for the GPU; 0 = 4 * files_per_submit
  int device = 0;                   // HIP ordinal
  bool entropy_only = false;        // diagnostic: run the workers only and count packets (no GPU call, no PCM, frames stay 0)
  bool share_setups = true;         // parse byte-identical setup headers once per run (SetupCache)
  bool checksum = true;             // fill CorpusFileResult::abs_sum (digest computed on the device, vsyn_pcm_abs_sum_host)
  bool pcm_s16 = false;             // deliver interleaved int16 (gotFilePcmS16) instead of planar f32 (SURVEY 8 f-3 on the host path)
};

struct CorpusStats {
  double wall_s = 0;
  double entropy_cpu_s = 0;   // summed over workers
  double gpu_call_s = 0;      // time inside vsyn_submit_host (PCIe both ways + kernels), summed over feeders
  double pack_s = 0;          // feeders: gathering batches into the submit buffers
  double deliver_s = 0;       // feeders: checksums + gotFilePcm
  uint64_t submits = 0, files = 0, audio_packets = 0, frames = 0;
  uint32_t handles = 0;       // distinct synthesis setups seen
  uint64_t setup_parses = 0, setup_reuses = 0;
};

// Decodes every item; results[i] belongs to items[i].  The returned status is an error only if the run itself could not
// proceed (no GPU, callback abort); per-file problems (corrupt file, unsupported stream) are reported in results[i].status
// and do not stop the other files.
OkOrError decode_corpus(const std::vector<CorpusItem>& items, const CorpusOptions& opts, CorpusCallbacks* callbacks,
                        std::vector<CorpusFileResult>& results, CorpusStats* stats);

extern "C" {
// C / ctypes form.  frames_out, abs_sum_out, ok_out: arrays of num_files (any may be NULL).  pcm_out (may be NULL): per file
// either NULL or a buffer of channels * pcm_capacity[i] floats that receives the planar PCM, channel c at c * pcm_capacity[i]
// (a file longer than its capacity is marked failed).  stats_out: 8 doubles {wall_s, entropy_cpu_s, gpu_call_s, pack_s,
// deliver_s, submits, audio_packets, frames} or NULL.
// Returns 0 if the run proceeded (look at ok_out per file), 1 otherwise with *error_out set as for ogg_vorbis_full_read.
int ogg_vorbis_decode_corpus(const uint8_t* const* datas, const size_t* lens, size_t num_files, int threads, int feeders,
                             uint32_t files_per_submit, int device, uint64_t* frames_out, double* abs_sum_out, uint8_t* ok_out, float* const* pcm_out,
                             const uint64_t* pcm_capacity, double* stats_out, const char** error_out);
// int16 output (CorpusOptions::pcm_s16): pcm16_out[i] = NULL or room for pcm_capacity_frames[i] interleaved frames
int ogg_vorbis_decode_corpus_s16(const uint8_t* const* datas, const size_t* lens, size_t num_files, int threads, int feeders,
                                 uint32_t files_per_submit, int device, uint64_t* frames_out, uint8_t* ok_out, int16_t* const* pcm16_out,
                                 const uint64_t* pcm_capacity_frames, double* stats_out, const char** error_out);
}

#endif
