// ParseOggVorbis.hpp — host side of the MI355X-native decoder, mirroring the reference's public C++ API
// (reference: src/ParseOggVorbis.hpp — ParseCallbacks 966-973, OggReader 1385-1485, the setup model 104-964, the
// C wrappers 1488-1494) so that code written against the reference compiles and behaves the same:
//
//   struct MyCallbacks : ParseCallbacks { bool gotPcmData(const std::vector<DataRange<const float>>&) override; ... };
//   OggReader reader(cb);  OkOrError r = reader.full_read("x.ogg");
//
// What differs is WHERE the work happens.  The sequential parts stay on the CPU here (Ogg paging + CRC, header and
// setup parse, per-packet Huffman / VQ entropy decode — reference hpp:1139-1211, 473-518, 670-762).  Everything after
// "after_residue" — floor-1 synthesis, coupling, floor product, IMDCT, window, overlap-add, PCM hand-off (reference
// hpp:521-590, 1213-1271, 1008-1109, src/mdct.cpp) — is batched and executed on the GPU through the C-ABI of
// include/vorbis_synth_hip.h; hooks and gotPcmData are then replayed in packet order.  There is no CPU version of that
// half in this library: without an MI355X the first audio batch fails with the C-ABI's error text.
//
// Independent implementation; names of public types / members follow the reference where user code touches them.
#ifndef PARSEOGGVORBIS_AMD_HOST_PARSEOGGVORBIS_HPP_
#define PARSEOGGVORBIS_AMD_HOST_PARSEOGGVORBIS_HPP_

#include <stdint.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vorbis_synth_hip.h"
#include "Callbacks.h"
#include "Utils.hpp"

enum { HeaderFlag_Continued = 0x1, HeaderFlag_First = 0x2, HeaderFlag_Last = 0x4 };

struct __attribute__((packed)) VorbisIdHeader {  // Vorbis I 4.2.2, 23 bytes after "\x01vorbis"
  uint32_t vorbis_version;
  uint8_t audio_channels;
  uint32_t audio_sample_rate;
  uint32_t bitrate_maximum;
  uint32_t bitrate_nominal;
  uint32_t bitrate_minimum;
  uint8_t blocksizes_exp;
  uint16_t get_blocksize_0() const { return uint16_t(1u << (blocksizes_exp & 0x0f)); }
  uint16_t get_blocksize_1() const { return uint16_t(1u << ((blocksizes_exp & 0xf0) >> 4)); }
  uint8_t framing_flag;
};

struct VorbisCodebook {  // Vorbis I 3.2.1
  uint16_t dimensions_ = 0;
  uint32_t num_entries_ = 0;
  bool ordered_ = false, sparse_ = false;
  uint8_t lookup_type_ = 0;
  double minimum_value_ = 0, delta_value_ = 0;
  uint8_t value_bits_ = 0;
  bool sequence_p_ = false;
  uint32_t num_lookup_values_ = 0;
  std::vector<uint8_t> lengths_;        // per entry, 0 = unused
  std::vector<uint32_t> multiplicands_;
  std::vector<float> lookup_table_;     // [num_entries_][dimensions_]
  // decode structures: 10-bit prefix table + binary tree for longer codes
  struct Node { int32_t child[2]; };    // >= 0: node index; < 0: ~entry
  std::vector<Node> tree_;
  std::vector<uint32_t> fast_;          // (entry << 8) | length, or 0 if longer than the table

  OkOrError parse(BitReader& reader);
  uint32_t decodeScalar(BitReader& reader) const;
  // `count` code words in a row into dst (entry numbers are < 65536 for VQ use); returns the largest value decodeScalar would
  // have returned (0xffffffff: no such code word)
  uint32_t decodeRun(BitReader& reader, uint32_t count, uint16_t* dst) const;
  const float* decodeVector(BitReader& reader) const;  // dimensions_ floats, or nullptr (no VQ table / bad entry)
};

struct VorbisFloorClass {
  uint8_t dimensions = 0, subclass = 0, masterbook = 0;
  std::vector<int> subclass_books;
};

struct VorbisFloor0 {  // parsed, never decodable (the reference does not implement it either, hpp:402)
  uint8_t order = 0, amplitude_bits = 0, amplitude_offset = 0;
  uint16_t rate = 0, bark_map_size = 0;
  std::vector<uint8_t> books;
  OkOrError parse(BitReader& reader, int max_books);
};

struct VorbisFloor1 {  // Vorbis I 7.2.2
  std::vector<uint8_t> partition_classes;
  std::vector<VorbisFloorClass> classes;
  uint8_t multiplier = 0;
  typedef uint32_t x_t;
  std::vector<x_t> xs;
  OkOrError parse(BitReader& reader, int num_codebooks);
  // entropy half of 7.2.3 only: coded Y values ("floor1 ys"); the curve itself is synthesised on the GPU
  OkOrError decode_ys(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, std::vector<uint32_t>& ys, bool& use_output) const;
};

struct VorbisFloor {
  uint16_t floor_type = 0;
  VorbisFloor0 floor0;
  VorbisFloor1 floor1;
  OkOrError parse(BitReader& reader, int num_codebooks);
};

struct VorbisResidue {  // Vorbis I 8.6
  uint16_t type = 0;
  uint32_t begin = 0, end = 0, partition_size = 0;
  uint8_t num_classifications = 0, classbook = 0;
  std::vector<uint32_t> cascades;
  std::vector<int16_t> books;  // [class][pass], -1 = none
  // decode_entries' view of `books` (built by prepare() once the codebooks are parsed; the pointers go into the setup's shared
  // codebook vector): per (class, pass) the prefix table, how many code words a partition holds and the entry count
  struct Run {
    const uint32_t* fast = nullptr;  // nullptr: no codebook in this pass
    const VorbisCodebook* book = nullptr;
    uint32_t count = 0, num_entries = 0;
  };
  std::vector<Run> runs;             // [class][pass]
  std::vector<uint32_t> cls_unpack;  // class word -> its <= 4 class numbers, 8 bits each, first partition lowest (empty: computed per word)
  void prepare(const std::vector<VorbisCodebook>& codebooks);
  OkOrError parse(BitReader& reader, int num_codebooks);
  // out: num_channel vectors of decode_len floats, zero-initialised by the caller; adds the VQ vectors (8.6.2-8.6.5)
  OkOrError decode(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, uint32_t num_channel, const std::vector<bool>& channel_used,
                   uint32_t decode_len, float* const* out, int type_override = -1) const;
  // bit-serial half only (for the device VQ stage, include/vorbis_synth_hip.h): appends the classification numbers
  // ([channel][partition]) and, in decode order, the codebook entry numbers; no value vector is looked up or added
  OkOrError decode_entries(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, uint32_t num_channel,
                           const std::vector<bool>& channel_used, uint32_t decode_len, std::vector<uint8_t>& cls_out,
                           std::vector<uint16_t>& entries_out, int type_override = -1) const;
};

struct VorbisMapping {
  uint16_t type = 0;
  struct Coupling { int magintude, angle; };  // (sic) spelled as upstream, user code may read it
  std::vector<Coupling> couplings;
  std::vector<uint8_t> muxs;
  struct Submap { uint8_t floor, residue; };
  std::vector<Submap> submaps;
  OkOrError parse(BitReader& reader, int num_channels, int num_floors, int num_residues);
};

struct VorbisModeNumber {
  bool block_flag = false;
  uint16_t window_type = 0, transform_type = 0;
  uint8_t mapping = 0;
  uint16_t blocksize = 0;
  OkOrError parse(BitReader& reader, int num_mappings, const VorbisIdHeader& header);
};

// A vector whose copies share the elements: the parsed codebooks (decode trees, prefix tables, VQ value tables — hundreds of KB)
// never change once VorbisStreamSetup::parse has returned, so a stream that takes its setup from the SetupCache shares them with
// every other stream of that setup instead of copying them (copying cost ~8 % of entropy-decoding a two-second file).
template <typename T>
struct SharedVec {
  std::shared_ptr<std::vector<T>> v = std::make_shared<std::vector<T>>();
  size_t size() const { return v->size(); }
  bool empty() const { return v->empty(); }
  const T& operator[](size_t i) const { return (*v)[i]; }
  T& operator[](size_t i) { return (*v)[i]; }
  void resize(size_t n) { v = std::make_shared<std::vector<T>>(n); }  // (a fresh vector: other holders keep theirs)
  typename std::vector<T>::iterator begin() { return v->begin(); }
  typename std::vector<T>::iterator end() { return v->end(); }
  typename std::vector<T>::const_iterator begin() const { return v->begin(); }
  typename std::vector<T>::const_iterator end() const { return v->end(); }
  operator const std::vector<T>&() const { return *v; }
};

struct VorbisStreamSetup {
  SharedVec<VorbisCodebook> codebooks;
  std::vector<VorbisFloor> floors;
  std::vector<VorbisResidue> residues;
  std::vector<VorbisMapping> mappings;
  std::vector<VorbisModeNumber> modes;
  OkOrError parse(BitReader& reader, const VorbisIdHeader& header);
};

struct ParseCallbacks {  // returning false stops the read with a check failure, as upstream
  virtual ~ParseCallbacks() {}
  virtual bool gotHeader(const VorbisIdHeader& header) { (void)header; return true; }
  virtual bool gotComments(const std::string& vendor, const std::vector<std::string> comments) { (void)vendor; (void)comments; return true; }
  virtual bool gotSetup(const VorbisStreamSetup& setup) { (void)setup; return true; }
  virtual bool gotPcmData(const std::vector<DataRange<const float>>& channelPcms) { (void)channelPcms; return true; }  // planar, valid during the call only
  virtual bool gotEof() { return true; }
};

// What the entropy half leaves behind for a run of consecutive audio packets of one stream: exactly the tensors of the
// C-ABI batch (include/vorbis_synth_hip.h).
struct PacketBatch {
  std::vector<vsyn_packet> pk;
  std::vector<uint16_t> ys;            // [packet][channel][ys_stride]
  std::vector<float> residue;          // packed [packet][channel][n/2]  ("after_residue"); empty in VQ mode
  std::vector<uint8_t> floor_number;   // [packet][channel], for the "floor_number" hook
  // VQ mode (the device rebuilds "after_residue" from these):
  bool vq = false;
  std::vector<vsyn_vq_packet> vq_pk;
  std::vector<uint8_t> cls;
  std::vector<uint16_t> entries;
  size_t residue_floats = 0;           // what `residue` would hold
  bool first = true;                   // first batch of its stream (no overlap carry-in)
};

// Parsed setup headers, shared between streams (optional; the corpus decoder installs one per run). Files produced by
// the same encoder settings carry byte-identical setup packets (38 codebooks for the fixtures: building their decode
// trees and VQ tables costs more than entropy-decoding two seconds of audio), so a stream whose setup packet is already
// known copies the parsed model instead of parsing again. Keyed by a hash of the packet bytes, confirmed by comparing
// the bytes and the id-header fields the parse depends on. Thread safe.
struct SetupCache {
  struct Entry {
    std::vector<uint8_t> bytes;
    uint32_t channels, blocksizes;
    std::shared_ptr<const VorbisStreamSetup> setup;
  };
  std::mutex mu;
  std::multimap<uint64_t, Entry> entries;
  uint64_t hits = 0, misses = 0;
  std::shared_ptr<const VorbisStreamSetup> find(uint64_t hash, const uint8_t* data, uint32_t len, const VorbisIdHeader& h);
  void insert(uint64_t hash, const uint8_t* data, uint32_t len, const VorbisIdHeader& h, const VorbisStreamSetup& parsed);
};

struct VorbisStream;
// Where finished batches go. Default (nullptr): the stream's own GPU handle, synchronously, followed by the hook /
// gotPcmData replay. The corpus decoder installs a collector instead and submits many files' batches in one GPU call.
struct SynthSink {
  virtual ~SynthSink() {}
  virtual OkOrError consume(VorbisStream& stream, PacketBatch&& batch) = 0;
  // called once when a stream is created: the sink may swap pre-reserved vectors into the stream's batch state
  virtual void prepare(VorbisStream& stream) { (void)stream; }
};

// One logical Vorbis stream: headers, the per-stream GPU handle and the batch of entropy-decoded packets waiting for it.
struct VorbisStream {
  VorbisIdHeader header;
  VorbisStreamSetup setup;
  uint32_t packet_counts_ = 0, audio_packet_counts_ = 0;

  VorbisStream();
  ~VorbisStream();
  VorbisStream(const VorbisStream&) = delete;
  VorbisStream& operator=(const VorbisStream&) = delete;

  OkOrError parse_id(const uint8_t* data, uint32_t len, ParseCallbacks& cb);
  OkOrError parse_comment(const uint8_t* data, uint32_t len, ParseCallbacks& cb);
  OkOrError parse_setup(const uint8_t* data, uint32_t len, ParseCallbacks& cb);
  OkOrError parse_audio(const uint8_t* data, uint32_t len, int64_t page_granule_or_minus1, ParseCallbacks& cb);
  OkOrError flush(ParseCallbacks& cb);  // run the pending batch on the GPU and replay hooks + gotPcmData in packet order

  SetupCache* setup_cache_ = nullptr;  // optional, see SetupCache

  // --- batch state ---
  SynthSink* sink_ = nullptr;
  vsyn_handle* synth_ = nullptr;
  uint32_t ys_stride_ = 0, batch_limit_ = 2048;
  bool first_batch_ = true;
  uint64_t abs_total_pos_ = 0;  // samples handed out so far (hook "abs_total_pos")
  std::vector<vsyn_packet> pk_;
  std::vector<uint16_t> ys_;
  std::vector<float> residue_;
  std::vector<uint8_t> floor_number_;  // [packet][channel], for the "floor_number" hook
  // VQ mode: the residue leaves the host as classification + entry numbers (decided once per stream in parse_setup:
  // every VQ book <= 65536 entries, vector lengths dividing the partition sizes; PARSEOGGVORBIS_VQ=0 turns it off)
  bool vq_mode_ = false;
  uint64_t setup_hash_ = 0;  // FNV-1a of the setup header packet: identifies the codebooks when streams share a handle
  std::vector<vsyn_vq_packet> vq_pk_;
  std::vector<uint8_t> cls_;
  std::vector<uint16_t> entries_;
  size_t residue_floats_ = 0;
  uint64_t packets_seen_ = 0;          // audio packets parsed so far (only counted under PARSEOGGVORBIS_TEST_FAIL_AT)
};

struct OggReader {
  std::map<uint32_t, std::unique_ptr<VorbisStream>> streams_;
  size_t packet_counts_;
  std::shared_ptr<IReader> reader_;
  ParseCallbacks& callbacks_;

  SynthSink* sink_ = nullptr;           // optional: where streams hand their batches (see SynthSink)
  SetupCache* setup_cache_ = nullptr;   // optional: parsed setup headers shared between readers (see SetupCache)
  uint32_t batch_limit_override_ = 0;   // optional: audio packets per batch (0: default / PARSEOGGVORBIS_BATCH)

  explicit OggReader(ParseCallbacks& callbacks) : packet_counts_(0), callbacks_(callbacks) {}
  OkOrError open_file(const std::string& filename);
  OkOrError set_reader(const std::shared_ptr<IReader>& reader);
  OkOrError read_next_page(bool& reached_eof);
  OkOrError read_until_end();
  OkOrError full_read(const std::string& filename);
  OkOrError full_read_from_memory(const uint8_t* data, size_t data_len);
};

// The synthesis-relevant part of a stream's setup in C-ABI form (keeps the arrays the vsyn_setup points into alive), and
// a byte string that is equal for two streams iff they can share one vsyn_handle.
struct SynthSetup {
  std::vector<vsyn_floor1> floors;
  std::vector<std::vector<uint32_t>> xs;
  std::vector<std::vector<vsyn_coupling>> coup;
  std::vector<std::vector<uint8_t>> chfloor;
  std::vector<vsyn_mapping> maps;
  std::vector<vsyn_mode> modes;
  vsyn_setup su;
  std::string key;
  // residue VQ stage (filled when the stream can use it; `key` then also covers the codebook tables)
  bool has_vq = false;
  std::vector<vsyn_codebook> books;
  std::vector<std::vector<float>> book_tables;  // owned copies: the setup must be usable after the stream is gone
  std::vector<vsyn_residue> residues;
  std::vector<std::vector<int16_t>> residue_books;
  std::vector<vsyn_vq_mapping> vq_maps;
  std::vector<std::vector<uint8_t>> mux, submap_residue;
  vsyn_vq_setup vq;
};
OkOrError build_synth_setup(const VorbisStream& st, SynthSetup& out);
bool stream_can_use_vq(const VorbisStream& st);  // the conditions of vsyn_attach_vq, checked on the host model

extern "C" {
// 0 on success; on failure 1 and *error_out (if non-NULL) points at a static, NUL-terminated 255-byte buffer
// (reference: src/ParseOggVorbis.cpp:12-42 — same contract, same lack of thread safety of that buffer)
int ogg_vorbis_full_read(const char* filename, const char** error_out);
int ogg_vorbis_full_read_from_memory(const char* data, size_t data_len, const char** error_out);
}

#endif
