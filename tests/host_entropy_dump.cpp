// tests/host_entropy_dump.cpp — probe of the host decoder's ENTROPY half (no GPU involved, works with or without one):
// parses an .ogg with the product's OggReader, collects through a SynthSink what the entropy half queues for the GPU —
// mode/flags/granule, coded floor posts and "after_residue" (PARSEOGGVORBIS_VQ=0) or, in VQ mode (default), the
// classification and entry numbers plus the VQ setup — and dumps it, so that tests can compare it with the reference
// decoder's hooks in tests/golden/ and feed the same tensors to the oracle / the GPU.
// Usage: host_entropy_dump in.ogg out.bin
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../parseoggvorbis_amd/host/ParseOggVorbis.hpp"

namespace {
struct Collect : SynthSink {
  PacketBatch batch;
  SynthSetup setup;
  VorbisIdHeader header;
  uint32_t ys_stride = 0, batches = 0;
  OkOrError consume(VorbisStream& st, PacketBatch&& b) override {
    ++batches;
    batch = std::move(b);
    header = st.header;
    ys_stride = st.ys_stride_;
    return build_synth_setup(st, setup);
  }
};
}  // namespace

// --check in.ogg: a damaged file may fail in the middle of a packet; whatever batch the reader then hands over (it flushes the
// complete packets in front of the error) must be self-consistent — every per-packet vector as long as pk says. A packet that
// failed half way and left its rows behind made CorpusDecoder's feeders copy past their staging buffers (round-1 advisor finding).
static int check_mode(const char* path) {
  ParseCallbacks cb;
  Collect sink;
  OggReader reader(cb);
  reader.sink_ = &sink;
  reader.batch_limit_override_ = 0xffffffffu;
  const OkOrError r = reader.full_read(path);
  const PacketBatch& b = sink.batch;
  const uint32_t C = sink.header.audio_channels;
  bool ok = true;
  if (sink.batches) {
    size_t floats = 0;
    for (const vsyn_packet& k : b.pk) {
      (void)k;
    }
    ok = ok && b.ys.size() == b.pk.size() * C * sink.ys_stride && b.floor_number.size() == b.pk.size() * C;
    if (b.vq) {
      ok = ok && b.vq_pk.size() == b.pk.size() && b.residue.empty();
      if (!b.vq_pk.empty()) ok = ok && b.vq_pk.back().entry_off + b.vq_pk.back().num_entries == b.entries.size() && b.vq_pk.back().cls_off <= b.cls.size();
      else ok = ok && b.entries.empty() && b.cls.empty();
    } else {
      ok = ok && b.residue.size() == b.residue_floats;
    }
    (void)floats;
  }
  printf("%s batches=%u packets=%zu error=%d %s\n", ok ? "consistent" : "INCONSISTENT", sink.batches, b.pk.size(), r.is_error_ ? 1 : 0,
         r.is_error_ ? r.err_msg_.c_str() : "");
  return ok ? 0 : 3;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--check") return check_mode(argv[2]);
  if (argc != 3) return 2;
  ParseCallbacks cb;
  Collect sink;
  OggReader reader(cb);
  reader.sink_ = &sink;
  reader.batch_limit_override_ = 0xffffffffu;  // the whole stream in one batch
  const OkOrError r = reader.full_read(argv[1]);
  if (r.is_error_ || sink.batches != 1) {
    fprintf(stderr, "expected one batch (%u), result: %s\n", sink.batches, r.err_msg_.c_str());
    return 1;
  }
  const PacketBatch& b = sink.batch;
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 1;
  auto u32 = [f](uint32_t v) { fwrite(&v, 4, 1, f); };
  const uint32_t hdr[6] = {(uint32_t)b.pk.size(), sink.header.audio_channels, sink.ys_stride, (uint32_t)b.residue.size(),
                           sink.header.get_blocksize_0(), sink.header.get_blocksize_1()};
  fwrite(hdr, sizeof(hdr), 1, f);
  fwrite(b.pk.data(), sizeof(vsyn_packet), b.pk.size(), f);
  fwrite(b.ys.data(), sizeof(uint16_t), b.ys.size(), f);
  fwrite(b.residue.data(), sizeof(float), b.residue.size(), f);
  if (b.vq) {
    const SynthSetup& ss = sink.setup;
    if (!ss.has_vq) return 1;
    u32(0x31305156u);  // "VQ01"
    u32((uint32_t)b.vq_pk.size());
    u32((uint32_t)b.cls.size());
    u32((uint32_t)b.entries.size());
    u32((uint32_t)b.residue_floats);
    fwrite(b.vq_pk.data(), sizeof(vsyn_vq_packet), b.vq_pk.size(), f);
    fwrite(b.cls.data(), 1, b.cls.size(), f);
    fwrite(b.entries.data(), 2, b.entries.size(), f);
    u32(ss.vq.num_codebooks);
    for (uint32_t i = 0; i < ss.vq.num_codebooks; ++i) {
      const vsyn_codebook& k = ss.vq.codebooks[i];
      u32(k.dimensions);
      u32(k.num_entries);
      u32(k.lookup ? 1u : 0u);
      if (k.lookup) fwrite(k.lookup, sizeof(float), (size_t)k.dimensions * k.num_entries, f);
    }
    u32(ss.vq.num_residues);
    for (uint32_t i = 0; i < ss.vq.num_residues; ++i) {
      const vsyn_residue& rs = ss.vq.residues[i];
      u32(rs.type); u32(rs.begin); u32(rs.end); u32(rs.partition_size); u32(rs.num_classifications); u32(rs.classwords);
      fwrite(rs.books, sizeof(int16_t), (size_t)rs.num_classifications * 8, f);
    }
    u32(ss.vq.num_mappings);
    for (uint32_t m = 0; m < ss.vq.num_mappings; ++m) {
      const vsyn_vq_mapping& mp = ss.vq.mappings[m];
      u32(mp.num_submaps);
      fwrite(mp.mux, 1, sink.header.audio_channels, f);
      fwrite(mp.submap_residue, 1, mp.num_submaps, f);
    }
  }
  fclose(f);
  printf("ok\n");
  return 0;
}
