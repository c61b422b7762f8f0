// tests/host_entropy_dump.cpp — CPU-only probe of the host decoder's ENTROPY half: parses an .ogg with the product's
// OggReader, lets the (GPU-less) synthesis batch fail, and dumps what the entropy half queued for the GPU
// (mode/flags/granule, coded floor posts, "after_residue") so tests/test_host_decoder.py can compare it with the
// reference decoder's hooks in tests/golden/.  Usage: host_entropy_dump in.ogg out.bin
#include <stdio.h>
#include <stdlib.h>

#include "../parseoggvorbis_amd/host/ParseOggVorbis.hpp"

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  setenv("PARSEOGGVORBIS_BATCH", "100000000", 1);  // never flush before the end of the stream
  ParseCallbacks cb;
  OggReader reader(cb);
  OkOrError r = reader.full_read(argv[1]);
  if (reader.streams_.size() != 1) {
    fprintf(stderr, "expected the stream to be still pending (%zu), result: %s\n", reader.streams_.size(), r.err_msg_.c_str());
    return 1;
  }
  const VorbisStream& st = *reader.streams_.begin()->second;
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 1;
  const uint32_t hdr[6] = {(uint32_t)st.pk_.size(), st.header.audio_channels, st.ys_stride_, (uint32_t)st.residue_.size(),
                           st.header.get_blocksize_0(), st.header.get_blocksize_1()};
  fwrite(hdr, sizeof(hdr), 1, f);
  fwrite(st.pk_.data(), sizeof(vsyn_packet), st.pk_.size(), f);
  fwrite(st.ys_.data(), sizeof(uint16_t), st.ys_.size(), f);
  fwrite(st.residue_.data(), sizeof(float), st.residue_.size(), f);
  fclose(f);
  printf("%s\n", r.is_error_ ? r.err_msg_.c_str() : "ok");
  return 0;
}
