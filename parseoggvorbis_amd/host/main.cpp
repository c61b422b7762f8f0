// main.cpp — command-line front end of the MI355X-backed decoder; same flags and exit codes as the reference's demo
// executable (reference: src/main.cpp:53-67, src/Callbacks.cpp:392-440), so tests/compare-debug-out.py can drive it:
//   ours_hip.bin --in file.ogg [--debug_out dump] [--debug_stdout]
#include <iostream>

#include "Callbacks.h"
#include "ParseOggVorbis.hpp"

namespace {
struct PrintingCallbacks : ParseCallbacks {
  uint64_t samples = 0;
  bool gotHeader(const VorbisIdHeader& h) override {
    std::cout << "Header: vorbis version: " << h.vorbis_version << ", channels: " << (int)h.audio_channels
              << ", sample rate: " << h.audio_sample_rate << std::endl;
    return true;
  }
  bool gotComments(const std::string& vendor, const std::vector<std::string> comments) override {
    std::cout << "Vendor: " << vendor << std::endl;
    for (const std::string& c : comments) std::cout << "Comment: " << c << std::endl;
    return true;
  }
  bool gotSetup(const VorbisStreamSetup& s) override {
    std::cout << "Setup: num codebooks: " << s.codebooks.size() << ", num floors: " << s.floors.size()
              << ", num mappings: " << s.mappings.size() << ", num modes: " << s.modes.size()
              << ", num residues: " << s.residues.size() << std::endl;
    return true;
  }
  bool gotPcmData(const std::vector<DataRange<const float>>& pcm) override {
    if (!pcm.empty()) samples += pcm[0].size();
    return true;
  }
  bool gotEof() override {
    std::cout << "got eof. sample count: " << samples << std::endl;
    return true;
  }
};
}  // namespace

int main(int argc, const char** argv) {
  ArgParser args;
  if (!args.parse_args(argc, argv)) return 1;
  PrintingCallbacks cb;
  OggReader reader(cb);
  const OkOrError r = reader.full_read(args.ogg_filename);
  if (r.is_error_) {
    std::cerr << "error: " << r.err_msg_ << std::endl;
    return 1;
  }
  std::cout << "ok" << std::endl;
  std::cout << "Ogg total packets count: " << reader.packet_counts_ << std::endl;
  return 0;
}
