// corpus_main.cpp — decode a corpus of Ogg Vorbis files with T entropy threads and one GPU, print one JSON line.
//   corpus_hip.bin [--threads T] [--feeders F] [--files_per_submit K] [--replicas N] [--device D] [--entropy_only] [--no_setup_cache] [--no_checksum] [--s16] file.ogg [file.ogg ...]
// --replicas N decodes every listed file N times (N independent decodes from the same bytes in memory): the way to get a
// corpus-sized run out of the two fixture files when there is no corpus on the box.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "CorpusDecoder.hpp"

static bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
  fclose(f);
  return got == out.size();
}

int main(int argc, const char** argv) {
  CorpusOptions opts;
  size_t replicas = 1;
  std::vector<std::string> paths;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto need = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        fprintf(stderr, "%s needs a value\n", name);
        exit(2);
      }
      return argv[++i];
    };
    if (a == "--threads") opts.threads = atoi(need("--threads"));
    else if (a == "--feeders") opts.feeders = atoi(need("--feeders"));
    else if (a == "--files_per_submit") opts.files_per_submit = (uint32_t)atoi(need("--files_per_submit"));
    else if (a == "--replicas") replicas = (size_t)atol(need("--replicas"));
    else if (a == "--device") opts.device = atoi(need("--device"));
    else if (a == "--entropy_only") opts.entropy_only = true;
    else if (a == "--no_setup_cache") opts.share_setups = false;
    else if (a == "--no_checksum") opts.checksum = false;
    else if (a == "--s16") opts.pcm_s16 = true;
    else if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    } else paths.push_back(a);
  }
  if (paths.empty() || replicas == 0) {
    fprintf(stderr, "usage: corpus_hip.bin [--threads T] [--feeders F] [--files_per_submit K] [--replicas N] [--device D] file.ogg ...\n");
    return 2;
  }
  std::vector<std::vector<uint8_t>> blobs(paths.size());
  for (size_t i = 0; i < paths.size(); ++i)
    if (!read_file(paths[i], blobs[i])) {
      fprintf(stderr, "cannot read %s\n", paths[i].c_str());
      return 1;
    }
  std::vector<CorpusItem> items;
  for (size_t r = 0; r < replicas; ++r)
    for (const auto& b : blobs) items.push_back(CorpusItem{b.data(), b.size()});

  std::vector<CorpusFileResult> results;
  CorpusStats st;
  const OkOrError run = decode_corpus(items, opts, nullptr, results, &st);
  if (run.is_error_) {
    fprintf(stderr, "error: %s\n", run.err_msg_.c_str());
    return 1;
  }
  size_t failed = 0;
  double seconds_of_audio = 0;
  for (size_t i = 0; i < results.size(); ++i) {
    if (results[i].status.is_error_) {
      if (failed++ < 5) fprintf(stderr, "file %zu (%s): %s\n", i, paths[i % paths.size()].c_str(), results[i].status.err_msg_.c_str());
    } else if (results[i].sample_rate) {
      seconds_of_audio += (double)results[i].frames / results[i].sample_rate;
    }
  }
  // replicas of one file must agree exactly: same bytes in, same kernels, same order of operations
  size_t mismatched = 0;
  for (size_t i = paths.size(); i < results.size(); ++i) {
    const CorpusFileResult &a = results[i % paths.size()], &b = results[i];
    if (a.frames != b.frames || (opts.checksum && a.abs_sum != b.abs_sum)) ++mismatched;
  }
  printf("{\"files\": %zu, \"failed\": %zu, \"replica_mismatches\": %zu, \"threads\": %d, \"feeders\": %d, \"files_per_submit\": %u, "
         "\"audio_packets\": %llu, \"frames\": %llu, \"wall_s\": %.4f, \"files_per_s\": %.1f, \"packets_per_s\": %.0f, \"frames_per_s\": %.0f, \"realtime_factor\": %.0f, "
         "\"entropy_cpu_s\": %.3f, \"gpu_call_s\": %.3f, \"pack_s\": %.3f, \"deliver_s\": %.3f, \"submits\": %llu, \"handles\": %u, \"setup_parses\": %llu, \"setup_reuses\": %llu, "
         "\"first_file\": {\"frames\": %llu, \"abs_sum\": %.9g}}\n",
         results.size(), failed, mismatched, opts.threads, opts.feeders, opts.files_per_submit, (unsigned long long)st.audio_packets,
         (unsigned long long)st.frames, st.wall_s, results.size() / st.wall_s,
         st.audio_packets / st.wall_s, st.frames / st.wall_s, seconds_of_audio / st.wall_s, st.entropy_cpu_s, st.gpu_call_s, st.pack_s,
         st.deliver_s, (unsigned long long)st.submits, st.handles, (unsigned long long)st.setup_parses, (unsigned long long)st.setup_reuses, (unsigned long long)results[0].frames, results[0].abs_sum);
  return failed || mismatched ? 1 : 0;
}
