// hooks.cpp — debug-hook registry and dump writer of the host decoder.
//
// Dump format (what tests/compare-debug-out.py of the reference reads; reference writer: src/Callbacks.cpp:136-201,
// 317-324): a sequence of records, each  u32 length (native endian) + bytes.  File header: the magic record
// "ParseOggVorbis-header-v1", then three key/value groups (decoder-name, decoder-sample-rate u32,
// decoder-num-channels u8).  A key/value group = key record, 1-byte type-id record, 1-byte element-size record,
// payload record.  An entry = "entry-name" group, optional "entry-channel" (u8) group, "entry-data" group.
#include "Callbacks.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace {

enum class Sink { Null, Stdout, File };

thread_local Sink tl_sink = Sink::Null;
thread_local std::string tl_file;
thread_local bool tl_filter_on = false;
thread_local std::set<std::string> tl_filter;

struct Decoder {
  int idx = 0;
  std::string name;
  long rate = 0;
  int channels = 0;
  Sink sink = Sink::Null;
  FILE* fp = nullptr;
  bool filter_on = false;
  std::set<std::string> filter;
  std::set<const void*> aliases;

  void close() {
    if (fp) fclose(fp);
    fp = nullptr;
    sink = Sink::Null;
  }
  void record(const void* p, uint32_t n) {
    fwrite(&n, sizeof(n), 1, fp);
    if (n) fwrite(p, 1, n, fp);
  }
  void group(const char* key, uint8_t type_id, uint8_t elem, const void* data, uint32_t bytes) {
    record(key, (uint32_t)strlen(key));
    record(&type_id, 1);
    record(&elem, 1);
    record(data, bytes);
  }
  void open(Sink s, const std::string& fn) {
    close();
    sink = s;
    if (s != Sink::File) return;
    fp = fopen(fn.c_str(), "wb");
    if (!fp) {  // the reference aborts here too (src/Callbacks.cpp:141-145)
      fprintf(stderr, "Callbacks: could not open file %s\n", fn.c_str());
      abort();
    }
    static const char magic[] = "ParseOggVorbis-header-v1";
    record(magic, sizeof(magic) - 1);
    group("decoder-name", DT_Uint8, 1, name.data(), (uint32_t)name.size());
    uint32_t r = (uint32_t)rate;
    group("decoder-sample-rate", DT_UInt32, 4, &r, 4);
    uint8_t c = (uint8_t)channels;
    group("decoder-num-channels", DT_Uint8, 1, &c, 1);
  }
};

std::mutex g_mu;
int g_next_idx = 1;
std::map<const void*, Decoder> g_decoders;
std::map<const void*, const void*> g_alias;

Decoder* find(const void* ref) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto a = g_alias.find(ref);
  if (a != g_alias.end()) ref = a->second;
  auto it = g_decoders.find(ref);
  return it == g_decoders.end() ? nullptr : &it->second;
}

template <typename T>
struct Tag;
template <> struct Tag<float> { static constexpr uint8_t id = DT_Float32; static const char* nm() { return "f32"; } };
template <> struct Tag<int32_t> { static constexpr uint8_t id = DT_Int32; static const char* nm() { return "i32"; } };
template <> struct Tag<uint32_t> { static constexpr uint8_t id = DT_UInt32; static const char* nm() { return "u32"; } };
template <> struct Tag<uint8_t> { static constexpr uint8_t id = DT_Uint8; static const char* nm() { return "u8"; } };
template <> struct Tag<int64_t> { static constexpr uint8_t id = DT_Int64; static const char* nm() { return "i64"; } };
template <> struct Tag<uint64_t> { static constexpr uint8_t id = DT_UInt64; static const char* nm() { return "u64"; } };

template <typename T>
void emit(const void* ref, const char* name, int channel, const T* data, size_t len, uint8_t type_id = Tag<T>::id) {
  Decoder* d = find(ref);
  if (!d) {
    fprintf(stderr, "Callbacks: push_data for an unregistered decoder ('%s')\n", name);
    abort();
  }
  if (d->filter_on && !d->filter.count(name)) return;
  if (d->sink == Sink::Null) return;
  if (d->sink == Sink::Stdout) {
    std::cout << "decoder=" << d->idx << " '" << d->name << "' name='" << name << "' channel=" << channel;
    if (!data) {
      std::cout << " data=NULL";
    } else {
      std::cout << " data=" << Tag<T>::nm() << "{";
      for (size_t i = 0; i < len && i < 10; ++i) std::cout << (i ? " " : "") << +data[i];
      if (len > 10) std::cout << " ...";
      std::cout << "} len=" << len;
    }
    std::cout << std::endl;
    return;
  }
  d->group("entry-name", DT_Uint8, 1, name, (uint32_t)strlen(name));
  if (channel >= 0) {
    uint8_t c = (uint8_t)channel;
    d->group("entry-channel", DT_Uint8, 1, &c, 1);
  }
  d->group("entry-data", type_id, (uint8_t)sizeof(T), data, data ? (uint32_t)(len * sizeof(T)) : 0u);
}

}  // namespace

extern "C" {

void register_decoder_ref(const void* ref, const char* decoder_name, long sample_rate, int num_channels) {
  std::lock_guard<std::mutex> lk(g_mu);
  Decoder& d = g_decoders[ref];
  if (!d.idx) d.idx = g_next_idx++;
  d.name = decoder_name ? decoder_name : "";
  d.rate = sample_rate;
  d.channels = num_channels;
  d.open(tl_sink, tl_file);
  d.filter_on = tl_filter_on;
  d.filter.swap(tl_filter);
  tl_filter.clear();
  tl_filter_on = false;  // the pending selection is consumed by this registration
  tl_sink = Sink::Null;
}

void register_decoder_alias(const void* orig_ref, const void* alias_ref) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto a = g_alias.find(orig_ref);
  const void* root = a != g_alias.end() ? a->second : orig_ref;
  auto it = g_decoders.find(root);
  if (it == g_decoders.end()) return;
  it->second.aliases.insert(alias_ref);
  g_alias[alias_ref] = root;
}

void unregister_decoder_ref(const void* ref) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto a = g_alias.find(ref);
  if (a != g_alias.end()) ref = a->second;
  auto it = g_decoders.find(ref);
  if (it == g_decoders.end()) return;
  for (const void* al : it->second.aliases) g_alias.erase(al);
  it->second.close();
  g_decoders.erase(it);
}

void set_data_output_null(void) { tl_sink = Sink::Null; }
void set_data_output_short_stdout(void) { tl_sink = Sink::Stdout; }
void set_data_output_file(const char* fn) {
  tl_sink = Sink::File;
  tl_file = fn ? fn : "";
}
void set_data_filter(const char** allowed_names) {
  tl_filter.clear();
  tl_filter_on = allowed_names != nullptr;
  if (allowed_names)
    for (const char** p = allowed_names; *p; ++p) tl_filter.insert(*p);
}

void push_data_float(const void* r, const char* n, int c, const float* d, size_t l) { emit(r, n, c, d, l); }
void push_data_u8(const void* r, const char* n, int c, const uint8_t* d, size_t l) { emit(r, n, c, d, l); }
void push_data_i32(const void* r, const char* n, int c, const int32_t* d, size_t l) { emit(r, n, c, d, l); }
void push_data_u32(const void* r, const char* n, int c, const uint32_t* d, size_t l) { emit(r, n, c, d, l); }
void push_data_i64(const void* r, const char* n, int c, const int64_t* d, size_t l) { emit(r, n, c, d, l); }
void push_data_u64(const void* r, const char* n, int c, const uint64_t* d, size_t l) { emit(r, n, c, d, l); }
void push_data_int(const void* r, const char* n, int c, const int* d, size_t l) {
  static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bit on every supported host");
  emit(r, n, c, (const int32_t*)d, l);
}

const char* generic_itoa(uint32_t val, int base, int len) {
  static thread_local char buf[40];
  if (base < 2 || base > 16) base = 10;
  if (len < 0 || len > 32) len = 32;
  char* p = buf + sizeof(buf) - 1;
  *p = 0;
  int digits = 0;
  do {
    *--p = "0123456789abcdef"[val % (uint32_t)base];
    val /= (uint32_t)base;
    ++digits;
  } while (val);
  while (digits++ < len) *--p = '0';
  return p;
}

}  // extern "C"

void push_data_bool(const void* ref, const char* name, int channel, const std::vector<bool>& data) {
  std::vector<uint8_t> tmp(data.size());
  for (size_t i = 0; i < data.size(); ++i) tmp[i] = data[i] ? 1 : 0;
  emit<uint8_t>(ref, name, channel, tmp.data(), tmp.size(), DT_Bool);
}

bool decoder_wants_data(const void* ref) {
  Decoder* d = find(ref);
  return d && d->sink != Sink::Null;
}

void ArgParser::print_usage(const char* argv0) {
  std::cout << argv0 << " --in ogg_filename [--help] [--debug_out filename] [--debug_stdout]" << std::endl;
}

bool ArgParser::parse_args(int argc, const char** argv) {
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&](const char* what) -> const char* {
      if (i + 1 >= argc) {
        std::cerr << "missing arg after " << what << std::endl;
        print_usage(argv[0]);
        return nullptr;
      }
      return argv[++i];
    };
    if (a == "--help") {
      print_usage(argv[0]);
      return false;
    } else if (a == "--in") {
      const char* v = value("--in");
      if (!v) return false;
      ogg_filename = v;
      if (ogg_filename.empty()) {
        std::cerr << "invalid empty filename" << std::endl;
        print_usage(argv[0]);
        return false;
      }
    } else if (a == "--debug_out") {
      const char* v = value("--debug_out");
      if (!v) return false;
      set_data_output_file(v);
    } else if (a == "--debug_stdout") {
      set_data_output_short_stdout();
    } else {
      std::cerr << "unexpected arg " << i << " \"" << a << "\"" << std::endl;
      print_usage(argv[0]);
      return false;
    }
  }
  if (ogg_filename.empty()) {
    std::cerr << "need to provide --in ogg_filename" << std::endl;
    print_usage(argv[0]);
    return false;
  }
  return true;
}
