// Utils.hpp — small host utilities of the MI355X-backed decoder: error type + CHECK macros with the reference's
// message convention (reference: src/Utils.hpp:33-44), byte readers (src/Utils.hpp:257-292), an LSb-first bit
// reader with the reference's read-past-the-end semantics (src/Utils.hpp:330-424), a non-owning range
// (src/Utils.hpp:426-449) and the Ogg CRC.  Independent implementation; only the public names are kept so that
// code written against the reference's headers compiles against these.
#ifndef PARSEOGGVORBIS_AMD_HOST_UTILS_HPP_
#define PARSEOGGVORBIS_AMD_HOST_UTILS_HPP_

#include <assert.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

struct OkOrError {
  bool is_error_;
  std::string err_msg_;
  explicit OkOrError() : is_error_(false) {}
  explicit OkOrError(const std::string& msg) : is_error_(true), err_msg_(msg) {}
};

#define POV_STR2(x) #x
#define POV_STR(x) POV_STR2(x)
// "file:line: check failed: expr" — callers and tests match on this text
#define CHECK(cond)                                                                         \
  do {                                                                                      \
    if (!(cond)) return OkOrError(__FILE__ ":" POV_STR(__LINE__) ": check failed: " #cond); \
  } while (0)
#define CHECK_ERR(expr)             \
  do {                              \
    OkOrError r_ = (expr);          \
    if (r_.is_error_) return r_;    \
  } while (0)

// Vorbis I 9.2.1 ilog: position of the highest set bit (0 for 0)
template <typename T>
inline int highest_bit(T v) {
  int n = 0;
  while (v) {
    ++n;
    v >>= 1;
  }
  return n;
}

uint32_t update_crc(uint32_t crc, const uint8_t* buffer, size_t size);  // Ogg CRC-32: poly 0x04c11db7, MSb first, no inversion

struct IReader {
  virtual ~IReader() {}
  virtual OkOrError isValid() = 0;
  virtual bool reachedEnd() = 0;
  virtual size_t read(void* ptr, size_t size, size_t nitems) = 0;  // fread semantics
};

struct FileReader : IReader {
  FILE* fp_;
  explicit FileReader(const std::string& filename) : fp_(fopen(filename.c_str(), "rb")) {}
  ~FileReader() override {
    if (fp_) fclose(fp_);
  }
  OkOrError isValid() override {
    CHECK(fp_ != NULL);
    return OkOrError();
  }
  bool reachedEnd() override { return feof(fp_) != 0; }
  size_t read(void* ptr, size_t size, size_t nitems) override { return fread(ptr, size, nitems, fp_); }
};

struct ConstDataReader : IReader {  // borrowed memory, valid for the duration of the read
  const uint8_t* data_;
  size_t len_;
  bool reached_end_;
  ConstDataReader(const uint8_t* data, size_t len) : data_(data), len_(len), reached_end_(false) {}
  OkOrError isValid() override { return OkOrError(); }
  bool reachedEnd() override { return reached_end_; }
  size_t read(void* ptr, size_t size, size_t nitems) override {
    size_t can = size ? len_ / size : 0;
    if (can < nitems) {
      nitems = can;
      reached_end_ = true;
    }
    memcpy(ptr, data_, size * nitems);
    data_ += size * nitems;
    len_ -= size * nitems;
    return nitems;
  }
};

// Vorbis bit packing: bit 0 of a byte comes first. Reading past the end yields zero bits and latches reachedEnd()
// (the reference treats that as "not an error", src/Utils.hpp:338).
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "BitReader's fast path reads the stream as little-endian 64-bit words");
struct BitReader {
  const uint8_t* p_;
  size_t len_, byte_;
  int bit_;
  bool reached_end_;
  BitReader(const uint8_t* data, size_t len) : p_(data), len_(len), byte_(0), bit_(0), reached_end_(false) {}

  inline uint32_t bit1() {
    if (byte_ >= len_) {
      reached_end_ = true;
      return 0;
    }
    uint32_t b = (p_[byte_] >> bit_) & 1u;
    if (++bit_ == 8) {
      bit_ = 0;
      ++byte_;
    }
    return b;
  }
  template <typename T>
  T readBits(int num) {
    if (num <= 32 && byte_ + 8 <= len_) {  // fast path, as peek + skip
      uint64_t w;
      memcpy(&w, p_ + byte_, 8);
      const uint64_t v = (w >> bit_) & ((1ull << num) - 1ull);
      const size_t pos = (size_t)bit_ + (size_t)num;
      byte_ += pos >> 3;
      bit_ = (int)(pos & 7);
      return (T)v;
    }
    uint64_t out = 0;
    int got = 0;
    while (got < num) {
      if (byte_ >= len_) {
        reached_end_ = true;
        break;
      }
      const int take = (8 - bit_) < (num - got) ? (8 - bit_) : (num - got);
      out |= (uint64_t)((p_[byte_] >> bit_) & ((1u << take) - 1u)) << got;
      got += take;
      bit_ += take;
      if (bit_ == 8) {
        bit_ = 0;
        ++byte_;
      }
    }
    return (T)out;
  }
  template <int N>
  uint32_t readBitsT() {
    static_assert(N > 0 && N <= 32, "1..32 bits");
    return readBits<uint32_t>(N);
  }
  // next `n` (<= 24) bits without consuming them, zero-padded past the end
  inline uint32_t peek(int n) const {
    if (byte_ + 8 <= len_) {  // fast path: one unaligned 64-bit little-endian load covers bit_ + n <= 31 bits
      uint64_t w;
      memcpy(&w, p_ + byte_, 8);
      return (uint32_t)((w >> bit_) & ((1ull << n) - 1ull));
    }
    uint64_t acc = 0;
    int have = 0;
    size_t b = byte_;
    int sh = bit_;
    while (have < n && b < len_) {
      acc |= (uint64_t)(p_[b] >> sh) << have;
      have += 8 - sh;
      sh = 0;
      ++b;
    }
    return (uint32_t)(acc & ((1ull << n) - 1ull));
  }
  inline void skip(int n) {
    if (byte_ + 8 <= len_) {
      const size_t q = (size_t)bit_ + (size_t)n;
      byte_ += q >> 3;
      bit_ = (int)(q & 7);
      return;
    }
    size_t pos = byte_ * 8 + (size_t)bit_ + (size_t)n;
    if (pos > len_ * 8) {
      reached_end_ = true;
      pos = len_ * 8;
    }
    byte_ = pos >> 3;
    bit_ = (int)(pos & 7);
  }
  size_t bitsLeft() const { return byte_ >= len_ ? 0 : (len_ - byte_) * 8 - (size_t)bit_; }
  bool reachedEnd() const { return reached_end_; }
};

template <typename T>
struct DataRange {  // non-owning view
  T* data_;
  size_t size_;
  DataRange() : data_(nullptr), size_(0) {}
  DataRange(T* data, size_t size) : data_(data), size_(size) {}
  T& operator[](size_t i) { return data_[i]; }
  const T& operator[](size_t i) const { return data_[i]; }
  T* begin() { return data_; }
  T* end() { return data_ + size_; }
  const T* begin() const { return data_; }
  const T* end() const { return data_ + size_; }
  size_t size() const { return size_; }
};

// Vorbis I 9.2.2 float32_unpack (reference: src/Utils.hpp:194-203)
double float32_unpack(uint32_t v);

#endif
