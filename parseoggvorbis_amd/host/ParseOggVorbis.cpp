// ParseOggVorbis.cpp — host decoder: Ogg framing, Vorbis header/setup parse and the per-packet ENTROPY half on the CPU;
// the synthesis half is batched onto the MI355X through include/vorbis_synth_hip.h.  See ParseOggVorbis.hpp.
// Written from the Vorbis I specification (section numbers cited); behaviour on the public surface follows the
// reference decoder (file:line cited where a reference convention is mirrored on purpose).
#include "ParseOggVorbis.hpp"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

static const uint32_t k_inverse_db_bits[256] = {
#include "../csrc/vorbis_floor1_inverse_db.inc"
};

// ------------------------------------------------------------------------------------------------
// utilities
// ------------------------------------------------------------------------------------------------
// Ogg page checksum (framing spec: CRC-32, polynomial 0x04c11db7, MSB first, no reflection, initial value and final xor 0).
// Slicing-by-8: T[k][i] is the checksum of byte i followed by k zero bytes, so eight input bytes fold with eight independent
// look-ups; the page checksum was a quarter of the entropy front-end's time with the byte-at-a-time loop.
namespace {
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t r = i << 24;
      for (int k = 0; k < 8; ++k) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : (r << 1);
      t[0][i] = r;
    }
    for (int k = 1; k < 8; ++k)
      for (uint32_t i = 0; i < 256; ++i) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
  }
};
}  // namespace

uint32_t update_crc(uint32_t crc, const uint8_t* buffer, size_t size) {
  static const CrcTables tables;  // thread-safe initialisation (C++11)
  const uint32_t (*t)[256] = tables.t;
  size_t i = 0;
  for (; i + 8 <= size; i += 8) {
    const uint32_t hi = crc ^ (((uint32_t)buffer[i] << 24) | ((uint32_t)buffer[i + 1] << 16) | ((uint32_t)buffer[i + 2] << 8) | buffer[i + 3]);
    crc = t[7][hi >> 24] ^ t[6][(hi >> 16) & 0xffu] ^ t[5][(hi >> 8) & 0xffu] ^ t[4][hi & 0xffu] ^ t[3][buffer[i + 4]] ^ t[2][buffer[i + 5]] ^
          t[1][buffer[i + 6]] ^ t[0][buffer[i + 7]];
  }
  for (; i < size; ++i) crc = (crc << 8) ^ t[0][((crc >> 24) ^ buffer[i]) & 0xffu];
  return crc;
}

double float32_unpack(uint32_t v) {  // Vorbis I 9.2.2
  double mant = (double)(v & 0x1fffffu);
  if (v & 0x80000000u) mant = -mant;
  long e = (long)((v & 0x7fe00000u) >> 21) - 788;
  if (e > 63) e = 63;  // exponent clamp as the reference applies it (src/Utils.hpp:200-201)
  if (e < -63) e = -63;
  return ldexp(mant, (int)e);
}

// ------------------------------------------------------------------------------------------------
// codebooks (3.2.1)
// ------------------------------------------------------------------------------------------------
static const int kFastBits = 10;

OkOrError VorbisCodebook::parse(BitReader& reader) {
  CHECK(reader.readBitsT<24>() == 0x564342);
  dimensions_ = (uint16_t)reader.readBitsT<16>();
  CHECK(dimensions_ > 0);
  num_entries_ = reader.readBitsT<24>();
  CHECK(num_entries_ > 0);
  lengths_.assign(num_entries_, 0);
  ordered_ = reader.readBitsT<1>() != 0;
  if (!ordered_) {
    sparse_ = reader.readBitsT<1>() != 0;
    for (uint32_t i = 0; i < num_entries_; ++i) {
      if (sparse_ && !reader.readBitsT<1>()) continue;
      lengths_[i] = (uint8_t)(reader.readBitsT<5>() + 1);
    }
  } else {
    sparse_ = false;
    uint32_t len = reader.readBitsT<5>() + 1, cur = 0;
    while (cur < num_entries_) {
      const uint32_t number = reader.readBits<uint32_t>(highest_bit(num_entries_ - cur));
      CHECK(cur + number <= num_entries_);
      CHECK(len <= 32);
      for (uint32_t i = cur; i < cur + number; ++i) lengths_[i] = (uint8_t)len;
      cur += number;
      ++len;
    }
  }

  // codeword assignment: each used entry, in order, takes the lowest still-free codeword of its length (3.2.1);
  // free[l] = a free codeword of length l, left-aligned in 32 bits, or 0
  {
    uint32_t free_at[33];
    memset(free_at, 0, sizeof(free_at));
    tree_.clear();
    tree_.push_back(Node{{0, 0}});
    fast_.assign(1u << kFastBits, 0);
    bool first = true;
    size_t used = 0;
    auto insert = [&](uint32_t code_left_aligned, int len, uint32_t entry) {
      int node = 0;
      for (int b = 0; b < len; ++b) {
        const int bit = (code_left_aligned >> (31 - b)) & 1;
        if (b == len - 1) {
          tree_[node].child[bit] = ~(int32_t)entry;
        } else {
          if (tree_[node].child[bit] <= 0) {
            tree_.push_back(Node{{0, 0}});
            tree_[node].child[bit] = (int32_t)tree_.size() - 1;
          }
          node = tree_[node].child[bit];
        }
      }
      if (len <= kFastBits) {  // table index bit i = i-th bit read = i-th most significant code bit
        uint32_t base = 0;
        for (int b = 0; b < len; ++b) base |= ((code_left_aligned >> (31 - b)) & 1u) << b;
        for (uint32_t hi = 0; hi < (1u << (kFastBits - len)); ++hi) fast_[base | (hi << len)] = (entry << 8) | (uint32_t)len;
      }
    };
    for (uint32_t i = 0; i < num_entries_; ++i) {
      const int len = lengths_[i];
      if (!len) continue;
      ++used;
      uint32_t code;
      if (first) {
        first = false;
        code = 0;
        for (int l = 1; l <= len; ++l) free_at[l] = 1u << (32 - l);
      } else {
        int z = len;
        while (z > 0 && !free_at[z]) --z;
        CHECK(z > 0);  // over-specified tree
        code = free_at[z];
        free_at[z] = 0;
        for (int y = len; y > z; --y) free_at[y] = code + (1u << (32 - y));
      }
      insert(code, len, i);
    }
    CHECK(used > 0);
    for (int l = 1; l <= 32; ++l) CHECK(free_at[l] == 0);  // under-specified tree (the reference rejects these too, hpp:182-184)
  }

  lookup_type_ = (uint8_t)reader.readBitsT<4>();
  CHECK(lookup_type_ <= 2);
  num_lookup_values_ = 0;
  if (lookup_type_) {
    minimum_value_ = float32_unpack(reader.readBitsT<32>());
    delta_value_ = float32_unpack(reader.readBitsT<32>());
    value_bits_ = (uint8_t)(reader.readBitsT<4>() + 1);
    sequence_p_ = reader.readBitsT<1>() != 0;
    if (lookup_type_ == 1) {  // lookup1_values (9.2.3)
      uint32_t n = 0;
      for (;;) {
        uint64_t pw = 1;
        bool over = false;
        for (uint16_t d = 0; d < dimensions_ && !over; ++d) {
          pw *= (uint64_t)(n + 1);
          over = pw > num_entries_;
        }
        if (over) break;
        ++n;
      }
      num_lookup_values_ = n;
    } else {
      num_lookup_values_ = num_entries_ * dimensions_;
    }
    multiplicands_.resize(num_lookup_values_);
    for (uint32_t& m : multiplicands_) m = reader.readBits<uint32_t>(value_bits_);
    // VQ table (3.2.1 "vector representation"); double arithmetic rounded to float per element and the float fed
    // back for sequence_p, as the reference's table does (hpp:217-243)
    lookup_table_.assign((size_t)num_entries_ * dimensions_, 0.f);
    for (uint32_t e = 0; e < num_entries_; ++e) {
      double last = 0;
      uint32_t div = 1;
      for (uint16_t d = 0; d < dimensions_; ++d) {
        const size_t at = (size_t)e * dimensions_ + d;
        const uint32_t mi = lookup_type_ == 1 ? (num_lookup_values_ ? (e / div) % num_lookup_values_ : 0) : (uint32_t)at;
        CHECK(mi < multiplicands_.size());
        const float val = (float)(multiplicands_[mi] * delta_value_ + minimum_value_ + last);
        lookup_table_[at] = val;
        if (sequence_p_) last = val;
        if (lookup_type_ == 1) div *= num_lookup_values_;
      }
    }
  }
  CHECK(!reader.reachedEnd());
  return OkOrError();
}

uint32_t VorbisCodebook::decodeScalar(BitReader& reader) const {
  const uint32_t e = fast_[reader.peek(kFastBits)];
  if (e) {
    reader.skip((int)(e & 0xff));
    return e >> 8;
  }
  int node = 0;
  for (int depth = 0; depth < 33; ++depth) {  // bits past the packet end read as 0, like the reference's reader
    const int32_t nx = tree_[node].child[reader.bit1()];
    if (nx < 0) return (uint32_t)~nx;
    if (nx == 0) break;  // cannot happen for a fully specified tree
    node = nx;
  }
  return 0xffffffffu;
}

// The same as `count` calls of decodeScalar, with the bit window held in a register between code words: one 64-bit load feeds
// several table look-ups (the per-word reload and byte/bit bookkeeping was most of decodeScalar's dependent chain).
uint32_t VorbisCodebook::decodeRun(BitReader& reader, uint32_t count, uint16_t* dst) const {
  uint32_t worst = 0, k = 0;
  const uint32_t* const fast = fast_.data();
  while (k < count) {
    if (reader.byte_ + 16 <= reader.len_) {
      uint64_t w;
      memcpy(&w, reader.p_ + reader.byte_, 8);
      w >>= reader.bit_;
      int avail = 64 - reader.bit_, used = 0;
      bool slow = false;
      while (k < count && avail - used >= kFastBits) {
        const uint32_t e = fast[w & ((1u << kFastBits) - 1u)];
        if (!e) {  // longer than the table covers (or not a code word): the tree walk decides
          slow = true;
          break;
        }
        const int len = (int)(e & 0xffu);
        w >>= len;
        used += len;
        const uint32_t v = e >> 8;
        worst = v > worst ? v : worst;
        dst[k++] = (uint16_t)v;
      }
      const size_t pos = (size_t)reader.bit_ + (size_t)used;
      reader.byte_ += pos >> 3;
      reader.bit_ = (int)(pos & 7);
      if (!slow) continue;
    }
    const uint32_t v = decodeScalar(reader);
    worst = v > worst ? v : worst;
    dst[k++] = (uint16_t)v;
  }
  return worst;
}

const float* VorbisCodebook::decodeVector(BitReader& reader) const {
  const uint32_t idx = decodeScalar(reader);
  if (!lookup_type_ || idx >= num_entries_) return nullptr;
  return &lookup_table_[(size_t)idx * dimensions_];
}

// ------------------------------------------------------------------------------------------------
// floors (6, 7)
// ------------------------------------------------------------------------------------------------
OkOrError VorbisFloor0::parse(BitReader& reader, int max_books) {
  order = (uint8_t)reader.readBitsT<8>();
  rate = (uint16_t)reader.readBitsT<16>();
  bark_map_size = (uint16_t)reader.readBitsT<16>();
  amplitude_bits = (uint8_t)reader.readBitsT<6>();
  amplitude_offset = (uint8_t)reader.readBitsT<8>();
  books.resize(reader.readBitsT<4>() + 1);
  for (uint8_t& b : books) {
    b = (uint8_t)reader.readBitsT<8>();
    CHECK(b < max_books);
  }
  return OkOrError();
}

OkOrError VorbisFloor1::parse(BitReader& reader, int num_codebooks) {
  partition_classes.resize(reader.readBitsT<5>());
  int max_class = -1;
  for (uint8_t& c : partition_classes) {
    c = (uint8_t)reader.readBitsT<4>();
    max_class = std::max<int>(max_class, c);
  }
  classes.resize((size_t)(max_class + 1));
  for (VorbisFloorClass& cl : classes) {
    cl.dimensions = (uint8_t)(reader.readBitsT<3>() + 1);
    cl.subclass = (uint8_t)reader.readBitsT<2>();
    if (cl.subclass) {
      cl.masterbook = (uint8_t)reader.readBitsT<8>();
      CHECK(cl.masterbook < num_codebooks);
    }
    cl.subclass_books.resize(1u << cl.subclass);
    for (int& b : cl.subclass_books) {
      b = (int)reader.readBitsT<8>() - 1;
      CHECK(b < num_codebooks);
    }
  }
  multiplier = (uint8_t)(reader.readBitsT<2>() + 1);
  const int rangebits = (int)reader.readBitsT<4>();
  xs.assign(2, 0);
  xs[1] = 1u << rangebits;
  for (uint8_t c : partition_classes)
    for (int j = 0; j < classes[c].dimensions; ++j) xs.push_back(rangebits ? reader.readBits<x_t>(rangebits) : 0);
  CHECK(xs.size() <= VSYN_MAX_POSTS);  // 7.2.2: at most 65 posts
  return OkOrError();
}

OkOrError VorbisFloor1::decode_ys(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, std::vector<uint32_t>& ys, bool& use_output) const {
  ys.clear();
  if (!reader.readBitsT<1>()) {  // 7.2.3: "nonzero" flag clear: this channel is silent in this packet
    use_output = false;
    return OkOrError();
  }
  use_output = true;
  static const uint32_t range_of[5] = {0, 256, 128, 86, 64};
  CHECK(multiplier >= 1 && multiplier <= 4);
  const int ybits = highest_bit(range_of[multiplier] - 1);
  ys.push_back(reader.readBits<uint32_t>(ybits));
  ys.push_back(reader.readBits<uint32_t>(ybits));
  for (uint8_t c : partition_classes) {
    const VorbisFloorClass& cl = classes[c];
    const uint32_t cbits = cl.subclass, csub = (1u << cbits) - 1u;
    uint32_t cval = cbits ? codebooks[cl.masterbook].decodeScalar(reader) : 0;
    for (int j = 0; j < cl.dimensions; ++j) {
      const int book = cl.subclass_books[cval & csub];
      cval >>= cbits;
      ys.push_back(book >= 0 ? codebooks[(size_t)book].decodeScalar(reader) : 0u);
    }
  }
  CHECK(ys.size() == xs.size());
  return OkOrError();
}

OkOrError VorbisFloor::parse(BitReader& reader, int num_codebooks) {
  floor_type = (uint16_t)reader.readBitsT<16>();
  if (floor_type == 0) CHECK_ERR(floor0.parse(reader, num_codebooks));
  else if (floor_type == 1) CHECK_ERR(floor1.parse(reader, num_codebooks));
  else CHECK(false);  // invalid floor type
  return OkOrError();
}

// ------------------------------------------------------------------------------------------------
// residues (8)
// ------------------------------------------------------------------------------------------------
OkOrError VorbisResidue::parse(BitReader& reader, int num_codebooks) {
  type = (uint16_t)reader.readBitsT<16>();
  CHECK(type <= 2);
  begin = reader.readBitsT<24>();
  end = reader.readBitsT<24>();
  CHECK(begin <= end);
  partition_size = reader.readBitsT<24>() + 1;
  num_classifications = (uint8_t)(reader.readBitsT<6>() + 1);
  classbook = (uint8_t)reader.readBitsT<8>();
  CHECK(classbook < num_codebooks);
  cascades.resize(num_classifications);
  for (uint32_t& x : cascades) {
    const uint32_t low = reader.readBitsT<3>();
    const uint32_t high = reader.readBitsT<1>() ? reader.readBitsT<5>() : 0;
    x = high * 8 + low;
  }
  books.assign((size_t)num_classifications * 8, -1);
  for (int i = 0; i < num_classifications; ++i)
    for (int j = 0; j < 8; ++j)
      if (cascades[(size_t)i] & (1u << j)) {
        // (not checked against the codebook count here: the reference reads it unchecked, hpp:656, and only a packet that USES
        // such an entry fails; decode / decode_entries check at that point)
        books[(size_t)i * 8 + j] = (int16_t)reader.readBitsT<8>();
      }
  return OkOrError();
}

OkOrError VorbisResidue::decode(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, uint32_t num_channel,
                                const std::vector<bool>& channel_used, uint32_t decode_len, float* const* out, int type_override) const {
  const int type = type_override >= 0 ? type_override : (int)this->type;
  CHECK(num_channel > 0 && channel_used.size() == num_channel);
  if (type == 2) {
    // 8.6.5: decode one interleaved vector of num_channel*decode_len as format 1, then de-interleave.
    // (Decoded even when no channel is marked used — the reference does the same, hpp:685-694.)
    std::vector<float> tmp((size_t)num_channel * decode_len, 0.f);
    float* one[1] = {tmp.data()};
    CHECK_ERR(decode(reader, codebooks, 1, std::vector<bool>{true}, num_channel * decode_len, one, 1));
    for (uint32_t j = 0; j < num_channel; ++j)
      for (uint32_t i = 0; i < decode_len; ++i) out[j][i] = tmp[j + (size_t)num_channel * i];
    return OkOrError();
  }
  const uint32_t lim_begin = std::min(begin, decode_len), lim_end = std::min(end, decode_len);  // limited by the vector, as upstream (hpp:696-698)
  CHECK(lim_begin <= lim_end);
  const VorbisCodebook& cbook = codebooks[classbook];
  const uint32_t cw = cbook.dimensions_;
  const uint32_t n_to_read = lim_end - lim_begin;
  if (!n_to_read) return OkOrError();
  const uint32_t parts = n_to_read / partition_size;
  const uint32_t per_ch = parts + cw;
  std::vector<uint8_t> cls((size_t)num_channel * per_ch, 0);
  for (int pass = 0; pass < 8; ++pass) {
    uint32_t pc = 0;
    while (pc < parts) {
      if (pass == 0)
        for (uint32_t j = 0; j < num_channel; ++j) {
          if (!channel_used[j]) continue;
          uint32_t temp = cbook.decodeScalar(reader);
          for (uint32_t i = cw; i > 0; --i) {
            cls[(size_t)j * per_ch + (i - 1) + pc] = (uint8_t)(temp % num_classifications);
            temp /= num_classifications;
          }
        }
      for (uint32_t i = 0; i < cw && pc < parts; ++i, ++pc) {  // 8.6.2: one partition per classword, all channels
        for (uint32_t j = 0; j < num_channel; ++j) {
          if (!channel_used[j]) continue;
          const int book = books[(size_t)cls[(size_t)j * per_ch + pc] * 8 + (size_t)pass];
          if (book < 0) continue;
          CHECK((size_t)book < codebooks.size());  // (the reference indexes out of range here, hpp:731)
          const VorbisCodebook& vq = codebooks[(size_t)book];
          float* v = out[j];
          const uint32_t offset = lim_begin + pc * partition_size;
          if (type == 0) {  // 8.6.3
            const uint32_t step = partition_size / vq.dimensions_;
            for (uint32_t k = 0; k < step; ++k) {
              const float* t = vq.decodeVector(reader);
              CHECK(t != nullptr);
              for (uint32_t l = 0; l < vq.dimensions_; ++l) {
                CHECK(offset + k + l * step < decode_len);
                v[offset + k + l * step] += t[l];
              }
            }
          } else {  // 8.6.4
            for (uint32_t k = 0; k < partition_size;) {
              const float* t = vq.decodeVector(reader);
              CHECK(t != nullptr);
              for (uint32_t l = 0; l < vq.dimensions_; ++l, ++k) {
                CHECK(offset + k < decode_len);
                v[offset + k] += t[l];
              }
            }
          }
        }
      }
    }
  }
  return OkOrError();
}

// Same control flow as decode() above with the table look-up and the adds left out: what remains is the bit-serial half.
void VorbisResidue::prepare(const std::vector<VorbisCodebook>& codebooks) {
  runs.assign((size_t)num_classifications * 8, Run());
  for (size_t k = 0; k < runs.size() && k < books.size(); ++k) {
    const int b = books[k];
    if (b < 0) continue;
    runs[k].count = 0xFFFFFFFFu;  // a book that cannot be decoded from: an error for the packet that uses it
    if ((size_t)b >= codebooks.size()) continue;
    const VorbisCodebook& vq = codebooks[(size_t)b];
    if (!vq.dimensions_ || vq.fast_.empty()) continue;
    runs[k].fast = vq.fast_.data();
    runs[k].book = &vq;
    runs[k].count = partition_size / vq.dimensions_;
    runs[k].num_entries = vq.num_entries_;
  }
  cls_unpack.clear();
  if (classbook < codebooks.size() && num_classifications) {
    const VorbisCodebook& cb = codebooks[classbook];
    if (cb.dimensions_ >= 1 && cb.dimensions_ <= 4 && cb.num_entries_ <= 65536) {
      cls_unpack.resize(cb.num_entries_);
      for (uint32_t w = 0; w < cb.num_entries_; ++w) {
        uint32_t temp = w, packed = 0;
        for (uint32_t i = cb.dimensions_; i > 0; --i) {  // hpp:717-721: the LAST partition of the group takes temp % classes
          packed |= (temp % num_classifications) << (8 * (i - 1));
          temp /= num_classifications;
        }
        cls_unpack[w] = packed;
      }
    }
  }
}

// The bit-serial half of VorbisResidue::decode (hpp:696-760) and nothing else: class words and codebook entry numbers, in decode
// order. This is where the host front-end spends its time, so the loop keeps the bit window in registers across code words AND
// across codebooks (a 64-bit buffer refilled by one unaligned load when fewer than kFastBits bits are left; VorbisCodebook::
// decodeRun reloaded it per run of ~9 words), reads what a (class, pass) pair means from a table built at setup time (no
// division, no codebook look-up per partition) and writes through a cursor into storage grown geometrically (the per-run
// vector::resize was 12 % of the profile). Code words longer than the prefix table, and the last 16 bytes of a packet, go through
// VorbisCodebook::decodeScalar on the BitReader.
OkOrError VorbisResidue::decode_entries(BitReader& reader, const std::vector<VorbisCodebook>& codebooks, uint32_t num_channel,
                                        const std::vector<bool>& channel_used, uint32_t decode_len, std::vector<uint8_t>& cls_out,
                                        std::vector<uint16_t>& entries_out, int type_override) const {
  const int type = type_override >= 0 ? type_override : (int)this->type;
  CHECK(num_channel > 0 && channel_used.size() == num_channel);
  if (type == 2) {
    static const std::vector<bool> one_used{true};
    return decode_entries(reader, codebooks, 1, one_used, num_channel * decode_len, cls_out, entries_out, 1);
  }
  const uint32_t lim_begin = std::min(begin, decode_len), lim_end = std::min(end, decode_len);
  CHECK(lim_begin <= lim_end);
  const VorbisCodebook& cbook = codebooks[classbook];
  const uint32_t cw = cbook.dimensions_;
  const uint32_t n_to_read = lim_end - lim_begin;
  if (!n_to_read) return OkOrError();
  CHECK(runs.size() == (size_t)num_classifications * 8);  // prepare() ran
  const uint32_t parts = n_to_read / partition_size;
  CHECK(lim_begin + parts * partition_size <= decode_len);
  const uint32_t per_ch = parts + cw;
  static thread_local std::vector<uint8_t> cls;  // (scratch: one decoder per thread, Callbacks.h:16-21)
  cls.assign((size_t)num_channel * per_ch, 0);
  uint32_t pass_mask = 0;  // passes in which some class has a codebook: the others read nothing and are skipped whole
  for (uint32_t x : cascades) pass_mask |= x;

  // ---- bit window ----
  const uint8_t* const p = reader.p_;
  const size_t len = reader.len_;
  const size_t fast_end = len >= 16 ? (len - 16) * 8 : 0;  // bit positions up to here may refill with an 8-byte load
  size_t pos = reader.byte_ * 8 + (size_t)reader.bit_;
  uint64_t buf = 0;
  int have = 0;
  const uint32_t fmask = (1u << kFastBits) - 1u;
  const uint32_t* const cfast = cbook.fast_.data();
  // one code word of `book` (prefix table `fast`); 0xffffffff: no such code word
  auto word = [&](const uint32_t* fast, const VorbisCodebook& book) -> uint32_t {
    if (have < kFastBits) {
      if (len >= 16 && pos <= fast_end) {
        memcpy(&buf, p + (pos >> 3), 8);
        buf >>= (pos & 7);
        have = 64 - (int)(pos & 7);
      } else {
        have = 0;
      }
    }
    if (have >= kFastBits) {
      const uint32_t e = fast[(uint32_t)buf & fmask];
      if (e) {
        const int l = (int)(e & 0xffu);
        buf >>= l;
        have -= l;
        pos += (size_t)l;
        return e >> 8;
      }
    }
    reader.byte_ = pos >> 3;  // the tree walk (or the tail of the packet): through the reader
    reader.bit_ = (int)(pos & 7);
    const uint32_t v = book.decodeScalar(reader);
    pos = reader.byte_ * 8 + (size_t)reader.bit_;
    have = 0;
    return v;
  };

  // ---- output cursor ----
  size_t at = entries_out.size();
  auto room = [&](size_t need) {
    if (at + need > entries_out.size()) entries_out.resize(at + need + 2048);  // (the vector's own growth policy keeps this amortised)
  };
  bool bad_word = false;
  const bool unpack = !cls_unpack.empty();
  for (int pass = 0; pass < 8 && !bad_word; ++pass) {
    if (pass > 0 && !((pass_mask >> pass) & 1u)) continue;
    uint32_t pc = 0;
    while (pc < parts && !bad_word) {
      if (pass == 0)
        for (uint32_t j = 0; j < num_channel; ++j) {
          if (!channel_used[j]) continue;
          uint32_t temp = word(cfast, cbook);
          uint8_t* row = &cls[(size_t)j * per_ch + pc];
          if (unpack && temp < cls_unpack.size()) {
            const uint32_t packed = cls_unpack[temp];
            for (uint32_t i = 0; i < cw; ++i) row[i] = (uint8_t)(packed >> (8 * i));
          } else {
            for (uint32_t i = cw; i > 0; --i) {
              row[i - 1] = (uint8_t)(temp % num_classifications);
              temp /= num_classifications;
            }
          }
        }
      for (uint32_t i = 0; i < cw && pc < parts; ++i, ++pc) {
        for (uint32_t j = 0; j < num_channel; ++j) {
          if (!channel_used[j]) continue;
          const Run& R = runs[(size_t)cls[(size_t)j * per_ch + pc] * 8 + (size_t)pass];
          if (!R.fast) {
            CHECK(R.count == 0u);  // (0xffffffff: the class names a codebook that does not exist / has no code words)
            continue;
          }
          CHECK(R.book->lookup_type_ != 0);  // decodeVector on a scalar-only book
          room(R.count);
          uint16_t* dst = entries_out.data() + at;
          uint32_t worst = 0, k = 0;
          const uint32_t* const fast = R.fast;
          const uint32_t count = R.count;
          have = 0;  // (the run loop below works on its own copy of the window)
          while (k < count) {
            if (len >= 16 && pos <= fast_end) {
              uint64_t w;
              memcpy(&w, p + (pos >> 3), 8);
              w >>= (pos & 7);
              const int avail = 64 - (int)(pos & 7);
              int used = 0;
              bool slow = false;
              while (k < count && avail - used >= kFastBits) {
                const uint32_t e = fast[(uint32_t)w & fmask];
                if (!e) {  // longer than the table covers (or not a code word): the tree walk decides
                  slow = true;
                  break;
                }
                const int l = (int)(e & 0xffu);
                w >>= l;
                used += l;
                const uint32_t v = e >> 8;
                worst = v > worst ? v : worst;
                dst[k++] = (uint16_t)v;
              }
              pos += (size_t)used;
              if (!slow) continue;
            }
            const uint32_t v = word(fast, *R.book);  // (with have == 0 and a code word the table does not hold: through the reader)
            worst = v > worst ? v : worst;
            dst[k++] = (uint16_t)v;
          }
          at += R.count;
          if (worst >= R.num_entries) bad_word = true;  // (0xffffffff = no such code word)
        }
      }
    }
  }
  reader.byte_ = pos >> 3;
  reader.bit_ = (int)(pos & 7);
  entries_out.resize(at);
  CHECK(!bad_word);
  for (uint32_t j = 0; j < num_channel; ++j) cls_out.insert(cls_out.end(), cls.begin() + (size_t)j * per_ch, cls.begin() + (size_t)j * per_ch + parts);
  return OkOrError();
}

// ------------------------------------------------------------------------------------------------
// mappings, modes, setup (4.2.4)
// ------------------------------------------------------------------------------------------------
OkOrError VorbisMapping::parse(BitReader& reader, int num_channels, int num_floors, int num_residues) {
  CHECK(num_channels > 0);
  const int bits = highest_bit(num_channels - 1);
  type = (uint16_t)reader.readBitsT<16>();
  CHECK(type == 0);
  const int num_submaps = reader.readBitsT<1>() ? (int)reader.readBitsT<4>() + 1 : 1;
  if (reader.readBitsT<1>()) {
    couplings.resize(reader.readBitsT<8>() + 1);
    for (Coupling& c : couplings) {
      c.magintude = bits ? (int)reader.readBits<uint32_t>(bits) : 0;
      c.angle = bits ? (int)reader.readBits<uint32_t>(bits) : 0;
      CHECK(c.magintude != c.angle);
      CHECK(c.magintude < num_channels);
      CHECK(c.angle < num_channels);
    }
  }
  CHECK(reader.readBitsT<2>() == 0);
  muxs.assign((size_t)num_channels, 0);
  if (num_submaps > 1)
    for (uint8_t& m : muxs) {
      m = (uint8_t)reader.readBitsT<4>();
      CHECK(m < num_submaps);
    }
  submaps.resize((size_t)num_submaps);
  for (Submap& s : submaps) {
    (void)reader.readBitsT<8>();  // unused time-domain configuration placeholder
    s.floor = (uint8_t)reader.readBitsT<8>();
    CHECK(s.floor < num_floors);
    s.residue = (uint8_t)reader.readBitsT<8>();
    CHECK(s.residue < num_residues);
  }
  return OkOrError();
}

OkOrError VorbisModeNumber::parse(BitReader& reader, int num_mappings, const VorbisIdHeader& header) {
  block_flag = reader.readBitsT<1>() != 0;
  window_type = (uint16_t)reader.readBitsT<16>();
  CHECK(window_type == 0);
  transform_type = (uint16_t)reader.readBitsT<16>();
  CHECK(transform_type == 0);
  mapping = (uint8_t)reader.readBitsT<8>();
  CHECK(mapping < num_mappings);
  blocksize = block_flag ? header.get_blocksize_1() : header.get_blocksize_0();
  return OkOrError();
}

OkOrError VorbisStreamSetup::parse(BitReader& reader, const VorbisIdHeader& header) {
  codebooks.resize(reader.readBitsT<8>() + 1);
  for (VorbisCodebook& c : codebooks) CHECK_ERR(c.parse(reader));
  CHECK(!reader.reachedEnd());
  for (uint32_t i = 0, n = reader.readBitsT<6>() + 1; i < n; ++i) CHECK(reader.readBitsT<16>() == 0);  // time-domain transforms: placeholders
  CHECK(!reader.reachedEnd());
  floors.resize(reader.readBitsT<6>() + 1);
  for (VorbisFloor& f : floors) CHECK_ERR(f.parse(reader, (int)codebooks.size()));
  CHECK(!reader.reachedEnd());
  residues.resize(reader.readBitsT<6>() + 1);
  for (VorbisResidue& r : residues) CHECK_ERR(r.parse(reader, (int)codebooks.size()));
  CHECK(!reader.reachedEnd());
  mappings.resize(reader.readBitsT<6>() + 1);
  for (VorbisMapping& m : mappings) CHECK_ERR(m.parse(reader, header.audio_channels, (int)floors.size(), (int)residues.size()));
  CHECK(!reader.reachedEnd());
  modes.resize(reader.readBitsT<6>() + 1);
  for (VorbisModeNumber& m : modes) CHECK_ERR(m.parse(reader, (int)mappings.size(), header));
  CHECK(!reader.reachedEnd());
  CHECK(reader.readBitsT<1>() == 1);  // framing
  CHECK(!reader.reachedEnd());
  // nothing but zero padding of the last byte may follow (the reference insists on this too, hpp:959-961)
  const size_t left = reader.bitsLeft();
  CHECK(left < 8);
  if (left) CHECK(reader.readBits<uint32_t>((int)left) == 0);
  for (VorbisResidue& r : residues) r.prepare(codebooks);
  return OkOrError();
}

// ------------------------------------------------------------------------------------------------
// stream: headers
// ------------------------------------------------------------------------------------------------
VorbisStream::VorbisStream() {
  memset(&header, 0, sizeof(header));
  if (const char* e = getenv("PARSEOGGVORBIS_BATCH"))
    if (atoi(e) > 0) batch_limit_ = (uint32_t)atoi(e);
}

VorbisStream::~VorbisStream() {
  if (synth_) vsyn_destroy(synth_);
  unregister_decoder_ref(this);
}

OkOrError VorbisStream::parse_id(const uint8_t* data, uint32_t len, ParseCallbacks& cb) {
  CHECK(len >= 16);
  CHECK(data[0] == 1);
  CHECK(memcmp(data + 1, "vorbis", 6) == 0);
  CHECK(len - 7 == sizeof(VorbisIdHeader));
  memcpy(&header, data + 7, sizeof(VorbisIdHeader));
  CHECK(header.framing_flag == 1);
  CHECK(header.vorbis_version == 0);
  CHECK(64 <= header.get_blocksize_0() && header.get_blocksize_0() <= 8192);
  CHECK(64 <= header.get_blocksize_1() && header.get_blocksize_1() <= 8192);
  CHECK(header.get_blocksize_0() <= header.get_blocksize_1());
  CHECK(cb.gotHeader(header));
  return OkOrError();
}

static bool rd_u32(const uint8_t* data, uint32_t len, size_t& off, uint32_t& v) {
  if (off + 4 > len) return false;
  v = (uint32_t)data[off] | ((uint32_t)data[off + 1] << 8) | ((uint32_t)data[off + 2] << 16) | ((uint32_t)data[off + 3] << 24);
  off += 4;
  return true;
}

OkOrError VorbisStream::parse_comment(const uint8_t* data, uint32_t len, ParseCallbacks& cb) {
  CHECK(len >= 16);
  CHECK(data[0] == 3);
  CHECK(memcmp(data + 1, "vorbis", 6) == 0);
  size_t off = 7;
  uint32_t n = 0;
  CHECK(rd_u32(data, len, off, n));
  CHECK(off + n <= len);
  const std::string vendor((const char*)data + off, n);
  off += n;
  uint32_t count = 0;
  CHECK(rd_u32(data, len, off, count));
  CHECK(off + (uint64_t)count * 4 <= len);
  std::vector<std::string> comments(count);
  for (std::string& c : comments) {
    CHECK(rd_u32(data, len, off, n));
    CHECK(off + n <= len);
    c.assign((const char*)data + off, n);
    off += n;
  }
  CHECK(off + 1 == len);
  CHECK(data[off] == 1);  // framing
  CHECK(cb.gotComments(vendor, comments));
  return OkOrError();
}

OkOrError VorbisStream::parse_setup(const uint8_t* data, uint32_t len, ParseCallbacks& cb) {
  CHECK(len >= 16);
  CHECK(data[0] == 5);
  CHECK(memcmp(data + 1, "vorbis", 6) == 0);
  setup_hash_ = 1469598103934665603ull;  // FNV-1a of the packet: identifies the codebooks when streams share a handle / a parsed setup
  for (uint32_t i = 0; i < len; ++i) setup_hash_ = (setup_hash_ ^ data[i]) * 1099511628211ull;
  std::shared_ptr<const VorbisStreamSetup> known = setup_cache_ ? setup_cache_->find(setup_hash_, data, len, header) : nullptr;
  if (known) {
    setup = *known;
  } else {
    BitReader reader(data + 7, len - 7);
    CHECK_ERR(setup.parse(reader, header));
    if (setup_cache_) setup_cache_->insert(setup_hash_, data, len, header, setup);
  }

  // hooks: same entries, same order as upstream (hpp:1360-1370)
  register_decoder_ref(this, "ParseOggVorbis", (long)header.audio_sample_rate, header.audio_channels);
  for (const VorbisFloor& f : setup.floors)
    if (f.floor_type == 1) {
      push_data_u8(this, "floor1_unpack multiplier", -1, &f.floor1.multiplier, 1);
      push_data_u32(this, "floor1_unpack xs", -1, f.floor1.xs.data(), f.floor1.xs.size());
    }
  push_data_u8(this, "finish_setup", -1, nullptr, 0);
  {  // row stride of the coded-post matrix handed to the GPU layer: posts rounded up to 4 (vsyn_ys_stride)
    size_t maxp = 2;
    for (const VorbisFloor& f : setup.floors)
      if (f.floor_type == 1) maxp = std::max(maxp, f.floor1.xs.size());
    ys_stride_ = (uint32_t)((maxp + 3) & ~(size_t)3);
  }
  vq_mode_ = stream_can_use_vq(*this);
  CHECK(cb.gotSetup(setup));
  return OkOrError();
}

std::shared_ptr<const VorbisStreamSetup> SetupCache::find(uint64_t hash, const uint8_t* data, uint32_t len, const VorbisIdHeader& h) {
  std::lock_guard<std::mutex> lk(mu);
  auto range = entries.equal_range(hash);
  for (auto it = range.first; it != range.second; ++it) {
    const Entry& e = it->second;
    if (e.bytes.size() == len && e.channels == h.audio_channels && e.blocksizes == h.blocksizes_exp && memcmp(e.bytes.data(), data, len) == 0) {
      ++hits;
      return e.setup;
    }
  }
  ++misses;
  return nullptr;
}

void SetupCache::insert(uint64_t hash, const uint8_t* data, uint32_t len, const VorbisIdHeader& h, const VorbisStreamSetup& parsed) {
  Entry e;
  e.bytes.assign(data, data + len);
  e.channels = h.audio_channels;
  e.blocksizes = h.blocksizes_exp;
  e.setup = std::make_shared<const VorbisStreamSetup>(parsed);
  std::lock_guard<std::mutex> lk(mu);
  if (entries.size() < 256) entries.emplace(hash, std::move(e));  // bounded: a corpus has a handful of encoder settings
}

// Whether the residue may leave the host as entry numbers: the limits of vsyn_attach_vq, checked on the host model.
bool stream_can_use_vq(const VorbisStream& st) {
  if (const char* e = getenv("PARSEOGGVORBIS_VQ"))
    if (e[0] == '0') return false;
  const uint32_t C = st.header.audio_channels, n2max = st.header.get_blocksize_1() / 2u;
  for (const VorbisResidue& r : st.setup.residues) {
    if (r.type > 2 || r.partition_size == 0 || r.num_classifications > 64) return false;
    if (r.classbook >= st.setup.codebooks.size() || st.setup.codebooks[r.classbook].dimensions_ == 0) return false;
    for (size_t i = 0; i < r.books.size(); ++i) {
      const int b = r.books[i];
      if (b < 0) continue;
      if ((size_t)b >= st.setup.codebooks.size()) return false;
      const VorbisCodebook& cb = st.setup.codebooks[(size_t)b];
      if (!cb.lookup_type_ || cb.dimensions_ == 0 || cb.num_entries_ > 65536u) return false;
      if (r.partition_size % cb.dimensions_) return false;
    }
  }
  for (const VorbisMapping& mp : st.setup.mappings)
    for (size_t s = 0; s < mp.submaps.size(); ++s) {
      uint32_t nch = 0;
      for (uint32_t ch = 0; ch < C; ++ch) nch += mp.muxs[ch] == s;
      if (!nch) continue;
      if (mp.submaps[s].residue >= st.setup.residues.size()) return false;
      const VorbisResidue& r = st.setup.residues[mp.submaps[s].residue];
      const uint32_t len = r.type == 2 ? nch * n2max : n2max, vch = r.type == 2 ? 1u : nch;
      const uint32_t parts = (std::min(r.end, len) - std::min(r.begin, len)) / r.partition_size;
      if ((uint64_t)8 * parts * vch > 8192u) return false;
    }
  return true;
}

// The synthesis-relevant part of the setup, handed to the GPU layer once per stream (or once per group of streams that
// share it: `key` serialises exactly what vsyn_create reads).
OkOrError build_synth_setup(const VorbisStream& st, SynthSetup& o) {
  const uint32_t C = st.header.audio_channels;
  CHECK(C >= 1 && C <= VSYN_MAX_CHANNELS);
  o.floors.resize(st.setup.floors.size());
  o.xs.resize(st.setup.floors.size());
  for (size_t i = 0; i < o.floors.size(); ++i) {
    const VorbisFloor& f = st.setup.floors[i];
    if (f.floor_type == 1) {
      o.xs[i] = f.floor1.xs;
      o.floors[i].multiplier = f.floor1.multiplier;
    } else {  // type 0 cannot be decoded (packets using it fail earlier); keep the table slot well-formed
      o.xs[i] = {0u, 1u};
      o.floors[i].multiplier = 1;
    }
    o.floors[i].num_posts = (uint32_t)o.xs[i].size();
    o.floors[i].xs = o.xs[i].data();
  }
  o.coup.resize(st.setup.mappings.size());
  o.chfloor.resize(st.setup.mappings.size());
  o.maps.resize(st.setup.mappings.size());
  for (size_t m = 0; m < o.maps.size(); ++m) {
    const VorbisMapping& mp = st.setup.mappings[m];
    for (const VorbisMapping::Coupling& c : mp.couplings) o.coup[m].push_back(vsyn_coupling{(uint16_t)c.magintude, (uint16_t)c.angle});
    for (uint32_t c = 0; c < C; ++c) o.chfloor[m].push_back(mp.submaps[mp.muxs[c]].floor);
    o.maps[m].num_couplings = (uint32_t)o.coup[m].size();
    o.maps[m].couplings = o.coup[m].data();
    o.maps[m].channel_floor = o.chfloor[m].data();
  }
  o.modes.resize(st.setup.modes.size());
  for (size_t k = 0; k < o.modes.size(); ++k) o.modes[k] = vsyn_mode{(uint8_t)(st.setup.modes[k].block_flag ? 1 : 0), st.setup.modes[k].mapping};
  o.su.channels = C;
  o.su.blocksize0 = st.header.get_blocksize_0();
  o.su.blocksize1 = st.header.get_blocksize_1();
  o.su.num_floors = (uint32_t)o.floors.size();
  o.su.floors = o.floors.data();
  o.su.num_mappings = (uint32_t)o.maps.size();
  o.su.mappings = o.maps.data();
  o.su.num_modes = (uint32_t)o.modes.size();
  o.su.modes = o.modes.data();
  std::string& k = o.key;
  auto put = [&k](uint32_t v) { k.append((const char*)&v, 4); };
  k.clear();
  put(C); put(o.su.blocksize0); put(o.su.blocksize1); put(o.su.num_floors); put(o.su.num_mappings); put(o.su.num_modes);
  for (size_t i = 0; i < o.floors.size(); ++i) {
    put(o.floors[i].multiplier);
    put(o.floors[i].num_posts);
    for (uint32_t x : o.xs[i]) put(x);
  }
  for (size_t m = 0; m < o.maps.size(); ++m) {
    put(o.maps[m].num_couplings);
    for (const vsyn_coupling& c : o.coup[m]) put(((uint32_t)c.magnitude << 16) | c.angle);
    for (uint8_t f : o.chfloor[m]) put(f);
  }
  for (const vsyn_mode& m : o.modes) put(((uint32_t)m.block_flag << 8) | m.mapping);

  // residue VQ stage: codebook value tables (copied: the setup outlives the stream in the corpus decoder), residue
  // descriptions, channel -> submap -> residue
  o.has_vq = st.vq_mode_;
  if (o.has_vq) {
    o.books.resize(st.setup.codebooks.size());
    o.book_tables.resize(st.setup.codebooks.size());
    for (size_t i = 0; i < o.books.size(); ++i) {
      const VorbisCodebook& cb = st.setup.codebooks[i];
      o.books[i].dimensions = cb.dimensions_;
      o.books[i].num_entries = cb.num_entries_;
      if (cb.lookup_type_ && !cb.lookup_table_.empty()) o.book_tables[i] = cb.lookup_table_;
      o.books[i].lookup = o.book_tables[i].empty() ? nullptr : o.book_tables[i].data();
    }
    o.residues.resize(st.setup.residues.size());
    o.residue_books.resize(st.setup.residues.size());
    for (size_t i = 0; i < o.residues.size(); ++i) {
      const VorbisResidue& r = st.setup.residues[i];
      o.residue_books[i].assign(r.books.begin(), r.books.end());
      o.residues[i].type = r.type;
      o.residues[i].begin = r.begin;
      o.residues[i].end = r.end;
      o.residues[i].partition_size = r.partition_size;
      o.residues[i].num_classifications = r.num_classifications;
      o.residues[i].classwords = st.setup.codebooks[r.classbook].dimensions_;
      o.residues[i].books = o.residue_books[i].data();
    }
    o.vq_maps.resize(st.setup.mappings.size());
    o.mux.resize(st.setup.mappings.size());
    o.submap_residue.resize(st.setup.mappings.size());
    for (size_t m = 0; m < o.vq_maps.size(); ++m) {
      const VorbisMapping& mp = st.setup.mappings[m];
      o.mux[m].assign(mp.muxs.begin(), mp.muxs.end());
      for (const VorbisMapping::Submap& sm : mp.submaps) o.submap_residue[m].push_back(sm.residue);
      o.vq_maps[m].num_submaps = (uint32_t)mp.submaps.size();
      o.vq_maps[m].mux = o.mux[m].data();
      o.vq_maps[m].submap_residue = o.submap_residue[m].data();
    }
    o.vq.num_codebooks = (uint32_t)o.books.size();
    o.vq.codebooks = o.books.data();
    o.vq.num_residues = (uint32_t)o.residues.size();
    o.vq.residues = o.residues.data();
    o.vq.num_mappings = (uint32_t)o.vq_maps.size();
    o.vq.mappings = o.vq_maps.data();
    // the key covers codebooks and residues through the hash of the setup header packet that defined them
    const uint64_t hsh = st.setup_hash_;
    put(0x56515631u);  // "VQV1"
    put((uint32_t)hsh);
    put((uint32_t)(hsh >> 32));
  }
  return OkOrError();
}

static OkOrError make_synth(VorbisStream& st) {
  SynthSetup ss;
  CHECK_ERR(build_synth_setup(st, ss));
  const char* err = nullptr;
  int dev = 0;
  if (const char* e = getenv("PARSEOGGVORBIS_DEVICE")) dev = atoi(e);
  const int rc = vsyn_create(&ss.su, dev, 1, &st.synth_, &err);
  if (rc != VSYN_OK) return OkOrError(std::string("GPU synthesis layer: ") + (err ? err : "vsyn_create failed"));
  CHECK(st.ys_stride_ == vsyn_ys_stride(st.synth_));
  if (ss.has_vq && vsyn_attach_vq(st.synth_, &ss.vq, &err) != VSYN_OK)
    return OkOrError(std::string("GPU synthesis layer: ") + (err ? err : "vsyn_attach_vq failed"));
  return OkOrError();
}

// ------------------------------------------------------------------------------------------------
// stream: audio packets — entropy half here, synthesis half batched
// ------------------------------------------------------------------------------------------------
OkOrError VorbisStream::parse_audio(const uint8_t* data, uint32_t len, int64_t granule, ParseCallbacks& cb) {
  const uint32_t C = header.audio_channels;
  BitReader reader(data, len);
  CHECK(reader.readBitsT<1>() == 0);  // 4.3.1 packet type: audio
  CHECK(setup.modes.size() > 0);
  const int mode_bits = highest_bit(setup.modes.size() - 1);
  const uint32_t mode_idx = mode_bits ? reader.readBits<uint32_t>(mode_bits) : 0;
  CHECK(mode_idx < setup.modes.size());
  const VorbisModeNumber& mode = setup.modes[mode_idx];
  const VorbisMapping& mapping = setup.mappings[mode.mapping];
  bool prev_flag = false, next_flag = false;
  if (mode.block_flag) {
    prev_flag = reader.readBitsT<1>() != 0;
    next_flag = reader.readBitsT<1>() != 0;
  }
  const uint32_t n = mode.blocksize, n2 = n / 2;

  // A packet that fails half way (a CHECK below) must leave the batch as it found it: the rows and entries it already appended
  // would otherwise make ys_ / residue_ / cls_ / entries_ longer than pk_ accounts for, and whoever packs the batch next
  // (flush, CorpusDecoder's feeders) sizes its buffers from pk_.
  struct Rollback {
    VorbisStream& s;
    const size_t ys, fl, res, cls, ent;
    bool armed = true;
    explicit Rollback(VorbisStream& st) : s(st), ys(st.ys_.size()), fl(st.floor_number_.size()), res(st.residue_.size()), cls(st.cls_.size()), ent(st.entries_.size()) {}
    ~Rollback() {
      if (!armed) return;
      s.ys_.resize(ys);
      s.floor_number_.resize(fl);
      s.residue_.resize(res);
      s.cls_.resize(cls);
      s.entries_.resize(ent);
    }
  } rollback(*this);

  // 4.3.2 floor curve decode: only the coded Y values; the curve is rendered on the GPU
  const size_t row0 = ys_.size();
  ys_.resize(row0 + (size_t)C * ys_stride_, 0);
  std::vector<bool> used(C, false);
  uint32_t own_mask = 0;
  std::vector<uint32_t> ys;
  for (uint32_t ch = 0; ch < C; ++ch) {
    const uint8_t floor_number = mapping.submaps[mapping.muxs[ch]].floor;
    floor_number_.push_back(floor_number);
    const VorbisFloor& floor = setup.floors[floor_number];
    CHECK(floor.floor_type == 1);  // floor 0 is not implemented (nor upstream: hpp:402)
    bool use = false;
    CHECK_ERR(floor.floor1.decode_ys(reader, setup.codebooks, ys, use));
    used[ch] = use;
    if (use) {
      if (ch < 32u) own_mask |= 1u << ch;  // (the synthesis layer takes at most 32 channels; a stream with more never reaches it)
      uint16_t* row = &ys_[row0 + (size_t)ch * ys_stride_];
      for (size_t i = 0; i < ys.size(); ++i) row[i] = (uint16_t)std::min<uint32_t>(ys[i], 0xffffu);
    }
  }
  // 4.3.3 nonzero vector propagate (also done on the device for the floor product; here it gates the residue decode)
  for (const VorbisMapping::Coupling& c : mapping.couplings)
    if (used[(size_t)c.angle] || used[(size_t)c.magintude]) used[(size_t)c.angle] = used[(size_t)c.magintude] = true;

  // 4.3.4 residue decode: straight into the batch buffer ("after_residue"), or — VQ mode — only its bit-serial half:
  // classification and entry numbers; the device looks the vectors up and adds them
  const size_t res0 = residue_.size();
  vsyn_vq_packet vqp;
  memset(&vqp, 0, sizeof(vqp));
  if (vq_mode_) {
    vqp.entry_off = entries_.size();
    CHECK(cls_.size() < 0xffffffffu);
    vqp.cls_off = (uint32_t)cls_.size();
  } else {
    residue_.resize(res0 + (size_t)C * n2, 0.f);
  }
  for (size_t s = 0; s < mapping.submaps.size(); ++s) {
    std::vector<float*> outs;
    std::vector<bool> ch_used;
    for (uint32_t ch = 0; ch < C; ++ch)
      if (mapping.muxs[ch] == s) {
        outs.push_back(vq_mode_ ? nullptr : &residue_[res0 + (size_t)ch * n2]);
        ch_used.push_back(used[ch]);
      }
    if (outs.empty()) continue;
    const VorbisResidue& res = setup.residues[mapping.submaps[s].residue];
    if (vq_mode_) CHECK_ERR(res.decode_entries(reader, setup.codebooks, (uint32_t)outs.size(), ch_used, n2, cls_, entries_));
    else CHECK_ERR(res.decode(reader, setup.codebooks, (uint32_t)outs.size(), ch_used, n2, outs.data()));
  }
#ifdef PARSEOGGVORBIS_TESTING
  {
    // Fault injection, compiled into the TESTING build of the library only (libparseoggvorbis_amd_testing.so, host/Makefile; the
    // product library holds none of this): the k-th audio packet of every stream fails HERE, after its floor rows and residue have
    // been appended — the worst place for the batch bookkeeping (a lookup-type-0 book in a residue, a floor-0 submap or a partition
    // overshoot fail at the same depth, but no fixture holds one).
    static const long fail_at = getenv("PARSEOGGVORBIS_TEST_FAIL_AT") ? atol(getenv("PARSEOGGVORBIS_TEST_FAIL_AT")) : -1;
    if (fail_at >= 0 && (long)(packets_seen_++) == fail_at) CHECK(false && "injected failure (PARSEOGGVORBIS_TEST_FAIL_AT)");
  }
#endif
  if (vq_mode_) {
    vqp.num_entries = (uint32_t)(entries_.size() - vqp.entry_off);
    vq_pk_.push_back(vqp);
  }
  residue_floats_ += (size_t)C * n2;

  vsyn_packet pk;
  memset(&pk, 0, sizeof(pk));
  pk.mode = (uint8_t)mode_idx;
  pk.prev_long = prev_flag;
  pk.next_long = next_flag;
  pk.floor_used = own_mask;
  pk.granule = granule;
  pk_.push_back(pk);
  rollback.armed = false;  // the packet is complete: it belongs to the batch now
  if (pk_.size() >= batch_limit_) CHECK_ERR(flush(cb));
  return OkOrError();
}

static std::string status_text(const vsyn_status& st) {
  std::string s = "GPU synthesis flagged audio packet " + std::to_string(st.first_bad_packet) + " of the batch:";
  if (st.flags & VSYN_ST_FLOOR_RANGE) s += " floor prediction out of range (hpp:536)";
  if (st.flags & VSYN_ST_FLOOR_VALUE) s += " floor value >= 256 (hpp:587)";
  if (st.flags & VSYN_ST_GRANULE) s += " page granule inconsistent with the packets (hpp:1029,1041)";
  if (st.flags & VSYN_ST_PLANE_OVERFLOW) s += " pcm plane overflow";
  if (st.flags & VSYN_ST_BAD_MODE) s += " bad mode";
  if (st.flags & VSYN_ST_BAD_SEGMENT) s += " bad segment";
  if (st.flags & VSYN_ST_BAD_VQ) s += " residue entry / classification number out of range";
  return s;
}

OkOrError VorbisStream::flush(ParseCallbacks& cb) {
  if (pk_.empty()) return OkOrError();
  if (sink_) {  // somebody else runs the GPU (e.g. the corpus decoder merges many streams into one submit)
    PacketBatch b;
    b.pk.swap(pk_);
    b.ys.swap(ys_);
    b.residue.swap(residue_);
    b.floor_number.swap(floor_number_);
    b.vq = vq_mode_;
    b.vq_pk.swap(vq_pk_);
    b.cls.swap(cls_);
    b.entries.swap(entries_);
    b.residue_floats = residue_floats_;
    residue_floats_ = 0;
    b.first = first_batch_;
    first_batch_ = false;
    return sink_->consume(*this, std::move(b));
  }
  // A hard failure of the GPU layer below drops the batch: the reader's error path flushes every stream once more so that packets in
  // front of a PARSE error are still delivered (hpp:1045-1054) — after a failure in here that second flush must find nothing to resubmit.
  auto drop_batch = [this]() {
    pk_.clear(); ys_.clear(); residue_.clear(); floor_number_.clear(); vq_pk_.clear(); cls_.clear(); entries_.clear();
    residue_floats_ = 0;
  };
  if (!synth_) {  // the GPU handle is created when the first batch is ready
    OkOrError e = make_synth(*this);
    if (e.is_error_) {
      drop_batch();
      return e;
    }
  }
  const uint32_t C = header.audio_channels, P = (uint32_t)pk_.size();
  const uint32_t bs0 = header.get_blocksize_0(), bs1 = header.get_blocksize_1();
  const uint64_t plane = (uint64_t)P * (bs1 / 2);
  std::vector<float> pcm((size_t)C * plane);
  std::vector<uint32_t> emit(P, 0);
  const bool hooks = decoder_wants_data(this);
  std::vector<float> env, blk;
  std::vector<uint16_t> ffin, fcurve;
  vsyn_taps taps = {nullptr, nullptr, nullptr, nullptr};
  if (vq_mode_ && hooks) residue_.assign(residue_floats_, 0.f);  // the "after_residue" hook needs the floats back from the device
  if (hooks) {
    env.resize(residue_floats_);
    blk.resize(residue_floats_ * 2);
    ffin.resize(ys_.size());
    taps.after_envelope = env.data();
    taps.pcm_after_mdct = blk.data();
    taps.floor_final = ffin.data();
    fcurve.resize(residue_floats_);
    taps.floor_curve = fcurve.data();
  }
  vsyn_segment seg;
  memset(&seg, 0, sizeof(seg));
  seg.stream = 0;
  seg.first_packet = 0;
  seg.num_packets = P;
  seg.flags = first_batch_ ? VSYN_SEG_RESET : 0u;
  seg.residue_off = 0;
  vsyn_status st = {0, 0xffffffffu};
  const char* err = nullptr;
  int rc;
  if (vq_mode_) {
    vsyn_vq_batch vqb;
    vqb.packets = vq_pk_.data();
    vqb.cls = cls_.data();
    vqb.entries = entries_.data();
    vqb.num_cls = cls_.size();
    vqb.num_entries = entries_.size();
    rc = vsyn_submit_host_vq(synth_, P, pk_.data(), 1, &seg, ys_.data(), &vqb, hooks ? residue_.data() : nullptr, residue_floats_, pcm.data(), plane,
                             emit.data(), hooks ? &taps : nullptr, 0, &st, &err);
  } else {
    rc = vsyn_submit_host(synth_, P, pk_.data(), 1, &seg, ys_.data(), residue_.data(), residue_.size(), pcm.data(), plane, emit.data(),
                          hooks ? &taps : nullptr, 0, &st, &err);
  }
  if (rc != VSYN_OK && rc != VSYN_ERR_STREAM) {
    drop_batch();
    return OkOrError(std::string("GPU synthesis layer: ") + (err ? err : "submit failed"));
  }
  const uint32_t good = rc == VSYN_ERR_STREAM ? std::min(P, st.first_bad_packet) : P;

  // replay, packet by packet, exactly the entries upstream pushes between hpp:1139 and hpp:1271
  size_t roff = 0, out_off = 0;
  std::vector<DataRange<const float>> chans(C);
  for (uint32_t q = 0; q < good; ++q) {
    const vsyn_packet& pk = pk_[q];
    const uint32_t n = setup.modes[pk.mode].block_flag ? bs1 : bs0, n2 = n / 2;
    if (hooks) {
      push_data_u8(this, "start_audio_packet", -1, nullptr, 0);
      push_data_u64(this, "abs_total_pos", -1, &abs_total_pos_, 1);
      push_data_i64(this, "expected_ending_total_pos", -1, &pk.granule, 1);
      for (uint32_t ch = 0; ch < C; ++ch) {
        const uint8_t fn = floor_number_[(size_t)q * C + ch];
        push_data_u8(this, "floor_number", (int)ch, &fn, 1);
        if ((pk.floor_used >> ch) & 1u) {
          const VorbisFloor1& f1 = setup.floors[fn].floor1;
          const size_t posts = f1.xs.size();
          const uint16_t* yrow = &ys_[((size_t)q * C + ch) * ys_stride_];
          const uint16_t* frow = &ffin[((size_t)q * C + ch) * ys_stride_];
          std::vector<uint32_t> y32(posts), fy(posts);
          std::vector<bool> flag(posts);
          for (size_t i = 0; i < posts; ++i) {
            y32[i] = yrow[i];
            fy[i] = (frow[i] & 0x7fffu) / f1.multiplier;  // tap holds final_y * multiplier
            flag[i] = (frow[i] >> 15) != 0;
          }
          push_data_u32(this, "floor1 ys", -1, y32.data(), posts);
          push_data_u32(this, "floor1 final_ys", -1, fy.data(), posts);
          push_data_bool(this, "floor1 step2_flag", -1, flag);
          // "floor1 floor" (hpp:585): n values upstream; the device tap holds the first n/2. The rest is flat (hpp:583-584
          // extends the last flagged post's y to the end of the vector): the y of the post at x = n/2 (header index 1) if it
          // is flagged, else the extension began earlier and bin n/2-1 already carries it
          {
            std::vector<uint32_t> curve(n);
            const uint16_t* crow = &fcurve[roff + (size_t)ch * n2];
            for (uint32_t i = 0; i < n2; ++i) curve[i] = crow[i];
            const uint32_t tail = (frow[1] >> 15) ? (uint32_t)(frow[1] & 0x7fffu) : (uint32_t)crow[n2 - 1];
            for (uint32_t i = n2; i < n; ++i) curve[i] = tail;
            push_data_u32(this, "floor1 floor", -1, curve.data(), n);
            // "floor_outputs" (hpp:1171): the curve through the inverse-dB table (Vorbis I 10.1, hpp:588)
            std::vector<float> fo(n);
            for (uint32_t i = 0; i < n; ++i) {
              const uint32_t bits = k_inverse_db_bits[curve[i] < 256u ? curve[i] : 255u];
              memcpy(&fo[i], &bits, 4);
            }
            push_data_float(this, "floor_outputs", (int)ch, fo.data(), n);
          }
        }
      }
      for (uint32_t ch = 0; ch < C; ++ch) push_data_float(this, "after_residue", (int)ch, &residue_[roff + (size_t)ch * n2], n2);
      for (uint32_t ch = 0; ch < C; ++ch) push_data_float(this, "after_envelope", (int)ch, &env[roff + (size_t)ch * n2], n2);
      for (uint32_t ch = 0; ch < C; ++ch) push_data_float(this, "pcm_after_mdct", (int)ch, &blk[2 * roff + (size_t)ch * n], n);
      push_data_u8(this, "finish_audio_packet", -1, nullptr, 0);
    }
    const uint32_t frames = emit[q];
    if (frames) {
      for (uint32_t ch = 0; ch < C; ++ch) {
        chans[ch] = DataRange<const float>(&pcm[(size_t)ch * plane + out_off], frames);
        if (hooks) push_data_float(this, "pcm", (int)ch, chans[ch].begin(), frames);
      }
      if (!cb.gotPcmData(chans)) {  // the sink asked to stop (hpp:1053): drop the rest of the batch, so that nothing is replayed twice
        pk_.clear(); ys_.clear(); residue_.clear(); floor_number_.clear(); vq_pk_.clear(); cls_.clear(); entries_.clear();
        residue_floats_ = 0;
        first_batch_ = false;
        CHECK(false && "gotPcmData returned false");
      }
      abs_total_pos_ += frames;
      out_off += frames;
    }
    roff += (size_t)C * n2;
  }
  pk_.clear();
  ys_.clear();
  residue_.clear();
  floor_number_.clear();
  vq_pk_.clear();
  cls_.clear();
  entries_.clear();
  residue_floats_ = 0;
  first_batch_ = false;
  if (rc == VSYN_ERR_STREAM) return OkOrError(status_text(st));
  return OkOrError();
}

// ------------------------------------------------------------------------------------------------
// Ogg container (framing: https://xiph.org/vorbis/doc/framing.html)
// ------------------------------------------------------------------------------------------------
OkOrError OggReader::open_file(const std::string& filename) { return set_reader(std::make_shared<FileReader>(filename)); }

OkOrError OggReader::set_reader(const std::shared_ptr<IReader>& reader) {
  reader_ = reader;
  CHECK_ERR(reader_->isValid());
  return OkOrError();
}

OkOrError OggReader::read_next_page(bool& reached_eof) {
  CHECK(reader_.get());
  uint8_t hdr[27];
  if (reader_->read(hdr, sizeof(hdr), 1) != 1) {
    if (reader_->reachedEnd()) {
      reached_eof = true;
      return OkOrError();
    }
    return OkOrError("read error");
  }
  CHECK(memcmp(hdr, "OggS", 4) == 0);
  CHECK(hdr[4] == 0);  // stream structure version
  const uint8_t flags = hdr[5];
  int64_t granule = 0;
  for (int i = 7; i >= 0; --i) granule = (int64_t)(((uint64_t)granule << 8) | hdr[6 + i]);
  const uint32_t serial = (uint32_t)hdr[14] | ((uint32_t)hdr[15] << 8) | ((uint32_t)hdr[16] << 16) | ((uint32_t)hdr[17] << 24);
  const uint32_t want_crc = (uint32_t)hdr[22] | ((uint32_t)hdr[23] << 8) | ((uint32_t)hdr[24] << 16) | ((uint32_t)hdr[25] << 24);
  const uint32_t nseg = hdr[26];
  uint8_t lacing[256];
  if (nseg) CHECK(reader_->read(lacing, nseg, 1) == 1);
  uint32_t data_len = 0;
  for (uint32_t i = 0; i < nseg; ++i) data_len += lacing[i];
  if (nseg) CHECK(lacing[nseg - 1] != 255);  // packets continued on the next page are not supported (nor upstream: hpp:89)
  std::vector<uint8_t> data(data_len);
  if (data_len) CHECK(reader_->read(data.data(), data_len, 1) == 1);
  memset(hdr + 22, 0, 4);  // the checksum field counts as zero
  uint32_t crc = update_crc(0, hdr, sizeof(hdr));
  crc = update_crc(crc, lacing, nseg);
  crc = update_crc(crc, data.data(), data_len);
  CHECK(want_crc == crc);

  if (flags & HeaderFlag_First) {
    CHECK(streams_.find(serial) == streams_.end());
    streams_[serial].reset(new VorbisStream());
    streams_[serial]->sink_ = sink_;
    streams_[serial]->setup_cache_ = setup_cache_;
    if (sink_) sink_->prepare(*streams_[serial]);
    if (batch_limit_override_) streams_[serial]->batch_limit_ = batch_limit_override_;
  }
  auto it = streams_.find(serial);
  CHECK(it != streams_.end());
  VorbisStream& stream = *it->second;

  uint32_t off = 0, len = 0;
  for (uint32_t s = 0; s < nseg; ++s) {
    len += lacing[s];
    if (lacing[s] == 255) continue;
    const uint8_t* pkt = data.data() + off;
    if (stream.packet_counts_ == 0) CHECK_ERR(stream.parse_id(pkt, len, callbacks_));
    else if (stream.packet_counts_ == 1) CHECK_ERR(stream.parse_comment(pkt, len, callbacks_));
    else if (stream.packet_counts_ == 2) CHECK_ERR(stream.parse_setup(pkt, len, callbacks_));
    else {
      // the page granule belongs to the last packet finishing on the page (hpp:1456-1459)
      CHECK_ERR(stream.parse_audio(pkt, len, s == nseg - 1 ? granule : -1, callbacks_));
      ++stream.audio_packet_counts_;
    }
    ++stream.packet_counts_;
    ++packet_counts_;
    off += len;
    len = 0;
  }
  CHECK(len == 0 && off == data_len);

  if (flags & HeaderFlag_Last) {
    CHECK_ERR(stream.flush(callbacks_));
    CHECK(callbacks_.gotEof());
    streams_.erase(it);
  }
  return OkOrError();
}

OkOrError OggReader::read_until_end() {
  bool eof = false;
  while (!eof) {
    const OkOrError r = read_next_page(eof);
    if (r.is_error_) {
      // Upstream hands out PCM packet by packet, so everything before the failing packet has reached gotPcmData and the hooks
      // when its CHECK fires (hpp:1045-1054). Here those packets may still sit in a batch (the failing one was rolled back by
      // parse_audio): synthesise and deliver them, then report the original error.
      for (auto& kv : streams_) (void)kv.second->flush(callbacks_);
      return r;
    }
  }
  for (auto& kv : streams_) CHECK_ERR(kv.second->flush(callbacks_));  // streams without an end-of-stream page
  return OkOrError();
}

OkOrError OggReader::full_read(const std::string& filename) {
  CHECK_ERR(open_file(filename));
  return read_until_end();
}

OkOrError OggReader::full_read_from_memory(const uint8_t* data, size_t data_len) {
  CHECK_ERR(set_reader(std::make_shared<ConstDataReader>(data, data_len)));
  return read_until_end();
}

// ------------------------------------------------------------------------------------------------
// C entry points (reference: src/ParseOggVorbis.cpp:12-42)
// ------------------------------------------------------------------------------------------------
static int finish(const OkOrError& r, const char** error_out) {
  if (!r.is_error_) return 0;
  if (error_out) {
    static char buf[255];
    strncpy(buf, r.err_msg_.c_str(), sizeof(buf));
    buf[sizeof(buf) - 1] = 0;
    *error_out = buf;
  }
  return 1;
}

extern "C" int ogg_vorbis_full_read(const char* filename, const char** error_out) {
  ParseCallbacks cb;
  OggReader reader(cb);
  return finish(reader.full_read(filename), error_out);
}

extern "C" int ogg_vorbis_full_read_from_memory(const char* data, size_t data_len, const char** error_out) {
  ParseCallbacks cb;
  OggReader reader(cb);
  return finish(reader.full_read_from_memory((const uint8_t*)data, data_len), error_out);
}
