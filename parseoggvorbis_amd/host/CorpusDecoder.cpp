// CorpusDecoder.cpp — see CorpusDecoder.hpp.
#include "CorpusDecoder.hpp"

#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One file after the entropy half.
struct FileRecord {
  void recycle() {  // keep the vectors' capacity: fresh multi-MB allocations per file serialise the workers in the kernel's mm
    status = OkOrError();
    batch.pk.clear();
    batch.ys.clear();
    batch.residue.clear();
    batch.floor_number.clear();
    batch.vq_pk.clear();
    batch.cls.clear();
    batch.entries.clear();
    batch.vq = false;
    batch.residue_floats = 0;
    batch.first = true;
    ys_stride = 0;
    has_audio = false;
    synth = SynthSetup();
  }
  size_t index = 0;
  OkOrError status;
  VorbisIdHeader header;
  SynthSetup synth;
  PacketBatch batch;
  uint32_t ys_stride = 0;
  bool has_audio = false;
};

// Collects the (single) batch of the (single) logical stream of one file.
struct CollectSink : SynthSink {
  FileRecord& rec;
  const VorbisStream* owner = nullptr;
  explicit CollectSink(FileRecord& r) : rec(r) {}
  void prepare(VorbisStream& st) override {  // lend the record's (recycled, capacity-keeping) vectors to the first stream
    if (owner) return;
    owner = &st;
    st.pk_.swap(rec.batch.pk);
    st.ys_.swap(rec.batch.ys);
    st.residue_.swap(rec.batch.residue);
    st.floor_number_.swap(rec.batch.floor_number);
    st.vq_pk_.swap(rec.batch.vq_pk);
    st.cls_.swap(rec.batch.cls);
    st.entries_.swap(rec.batch.entries);
  }
  OkOrError consume(VorbisStream& st, PacketBatch&& b) override {
    if (owner && owner != &st) return OkOrError("corpus path: files with more than one logical Vorbis stream are not supported");
    if (rec.has_audio) return OkOrError("corpus path: stream delivered in more than one batch");
    owner = &st;
    rec.header = st.header;
    rec.ys_stride = st.ys_stride_;
    CHECK_ERR(build_synth_setup(st, rec.synth));
    rec.batch = std::move(b);
    rec.has_audio = true;
    return OkOrError();
  }
};

struct NullCallbacks : ParseCallbacks {};

void entropy_decode_file(const CorpusItem& item, FileRecord& rec, SetupCache* cache) {
  NullCallbacks cb;
  CollectSink sink(rec);
  OggReader reader(cb);
  reader.sink_ = &sink;
  reader.setup_cache_ = cache;
  reader.batch_limit_override_ = 0xffffffffu;  // the whole file is one batch; it is cut into runs on the GPU
  rec.status = reader.full_read_from_memory(item.data, item.len);
  if (rec.status.is_error_)  // keep what was decoded before the failure, as the reference's gotPcmData calls would have
    for (auto& kv : reader.streams_) {
      const OkOrError r = kv.second->flush(cb);
      if (r.is_error_) break;
    }
}

// Bounded hand-off between the workers and the feeder.
struct RecordQueue {
  std::mutex mu;
  std::condition_variable not_empty, not_full;
  std::deque<std::unique_ptr<FileRecord>> q;
  std::vector<std::unique_ptr<FileRecord>> free_list;
  size_t cap;
  size_t producers;
  bool aborted = false;
  RecordQueue(size_t cap_, size_t producers_) : cap(cap_), producers(producers_) {}
  bool push(std::unique_ptr<FileRecord> r) {
    std::unique_lock<std::mutex> lk(mu);
    not_full.wait(lk, [&] { return q.size() < cap || aborted; });
    if (aborted) return false;
    q.push_back(std::move(r));
    not_empty.notify_one();
    return true;
  }
  std::unique_ptr<FileRecord> fresh() {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_list.empty()) {
        std::unique_ptr<FileRecord> r = std::move(free_list.back());
        free_list.pop_back();
        return r;
      }
    }
    return std::unique_ptr<FileRecord>(new FileRecord());
  }
  void give_back(std::unique_ptr<FileRecord> r) {
    r->recycle();
    std::lock_guard<std::mutex> lk(mu);
    free_list.push_back(std::move(r));
  }
  void producer_done() {
    std::lock_guard<std::mutex> lk(mu);
    --producers;
    not_empty.notify_all();
  }
  // nullptr: all producers are done and the queue is drained
  std::unique_ptr<FileRecord> pop() {
    std::unique_lock<std::mutex> lk(mu);
    not_empty.wait(lk, [&] { return !q.empty() || producers == 0 || aborted; });
    if (q.empty() || aborted) return nullptr;
    std::unique_ptr<FileRecord> r = std::move(q.front());
    q.pop_front();
    not_full.notify_one();
    return r;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    not_full.notify_all();
    not_empty.notify_all();
  }
};

// Grow-only page-locked array (contents are rebuilt for every submit, nothing is preserved on growth).
template <typename T>
struct PinnedArray {
  T* p = nullptr;
  size_t cap = 0;
  PinnedArray() {}
  PinnedArray(const PinnedArray&) = delete;
  PinnedArray& operator=(const PinnedArray&) = delete;
  ~PinnedArray() { vsyn_host_free(p); }
  OkOrError ensure(size_t n) {
    if (n <= cap) return OkOrError();
    vsyn_host_free(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 64;
    const char* err = nullptr;
    void* q = nullptr;
    if (vsyn_host_alloc(want * sizeof(T), &q, &err) != VSYN_OK) return OkOrError(std::string("GPU synthesis layer: ") + (err ? err : "host alloc failed"));
    p = (T*)q;
    cap = want;
    return OkOrError();
  }
  T& operator[](size_t i) { return p[i]; }
};

// Files that share one synthesis setup, the handle serving them and the submit buffers (reused between submits).
struct Group {
  vsyn_handle* handle = nullptr;
  uint32_t channels = 0, bs1 = 0, ys_stride = 0;
  std::vector<std::unique_ptr<FileRecord>> pending;
  PinnedArray<vsyn_packet> pk;
  PinnedArray<vsyn_segment> seg;
  PinnedArray<uint16_t> ys;
  PinnedArray<float> residue, pcm;
  PinnedArray<int16_t> pcm16;  // CorpusOptions::pcm_s16: [S][plane][C]
  PinnedArray<uint32_t> emit;
  bool vq = false;  // files of this group ship classification + entry numbers instead of residue floats
  PinnedArray<vsyn_vq_packet> vq_pk;
  PinnedArray<uint8_t> cls;
  PinnedArray<uint16_t> entries;
  ~Group() {
    if (handle) vsyn_destroy(handle);
  }
};

// sum |x| in double, 8 independent partial sums so that the compiler can keep it in vector registers
double abs_sum_f32(const float* x, uint64_t n) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; ++k) acc[k] += (double)__builtin_fabsf(x[i + k]);
  double a = 0;
  for (; i < n; ++i) a += (double)__builtin_fabsf(x[i]);
  for (int k = 0; k < 8; ++k) a += acc[k];
  return a;
}

std::string gpu_status_text(const vsyn_status& st) {
  std::string s = "GPU synthesis check failed (flags";
  if (st.flags & VSYN_ST_FLOOR_RANGE) s += " floor-range";
  if (st.flags & VSYN_ST_FLOOR_VALUE) s += " floor-value";
  if (st.flags & VSYN_ST_GRANULE) s += " granule";
  if (st.flags & VSYN_ST_PLANE_OVERFLOW) s += " plane-overflow";
  if (st.flags & VSYN_ST_BAD_MODE) s += " bad-mode";
  if (st.flags & VSYN_ST_BAD_SEGMENT) s += " bad-segment";
  return s + ")";
}

struct Feeder {
  const CorpusOptions& opts;
  CorpusCallbacks* callbacks;
  std::vector<CorpusFileResult>& results;
  CorpusStats& stats;
  RecordQueue& queue;
  std::mutex& callbacks_mu;
  std::map<std::string, std::unique_ptr<Group>> groups;

  OkOrError submit(Group& g) {
    if (g.pending.empty()) return OkOrError();
    const uint32_t C = g.channels, S = (uint32_t)g.pending.size();
    double t0 = now_s();
    size_t P = 0, rfloats = 0, ncls = 0, nent = 0;
    uint32_t max_p = 0;
    for (const auto& r : g.pending) {
      P += r->batch.pk.size();
      rfloats += r->batch.residue_floats;
      ncls += r->batch.cls.size();
      nent += r->batch.entries.size();
      max_p = std::max<uint32_t>(max_p, (uint32_t)r->batch.pk.size());
    }
    CHECK(ncls < 0xffffffffu);
    CHECK(P < 0xffffffffu);
    const uint64_t plane = (uint64_t)max_p * (g.bs1 / 2);
    CHECK_ERR(g.pk.ensure(P));
    CHECK_ERR(g.seg.ensure(S));
    CHECK_ERR(g.ys.ensure(P * C * g.ys_stride));
    if (g.vq) {
      CHECK_ERR(g.vq_pk.ensure(P));
      CHECK_ERR(g.cls.ensure(ncls));
      CHECK_ERR(g.entries.ensure(nent));
    } else {
      CHECK_ERR(g.residue.ensure(rfloats));
    }
    CHECK_ERR(g.emit.ensure(P));
    if (opts.pcm_s16) CHECK_ERR(g.pcm16.ensure((size_t)S * C * plane));
    else CHECK_ERR(g.pcm.ensure((size_t)S * C * plane));
    size_t p0 = 0, r0 = 0, c0 = 0, e0 = 0;
    for (uint32_t s = 0; s < S; ++s) {
      const PacketBatch& b = g.pending[s]->batch;
      CHECK(b.vq == g.vq);
      // the staging arrays were sized from pk / residue_floats: a batch whose per-packet vectors disagree with them (a packet
      // that failed half way and was not rolled back) must not be copied
      CHECK(b.ys.size() == b.pk.size() * C * g.ys_stride);
      CHECK(b.vq || b.residue.size() == b.residue_floats);
      memcpy(&g.pk[p0], b.pk.data(), b.pk.size() * sizeof(vsyn_packet));
      memcpy(&g.ys[p0 * C * g.ys_stride], b.ys.data(), b.ys.size() * sizeof(uint16_t));
      if (g.vq) {
        CHECK(b.vq_pk.size() == b.pk.size());
        for (size_t q = 0; q < b.vq_pk.size(); ++q) {  // rebase the file's offsets into the merged arrays
          vsyn_vq_packet v = b.vq_pk[q];
          v.entry_off += e0;
          v.cls_off += (uint32_t)c0;
          g.vq_pk[p0 + q] = v;
        }
        memcpy(&g.cls[c0], b.cls.data(), b.cls.size());
        memcpy(&g.entries[e0], b.entries.data(), b.entries.size() * sizeof(uint16_t));
        c0 += b.cls.size();
        e0 += b.entries.size();
      } else {
        memcpy(&g.residue[r0], b.residue.data(), b.residue.size() * sizeof(float));
      }
      vsyn_segment& sg = g.seg[s];
      memset(&sg, 0, sizeof(sg));
      sg.stream = s;
      sg.first_packet = (uint32_t)p0;
      sg.num_packets = (uint32_t)b.pk.size();
      sg.flags = VSYN_SEG_RESET;
      sg.residue_off = r0;
      p0 += b.pk.size();
      r0 += b.residue_floats;
    }
    double t1 = now_s();
    stats.pack_s += t1 - t0;
    vsyn_status st = {0, 0xffffffffu};
    const char* err = nullptr;
    int rc;
    const uint32_t sflags = opts.pcm_s16 ? VSYN_SUBMIT_KEEP_PCM : 0u;
    if (g.vq) {
      vsyn_vq_batch vqb;
      vqb.packets = g.vq_pk.p;
      vqb.cls = g.cls.p;
      vqb.entries = g.entries.p;
      vqb.num_cls = ncls;
      vqb.num_entries = nent;
      rc = vsyn_submit_host_vq(g.handle, (uint32_t)P, g.pk.p, S, g.seg.p, g.ys.p, &vqb, nullptr, rfloats, opts.pcm_s16 ? nullptr : g.pcm.p, plane, g.emit.p, nullptr, sflags, &st, &err);
    } else {
      rc = vsyn_submit_host(g.handle, (uint32_t)P, g.pk.p, S, g.seg.p, g.ys.p, g.residue.p, rfloats, opts.pcm_s16 ? nullptr : g.pcm.p, plane, g.emit.p, nullptr, sflags, &st, &err);
    }
    if (opts.pcm_s16 && rc == VSYN_OK) {
      const char* ferr = nullptr;
      if (vsyn_pcm_fetch_host(g.handle, VSYN_PCM_S16, g.pcm16.p, plane, nullptr, &ferr) != VSYN_OK)
        return OkOrError(std::string("GPU synthesis layer: ") + (ferr ? ferr : "pcm fetch failed"));
    }
    // per-(file, channel) digests from the device, where the PCM still is (a host pass over it cost more than the decode's GPU calls)
    std::vector<double> digest;
    if (opts.checksum && rc == VSYN_OK) {
      digest.resize((size_t)S * C);
      const char* derr = nullptr;
      if (vsyn_pcm_abs_sum_host(g.handle, digest.data(), &derr) != VSYN_OK)
        return OkOrError(std::string("GPU synthesis layer: ") + (derr ? derr : "digest failed"));
    }
    double t2 = now_s();
    stats.gpu_call_s += t2 - t1;
    stats.submits++;
    if (rc != VSYN_OK && rc != VSYN_ERR_STREAM) return OkOrError(std::string("GPU synthesis layer: ") + (err ? err : "submit failed"));
    if (rc == VSYN_ERR_STREAM && S > 1) {
      // Some file of the batch is bad; the status does not say which beyond the first. Re-run the files one by one so that
      // every good file still gets its PCM and every bad one its own message.
      std::vector<std::unique_ptr<FileRecord>> files;
      files.swap(g.pending);
      for (auto& f : files) {
        g.pending.clear();
        g.pending.push_back(std::move(f));
        CHECK_ERR(submit(g));
      }
      return OkOrError();
    }
    // deliver
    std::vector<DataRange<const float>> chans(C);
    p0 = 0;
    for (uint32_t s = 0; s < S; ++s) {
      FileRecord& r = *g.pending[s];
      CorpusFileResult& out = results[r.index];
      out.channels = C;
      out.sample_rate = r.header.audio_sample_rate;
      out.audio_packets = (uint32_t)r.batch.pk.size();
      if (rc == VSYN_ERR_STREAM) {
        out.status = OkOrError(gpu_status_text(st));
      } else {
        uint64_t frames = 0;
        for (size_t q = 0; q < r.batch.pk.size(); ++q) frames += g.emit[p0 + q];
        CHECK(frames <= plane);
        double acc = 0;
        for (uint32_t c = 0; c < C; ++c) {
          if (!opts.pcm_s16) {
            const float* x = &g.pcm[((size_t)s * C + c) * plane];
            chans[c] = DataRange<const float>(x, frames);
            if (opts.checksum && digest.empty()) acc += abs_sum_f32(x, frames);
          }
          if (opts.checksum && !digest.empty()) acc += digest[(size_t)s * C + c];
        }
        out.frames = frames;
        out.abs_sum = acc;
        out.status = r.status;
        stats.frames += frames;
        if (callbacks) {
          std::lock_guard<std::mutex> lk(callbacks_mu);
          if (opts.pcm_s16) {
            if (!callbacks->gotFilePcmS16(r.index, r.header, &g.pcm16[(size_t)s * plane * C], frames)) return OkOrError("aborted by gotFilePcmS16");
          } else if (!callbacks->gotFilePcm(r.index, r.header, chans)) {
            return OkOrError("aborted by gotFilePcm");
          }
        }
      }
      stats.audio_packets += r.batch.pk.size();
      stats.files++;
      p0 += r.batch.pk.size();
    }
    for (auto& r : g.pending) queue.give_back(std::move(r));
    g.pending.clear();
    stats.deliver_s += now_s() - t2;
    return OkOrError();
  }

  OkOrError take(std::unique_ptr<FileRecord> rec) {
    CorpusFileResult& out = results[rec->index];
    if (!rec->has_audio) {  // failed before any audio, or a file without audio packets
      out.status = rec->status;
      out.channels = rec->status.is_error_ ? 0 : rec->header.audio_channels;
      stats.files++;
      queue.give_back(std::move(rec));
      return OkOrError();
    }
    if (opts.entropy_only) {
      out.status = rec->status;
      out.channels = rec->header.audio_channels;
      out.sample_rate = rec->header.audio_sample_rate;
      out.audio_packets = (uint32_t)rec->batch.pk.size();
      stats.audio_packets += rec->batch.pk.size();
      stats.files++;
      queue.give_back(std::move(rec));
      return OkOrError();
    }
    // A file that failed half-way (truncated, corrupt packet) still delivers what was decoded before the failure, as the
    // reference does through gotPcmData before its CHECK fires; the error is kept in the result.
    std::unique_ptr<Group>& gp = groups[rec->synth.key];
    if (!gp) {
      gp.reset(new Group());
      const char* err = nullptr;
      const int rc = vsyn_create(&rec->synth.su, opts.device, opts.files_per_submit, &gp->handle, &err);
      if (rc != VSYN_OK) {
        // a setup the GPU layer rejects is this file's problem; no device at all is everybody's
        std::string msg = std::string("GPU synthesis layer: ") + (err ? err : "vsyn_create failed");
        groups.erase(rec->synth.key);
        if (rc == VSYN_ERR_NO_DEVICE || rc == VSYN_ERR_HIP) return OkOrError(msg);
        out.status = OkOrError(msg);
        stats.files++;
        return OkOrError();
      }
      if (rec->synth.has_vq && vsyn_attach_vq(gp->handle, &rec->synth.vq, &err) != VSYN_OK) {
        std::string msg = std::string("GPU synthesis layer: ") + (err ? err : "vsyn_attach_vq failed");
        groups.erase(rec->synth.key);
        return OkOrError(msg);
      }
      gp->vq = rec->synth.has_vq;
      gp->channels = rec->header.audio_channels;
      gp->bs1 = rec->header.get_blocksize_1();
      gp->ys_stride = vsyn_ys_stride(gp->handle);
      stats.handles++;
    }
    CHECK(gp->ys_stride == rec->ys_stride);
    gp->pending.push_back(std::move(rec));
    if (gp->pending.size() >= opts.files_per_submit) CHECK_ERR(submit(*gp));
    return OkOrError();
  }

  OkOrError finish() {
    for (auto& kv : groups) CHECK_ERR(submit(*kv.second));
    return OkOrError();
  }
};

}  // namespace

OkOrError decode_corpus(const std::vector<CorpusItem>& items, const CorpusOptions& opts_in, CorpusCallbacks* callbacks,
                        std::vector<CorpusFileResult>& results, CorpusStats* stats_out) {
  CorpusOptions opts = opts_in;
  if (opts.threads <= 0) opts.threads = (int)std::max(1u, std::thread::hardware_concurrency());
  if (opts.feeders <= 0) opts.feeders = 3;
  if (opts.files_per_submit == 0) opts.files_per_submit = 64;
  if (opts.max_pending_files == 0) opts.max_pending_files = 4 * opts.files_per_submit;
  results.assign(items.size(), CorpusFileResult());
  CorpusStats stats;
  const double t_start = now_s();

  SetupCache setup_cache;
  RecordQueue queue(opts.max_pending_files, (size_t)opts.threads);
  std::atomic<size_t> next(0);
  std::atomic<bool> stop(false);
  std::vector<double> worker_cpu((size_t)opts.threads, 0.0);
  std::vector<std::thread> workers;
  for (int t = 0; t < opts.threads; ++t) {
    workers.emplace_back([&, t] {
      for (;;) {
        if (stop.load(std::memory_order_relaxed)) break;
        const size_t i = next.fetch_add(1);
        if (i >= items.size()) break;
        std::unique_ptr<FileRecord> rec = queue.fresh();
        rec->index = i;
        const double t0 = now_s();
        entropy_decode_file(items[i], *rec, opts.share_setups ? &setup_cache : nullptr);
        worker_cpu[(size_t)t] += now_s() - t0;
        if (!queue.push(std::move(rec))) break;
      }
      queue.producer_done();
    });
  }

  // feeder lanes: each owns its handles (one per synthesis setup it meets) and its page-locked submit buffers
  std::mutex callbacks_mu, status_mu;
  OkOrError run_status;
  std::vector<CorpusStats> lane_stats((size_t)opts.feeders);
  std::vector<std::thread> feeders;
  for (int f = 0; f < opts.feeders; ++f) {
    feeders.emplace_back([&, f] {
      Feeder feeder{opts, callbacks, results, lane_stats[(size_t)f], queue, callbacks_mu, {}};
      OkOrError st;
      for (;;) {
        std::unique_ptr<FileRecord> rec = queue.pop();
        if (!rec) break;
        st = feeder.take(std::move(rec));
        if (st.is_error_) break;
      }
      if (!st.is_error_ && !stop.load()) st = feeder.finish();
      if (st.is_error_) {
        stop.store(true);
        queue.abort();
        std::lock_guard<std::mutex> lk(status_mu);
        if (!run_status.is_error_) run_status = st;
      }
    });
  }
  for (std::thread& w : workers) w.join();
  for (std::thread& f : feeders) f.join();
  for (const CorpusStats& l : lane_stats) {
    stats.gpu_call_s += l.gpu_call_s;
    stats.pack_s += l.pack_s;
    stats.deliver_s += l.deliver_s;
    stats.submits += l.submits;
    stats.files += l.files;
    stats.audio_packets += l.audio_packets;
    stats.frames += l.frames;
    stats.handles += l.handles;
  }

  stats.setup_parses = setup_cache.misses;
  stats.setup_reuses = setup_cache.hits;
  stats.wall_s = now_s() - t_start;
  for (double c : worker_cpu) stats.entropy_cpu_s += c;
  if (stats_out) *stats_out = stats;
  return run_status;
}

extern "C" int ogg_vorbis_decode_corpus(const uint8_t* const* datas, const size_t* lens, size_t num_files, int threads, int feeders,
                                        uint32_t files_per_submit, int device, uint64_t* frames_out, double* abs_sum_out, uint8_t* ok_out, float* const* pcm_out,
                                        const uint64_t* pcm_capacity, double* stats_out, const char** error_out) {
  static char error_buf[256];
  std::vector<CorpusItem> items(num_files);
  for (size_t i = 0; i < num_files; ++i) items[i] = CorpusItem{datas[i], lens[i]};
  CorpusOptions opts;
  opts.threads = threads;
  opts.feeders = feeders;
  opts.files_per_submit = files_per_submit;
  opts.device = device;
  std::vector<CorpusFileResult> results;
  CorpusStats st;
  struct CopyOut : CorpusCallbacks {
    float* const* pcm_out;
    const uint64_t* cap;
    std::vector<uint8_t> too_long;
    bool gotFilePcm(size_t i, const VorbisIdHeader&, const std::vector<DataRange<const float>>& ch) override {
      if (!pcm_out || !pcm_out[i]) return true;
      for (size_t c = 0; c < ch.size(); ++c) {
        if (ch[c].size() > cap[i]) {
          too_long[i] = 1;
          return true;
        }
        memcpy(pcm_out[i] + c * cap[i], ch[c].begin(), ch[c].size() * sizeof(float));
      }
      return true;
    }
  } copy_out;
  copy_out.pcm_out = pcm_out;
  copy_out.cap = pcm_capacity;
  copy_out.too_long.assign(num_files, 0);
  OkOrError r = decode_corpus(items, opts, pcm_out && pcm_capacity ? &copy_out : nullptr, results, &st);
  for (size_t i = 0; i < results.size() && i < num_files; ++i) {
    if (frames_out) frames_out[i] = results[i].frames;
    if (abs_sum_out) abs_sum_out[i] = results[i].abs_sum;
    if (ok_out) ok_out[i] = results[i].status.is_error_ || copy_out.too_long[i] ? 0 : 1;
  }
  if (stats_out) {
    stats_out[0] = st.wall_s;
    stats_out[1] = st.entropy_cpu_s;
    stats_out[2] = st.gpu_call_s;
    stats_out[3] = st.pack_s;
    stats_out[4] = st.deliver_s;
    stats_out[5] = (double)st.submits;
    stats_out[6] = (double)st.audio_packets;
    stats_out[7] = (double)st.frames;
  }
  if (r.is_error_) {
    strncpy(error_buf, r.err_msg_.c_str(), sizeof(error_buf) - 1);
    error_buf[sizeof(error_buf) - 1] = 0;
    if (error_out) *error_out = error_buf;
    return 1;
  }
  if (error_out) *error_out = nullptr;
  return 0;
}

// The same with int16 output (CorpusOptions::pcm_s16): pcm16_out[i] (may be NULL) receives the file's interleaved frames,
// at most pcm_capacity_frames[i] of them (a longer file is marked failed).
extern "C" int ogg_vorbis_decode_corpus_s16(const uint8_t* const* datas, const size_t* lens, size_t num_files, int threads, int feeders,
                                            uint32_t files_per_submit, int device, uint64_t* frames_out, uint8_t* ok_out, int16_t* const* pcm16_out,
                                            const uint64_t* pcm_capacity_frames, double* stats_out, const char** error_out) {
  static char error_buf[256];
  std::vector<CorpusItem> items(num_files);
  for (size_t i = 0; i < num_files; ++i) items[i] = CorpusItem{datas[i], lens[i]};
  CorpusOptions opts;
  opts.threads = threads;
  opts.feeders = feeders;
  opts.files_per_submit = files_per_submit;
  opts.device = device;
  opts.pcm_s16 = true;
  std::vector<CorpusFileResult> results;
  CorpusStats st;
  struct CopyOut : CorpusCallbacks {
    int16_t* const* out;
    const uint64_t* cap;
    std::vector<uint8_t> too_long;
    bool gotFilePcmS16(size_t i, const VorbisIdHeader& h, const int16_t* x, uint64_t frames) override {
      if (!out || !out[i]) return true;
      if (frames > cap[i]) {
        too_long[i] = 1;
        return true;
      }
      memcpy(out[i], x, (size_t)frames * h.audio_channels * sizeof(int16_t));
      return true;
    }
  } copy_out;
  copy_out.out = pcm16_out;
  copy_out.cap = pcm_capacity_frames;
  copy_out.too_long.assign(num_files, 0);
  OkOrError r = decode_corpus(items, opts, pcm16_out && pcm_capacity_frames ? &copy_out : nullptr, results, &st);
  for (size_t i = 0; i < results.size() && i < num_files; ++i) {
    if (frames_out) frames_out[i] = results[i].frames;
    if (ok_out) ok_out[i] = results[i].status.is_error_ || copy_out.too_long[i] ? 0 : 1;
  }
  if (stats_out) {
    const double v[8] = {st.wall_s, st.entropy_cpu_s, st.gpu_call_s, st.pack_s, st.deliver_s, (double)st.submits, (double)st.audio_packets, (double)st.frames};
    for (int i = 0; i < 8; ++i) stats_out[i] = v[i];
  }
  if (r.is_error_) {
    snprintf(error_buf, sizeof(error_buf), "%s", r.err_msg_.c_str());
    if (error_out) *error_out = error_buf;
    return 1;
  }
  return 0;
}
