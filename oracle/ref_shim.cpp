// ref_shim.cpp — TEST INFRASTRUCTURE.  Thin extern "C" wrappers that drive the REFERENCE's own code
// (included from where it lies under /root/reference/src; nothing is copied) so that the oracle
// restatement can be pinned against it function by function.  Built only in the authoring container by
// oracle/Makefile into oracle/_ref/libref_shim.so (git-ignored; travels to the GPU box as a binary).
//
// Exposes: mdct_init/mdct_backward (src/mdct.cpp), VorbisModeNumber::precalc windows (hpp:837-886),
// Utils.hpp render/neighbour helpers, VorbisStreamDecodeState overlap-add (hpp:975-1115) and the compute
// tail of VorbisFloor1::decode (hpp:473-591) fed through a synthetic fixed-length codebook.
#include <cstdint>
#include <cstring>
#include <vector>

#include "ParseOggVorbis.hpp"

namespace {

struct BitWriter { // LSb-first packing, matching BitReader (Utils.hpp:330-424)
	std::vector<uint8_t> bytes;
	int nbits = 0;
	void bit(int b) {
		if((nbits & 7) == 0) bytes.push_back(0);
		if(b) bytes.back() |= uint8_t(1u << (nbits & 7));
		++nbits;
	}
	void bits_lsb_first(uint32_t v, int n) { for(int i = 0; i < n; ++i) bit((v >> i) & 1); }
	void bits_msb_first(uint32_t v, int n) { for(int i = n - 1; i >= 0; --i) bit((v >> i) & 1); }
};

struct CollectPcm : ParseCallbacks {
	std::vector<std::vector<float>> out;
	size_t last_frames = 0;
	bool gotPcmData(const std::vector<DataRange<const float>>& ch) override {
		if(out.size() < ch.size()) out.resize(ch.size());
		for(size_t c = 0; c < ch.size(); ++c) out[c].insert(out[c].end(), ch[c].begin(), ch[c].end());
		last_frames = ch.empty() ? 0 : ch[0].size();
		return true;
	}
};

} // namespace

extern "C" {

void ref_mdct_backward(int n, const float* in, float* out) {
	Mdct m;
	m.init((unsigned)n);
	m.backward(in, out);
}

void ref_mdct_backward_batch(int n, uint32_t count, const float* in, float* out) {
	Mdct m;
	m.init((unsigned)n);
	for(uint32_t i = 0; i < count; ++i) m.backward(in + size_t(i) * (n / 2), out + size_t(i) * n);
}

void ref_mdct_tables(int n, float* trig /* n+n/4 */, int* bitrev /* n/4 */) {
	mdct_lookup l;
	mdct_init(&l, n);
	memcpy(trig, l.trig, sizeof(float) * size_t(n + n / 4));
	memcpy(bitrev, l.bitrev, sizeof(int) * size_t(n / 4));
	mdct_clear(&l);
}

static uint8_t blocksizes_exp(int bs0, int bs1) {
	int e0 = 0, e1 = 0;
	while((1 << e0) < bs0) ++e0;
	while((1 << e1) < bs1) ++e1;
	return uint8_t(e0 | (e1 << 4));
}

// window table exactly as VorbisModeNumber::precalc/getWindow produce it
int ref_window(int bs0, int bs1, int block_flag, int prev, int next, float* out) {
	VorbisIdHeader header;
	memset(&header, 0, sizeof(header));
	header.blocksizes_exp = blocksizes_exp(bs0, bs1);
	VorbisModeNumber mode;
	mode.block_flag = block_flag != 0;
	mode.window_type = mode.transform_type = 0;
	mode.mapping = 0;
	if(mode.precalc(header).is_error_) return 1;
	DataRange<const float> w = mode.getWindow(prev != 0, next != 0);
	memcpy(out, w.begin(), sizeof(float) * w.size());
	return 0;
}

int ref_low_neighbor(const uint32_t* v, int len, int idx) {
	std::vector<uint32_t> vec(v, v + len);
	return (int)(ptrdiff_t)low_neighbor(vec, (size_t)idx);
}
int ref_high_neighbor(const uint32_t* v, int len, int idx) {
	std::vector<uint32_t> vec(v, v + len);
	return (int)(ptrdiff_t)high_neighbor(vec, (size_t)idx);
}
uint32_t ref_render_point(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t X) {
	return render_point<uint32_t>(x0, y0, x1, y1, X);
}
void ref_render_line(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t* vec, uint32_t len) {
	std::vector<uint32_t> v(vec, vec + len);
	render_line<uint32_t>(x0, y0, x1, y1, v);
	memcpy(vec, v.data(), sizeof(uint32_t) * len);
}
const float* ref_inverse_db_table(void) { return inverse_db_table; }

// Overlap-add: feeds blocks[p][c][n_p] (already IMDCT'ed) through the reference decode state.
// win_idx[p] = prev + 2*next for long blocks. Returns 0 ok / 1 on a reference CHECK failure (at packet *bad).
int ref_overlap_add(int channels, int bs0, int bs1, int num_packets, const uint8_t* block_flag, const uint8_t* win_idx,
                    const int64_t* granule, const float* blocks, float* pcm /* [channels][cap] */, uint64_t cap,
                    uint32_t* emit_len, int* bad) {
	VorbisIdHeader header;
	memset(&header, 0, sizeof(header));
	header.blocksizes_exp = blocksizes_exp(bs0, bs1);
	VorbisModeNumber modes[2];
	for(int b = 0; b < 2; ++b) {
		modes[b].block_flag = b;
		modes[b].window_type = modes[b].transform_type = 0;
		modes[b].mapping = 0;
		if(modes[b].precalc(header).is_error_) return 1;
	}
	VorbisStreamDecodeState st;
	st.init((uint8_t)channels, uint32_t(bs0) * 5 + uint32_t(bs1) * 5); // hpp:1354-1359
	register_decoder_ref(&st, "ref_shim", 0, channels); // hooks inside forwardReadyPcm need a registered ref (null sink)
	CollectPcm cb;
	cb.out.resize(channels);
	size_t off = 0;
	int rc = 0;
	for(int p = 0; p < num_packets && !rc; ++p) {
		const VorbisModeNumber& mode = modes[block_flag[p] ? 1 : 0];
		DataRange<const float> window = mode.getWindow(win_idx[p] & 1, win_idx[p] & 2);
		const size_t n = window.size();
		if(st.advancePcmOffsetBeginAudioPacket((uint32_t)n).is_error_) { rc = 1; *bad = p; break; }
		for(int c = 0; c < channels; ++c) {
			if(st.addPcmFrame((uint8_t)c, DataRange<const float>(blocks + off, n), window).is_error_) { rc = 1; *bad = p; break; }
			off += n;
		}
		if(rc) break;
		st.setExpectedEndingPos(granule[p]);
		cb.last_frames = 0;
		if(st.forwardReadyPcm(cb).is_error_) { rc = 1; *bad = p; break; }
		emit_len[p] = (uint32_t)cb.last_frames;
	}
	unregister_decoder_ref(&st);
	for(int c = 0; c < channels; ++c) {
		size_t k = cb.out[c].size() < cap ? cb.out[c].size() : (size_t)cap;
		memcpy(pcm + size_t(c) * cap, cb.out[c].data(), sizeof(float) * k);
	}
	return rc;
}

// The compute tail of VorbisFloor1::decode on arbitrary coded ys (each < 256): the ys are packed into a
// bitstream using one synthetic 256-entry, 8-bit fixed-length codebook, one 1-dimensional partition class per post.
// returns 0 ok, 1 reference CHECK failure, 2 floor unused.
int ref_floor1_synth(const uint32_t* xs, int posts, int multiplier, const uint32_t* ys, int n, float* out) {
	std::vector<VorbisCodebook> books(1);
	VorbisCodebook& bk = books[0];
	bk.dimensions_ = 1;
	bk.num_entries_ = 256;
	bk.ordered_ = false;
	bk.sparse_ = false;
	bk.entries_.resize(256);
	for(uint32_t i = 0; i < 256; ++i) bk.entries_[i].init(i, 8);
	bk.lookup_type_ = 0;
	if(bk._assignCodewords().is_error_) return 1;

	VorbisFloor1 fl;
	fl.multiplier = (uint8_t)multiplier;
	fl.partition_classes.assign(size_t(posts - 2), 0);
	fl.classes.resize(1);
	fl.classes[0].dimensions = 1;
	fl.classes[0].subclass = 0;
	fl.classes[0].masterbook = 0;
	fl.classes[0].subclass_books.assign(1, 0);
	fl.xs.assign(xs, xs + posts);
	fl.xs_sorted_idx.resize(posts); // as VorbisFloor1::parse does, hpp:459-469
	for(int i = 0; i < posts; ++i) fl.xs_sorted_idx[i] = i;
	std::sort(fl.xs_sorted_idx.begin(), fl.xs_sorted_idx.end(), [&](const size_t& a, const size_t& b) { return fl.xs[a] < fl.xs[b]; });
	fl.xs_sorted.resize(posts);
	for(int i = 0; i < posts; ++i) fl.xs_sorted[i] = fl.xs[fl.xs_sorted_idx[i]];

	static const uint32_t range_of[5] = {0, 256, 128, 86, 64};
	const int ybits = highest_bit(range_of[multiplier] - 1);
	BitWriter bw;
	bw.bit(1); // nonzero flag, hpp:478
	bw.bits_lsb_first(ys[0], ybits);
	bw.bits_lsb_first(ys[1], ybits);
	for(int i = 2; i < posts; ++i) bw.bits_msb_first(ys[i] & 255u, 8); // Huffman walk is MSb-first, hpp:347-360
	bw.bytes.push_back(0);

	ConstDataReader rd(bw.bytes.data(), bw.bytes.size());
	BitReader br(&rd);
	std::vector<float> buf(n);
	DataRange<float> o(buf.data(), buf.size());
	bool use = false;
	register_decoder_ref(&fl, "ref_shim", 0, 1);
	OkOrError res = fl.decode(br, books, o, use);
	unregister_decoder_ref(&fl);
	if(res.is_error_) return 1;
	if(!use) return 2;
	memcpy(out, buf.data(), sizeof(float) * size_t(n));
	return 0;
}

} // extern "C"
