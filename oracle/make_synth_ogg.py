#!/usr/bin/env python3
"""Synthetic Ogg Vorbis I streams for differential tests against the REFERENCE decoder.  TEST INFRASTRUCTURE (authoring
container only): writes random but well-formed streams — setups the two real fixtures do not have (1-3 channels, other block
sizes, floors with other multipliers / post counts / subclass structures, residue formats 0 / 1 / 2, vector lengths that are not
powers of two, lookup types 1 and 2 with and without sequence_p, sparse and ordered codebooks, several submaps, several coupling
steps, clipped last page) — runs the reference decoder built from its own sources (oracle/_ref/ours.bin, oracle/Makefile) on
them and stores what it produced as golden vectors:

    tests/golden/synth_NN.ogg   the stream (data, written by this script from the Vorbis I specification)
    tests/golden/synth_NN.npz   channels, the reference's streamed "pcm", and per audio packet its "floor1 ys" and
                                "after_residue" hooks (src/ParseOggVorbis.hpp:518, 1211, 1051)

A stream the reference rejects (e.g. a floor curve that leaves the inverse-dB table, hpp:587) is discarded and the next seed is
tried: every committed stream is one the reference decodes without error.

    python oracle/make_synth_ogg.py [count] [first_seed]
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)
from make_golden import read_dump  # noqa: E402  (TLV reader of the reference's dump format)


# ------------------------------------------------------------------------------------------------
# bit packing (Vorbis I 2.1: LSb first within a byte, bytes in stream order)
# ------------------------------------------------------------------------------------------------
class BitWriter:
    def __init__(self):
        self.bytes = bytearray()
        self.acc = 0
        self.n = 0

    def write(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0, (value, bits)
        self.acc |= value << self.n
        self.n += bits
        while self.n >= 8:
            self.bytes.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def done(self):
        if self.n:
            self.bytes.append(self.acc & 0xFF)
            self.acc = 0
            self.n = 0
        return bytes(self.bytes)


def ilog(x):
    return int(x).bit_length() if x > 0 else 0


def float32_pack(mant, exp2):
    """value = mant * 2**exp2 (mant integer, |mant| < 2**21) in the packing of Vorbis I 9.2.2."""
    sign = 0x80000000 if mant < 0 else 0
    mant = abs(mant)
    assert mant < (1 << 21) and 0 <= exp2 + 788 < 1024
    return sign | ((exp2 + 788) << 21) | mant


def lookup1_values(entries, dims):
    r = 0
    while (r + 1) ** dims <= entries:
        r += 1
    return r


# ------------------------------------------------------------------------------------------------
# codebooks
# ------------------------------------------------------------------------------------------------
class Book:
    def __init__(self, rng, dims, entries, lookup, unused_frac=0.0):
        self.dims, self.entries, self.lookup = dims, entries, lookup
        # a complete prefix code: split random leaves until there are enough
        used = max(2, int(round(entries * (1.0 - unused_frac))))
        lens = [1, 1]
        while len(lens) < used:
            cand = [i for i, l in enumerate(lens) if l < 18]
            i = cand[int(rng.integers(0, len(cand)))]
            l = lens.pop(i) + 1
            lens += [l, l]
        rng.shuffle(lens)
        self.lengths = [0] * entries
        slots = sorted(rng.choice(entries, used, replace=False).tolist())
        for s, l in zip(slots, lens):
            self.lengths[s] = int(l)
        self.ordered = False
        if unused_frac == 0.0 and rng.random() < 0.25:  # "ordered" length coding needs ascending lengths in entry order
            self.lengths = sorted(self.lengths)
            self.ordered = True
        self.sparse = any(l == 0 for l in self.lengths)
        self.codes = self._assign()
        self.usable = [i for i, l in enumerate(self.lengths) if l]
        if lookup:
            self.value_bits = int(rng.integers(2, 5))
            self.sequence_p = bool(rng.random() < 0.15)
            nvals = lookup1_values(entries, dims) if lookup == 1 else entries * dims
            self.mults = rng.integers(0, 1 << self.value_bits, nvals).tolist()
            e = int(rng.integers(-3, 1))  # delta = 2**e
            self.delta = (1, e)
            centre = (1 << self.value_bits) // 2
            self.minimum = (-centre if not self.sequence_p else 0, e)

    def _assign(self):
        """3.2.1: every used entry, in order, takes the lowest free codeword of its length."""
        free = [0] * 33
        codes = [None] * self.entries
        first = True
        for i, ln in enumerate(self.lengths):
            if not ln:
                continue
            if first:
                first = False
                code = 0
                for l in range(1, ln + 1):
                    free[l] = 1 << (32 - l)
            else:
                z = ln
                while z > 0 and not free[z]:
                    z -= 1
                assert z > 0
                code = free[z]
                free[z] = 0
                for y in range(ln, z, -1):
                    free[y] = code + (1 << (32 - y))
            codes[i] = code >> (32 - ln)
        assert not any(free[1:]), "incomplete tree"
        return codes

    def write_header(self, w):
        w.write(0x564342, 24)
        w.write(self.dims, 16)
        w.write(self.entries, 24)
        w.write(1 if self.ordered else 0, 1)
        if not self.ordered:
            w.write(1 if self.sparse else 0, 1)
            for l in self.lengths:
                if self.sparse:
                    w.write(1 if l else 0, 1)
                    if not l:
                        continue
                w.write(l - 1, 5)
        else:
            cur, ln = 0, self.lengths[0]
            w.write(ln - 1, 5)
            while cur < self.entries:
                num = sum(1 for l in self.lengths[cur:] if l == ln)
                assert all(l == ln for l in self.lengths[cur:cur + num])
                w.write(num, ilog(self.entries - cur))
                cur += num
                ln += 1
        w.write(self.lookup, 4)
        if self.lookup:
            w.write(float32_pack(*self.minimum), 32)
            w.write(float32_pack(*self.delta), 32)
            w.write(self.value_bits - 1, 4)
            w.write(1 if self.sequence_p else 0, 1)
            for m in self.mults:
                w.write(int(m), self.value_bits)

    def put(self, w, entry):
        ln, code = self.lengths[entry], self.codes[entry]
        assert ln
        for b in range(ln):  # the decoder walks the tree from the most significant code bit
            w.write((code >> (ln - 1 - b)) & 1, 1)

    def random_entry(self, rng):
        return self.usable[int(rng.integers(0, len(self.usable)))]


# ------------------------------------------------------------------------------------------------
# a random stream setup
# ------------------------------------------------------------------------------------------------
class Setup:
    pass


def make_setup(rng):
    s = Setup()
    s.channels = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 6]))
    if os.environ.get("SYNTH_CHANNELS"):  # stress knobs (one-off runs; the committed set uses none)
        s.channels = int(os.environ["SYNTH_CHANNELS"])
    s.bs0, s.bs1 = [(64, 256), (128, 1024), (256, 2048), (256, 2048), (512, 512), (64, 64), (128, 512), (256, 2048), (512, 4096)][int(rng.integers(0, 9))]
    if os.environ.get("SYNTH_BLOCKS"):
        s.bs0, s.bs1 = (int(v) for v in os.environ["SYNTH_BLOCKS"].split("/"))
    s.rate = 44100
    books = []

    def add(b):
        books.append(b)
        return len(books) - 1

    # ---- floors: one per block size ----
    s.floors = []
    for n in (s.bs0, s.bs1):
        f = Setup()
        n2 = n // 2
        f.rangebits = ilog(n2 - 1) if n2 > 1 else 1
        assert (1 << f.rangebits) == n2
        f.multiplier = int(rng.integers(1, 5))
        rng_y = [256, 128, 86, 64][f.multiplier - 1]
        nclasses = int(rng.integers(1, 4))
        f.classes = []
        for _ in range(nclasses):
            c = Setup()
            c.dims = int(rng.integers(1, 5))
            c.subclass = int(rng.integers(0, 3))
            nsub = 1 << c.subclass
            # y residuals are entry numbers: small books keep the curve inside its range
            c.subbooks = [add(Book(rng, 1, int(rng.integers(4, 13)), 0)) if rng.random() < 0.85 else -1 for _ in range(nsub)]
            c.masterbook = add(Book(rng, 1, nsub ** c.dims if nsub ** c.dims >= 2 else 2, 0)) if c.subclass else 0
            f.classes.append(c)
        maxposts = min(40, n2 - 1)
        f.part_classes = []
        posts = 2
        while True:
            c = int(rng.integers(0, nclasses))
            if posts + f.classes[c].dims > maxposts or len(f.part_classes) >= 31:
                break
            f.part_classes.append(c)
            posts += f.classes[c].dims
            if rng.random() < 0.12 and len(f.part_classes) >= 2:
                break
        # every class index up to the largest one used must exist: the header declares max+1 classes
        used_max = max(f.part_classes) if f.part_classes else -1
        f.classes = f.classes[:used_max + 1]
        xs_inner = rng.choice(np.arange(1, n2), posts - 2, replace=False).tolist() if posts > 2 else []
        f.xs = [0, n2] + [int(x) for x in xs_inner]
        f.range = rng_y
        s.floors.append(f)

    # ---- residues: one per block size (plus sometimes a second pair for a second submap) ----
    def make_residue(n, nch_for_type2):
        r = Setup()
        n2 = n // 2
        # formats 0 / 1 only for a submap of ONE channel: with more, the reference advances its partition counter once per channel
        # instead of once per classword (hpp:756) and reads past its buffers; the product follows the specification there
        # (DESIGN.md section 7), so such streams have no reference answer to compare with
        r.type = int(rng.integers(0, 3)) if nch_for_type2 == 1 else 2
        ln = n2 * (nch_for_type2 if r.type == 2 else 1)
        r.psize = int(rng.choice([p for p in (4, 6, 8, 12, 16, 32) if p <= max(4, ln // 2)]))
        r.begin = int(rng.integers(0, max(1, ln // 4)))
        r.end = int(rng.integers(r.begin, ln + max(1, ln // 8)))
        r.nclass = int(rng.integers(1, 7))
        # The partition count is kept a multiple of the classword length: the reference writes the classes of a whole
        # classword even past the last partition (its buffer holds exactly `parts` of them, hpp:708-722), so other streams
        # corrupt its heap; libvorbis-made streams always satisfy this.
        parts = (min(r.end, ln) - min(r.begin, ln)) // r.psize
        r.classwords = int(rng.choice([c for c in (1, 2, 3) if parts % c == 0]))
        r.classbook = add(Book(rng, r.classwords, r.nclass ** r.classwords if r.nclass ** r.classwords >= 2 else 2, 0))
        dims_ok = [d for d in (1, 2, 3, 4, 6, 8) if r.psize % d == 0]
        vq = []
        for _ in range(int(rng.integers(2, 5))):
            d = int(rng.choice(dims_ok))
            lk = int(rng.choice([1, 1, 2]))
            if lk == 1:
                q = int(rng.integers(2, 6))
                ent = q ** d
                if ent > 4096:
                    lk, ent = 2, int(rng.integers(4, 120))
            else:
                ent = int(rng.integers(4, 120))
            vq.append(add(Book(rng, d, ent, lk, unused_frac=0.2 if rng.random() < 0.2 else 0.0)))
        r.cascade = np.full((r.nclass, 8), -1, np.int64)
        for c in range(r.nclass):
            for ps in range(8):
                if rng.random() < (0.45 if ps < 3 else 0.08):
                    r.cascade[c, ps] = vq[int(rng.integers(0, len(vq)))]
        return r

    s.residues = []
    s.mappings = []
    blocks = [(0, s.bs0), (1, s.bs1)]
    if rng.random() < 0.35:
        blocks.append((1, s.bs1))  # a second long-block mode with its own mapping (other coupling / submaps / residue)
    for blk, n in blocks:
        m = Setup()
        m.submaps = 2 if (s.channels >= 2 and rng.random() < 0.3) else 1
        m.mux = [0] * s.channels
        if m.submaps > 1:  # every submap gets a channel (the reference dereferences an empty submap's first channel, hpp:1191-1200)
            while len(set(m.mux)) < m.submaps:
                m.mux = [int(rng.integers(0, m.submaps)) for _ in range(s.channels)]
        m.coupling = []
        if s.channels >= 2 and rng.random() < 0.7:
            steps = 1 if s.channels == 2 else int(rng.integers(1, min(4, s.channels)))
            for _ in range(steps):
                a, b = rng.choice(s.channels, 2, replace=False).tolist()
                m.coupling.append((int(a), int(b)))
        m.sub = []
        for sm in range(m.submaps):
            nch = sum(1 for c in range(s.channels) if m.mux[c] == sm)
            s.residues.append(make_residue(n, max(1, nch)))
            m.sub.append((blk, len(s.residues) - 1))  # (floor, residue)
        s.mappings.append(m)
    s.modes = [(blk, k) for k, (blk, _) in enumerate(blocks)]  # (blockflag, mapping)
    s.books = books
    return s


def write_setup(s):
    w = BitWriter()
    w.write(5, 8)
    for ch in b"vorbis":
        w.write(ch, 8)
    w.write(len(s.books) - 1, 8)
    for b in s.books:
        b.write_header(w)
    w.write(0, 6)  # one time-domain transform placeholder
    w.write(0, 16)
    w.write(len(s.floors) - 1, 6)
    for f in s.floors:
        w.write(1, 16)
        w.write(len(f.part_classes), 5)
        for c in f.part_classes:
            w.write(c, 4)
        for c in f.classes:
            w.write(c.dims - 1, 3)
            w.write(c.subclass, 2)
            if c.subclass:
                w.write(c.masterbook, 8)
            for b in c.subbooks:
                w.write(b + 1, 8)
        w.write(f.multiplier - 1, 2)
        w.write(f.rangebits, 4)
        for x in f.xs[2:]:
            w.write(x, f.rangebits)
    w.write(len(s.residues) - 1, 6)
    for r in s.residues:
        w.write(r.type, 16)
        w.write(r.begin, 24)
        w.write(r.end, 24)
        w.write(r.psize - 1, 24)
        w.write(r.nclass - 1, 6)
        w.write(r.classbook, 8)
        for c in range(r.nclass):
            bits = sum(1 << ps for ps in range(8) if r.cascade[c, ps] >= 0)
            w.write(bits & 7, 3)
            if bits >> 3:
                w.write(1, 1)
                w.write(bits >> 3, 5)
            else:
                w.write(0, 1)
        for c in range(r.nclass):
            for ps in range(8):
                if r.cascade[c, ps] >= 0:
                    w.write(int(r.cascade[c, ps]), 8)
    w.write(len(s.mappings) - 1, 6)
    for m in s.mappings:
        w.write(0, 16)
        if m.submaps > 1:
            w.write(1, 1)
            w.write(m.submaps - 1, 4)
        else:
            w.write(0, 1)
        if m.coupling:
            w.write(1, 1)
            w.write(len(m.coupling) - 1, 8)
            bits = ilog(s.channels - 1)
            for a, b in m.coupling:
                w.write(a, bits)
                w.write(b, bits)
        else:
            w.write(0, 1)
        w.write(0, 2)
        if m.submaps > 1:
            for c in range(s.channels):
                w.write(m.mux[c], 4)
        for fl, rs in m.sub:
            w.write(0, 8)
            w.write(fl, 8)
            w.write(rs, 8)
    w.write(len(s.modes) - 1, 6)
    for bf, mp in s.modes:
        w.write(bf, 1)
        w.write(0, 16)
        w.write(0, 16)
        w.write(mp, 8)
    w.write(1, 1)  # framing
    return w.done()


# ------------------------------------------------------------------------------------------------
# audio packets: the decode order of Vorbis I 4.3 / 7.2.3 / 8.6.2, with random symbols
# ------------------------------------------------------------------------------------------------
def write_audio(s, rng, mode, prev_long, next_long):
    w = BitWriter()
    w.write(0, 1)
    w.write(mode, ilog(len(s.modes) - 1))
    bf, mp = s.modes[mode]
    if bf:
        w.write(prev_long, 1)
        w.write(next_long, 1)
    n = s.bs1 if bf else s.bs0
    m = s.mappings[mp]
    used = []
    for c in range(s.channels):
        f = s.floors[m.sub[m.mux[c]][0]]
        nonzero = rng.random() < 0.85
        w.write(1 if nonzero else 0, 1)
        used.append(nonzero)
        if not nonzero:
            continue
        ybits = ilog(f.range - 1)
        hi = f.range
        w.write(int(rng.integers(hi // 3, hi - hi // 6)), ybits)
        w.write(int(rng.integers(hi // 3, hi - hi // 6)), ybits)
        for pc in f.part_classes:
            cl = f.classes[pc]
            cval = 0
            if cl.subclass:
                mb = s.books[cl.masterbook]
                cval = mb.random_entry(rng)
                mb.put(w, cval)
            for _ in range(cl.dims):
                book = cl.subbooks[cval & ((1 << cl.subclass) - 1)]
                cval >>= cl.subclass
                if book >= 0:
                    b = s.books[book]
                    # mostly "no change" (entry 0 if it exists), otherwise a small residual
                    e = b.random_entry(rng)
                    if 0 in b.usable and rng.random() < 0.5:
                        e = 0
                    b.put(w, e)
    # nonzero propagation through the coupling steps (4.3.3)
    for a, b in m.coupling:
        if used[a] or used[b]:
            used[a] = used[b] = True
    for sm, (_, ri) in enumerate(m.sub):
        chans = [c for c in range(s.channels) if m.mux[c] == sm]
        if not chans:
            continue
        r = s.residues[ri]
        n2 = n // 2
        if r.type == 2:
            vec_used, ln = [True], n2 * len(chans)  # (decoded even when no channel is marked used: the reference does, hpp:685-694)
        else:
            vec_used, ln = [used[c] for c in chans], n2
        lb, le = min(r.begin, ln), min(r.end, ln)
        nread = le - lb
        if nread == 0:
            continue
        parts = nread // r.psize
        cb = s.books[r.classbook]
        cls = [[0] * (parts + r.classwords) for _ in vec_used]
        for ps in range(8):
            pc = 0
            while pc < parts:
                if ps == 0:
                    for j, u in enumerate(vec_used):
                        if not u:
                            continue
                        e = cb.random_entry(rng)
                        cb.put(w, e)
                        t = e
                        for i in range(r.classwords - 1, -1, -1):
                            cls[j][pc + i] = t % r.nclass
                            t //= r.nclass
                for _ in range(r.classwords):
                    if pc >= parts:
                        break
                    for j, u in enumerate(vec_used):
                        if not u:
                            continue
                        book = int(r.cascade[cls[j][pc], ps])
                        if book >= 0:
                            b = s.books[book]
                            for _k in range(r.psize // b.dims):
                                b.put(w, b.random_entry(rng))
                    pc += 1
    return w.done()


# ------------------------------------------------------------------------------------------------
# Ogg framing
# ------------------------------------------------------------------------------------------------
def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        t.append(r)
    return t


_CRC = _crc_table()


def ogg_page(serial, seq, granule, packets, bos=False, eos=False):
    table = bytearray()
    for p in packets:
        ln = len(p)
        while ln >= 255:
            table.append(255)
            ln -= 255
        table.append(ln)
    assert len(table) <= 255
    hdr = bytearray(b"OggS") + bytes([0, (2 if bos else 0) | (4 if eos else 0)]) + struct.pack("<qIII", granule, serial, seq, 0) + \
        bytes([len(table)]) + table
    page = hdr + b"".join(packets)
    crc = 0
    for byte in page:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ _CRC[((crc >> 24) & 0xFF) ^ byte]
    page[22:26] = struct.pack("<I", crc)
    return bytes(page)


def make_stream(seed):
    rng = np.random.default_rng(seed)
    s = make_setup(rng)
    ident = b"\x01vorbis" + struct.pack("<IBIiiiB", 0, s.channels, s.rate, 0, 128000, 0, (ilog(s.bs0) - 1) | ((ilog(s.bs1) - 1) << 4)) + b"\x01"
    vendor = b"synthetic stream for differential tests"
    comment = b"\x03vorbis" + struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 1) + struct.pack("<I", 6) + b"SEED=%d" % (seed % 10) + b"\x01"
    comment = b"\x03vorbis" + struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", 0) + b"\x01"
    setup = write_setup(s)
    npk = int(os.environ.get("SYNTH_PACKETS", 0)) or int(rng.integers(8, 20))
    flags = [int(rng.random() < 0.55) for _ in range(npk)]
    pages = [ogg_page(77, 0, 0, [ident], bos=True), ogg_page(77, 1, 0, [comment, setup])]
    seq, pos, prev_n, batch = 2, 0, 0, []
    for q in range(npk):
        lng = flags[q]
        prev_long = flags[q - 1] if q else 1
        next_long = flags[q + 1] if q + 1 < npk else 1
        long_modes = [k for k, (bf, _) in enumerate(s.modes) if bf]
        mode = long_modes[int(rng.integers(0, len(long_modes)))] if lng else 0
        pkt = write_audio(s, rng, mode, prev_long, next_long)
        assert len(pkt) < 255 * 200
        n = s.bs1 if lng else s.bs0
        pos += (prev_n // 4 + n // 4) if prev_n else 0
        prev_n = n
        batch.append(pkt)
        last = q == npk - 1
        if last or rng.random() < 0.6 or sum(len(p) // 255 + 1 for p in batch) > 200:
            granule = pos
            if last and pos > 40 and rng.random() < 0.5:
                granule = pos - int(rng.integers(1, min(40, (prev_n // 4) or 1) + 1))  # clipped end (hpp:1028-1044)
            pages.append(ogg_page(77, seq, granule, batch, eos=last))
            seq += 1
            batch = []
    return b"".join(pages), s, npk


def reference_vectors(path):
    ours = os.path.join(HERE, "_ref", "ours.bin")
    asan = os.path.join(HERE, "_ref", "ours_asan.bin")  # `make -C oracle _ref/ours_asan.bin`
    r = subprocess.run([asan, "--in", path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:  # an error exit or a memory error inside the reference: no golden from this stream
        out = r.stdout.decode(errors="replace")
        i = out.find("ERROR")
        return None, out[i:i + 300] if i >= 0 else out[-300:]
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "d.bin")
        r = subprocess.run([ours, "--in", path, "--debug_out", dump], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            return None, r.stdout.decode(errors="replace")[-300:]
        header, entries = read_dump(dump)
    C = int(header["decoder-num-channels"][0])
    pcm = [[] for _ in range(C)]
    ys, ys_len, ys_ch, res, res_len = [], [], [], [], []
    packets = 0
    last_floor_ch = -1
    for nm, ch, v in entries:
        if nm == "start_audio_packet":
            packets += 1
        elif nm == "floor_number":
            last_floor_ch = ch
        elif nm == "floor1 ys":
            ys.append(v.astype(np.uint32))
            ys_len.append(len(v))
            ys_ch.append(last_floor_ch + 1000 * (packets - 1))
        elif nm == "after_residue":
            res.append(v.astype(np.float32))
            res_len.append(len(v))
        elif nm == "pcm":
            pcm[ch].append(v.astype(np.float32))
    pcm = np.stack([np.concatenate(p) if p else np.zeros(0, np.float32) for p in pcm])
    # the whole hook stream in digest form — name, channel, length and (integers) a CRC of the values as int64 / (floats) sum
    # and sum of magnitudes in double —, "pcm" entries left out (the chunking of PCM across calls may differ)
    import zlib
    hk = [(nm, ch, v) for nm, ch, v in entries if nm != "pcm"]
    hook = dict(hook_names=np.asarray([nm.encode() for nm, _, _ in hk]), hook_ch=np.asarray([ch for _, ch, _ in hk], np.int16),
                hook_len=np.asarray([len(v) for _, _, v in hk], np.int32),
                hook_float=np.asarray([v.dtype.kind == "f" for _, _, v in hk], np.bool_),
                hook_crc=np.asarray([0 if v.dtype.kind == "f" else zlib.crc32(v.astype(np.int64).tobytes()) for _, _, v in hk], np.uint32),
                hook_sum=np.asarray([float(v.astype(np.float64).sum()) if v.dtype.kind == "f" else 0.0 for _, _, v in hk]),
                hook_abs=np.asarray([float(np.abs(v.astype(np.float64)).sum()) if v.dtype.kind == "f" else 0.0 for _, _, v in hk]))
    return dict(channels=np.int32(C), packets=np.int32(packets), pcm=pcm, **hook,
                ys=np.concatenate(ys) if ys else np.zeros(0, np.uint32), ys_len=np.asarray(ys_len, np.int32),
                ys_where=np.asarray(ys_ch, np.int32),  # packet * 1000 + channel of every "floor1 ys" entry
                residue=np.concatenate(res) if res else np.zeros(0, np.float32), residue_len=np.asarray(res_len, np.int32)), ""


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    made = 0
    while made < count:
        data, s, npk = make_stream(seed)
        path = os.path.join(OUT, "synth_%02d.ogg" % made)
        open(path, "wb").write(data)
        vec, why = reference_vectors(path)
        too_big = vec is not None and vec["pcm"].nbytes > int(os.environ.get("SYNTH_MAX_PCM_BYTES", "1000000000"))
        if vec is None or too_big or vec["pcm"].shape[1] == 0 or not np.isfinite(vec["pcm"]).all() or np.abs(vec["pcm"]).max() > 1e4:
            print("seed %d skipped: %s" % (seed, (why or "empty / non-finite / huge pcm / over the size limit").strip().replace("\n", " | ")[-200:]))
            os.remove(path)
            seed += 1
            continue
        # what a test needs to redo the nonzero propagate (4.3.3) per packet: mode -> mapping -> coupling steps
        extra = dict(blocksize0=np.int32(s.bs0), blocksize1=np.int32(s.bs1), mode_mapping=np.asarray([m for _, m in s.modes], np.int32),
                     mode_blockflag=np.asarray([bf for bf, _ in s.modes], np.int32))
        for k, mp in enumerate(s.mappings):
            extra["coupling_m%d" % k] = np.asarray(mp.coupling, np.int32).reshape(-1, 2)
            extra["chfloor_m%d" % k] = np.asarray([mp.sub[mp.mux[c]][0] for c in range(s.channels)], np.int32)
        for k, f in enumerate(s.floors):  # the synthesis-side setup (what vsyn_setup / the oracle take)
            extra["floor%d_mult" % k] = np.int32(f.multiplier)
            extra["floor%d_xs" % k] = np.asarray(f.xs, np.int32)
        extra["num_floors"] = np.int32(len(s.floors))
        np.savez_compressed(os.path.join(OUT, "synth_%02d.npz" % made), seed=np.int32(seed), **vec, **extra)
        print("synth_%02d: seed %d, %d ch, blocks %d/%d, %d packets, %d frames, |pcm| <= %.3g, %d bytes"
              % (made, seed, s.channels, s.bs0, s.bs1, npk, vec["pcm"].shape[1], float(np.abs(vec["pcm"]).max()), len(data)))
        made += 1
        seed += 1


if __name__ == "__main__":
    main()
