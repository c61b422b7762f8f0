"""ctypes view of the C-ABI in include/vorbis_synth_hip.h (libvorbis_synth_hip.so).

Plumbing only: POD struct mirrors, the symbol table and a loader that FAILS LOUDLY when the HIP
library is missing.  There is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvorbis_synth_hip.so")

VSYN_MAX_CHANNELS = 32
VSYN_MAX_POSTS = 65
VSYN_OK, VSYN_ERR_INVALID, VSYN_ERR_NO_DEVICE, VSYN_ERR_HIP, VSYN_ERR_STREAM = 0, 1, 2, 3, 4
VSYN_ST_FLOOR_RANGE, VSYN_ST_FLOOR_VALUE, VSYN_ST_GRANULE, VSYN_ST_PLANE_OVERFLOW, VSYN_ST_BAD_MODE = 1, 2, 4, 8, 16
VSYN_ST_BAD_SEGMENT, VSYN_ST_BAD_VQ = 32, 64
VSYN_SEG_RESET = 1
VSYN_SUBMIT_STAGED = 1
VSYN_SUBMIT_INPUTS_READY = 2
VSYN_SUBMIT_KEEP_PCM = 4
VSYN_SUBMIT_PRE_KERNELS = 8
VSYN_PCM_S16, VSYN_PCM_F32 = 1, 2


class Floor1(C.Structure):
    _fields_ = [("multiplier", C.c_uint32), ("num_posts", C.c_uint32), ("xs", C.POINTER(C.c_uint32))]


class Coupling(C.Structure):
    _fields_ = [("magnitude", C.c_uint16), ("angle", C.c_uint16)]


class Mapping(C.Structure):
    _fields_ = [("num_couplings", C.c_uint32), ("couplings", C.POINTER(Coupling)),
                ("channel_floor", C.POINTER(C.c_uint8))]


class Mode(C.Structure):
    _fields_ = [("block_flag", C.c_uint8), ("mapping", C.c_uint8)]


class Setup(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("blocksize0", C.c_uint32), ("blocksize1", C.c_uint32),
                ("num_floors", C.c_uint32), ("floors", C.POINTER(Floor1)),
                ("num_mappings", C.c_uint32), ("mappings", C.POINTER(Mapping)),
                ("num_modes", C.c_uint32), ("modes", C.POINTER(Mode))]


class Taps(C.Structure):
    _fields_ = [("after_envelope", C.c_void_p), ("pcm_after_mdct", C.c_void_p), ("floor_final", C.c_void_p),
                ("floor_curve", C.c_void_p)]


class Status(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("first_bad_packet", C.c_uint32)]


class Codebook(C.Structure):
    _fields_ = [("dimensions", C.c_uint32), ("num_entries", C.c_uint32), ("lookup", C.POINTER(C.c_float))]


class Residue(C.Structure):
    _fields_ = [("type", C.c_uint32), ("begin", C.c_uint32), ("end", C.c_uint32), ("partition_size", C.c_uint32),
                ("num_classifications", C.c_uint32), ("classwords", C.c_uint32), ("books", C.POINTER(C.c_int16))]


class VqMapping(C.Structure):
    _fields_ = [("num_submaps", C.c_uint32), ("mux", C.POINTER(C.c_uint8)), ("submap_residue", C.POINTER(C.c_uint8))]


class VqSetup(C.Structure):
    _fields_ = [("num_codebooks", C.c_uint32), ("codebooks", C.POINTER(Codebook)),
                ("num_residues", C.c_uint32), ("residues", C.POINTER(Residue)),
                ("num_mappings", C.c_uint32), ("mappings", C.POINTER(VqMapping))]


class VqBatch(C.Structure):
    _fields_ = [("packets", C.c_void_p), ("cls", C.c_void_p), ("entries", C.c_void_p),
                ("num_cls", C.c_uint64), ("num_entries", C.c_uint64)]


class VqSpec:
    """Plain-python description of the residue VQ setup; `.c_setup()` builds the vsyn_vq_setup tree (keeps it alive).
    codebooks: list of (dimensions, num_entries, float32 table [entries*dims] or None)
    residues:  list of dict(type, begin, end, partition_size, num_classifications, classwords, books int16 [nclass*8])
    mappings:  list of (mux list [channels], submap_residue list)"""

    def __init__(self, codebooks, residues, mappings):
        self.codebooks, self.residues, self.mappings = codebooks, residues, mappings
        self._keep = []

    def c_setup(self):
        keep = []
        cb = (Codebook * len(self.codebooks))()
        for i, (dims, n, tab) in enumerate(self.codebooks):
            cb[i].dimensions, cb[i].num_entries = dims, n
            if tab is not None:
                t = np.ascontiguousarray(tab, np.float32)
                keep.append(t)
                cb[i].lookup = t.ctypes.data_as(C.POINTER(C.c_float))
        rs = (Residue * len(self.residues))()
        for i, r in enumerate(self.residues):
            b = np.ascontiguousarray(r["books"], np.int16)
            keep.append(b)
            rs[i].type, rs[i].begin, rs[i].end, rs[i].partition_size = r["type"], r["begin"], r["end"], r["partition_size"]
            rs[i].num_classifications, rs[i].classwords = r["num_classifications"], r["classwords"]
            rs[i].books = b.ctypes.data_as(C.POINTER(C.c_int16))
        mp = (VqMapping * len(self.mappings))()
        for i, (mux, sres) in enumerate(self.mappings):
            m = (C.c_uint8 * len(mux))(*mux)
            sr = (C.c_uint8 * len(sres))(*sres)
            keep += [m, sr]
            mp[i].num_submaps, mp[i].mux, mp[i].submap_residue = len(sres), m, sr
        keep += [cb, rs, mp]
        self._keep.append(keep)
        return VqSetup(len(self.codebooks), cb, len(self.residues), rs, len(self.mappings), mp)


VQ_PACKET_DTYPE = np.dtype([("entry_off", "<u8"), ("num_entries", "<u4"), ("cls_off", "<u4")], align=True)
assert VQ_PACKET_DTYPE.itemsize == 16

# numpy record layouts of the batch PODs (sizes asserted against the header's comments)
PACKET_DTYPE = np.dtype([("mode", "u1"), ("prev_long", "u1"), ("next_long", "u1"), ("reserved0", "u1"),
                         ("floor_used", "<u4"), ("granule", "<i8")], align=True)
SEGMENT_DTYPE = np.dtype([("stream", "<u4"), ("first_packet", "<u4"), ("num_packets", "<u4"), ("flags", "<u4"),
                          ("residue_off", "<u8")], align=True)
assert PACKET_DTYPE.itemsize == 16 and SEGMENT_DTYPE.itemsize == 24


class SetupSpec:
    """Plain-python description of a stream setup; `.c_setup()` builds the vsyn_setup tree (keeps it alive)."""

    def __init__(self, channels, blocksize0, blocksize1, floors, mappings, modes):
        # floors: list of (multiplier, xs list); mappings: list of (couplings [(mag,ang)], channel_floor list)
        # modes: list of (block_flag, mapping)
        self.channels, self.blocksize0, self.blocksize1 = channels, blocksize0, blocksize1
        self.floors, self.mappings, self.modes = floors, mappings, modes
        self._keep = []

    def c_setup(self):
        keep = []
        fl = (Floor1 * len(self.floors))()
        for i, (mult, xs) in enumerate(self.floors):
            arr = (C.c_uint32 * len(xs))(*xs)
            keep.append(arr)
            fl[i].multiplier, fl[i].num_posts, fl[i].xs = mult, len(xs), arr
        mp = (Mapping * len(self.mappings))()
        for i, (coups, chfloor) in enumerate(self.mappings):
            ca = (Coupling * max(1, len(coups)))()
            for k, (m, a) in enumerate(coups):
                ca[k].magnitude, ca[k].angle = m, a
            cf = (C.c_uint8 * self.channels)(*chfloor)
            keep += [ca, cf]
            mp[i].num_couplings, mp[i].couplings, mp[i].channel_floor = len(coups), ca, cf
        md = (Mode * len(self.modes))()
        for i, (bf, m) in enumerate(self.modes):
            md[i].block_flag, md[i].mapping = bf, m
        su = Setup(self.channels, self.blocksize0, self.blocksize1, len(self.floors), fl, len(self.mappings), mp,
                   len(self.modes), md)
        keep += [fl, mp, md]
        self._keep.append(keep)
        return su

    @property
    def ys_stride(self):
        return (max(len(xs) for _, xs in self.floors) + 3) & ~3

    def blocksize_of_mode(self, mode):
        return self.blocksize1 if self.modes[mode][0] else self.blocksize0


_SYMBOLS = [
    "vsyn_version", "vsyn_abi_version", "vsyn_create", "vsyn_destroy", "vsyn_ys_stride", "vsyn_channels", "vsyn_fused_paths",
    "vsyn_const_block_bytes", "vsyn_submit_device", "vsyn_submit_host", "vsyn_sync_status", "vsyn_reset_streams",
    "vsyn_profile_enable", "vsyn_profile_read", "vsyn_imdct_device", "vsyn_host_alloc", "vsyn_host_free",
    "vsyn_attach_vq", "vsyn_submit_device_vq", "vsyn_submit_host_vq", "vsyn_pcm_interleave_device", "vsyn_pcm_abs_sum_host", "vsyn_pcm_fetch_host",
]


def declared_symbols():
    return list(_SYMBOLS)


_lib = None


def load():
    """dlopen the HIP library. Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch-ROCm bundles its own libamdhip64: when torch shares the process it must be the HIP runtime that
        # gets loaded (two runtimes in one process cannot both own the GPU). Plumbing only; torch is optional.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension missing: %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, cpp = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_char_p)
    lib.vsyn_version.restype = C.c_char_p
    lib.vsyn_abi_version.restype = C.c_int
    lib.vsyn_create.argtypes = [C.POINTER(Setup), C.c_int, u32, C.POINTER(vp), cpp]
    lib.vsyn_destroy.argtypes = [vp]
    lib.vsyn_destroy.restype = None
    lib.vsyn_ys_stride.argtypes = [vp]
    lib.vsyn_ys_stride.restype = u32
    lib.vsyn_channels.argtypes = [vp]
    lib.vsyn_channels.restype = u32
    lib.vsyn_const_block_bytes.argtypes = [vp]
    lib.vsyn_const_block_bytes.restype = C.c_size_t
    lib.vsyn_fused_paths.argtypes = [vp]
    lib.vsyn_fused_paths.restype = u32
    lib.vsyn_submit_device.argtypes = [vp, u32, vp, u32, vp, u32, vp, vp, vp, u64, vp, C.POINTER(Taps), u32, vp, cpp]
    lib.vsyn_submit_host.argtypes = [vp, u32, vp, u32, vp, vp, vp, C.c_size_t, vp, u64, vp, C.POINTER(Taps), u32,
                                     C.POINTER(Status), cpp]
    lib.vsyn_sync_status.argtypes = [vp, vp, C.POINTER(Status), cpp]
    lib.vsyn_reset_streams.argtypes = [vp, vp, cpp]
    lib.vsyn_profile_enable.argtypes = [vp, C.c_int]
    lib.vsyn_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u32), cpp]
    lib.vsyn_imdct_device.argtypes = [vp, u32, u32, vp, vp, vp, cpp]
    lib.vsyn_attach_vq.argtypes = [vp, C.POINTER(VqSetup), cpp]
    lib.vsyn_submit_device_vq.argtypes = [vp, u32, vp, u32, vp, u32, vp, C.POINTER(VqBatch), vp, vp, u64, vp,
                                          C.POINTER(Taps), u32, vp, cpp]
    lib.vsyn_submit_host_vq.argtypes = [vp, u32, vp, u32, vp, vp, C.POINTER(VqBatch), vp, C.c_size_t, vp, u64, vp,
                                        C.POINTER(Taps), u32, C.POINTER(Status), cpp]
    lib.vsyn_pcm_interleave_device.argtypes = [vp, C.c_int, vp, u64, vp, u64, vp, vp, cpp]
    lib.vsyn_pcm_abs_sum_host.argtypes = [vp, C.POINTER(C.c_double), cpp]
    lib.vsyn_pcm_fetch_host.argtypes = [vp, C.c_int, vp, u64, vp, cpp]
    lib.vsyn_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp), cpp]
    lib.vsyn_host_free.argtypes = [vp]
    lib.vsyn_host_free.restype = None
    _lib = lib
    return lib


class VsynError(RuntimeError):
    def __init__(self, code, msg, status=None):
        super().__init__("vsyn error %d: %s" % (code, msg))
        self.code, self.status = code, status


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Synth:
    """Owns one vsyn_handle. Thin: every method is one C-ABI call."""

    def __init__(self, spec, device=0, max_streams=64):
        self.lib = load()
        self.spec = spec
        self._su = spec.c_setup()
        h, err = C.c_void_p(), C.c_char_p()
        rc = self.lib.vsyn_create(C.byref(self._su), device, max_streams, C.byref(h), C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())
        self.h = h
        self.channels = spec.channels
        self.ys_stride = self.lib.vsyn_ys_stride(h)
        self.fused_paths = self.lib.vsyn_fused_paths(h)  # bit 0: fused kernel for long-block runs, bit 1: for mixed-block runs too

    def close(self):
        if getattr(self, "h", None):
            self.lib.vsyn_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, stream=None):
        err = C.c_char_p()
        rc = self.lib.vsyn_reset_streams(self.h, stream, C.byref(err))
        if rc:
            raise VsynError(rc, (err.value or b"").decode())

    def submit_host(self, packets, segments, ys, residue, plane_stride, want_taps=False, flags=0):
        """numpy in, numpy out: returns dict(pcm [S][C][plane_stride], emit_len [P], taps..., status)."""
        P, S, Cn = len(packets), len(segments), self.channels
        packets = np.ascontiguousarray(packets, dtype=PACKET_DTYPE)
        segments = np.ascontiguousarray(segments, dtype=SEGMENT_DTYPE)
        ys = np.ascontiguousarray(ys, dtype=np.uint16)
        residue = np.ascontiguousarray(residue, dtype=np.float32)
        assert ys.size == P * Cn * self.ys_stride
        pcm = np.zeros((S, Cn, plane_stride), np.float32)
        emit = np.zeros(P, np.uint32)
        taps, tp = None, None
        if want_taps == "features":  # the two feature taps only (they do not force the staged kernels)
            taps = dict(floor_final=np.zeros(ys.size, np.uint16), floor_curve=np.zeros(residue.size, np.uint16))
            tp = Taps(None, None, taps["floor_final"].ctypes.data, taps["floor_curve"].ctypes.data)
        elif want_taps:
            taps = dict(after_envelope=np.zeros(residue.size, np.float32),
                        pcm_after_mdct=np.zeros(residue.size * 2, np.float32),
                        floor_final=np.zeros(ys.size, np.uint16),
                        floor_curve=np.zeros(residue.size, np.uint16))
            tp = Taps(taps["after_envelope"].ctypes.data, taps["pcm_after_mdct"].ctypes.data,
                      taps["floor_final"].ctypes.data, taps["floor_curve"].ctypes.data)
        st, err = Status(), C.c_char_p()
        rc = self.lib.vsyn_submit_host(self.h, P, _ptr(packets), S, _ptr(segments), _ptr(ys), _ptr(residue),
                                       residue.size, _ptr(pcm), plane_stride, _ptr(emit),
                                       C.byref(tp) if tp else None, flags, C.byref(st), C.byref(err))
        if rc not in (VSYN_OK, VSYN_ERR_STREAM):
            raise VsynError(rc, (err.value or b"").decode())
        return dict(rc=rc, pcm=pcm, emit_len=emit, taps=taps, flags=st.flags, first_bad=st.first_bad_packet)

    def attach_vq(self, vq_spec):
        """vsyn_attach_vq: codebook value tables + residue descriptions for the device VQ stage."""
        self._vq = vq_spec.c_setup()
        err = C.c_char_p()
        rc = self.lib.vsyn_attach_vq(self.h, C.byref(self._vq), C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())
        self.fused_paths = self.lib.vsyn_fused_paths(self.h)  # (+ bit 8: the VQ kernel keeps this setup's value tables in LDS)

    def submit_host_vq(self, packets, segments, ys, vq_packets, cls, entries, residue_floats, plane_stride,
                       want_residue=True, flags=0):
        """Like submit_host with the residue given as classification + entry numbers; returns 'residue' (the
        rebuilt after_residue tensor) when want_residue."""
        P, S, Cn = len(packets), len(segments), self.channels
        packets = np.ascontiguousarray(packets, dtype=PACKET_DTYPE)
        segments = np.ascontiguousarray(segments, dtype=SEGMENT_DTYPE)
        ys = np.ascontiguousarray(ys, dtype=np.uint16)
        vq_packets = np.ascontiguousarray(vq_packets, dtype=VQ_PACKET_DTYPE)
        cls = np.ascontiguousarray(cls, dtype=np.uint8)
        entries = np.ascontiguousarray(entries, dtype=np.uint16)
        pcm = np.zeros((S, Cn, plane_stride), np.float32)
        emit = np.zeros(P, np.uint32)
        res = np.zeros(residue_floats, np.float32) if want_residue else None
        vb = VqBatch(vq_packets.ctypes.data, cls.ctypes.data if cls.size else None,
                     entries.ctypes.data if entries.size else None, cls.size, entries.size)
        st, err = Status(), C.c_char_p()
        rc = self.lib.vsyn_submit_host_vq(self.h, P, _ptr(packets), S, _ptr(segments), _ptr(ys), C.byref(vb), _ptr(res),
                                          residue_floats, _ptr(pcm), plane_stride, _ptr(emit), None, flags,
                                          C.byref(st), C.byref(err))
        if rc not in (VSYN_OK, VSYN_ERR_STREAM):
            raise VsynError(rc, (err.value or b"").decode())
        return dict(rc=rc, pcm=pcm, emit_len=emit, residue=res, flags=st.flags, first_bad=st.first_bad_packet)

    def submit_device_vq(self, P, d_packets, S, d_segments, max_seg_packets, d_ys, d_vq_packets, d_cls, num_cls, d_entries,
                         num_entries, d_residue, d_pcm, plane_stride, d_emit=None, flags=0, stream=None):
        """All pointers are raw device addresses (ints)."""
        err = C.c_char_p()
        vb = VqBatch(d_vq_packets, d_cls, d_entries, num_cls, num_entries)
        rc = self.lib.vsyn_submit_device_vq(self.h, P, d_packets, S, d_segments, max_seg_packets, d_ys, C.byref(vb),
                                            d_residue, d_pcm, plane_stride, d_emit, None, flags, stream, C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())

    def submit_device(self, P, d_packets, S, d_segments, max_seg_packets, d_ys, d_residue, d_pcm, plane_stride,
                      d_emit=None, taps=None, flags=0, stream=None):
        """All arguments are raw device addresses (ints)."""
        err = C.c_char_p()
        tp = C.byref(Taps(*taps)) if taps else None
        rc = self.lib.vsyn_submit_device(self.h, P, d_packets, S, d_segments, max_seg_packets, d_ys, d_residue,
                                         d_pcm, plane_stride, d_emit, tp, flags, stream, C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())

    def pcm_interleave_device(self, fmt, d_pcm, plane_stride, d_out, out_stride_frames, d_frames=None, stream=None):
        """Interleave / convert the PCM of the most recent submit_device* (raw device addresses)."""
        err = C.c_char_p()
        rc = self.lib.vsyn_pcm_interleave_device(self.h, fmt, d_pcm, plane_stride, d_out, out_stride_frames, d_frames, stream,
                                                 C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())

    def pcm_fetch_host(self, fmt, num_segments, out_stride_frames):
        """The PCM of the most recent submit_host*, converted on the device -> ([S][out_stride_frames][C] int16 / float32, frames [S])."""
        out = np.zeros((num_segments, out_stride_frames, self.channels), np.int16 if fmt == VSYN_PCM_S16 else np.float32)
        frames = np.zeros(num_segments, np.uint32)
        err = C.c_char_p()
        rc = self.lib.vsyn_pcm_fetch_host(self.h, fmt, out.ctypes.data, out_stride_frames, frames.ctypes.data, C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())
        return out, frames

    def pcm_abs_sum_host(self, num_segments):
        """Per-(segment, channel) sum |x| of the PCM of the most recent submit_host*, computed on the device -> [S][C] float64."""
        out = np.zeros((num_segments, self.channels), np.float64)
        err = C.c_char_p()
        rc = self.lib.vsyn_pcm_abs_sum_host(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())
        return out

    def sync_status(self, stream=None):
        st, err = Status(), C.c_char_p()
        rc = self.lib.vsyn_sync_status(self.h, stream, C.byref(st), C.byref(err))
        if rc not in (VSYN_OK, VSYN_ERR_STREAM):
            raise VsynError(rc, (err.value or b"").decode())
        return st.flags, st.first_bad_packet

    def imdct_device(self, n, count, d_in, d_out, stream=None):
        err = C.c_char_p()
        rc = self.lib.vsyn_imdct_device(self.h, n, count, d_in, d_out, stream, C.byref(err))
        if rc != VSYN_OK:
            raise VsynError(rc, (err.value or b"").decode())

    def profile(self, on=1):
        """0/False off, 1/True long-run fused kernel, 2 mixed-block fused kernel."""
        self.lib.vsyn_profile_enable(self.h, int(on))

    def profile_read(self):
        ms, n, name = C.c_double(), C.c_uint32(), C.c_char_p()
        self.lib.vsyn_profile_read(self.h, C.byref(ms), C.byref(n), C.byref(name))
        return ms.value, n.value, (name.value or b"").decode()
