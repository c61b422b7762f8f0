"""GPU: PCM post-stage — planar f32 of the last submit -> interleaved int16 / f32.
Checked exactly against the oracle's restatement of ov_read's conversion (round to nearest even of x*32768.f, clamp),
including the half-way cases, the clamp edges and ragged segment lengths. Parity with the reference itself is UNPINNED
for this stage (see oracle/vorbis_synth_oracle.h): the rule is restated from vorbis_vorbisfile.c:2026-2029."""
import numpy as np
import pytest
import torch

from oracle import oracle_binding as ob
from parseoggvorbis_amd.binding import VSYN_PCM_F32, VSYN_PCM_S16, Synth
from tests.workloads import fixture_like_spec, synth_batch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


@pytest.mark.parametrize("channels,pattern", [(2, "long"), (2, "mixed"), (1, "mixed")])
def test_pcm_stage_matches_oracle(channels, pattern):
    spec = fixture_like_spec(channels)
    b = synth_batch(spec, 5, 23, pattern=pattern, seed=11, granule_last=True)
    syn = Synth(spec, max_streams=5)
    P, S, plane = len(b["packets"]), len(b["segments"]), b["plane_stride"]
    d_pk, d_sg, d_ys, d_res = _dev(b["packets"]), _dev(b["segments"]), _dev(b["ys"]), _dev(b["residue"])
    d_pcm = torch.zeros((S, channels, plane), device="cuda")
    d_emit = torch.zeros(P, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    syn.submit_device(P, d_pk.data_ptr(), S, d_sg.data_ptr(), 23, d_ys.data_ptr(), d_res.data_ptr(), d_pcm.data_ptr(), plane,
                      d_emit.data_ptr(), None, 0, stream)
    # plant values that exercise the rounding rule and the clamp in the emitted region of segment 0
    special = torch.tensor([0.5 / 32768, 1.5 / 32768, 2.5 / 32768, -0.5 / 32768, -1.5 / 32768, -2.5 / 32768, 1.0, -1.0, 0.99999, 1.7,
                            -1.7, 32766.5 / 32768, 32767.5 / 32768, -32768.5 / 32768, 0.0, -0.0, 3.0e-7, 123.456 / 32768],
                           device="cuda")
    d_pcm[0, 0, :special.numel()] = special
    d_pcm[0, channels - 1, 40:40 + special.numel()] = -special
    out_stride = plane - 7  # a different (odd) stride on the way out
    d_s16 = torch.full((S, out_stride, channels), 12345, dtype=torch.int16, device="cuda")
    d_f32 = torch.full((S, out_stride, channels), 7.0, device="cuda")
    d_frames = torch.zeros(S, dtype=torch.int32, device="cuda")
    syn.pcm_interleave_device(VSYN_PCM_S16, d_pcm.data_ptr(), plane, d_s16.data_ptr(), out_stride, d_frames.data_ptr(), stream)
    syn.pcm_interleave_device(VSYN_PCM_F32, d_pcm.data_ptr(), plane, d_f32.data_ptr(), out_stride, None, stream)
    torch.cuda.synchronize()
    assert syn.sync_status(stream)[0] == 0
    emit = d_emit.cpu().numpy().astype(np.int64)
    frames = d_frames.cpu().numpy()
    pcm = d_pcm.cpu().numpy()
    s16, f32 = d_s16.cpu().numpy(), d_f32.cpu().numpy()
    for g in range(S):
        n = int(emit[g * 23:(g + 1) * 23].sum())
        assert frames[g] == n and 0 < n <= out_stride
        want16 = ob.pcm_interleave(VSYN_PCM_S16, pcm[g], n)
        want32 = ob.pcm_interleave(VSYN_PCM_F32, pcm[g], n)
        assert np.array_equal(s16[g, :n], want16)
        assert np.array_equal(f32[g, :n].view(np.uint32), want32.view(np.uint32))
        assert (s16[g, n:] == 12345).all() and (f32[g, n:] == 7.0).all()  # nothing written past the emitted frames
    # the half-way cases round to even, the edges clamp
    got = s16[0, :special.numel(), 0].tolist()
    assert got[:8] == [0, 2, 2, 0, -2, -2, 32767, -32768] and got[9:14] == [32767, -32768, 32766, 32767, -32768]


def test_pcm_abs_sum_host_matches_numpy():
    """The per-(segment, channel) digest of the last host submit (what the corpus decoder reports per file): equals the sum of |x|
    over the emitted frames to double rounding, is reproducible bit for bit, and ignores what lies beyond the emitted frames."""
    import numpy as np
    from parseoggvorbis_amd import binding
    from tests.workloads import fixture_like_spec, synth_batch
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 5, 23, "mixed", seed=4, granule_last=True)
    gpu = binding.Synth(spec, max_streams=5)
    with pytest.raises(binding.VsynError):
        gpu.pcm_abs_sum_host(5)  # nothing submitted yet
    r = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert r["rc"] == 0
    d1 = gpu.pcm_abs_sum_host(5)
    per = len(b["packets"]) // 5
    for s in range(5):
        frames = int(r["emit_len"][s * per:(s + 1) * per].sum())
        want = np.abs(r["pcm"][s][:, :frames].astype(np.float64)).sum(axis=1)
        assert np.allclose(d1[s], want, rtol=1e-12, atol=0)
    r2 = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert r2["rc"] == 0 and np.array_equal(gpu.pcm_abs_sum_host(5).view(np.uint64), d1.view(np.uint64))


def test_pcm_fetch_host_converted_on_the_device():
    """Host path of the post-stage: VSYN_SUBMIT_KEEP_PCM leaves the planar f32 PCM on the device (nothing comes back with the
    submit), vsyn_pcm_fetch_host brings it back interleaved as int16 (half the bytes; ov_read's conversion, exact against the
    oracle's restatement) or as f32 (the planar values, interleaved)."""
    import numpy as np
    from parseoggvorbis_amd import binding
    from parseoggvorbis_amd.binding import VSYN_SUBMIT_KEEP_PCM
    spec = fixture_like_spec(2)
    b = synth_batch(spec, 3, 21, "mixed", seed=12, granule_last=True)
    gpu = binding.Synth(spec, max_streams=3)
    ref = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"])
    assert ref["rc"] == 0
    gpu.reset()
    kept = gpu.submit_host(b["packets"], b["segments"], b["ys"], b["residue"], b["plane_stride"], flags=VSYN_SUBMIT_KEEP_PCM)
    assert kept["rc"] == 0 and not kept["pcm"].any() and np.array_equal(kept["emit_len"], ref["emit_len"])
    per = len(b["packets"]) // 3
    stride = b["plane_stride"]
    s16, frames = gpu.pcm_fetch_host(VSYN_PCM_S16, 3, stride)
    f32, frames2 = gpu.pcm_fetch_host(VSYN_PCM_F32, 3, stride)
    assert np.array_equal(frames, frames2)
    for s in range(3):
        n = int(ref["emit_len"][s * per:(s + 1) * per].sum())
        assert int(frames[s]) == n
        assert np.array_equal(f32[s, :n], ref["pcm"][s][:, :n].T)
        want = ob.pcm_interleave(VSYN_PCM_S16, np.ascontiguousarray(ref["pcm"][s][:, :n]), n)
        assert np.array_equal(s16[s, :n], want)
        assert not s16[s, n:].any()
