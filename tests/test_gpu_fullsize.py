"""GPU, BASELINE.json full sizes (config 3: 65 536 stereo packets of blocksize 2048; config 4: 32 768 mixed packets),
where the CPU oracle would take too long: size-independent properties of the path instead.

 * homogeneity: the whole path is linear in the residue for a fixed floor, and scaling by a power of two is exact in
   binary32 (coupling only looks at signs) -> pcm(2*residue) == 2*pcm(residue) BIT FOR BIT;
 * stream independence: streams that are copies of each other produce identical PCM wherever they sit in the batch
   (a checksum of checksums over the 64 streams);
 * locality of the overlap: a stream restarted at packet k reproduces the original PCM from packet k+1 on;
 * plus an oracle spot check on two streams and exact emitted-frame counts.
"""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from parseoggvorbis_amd import binding

pytestmark = pytest.mark.gpu


def _submit(gpu, b, res, pcm, emit, stream, segs=None, S=None):
    gpu.submit_device(b["P"], b["packets"].data_ptr(), S or b["S"], (segs if segs is not None else b["segments"]).data_ptr(), b["ppk"],
                      b["ys"].data_ptr(), res.data_ptr(), pcm.data_ptr(), b["plane"], emit.data_ptr(), None, 0, stream)


@pytest.mark.parametrize("workload,ppk,bs", [("long", 1024, (256, 2048)), ("mixed", 512, (256, 2048)),
                                             ("long", 1024, (128, 1024)), ("mixed", 512, (128, 1024)), ("long", 512, (512, 4096))])
def test_full_size_properties(workload, ppk, bs):
    """(256, 2048): the tuned kernel's steady and mixed paths; (128, 1024), the long block of 16-22 kHz material, and (512, 4096):
    the size-generic kernel (one / two register sets) at the same batch sizes."""
    import torch
    import bench
    from tests.workloads import fixture_like_spec
    spec = fixture_like_spec(2, *bs)
    dev = torch.device("cuda", 0)
    S = 64
    b = bench.build_batch(spec, S, ppk, workload, 1234, dev)
    # make streams 32..63 exact copies of streams 0..31 (descriptors already identical per stream)
    psf = b["per_stream_floats"]
    res = b["residue"].view(S, psf)
    res[32:] = res[:32]
    ysv = b["ys"].view(S, ppk, 2, -1)
    ysv[32:] = ysv[:32]
    gpu = binding.Synth(spec, max_streams=S)
    assert gpu.fused_paths & 2, gpu.fused_paths  # every one of these setups stays on a fused kernel
    stream = torch.cuda.current_stream().cuda_stream
    pcm1 = torch.zeros((S, 2, b["plane"]), device=dev)
    pcm2 = torch.zeros_like(pcm1)
    emit1 = torch.zeros(b["P"], dtype=torch.int32, device=dev)
    emit2 = torch.zeros_like(emit1)
    _submit(gpu, b, b["residue"], pcm1, emit1, stream)
    res2 = (b["residue"] * 2.0).contiguous()
    _submit(gpu, b, res2, pcm2, emit2, stream)
    fl, bad = gpu.sync_status(stream)
    assert fl == 0, (fl, bad)

    # emitted frames: every packet but the first of a stream emits prev/4 + cur/4
    n_of = torch.from_numpy(b["n_of"].astype(np.int64)).to(dev)
    want_emit = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), n_of[:-1] // 4 + n_of[1:] // 4]).repeat(S)
    assert torch.equal(emit1.to(torch.int64), want_emit) and torch.equal(emit1, emit2)
    total = int(want_emit[:ppk].sum())
    assert total <= b["plane"]

    # homogeneity, bit for bit
    assert torch.equal(pcm2.view(torch.int32), (pcm1 * 2.0).view(torch.int32))
    assert float(pcm1.abs().max()) > 0.05 and not bool(torch.isnan(pcm1).any())

    # stream independence: checksum of checksums
    sums = pcm1[:, :, :total].view(torch.int32).to(torch.int64).sum(dim=(1, 2))
    assert torch.equal(sums[:32], sums[32:])
    assert torch.equal(pcm1[:32], pcm1[32:])
    assert len(set(sums[:32].tolist())) > 16  # and the base streams do differ from each other

    # oracle spot check on streams 0 and 63 (first and last workgroups of the grid)
    hp = b["host_packets"]
    for s in (0, S - 1):
        seg = b["host_segments"][s:s + 1].copy()
        seg["first_packet"], seg["residue_off"], seg["stream"] = 0, 0, 0
        want = ob.OracleSynth(spec, 1).submit_host(hp[s * ppk:(s + 1) * ppk], seg, b["ys"][s * ppk:(s + 1) * ppk].cpu().numpy().view(np.uint16),
                                                   res[s].cpu().numpy(), b["plane"])
        assert want["rc"] == 0
        assert np.abs(pcm1[s].cpu().numpy() - want["pcm"][0]).max() < 1e-5

    # locality: restart every stream at packet k; from packet k+1 on the PCM is the original one
    k = 37
    segs = b["host_segments"].copy()
    res_skip = int((2 * b["n_of"][:k] // 2).sum())
    segs["first_packet"] += k
    segs["num_packets"] -= k
    segs["residue_off"] += res_skip
    dsegs = torch.from_numpy(segs.view(np.uint8)).to(dev)
    pcm3 = torch.zeros_like(pcm1)
    gpu.reset(stream)
    _submit(gpu, b, b["residue"], pcm3, emit2, stream, segs=dsegs)
    fl, bad = gpu.sync_status(stream)
    assert fl == 0
    skipped = int(want_emit[:k + 1].sum())          # samples of packets 0..k in the original
    first = int(want_emit[k + 1])                    # packet k+1 is the restarted stream's first emitting packet
    tail = total - skipped
    assert tail > 0 and first > 0
    assert torch.equal(pcm3[:, :, :tail], pcm1[:, :, skipped:total])
