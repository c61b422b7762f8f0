"""tools/leak_check.py — GPU box: 60 handles created, used through every host entry point and destroyed; device memory must not shrink."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from parseoggvorbis_amd import binding
from parseoggvorbis_amd.binding import VSYN_PCM_S16, VSYN_SUBMIT_KEEP_PCM
from tests.workloads import fixture_like_spec, synth_batch
spec=fixture_like_spec(2)
b=synth_batch(spec,4,40,"mixed",seed=1)
free0=None
for i in range(60):
    g=binding.Synth(spec,max_streams=4)
    r=g.submit_host(b["packets"],b["segments"],b["ys"],b["residue"],b["plane_stride"],flags=VSYN_SUBMIT_KEEP_PCM)
    g.pcm_fetch_host(VSYN_PCM_S16,4,b["plane_stride"]); g.pcm_abs_sum_host(4)
    g.close() if hasattr(g,'close') else None
    del g
    if i==9: free0=torch.cuda.mem_get_info()[0]
free1=torch.cuda.mem_get_info()[0]
print("free after 10 handles %d MB, after 60 handles %d MB, delta %d KB"%(free0>>20, free1>>20, (free0-free1)>>10))
assert free0-free1 < 8<<20, "device memory leak across handle create/destroy"
print("no leak")
