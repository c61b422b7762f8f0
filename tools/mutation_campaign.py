"""tools/mutation_campaign.py <seed> <count> — authoring container only (needs oracle/_ref/ours.bin + ours_asan.bin): the committed
.ogg fixtures with 1-3 damaged bytes (page CRCs recomputed so that the damage reaches the codec layer) through the REFERENCE decoder
and through the host entropy half (tests/host_entropy_dump.cpp). Counts: both accept with identical hooks / both reject / one
accepts. Round 1: 600 files — 364 both accept, always identical; 208 both reject; 0 where the reference accepts and the host
rejects; 19 that only the reference rejects, all in its synthesis half (hpp:536 / 1041 / Utils.hpp:145), which the full CLI
rejects on the GPU with the matching reason; 9 skipped (memory error inside the reference)."""
import sys, os, subprocess, tempfile, struct
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import make_synth_ogg as m
from tests import workloads as w
first, count = int(sys.argv[1]), int(sys.argv[2])
td=tempfile.mkdtemp()
exe=w.build_probe(td)
GOLD='/root/repo/tests/golden'
bases=[open(os.path.join(GOLD,f),'rb').read() for f in sorted(os.listdir(GOLD)) if f.endswith('.ogg')]
def refix_crc(buf):
    # recompute every page CRC so that the damage reaches the codec layer
    b=bytearray(buf); pos=0
    while pos+27<=len(b) and b[pos:pos+4]==b'OggS':
        nseg=b[pos+26]
        if pos+27+nseg>len(b): break
        ln=sum(b[pos+27:pos+27+nseg]); end=pos+27+nseg+ln
        if end>len(b): break
        b[pos+22:pos+26]=b'\0\0\0\0'
        crc=0
        for byte in b[pos:end]:
            crc=((crc<<8)&0xFFFFFFFF)^m._CRC[((crc>>24)&0xFF)^byte]
        b[pos+22:pos+26]=struct.pack('<I',crc)
        pos=end
    return bytes(b)
stats=dict(both_ok_equal=0, both_ok_diff=0, both_reject=0, ref_ok_ours_reject=0, ref_reject_ours_ok=0, ref_memerr=0)
rng=np.random.default_rng(first)
for it in range(count):
    base=bytearray(bases[int(rng.integers(0,len(bases)))])
    # skip the first page (identification) mostly; damage 1-3 bytes/bits somewhere after
    for _ in range(int(rng.integers(1,4))):
        pos=int(rng.integers(60,len(base)))
        if rng.random()<0.5: base[pos]^=1<<int(rng.integers(0,8))
        else: base[pos]=int(rng.integers(0,256))
    data=refix_crc(bytes(base))
    path=os.path.join(td,'t.ogg'); open(path,'wb').write(data)
    vec,why=m.reference_vectors(path)
    ref_ok = vec is not None
    if not ref_ok and 'AddressSanitizer' in why: stats['ref_memerr']+=1; continue
    out=os.path.join(td,'d.bin')
    r=subprocess.run([exe,path,out],capture_output=True,env=dict(os.environ,PARSEOGGVORBIS_VQ='0'))
    ours_ok = r.returncode==0
    if ref_ok and ours_ok:
        d=w.read_entropy_dump(out)
        same = d['P']==int(vec['packets']) and d['residue'].size==vec['residue'].size and np.array_equal(d['residue'].view(np.uint32),vec['residue'].view(np.uint32))
        if same:
            off=0
            for ln,where in zip(vec['ys_len'],vec['ys_where']):
                p,c=divmod(int(where),1000)
                if not np.array_equal(d['ys'][p,c,:ln].astype(np.uint32),vec['ys'][off:off+ln]): same=False; break
                off+=ln
        stats['both_ok_equal' if same else 'both_ok_diff']+=1
        if not same: print('DIFF seed',first,'it',it, flush=True); open('/tmp/diff_%d_%d.ogg'%(first,it),'wb').write(data)
    elif not ref_ok and not ours_ok: stats['both_reject']+=1
    elif ref_ok: stats['ref_ok_ours_reject']+=1; print('ours rejects, ref ok: it',it,r.stderr.decode()[-150:].strip(), flush=True); open('/tmp/rej_%d_%d.ogg'%(first,it),'wb').write(data)
    else: stats['ref_reject_ours_ok']+=1; print('ref rejects, ours ok: it',it,why[-150:].replace('\n',' | '), flush=True); open('/tmp/acc_%d_%d.ogg'%(first,it),'wb').write(data)
print(stats, flush=True)
