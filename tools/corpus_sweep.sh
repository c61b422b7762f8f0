# tools/corpus_sweep.sh — end-to-end corpus throughput (host entropy threads + GPU feeders) on the GPU box; run from the repo root.
# VQ=1 (default): the residue leaves the host as entry numbers (device VQ stage); VQ=0: expanded floats.
set -e
B=parseoggvorbis_amd/host/corpus_hip.bin
F=tests/golden/test.stereo44khz.ogg
nproc
for vq in 1 0; do
  export PARSEOGGVORBIS_VQ=$vq
  echo "== PARSEOGGVORBIS_VQ=$vq"
  for t in 1 16; do timeout -k 10 120 $B --threads $t --replicas 20000 --entropy_only $F | cut -c1-330; done
  for cfg in "14 1 64" "14 2 64" "16 2 64" "16 3 64" "24 2 64" "32 3 64"; do
    set -- $cfg
    timeout -k 10 120 $B --threads $1 --feeders $2 --files_per_submit $3 --replicas 20000 $F
  done
done
unset PARSEOGGVORBIS_VQ
timeout -k 10 120 $B --threads 16 --feeders 2 --files_per_submit 64 --replicas 10000 $F tests/golden/test.mono44khz.ogg
