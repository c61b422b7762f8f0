# tools/corpus_sweep.sh — end-to-end corpus throughput (host entropy threads + GPU feeders) on the GPU box; run from the repo root.
set -e
B=parseoggvorbis_amd/host/corpus_hip.bin
F=tests/golden/test.stereo44khz.ogg
nproc
for t in 1 16 64; do timeout -k 10 120 $B --threads $t --replicas 20000 --entropy_only $F | cut -c1-330; done
for cfg in "16 1 64" "16 2 64" "32 1 64" "32 2 64" "32 3 64" "32 4 64" "64 3 64" "64 4 32" "64 4 128" "96 6 64"; do
  set -- $cfg
  timeout -k 10 120 $B --threads $1 --feeders $2 --files_per_submit $3 --replicas 20000 $F
done
timeout -k 10 120 $B --threads 32 --feeders 3 --files_per_submit 64 --replicas 10000 $F tests/golden/test.mono44khz.ogg
