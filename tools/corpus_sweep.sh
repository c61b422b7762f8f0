# tools/corpus_sweep.sh — end-to-end corpus throughput (host entropy threads + GPU feeders) on the GPU box; run from the repo root.
# VQ=1 (default): the residue leaves the host as entry numbers (device VQ stage); VQ=0: expanded floats.
set -e
B=parseoggvorbis_amd/host/corpus_hip.bin
F=tests/golden/test.stereo44khz.ogg
nproc
echo "== entropy half only (no GPU): threads 1 / 16, with and without the shared setup-header cache"
for t in 1 16; do timeout -k 10 120 $B --threads $t --replicas 20000 --entropy_only $F | cut -c1-420; done
timeout -k 10 120 $B --threads 16 --replicas 20000 --entropy_only --no_setup_cache $F | cut -c1-420
echo "== end to end"
for cfg in "12 2 64" "13 3 64" "14 2 64" "14 3 64" "16 3 64" "16 4 64" "24 4 64"; do
  set -- $cfg
  timeout -k 10 120 $B --threads $1 --feeders $2 --files_per_submit $3 --replicas 20000 $F
done
echo "== end to end without the per-file checksum pass"
for cfg in "13 3 64" "14 2 64" "16 3 64"; do
  set -- $cfg
  timeout -k 10 120 $B --threads $1 --feeders $2 --files_per_submit $3 --replicas 20000 --no_checksum $F
done
echo "== float residue instead of entry numbers (PARSEOGGVORBIS_VQ=0)"
PARSEOGGVORBIS_VQ=0 timeout -k 10 120 $B --threads 14 --feeders 3 --files_per_submit 64 --replicas 20000 $F
echo "== two different setups in one run"
timeout -k 10 120 $B --threads 14 --feeders 3 --files_per_submit 64 --replicas 10000 $F tests/golden/test.mono44khz.ogg
