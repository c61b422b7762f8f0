# tools/soak.sh — stability run on the GPU box: the corpus decoder over 120 000 files (both fixtures, 16 workers + 4 feeders)
# three times, then 2000-step runs of the mixed-block and the VQ workloads. Run from the repo root.
set -e
B=parseoggvorbis_amd/host/corpus_hip.bin
for i in 1 2 3; do
  timeout -k 10 300 $B --threads 16 --feeders 4 --files_per_submit 64 --replicas 60000 tests/golden/test.stereo44khz.ogg tests/golden/test.mono44khz.ogg
done
timeout -k 10 300 $B --threads 16 --feeders 3 --files_per_submit 64 --replicas 60000 --s16 tests/golden/test.stereo44khz.ogg tests/golden/test.mono44khz.ogg
timeout -k 10 300 python bench.py --steps 2000 --no-cpu-baseline --feature-taps
for w in config4 config3_vq config4 config3 config4 config3_vq; do
  timeout -k 10 300 python bench.py --steps 2000 --no-cpu-baseline --workload $w
done
