# tools/soak_round.sh — soak of the round's final build on the GPU box: a large corpus run (120 000 files, both fixtures alternating) three
# times, then 2000-step runs of every kernel workload (results to gpurun_out/soak_round.log; every bench line is gated on its oracle check).
# Stops at the first step that fails or times out: no further GPU step is started behind a failed one.
B=parseoggvorbis_amd/host/corpus_hip.bin
O=gpurun_out/soak_round.log; : > $O
for i in 1 2 3; do timeout -k 10 300 $B --threads 16 --feeders 3 --replicas 60000 tests/golden/test.stereo44khz.ogg tests/golden/test.mono44khz.ogg >> $O 2>&1 || { echo "corpus run $i FAILED rc=$?"; exit 1; }; echo "corpus run $i ok"; done
timeout -k 10 300 $B --threads 16 --feeders 3 --s16 --replicas 60000 tests/golden/test.stereo44khz.ogg tests/golden/test.mono44khz.ogg >> $O 2>&1 || { echo "corpus s16 FAILED rc=$?"; exit 1; }; echo "corpus s16 ok"
for w in "" "--workload config4" "--workload config3_vq" "--workload config3_vq --vq-books fixture" "--blocksizes 128,1024" "--blocksizes 128,1024 --workload config4" "--feature-taps" "--pcm-s16"; do
  timeout -k 10 400 python bench.py --steps 2000 --warmup 10 --no-cpu-baseline $w >> $O 2>> gpurun_out/soak_round.err || { echo "bench [$w] FAILED rc=$?"; exit 1; }; echo "bench [$w] ok"
done
