mkdir -p gpurun_out/r2f
bash tools/ab_variants.sh r2f w1
cp build_ab/lib_w1st.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-overlap 2> gpurun_out/r2f/stamps_noov.txt | python tools/bench_line.py stamped-noov >> gpurun_out/r2f/summary.txt
cat gpurun_out/r2f/stamps_noov.txt
cp build_ab/lib_w1.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python tools/bench_line.py driver-like
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --workload config4 | python tools/bench_line.py config4
