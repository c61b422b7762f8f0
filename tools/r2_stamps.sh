mkdir -p gpurun_out/r2e
cp build_ab/lib_st.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-overlap 2> gpurun_out/r2e/stamps_noov.txt | python tools/bench_line.py noov > gpurun_out/r2e/summary.txt
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/r2e/stamps.txt | python tools/bench_line.py ov >> gpurun_out/r2e/summary.txt
cat gpurun_out/r2e/summary.txt gpurun_out/r2e/stamps_noov.txt gpurun_out/r2e/stamps.txt
