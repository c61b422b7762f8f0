B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10"
cp build_ab/lib_p2.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_host_decoder.py -x -q > gpurun_out/pytest_p2.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_p2.log)"
$B --workload config4 | python tools/bench_line.py "config4"
$B --workload config4 --no-overlap | python tools/bench_line.py "config4, no overlap"
cp build_ab/lib_p2st.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --workload config4 --no-overlap 2> gpurun_out/stamps_c4b.txt | python tools/bench_line.py "config4 stamped"
grep -v amdgpu gpurun_out/stamps_c4b.txt | tail -10
