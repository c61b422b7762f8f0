cp parseoggvorbis_amd/csrc/libvorbis_synth_hip.so /tmp/keep.so
cp build_ab/lib_dmast.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --workload config4 --no-overlap 2> gpurun_out/stamps_c4.txt | python tools/bench_line.py "config4 stamped"
grep -v amdgpu gpurun_out/stamps_c4.txt
cp /tmp/keep.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_host_decoder.py -x -q -k "mid_stream" 2>&1 | tail -2
