mkdir -p gpurun_out/r2s
timeout -k 10 900 python tools/stress_mixed.py 1000 120 > gpurun_out/r2s/stress_mixed.log 2>&1; echo "stress_mixed rc=$? $(tail -1 gpurun_out/r2s/stress_mixed.log)"
VSYN_U_MIXED=1 timeout -k 10 900 python tools/stress_mixed.py 2000 60 > gpurun_out/r2s/stress_mixed_u.log 2>&1; echo "stress_mixed (generic kernel) rc=$? $(tail -1 gpurun_out/r2s/stress_mixed_u.log)"
timeout -k 10 600 python tools/stress_vq.py > gpurun_out/r2s/stress_vq.log 2>&1; echo "stress_vq rc=$? $(tail -1 gpurun_out/r2s/stress_vq.log)"
timeout -k 10 600 python tools/leak_check.py > gpurun_out/r2s/leak.log 2>&1; echo "leak_check rc=$? $(tail -1 gpurun_out/r2s/leak.log)"
