# tools/stress_round.sh — the round's stress set on the GPU box (each step bounded by its own timeout); logs under gpurun_out/stress_round/
mkdir -p gpurun_out/stress_round
timeout -k 10 900 python tools/stress_mixed.py 1000 120 > gpurun_out/stress_round/stress_mixed.log 2>&1; echo "stress_mixed rc=$? $(tail -1 gpurun_out/stress_round/stress_mixed.log)"
VSYN_U_MIXED=1 timeout -k 10 900 python tools/stress_mixed.py 2000 60 > gpurun_out/stress_round/stress_mixed_u.log 2>&1; echo "stress_mixed (generic kernel) rc=$? $(tail -1 gpurun_out/stress_round/stress_mixed_u.log)"
timeout -k 10 600 python tools/stress_vq.py > gpurun_out/stress_round/stress_vq.log 2>&1; echo "stress_vq rc=$? $(tail -1 gpurun_out/stress_round/stress_vq.log)"
VSYN_VQ_NO_LDS_TABLES=1 timeout -k 10 600 python tools/stress_vq.py > gpurun_out/stress_round/stress_vq_global_tables.log 2>&1; echo "stress_vq (tables in global memory) rc=$? $(tail -1 gpurun_out/stress_round/stress_vq_global_tables.log)"
timeout -k 10 900 python tools/stress_u.py > gpurun_out/stress_round/stress_u.log 2>&1; echo "stress_u rc=$? $(tail -1 gpurun_out/stress_round/stress_u.log)"
timeout -k 10 600 python tools/leak_check.py > gpurun_out/stress_round/leak.log 2>&1; echo "leak_check rc=$? $(tail -1 gpurun_out/stress_round/leak.log)"
