mkdir -p gpurun_out/hunt
( timeout 1500 python tools/det_hunt.py --iters 700 --trace ) > gpurun_out/hunt/K_trace.log 2>&1
grep -c "^it" gpurun_out/hunt/K_trace.log; grep -A6 "^it" gpurun_out/hunt/K_trace.log | head -60; tail -n 2 gpurun_out/hunt/K_trace.log
