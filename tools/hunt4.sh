mkdir -p gpurun_out/hunt
( timeout 1200 python tools/det_hunt.py --iters 600 ) > gpurun_out/hunt/H_two_det_600.log 2>&1
( timeout 1200 python tools/det_hunt.py --iters 300 --batch 4 ) > gpurun_out/hunt/I_two_det_b4_300.log 2>&1
( timeout 1200 python tools/det_hunt.py --iters 300 --watch --det 0 2>&1 | grep -v "modified=0" ) > gpurun_out/hunt/J_two_nodet_watch_300.log 2>&1
tail -n 5 gpurun_out/hunt/H_*.log gpurun_out/hunt/I_*.log; grep -c . gpurun_out/hunt/J_*.log;  grep "MODIFIED\|RESULT" gpurun_out/hunt/J_*.log | head
