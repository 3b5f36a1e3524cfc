mkdir -p gpurun_out/hunt
for i in $(seq 1 60); do
  timeout 300 python tools/det_hunt.py --iters 2 2>&1 | grep -h "RESULT\|differ" | sed "s/^/P$i: /"
done > gpurun_out/hunt/G_fresh.log 2>&1
grep -c "bad_iters=0" gpurun_out/hunt/G_fresh.log; grep -v "bad_iters=0" gpurun_out/hunt/G_fresh.log | head -20
