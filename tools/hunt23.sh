mkdir -p gpurun_out/hunt
n=0
for i in $(seq 1 ${RUNS:-14}); do
timeout 300 python -m pytest tests/test_trainer.py -m gpu -q -x -k "two_stream_step_gives or wgan_gp or phase_lane_matches or capture_keeps" > gpurun_out/hunt/X_$i.log 2>&1
rc=$?
[ $rc -ne 0 ] && n=$((n+1)) && echo "run $i rc=$rc" && grep -v "^  File\|Warn" gpurun_out/hunt/X_$i.log | head -5 | cut -c1-200
done
echo "GC guard: $n crashes of ${RUNS:-14}"
