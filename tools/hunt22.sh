mkdir -p gpurun_out/hunt
for v in f32 0; do
for i in 1 2 3 4 5 6 7 8; do
VMASR_MPD_CONV_L1=$v timeout 300 python -m pytest tests/test_trainer.py -m gpu -q -x -k "two_stream_step_gives or wgan_gp or phase_lane_matches or capture_keeps" > gpurun_out/hunt/V_${v}_$i.log 2>&1
rc=$?
echo "L1=$v run $i rc=$rc $(grep -h 'passed\|failed' gpurun_out/hunt/V_${v}_$i.log | tail -1)"
if [ $rc -ne 0 ]; then grep -v "^  File\|Warn" gpurun_out/hunt/V_${v}_$i.log | grep -i "fatal\|fault\|abort\|error\|HSA\|hip" | head -8 | cut -c1-300; fi
done; done
