for v in 0 f32 0 f32; do
echo "=== VMASR_MPD_CONV_L1=$v"
VMASR_MPD_CONV_L1=$v timeout 900 python -m pytest tests/test_trainer.py tests/test_fullsize.py -m gpu -q -x -k "b4 or lane or gan or two_stream or capture" 2>&1 | grep -v Warn | grep -v "^  File\|^$" | head -14 | cut -c1-400
done
