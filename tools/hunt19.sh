timeout 1500 python -m pytest tests/test_convgemm.py tests/test_mpd.py tests/test_trainstep.py tests/test_determinism.py -m gpu -q -x 2>&1 | grep -v Warn | tail -6
timeout 1500 python -m pytest tests/test_trainer.py tests/test_fullsize.py -m gpu -q -x -k "b4 or lane or gan or two_stream or capture" 2>&1 | grep -v Warn | tail -4
for rep in 1 2; do
for v in 0 f32; do
  VMASR_MPD_CONV_L1=$v timeout 600 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail /tmp/b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('L1=$v', round(d['value'],2), round(d['ms_per_step'],3))"
done; done
