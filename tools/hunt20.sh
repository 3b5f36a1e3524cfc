timeout 1500 python -m pytest tests/test_mpd.py tests/test_trainstep.py tests/test_determinism.py -m gpu -q -x 2>&1 | grep -v Warn | tail -8
timeout 1500 python -m pytest tests/test_trainer.py -m gpu -q -x 2>&1 | grep -v Warn | grep -v "^$" | head -60
