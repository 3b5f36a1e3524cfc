mkdir -p gpurun_out/hunt
for i in $(seq 1 12); do
  timeout 300 python -m pytest tests/test_abi.py tests/test_benchfmt.py tests/test_ckpt_fixture.py tests/test_config.py tests/test_convgemm.py tests/test_determinism.py -q -m gpu -x 2>&1 | tail -n 4 > gpurun_out/hunt/E_pytest_$i.log
  grep -h "passed\|failed" gpurun_out/hunt/E_pytest_$i.log
done
for i in $(seq 1 6); do
  timeout 300 python -m pytest tests/test_determinism.py -q -m gpu -x 2>&1 | tail -n 4 > gpurun_out/hunt/F_pytest_$i.log
  grep -h "passed\|failed" gpurun_out/hunt/F_pytest_$i.log
done
