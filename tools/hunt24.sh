# CU-limit sweep of the discriminator's convolution kernels with round 6's kernels (lane on: VMASR_STEP_VARIANT pins the phase lane)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/side_cus_sweep_b4.log; : > $out
for bw in 176 192 208 224; do for fw in 80 96 112; do
  v=$(VMASR_STEP_VARIANT=lane VMASR_SIDE_CUS=$bw VMASR_SIDE_CUS_FWD=$fw python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
  echo "fwd $fw bwd $bw : $v" | tee -a $out
done; done
v=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
echo "default (derived limits, layout chosen by timing): $v" | tee -a $out
