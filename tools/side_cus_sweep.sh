# Sweep of the CU limit of the discriminator's convolution kernels while the two streams overlap (trainer._side_cus), per workload / batch:
# forward limit (beside the generator's forward) x backward limit (beside its backward).  -> gpurun_out/r05/side_cus_sweep_<tag>.log
mkdir -p gpurun_out/r05
run() {  # tag, extra bench args
  tag=$1; shift
  out=gpurun_out/r05/side_cus_sweep_$tag.log; : > $out
  for bw in 0 128 160 192 224; do for fw in 96 128 160; do
    if [ $bw = 0 ] && [ $fw != 96 ]; then continue; fi
    if [ $bw = 0 ]; then f=0; else f=$fw; fi
    v=$(VMASR_SIDE_CUS=$bw VMASR_SIDE_CUS_FWD=$f python bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-extra-points --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "fwd $f bwd $bw : $v" | tee -a $out
  done; done
}
run b4
run b8 --batch 8
run vssm32 --workload vm_asr_48k_16k_MPD_VSSM32
run b2 --batch 2
