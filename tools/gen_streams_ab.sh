# A/B of the generator's phase branch on a second stream (VMASR_GEN_STREAMS=2) in captured steps: the full GAN step and the generator-only step
B="--no-cpu-baseline --no-extra-points --no-kernel-timing --steps 30 --warmup 8"
run() { name=$1; shift; printf "%-44s" "$name"; timeout 200 env "$@" python bench.py $B ${EXTRA} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), 'clips/s', d.get('ms_per_step'), 'ms', d.get('error') or '')" ; }
for g in 1 2 auto; do
EXTRA="" run "full step B=4 gen_streams=$g" VMASR_GEN_STREAMS=$g
EXTRA="--batch 8" run "full step B=8 gen_streams=$g" VMASR_GEN_STREAMS=$g
EXTRA="--batch 35" run "full step B=35 gen_streams=$g" VMASR_GEN_STREAMS=$g
for b in 4 8 16 35; do EXTRA="--workload vm_asr_48k --batch $b" run "generator only B=$b gen_streams=$g" VMASR_GEN_STREAMS=$g; done
done
EXTRA="--batch 2" run "full step B=2 gen_streams=1" VMASR_GEN_STREAMS=1
EXTRA="--batch 2" run "full step B=2 gen_streams=2" VMASR_GEN_STREAMS=2
for w in vm_asr_48k_16k_MPD_VSSM32 vm_asr_48k_16k_nfft2048; do for g in 1 2; do EXTRA="--workload $w" run "$w gen_streams=$g" VMASR_GEN_STREAMS=$g; done; done
