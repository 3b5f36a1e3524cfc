for rep in 1 2; do
for c in 2048 512 256; do
  VMASR_SPLITK_CHUNK=$c timeout 600 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail /tmp/b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('chunk=$c', round(d['value'],2), round(d['ms_per_step'],3))"
done; done
