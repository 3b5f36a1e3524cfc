mkdir -p gpurun_out/hunt
export PYTORCH_TUNABLEOP_FILENAME=/tmp/tunableop_%d.csv
( PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20 timeout 1500 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail /tmp/b.json 2>gpurun_out/hunt/tunable.err | tail -n 1 | cut -c1-300 ) 
ls -la /tmp/tunableop_*.csv 2>/dev/null; wc -l /tmp/tunableop_0.csv; head -5 /tmp/tunableop_0.csv; cp /tmp/tunableop_0.csv gpurun_out/hunt/tunableop_0.csv
tail -n 5 gpurun_out/hunt/tunable.err
# second run: tuned results from the file, no tuning
( PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 timeout 600 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail /tmp/b.json 2>/dev/null | tail -n 1 | cut -c1-200 )
( timeout 600 python bench.py --no-cpu-baseline --no-extra-points --no-kernel-timing --steps 60 --warmup 10 --detail /tmp/b.json 2>/dev/null | tail -n 1 | cut -c1-200 )
