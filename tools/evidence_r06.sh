# round-6 evidence: the full GPU suite as the driver runs it, on this lease
mkdir -p gpurun_out/r06
tag=${1:-a}
( timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -n 25 ) > gpurun_out/r06/gpu_suite_$tag.log 2>&1
( echo "host $(hostname) $(date -u +%FT%TZ)"; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2 ) >> gpurun_out/r06/gpu_suite_$tag.log
tail -n 6 gpurun_out/r06/gpu_suite_$tag.log
