# round-6 evidence: the full GPU suite as the driver runs it, on this lease (tag a / b / c ...); with the achieved-error table
mkdir -p gpurun_out/r06
tag=${1:-a}
export VMASR_PARITY_TABLE=$PWD/gpurun_out/r06/parity_table_$tag.md
( timeout 2400 python -m pytest tests/ -x -q -m gpu -rs 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -n 25 ) > gpurun_out/r06/gpu_suite_$tag.log 2>&1
unset VMASR_PARITY_TABLE
( echo "host $(hostname) $(date -u +%FT%TZ)"; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2 ) >> gpurun_out/r06/gpu_suite_$tag.log
tail -n 8 gpurun_out/r06/gpu_suite_$tag.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 2 | tee gpurun_out/r06/smoke_$tag.log
