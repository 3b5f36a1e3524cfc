R=$PWD; cd /tmp && export TMPDIR=/tmp
for d in 16 128; do for m in fused unfused; do
ONLY=$d MODE=$m REP=20 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_${d}_$m -o p -- python $R/tools/bench_mlp.py > /dev/null 2>&1
echo "== d=$d $m"; f=$(find /tmp/pm_${d}_$m -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]: print(f"{float(r['TotalDurationNs'])/1e3:10.1f} us total {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:8.1f} us avg  {r['Name'][:90]}")
PY
done; done
