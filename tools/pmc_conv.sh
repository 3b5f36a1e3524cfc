cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o c -- python $R/tools/bench_convgemm.py 4 > /dev/null 2> /tmp/pc.err
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pc/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'conv_mfma' not in k: continue
    key = ('wgrad' if 'wgrad' in k else 'nt') + ' grid ' + r.get('Grid_Size','?')
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[key] += 1
for k, c in sorted(agg.items()):
    wc = c['SQ_WAVE_CYCLES']
    print(k, 'n', n[k], {a: round(b / wc, 3) for a, b in c.items() if a != 'SQ_WAVE_CYCLES'}, 'wave_cycles/launch', int(wc / max(1, n[k])))
PY
