"""Categorise a rocprofv3 kernel_stats.csv: ms/step and launches/step per kernel family (dev tool).
usage: python tools/kcat.py stats.csv n_steps [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cat, cnt = collections.Counter(), collections.Counter()
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs']) / 1e6 / steps; c = int(r['Calls']) / steps
    if n.startswith('Cijk'): k = 'hipblaslt GEMM'
    elif 'vmasr' in n:
        m = re.search(r'(\w+_kernel)', n); k = 'vmasr:' + (m.group(1) if m else n[:40])
    elif 'elementwise' in n.lower():
        m = re.search(r'(\w+(?:Functor|_kernel_cuda|KernelImpl|Impl))', n.split('elementwise_kernel', 1)[-1]); k = 'aten ew:' + (m.group(1) if m else '?')
    elif 'reduce_kernel' in n: k = 'aten reduce'
    elif 'unfold' in n: k = 'aten unfold_backward'
    elif 'Cat' in n: k = 'aten cat'
    elif 'fillBuffer' in n or 'copyBuffer' in n: k = 'rocclr fill/copy'
    elif 'multi_tensor' in n: k = 'multi_tensor (adam/foreach)'
    else: k = 'other:' + n[:50]
    cat[k] += t; cnt[k] += c
print(f'total {sum(cat.values()):.2f} ms/step, {sum(cnt.values()):.0f} launches/step')
for k, v in cat.most_common(top):
    print(f'{v:7.2f} ms {cnt[k]:7.0f}  {k}')
