#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) -> the text table kept under profiles/.
    python tools/kernel_stats_txt.py <dir>/<name>_kernel_stats.csv profiles/<out>.txt "<title>" [steps]"""
import csv
import re
import sys


def main(src, dst, title, steps=1):
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    out = [f'# {title}', f'# rocprofv3 --kernel-trace --stats: {calls} dispatches, {tot / 1e6:.2f} ms GPU busy over {steps} step(s) '
           f'= {tot / 1e6 / steps:.2f} ms per step', '',
           f'{"kernel":96s} {"calls/step":>10s} {"ms/step":>9s} {"avg_us":>10s} {"min_us":>9s} {"max_us":>10s} {"pct":>6s}']
    for r in rows[:48]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'at::native::', 'ATen:', n)
        n = re.sub(r'\(.*$', '', n)
        out.append(f'{n[:96]:96s} {int(r["Calls"]) / steps:10.1f} {float(r["TotalDurationNs"]) / 1e6 / steps:9.3f} '
                   f'{float(r["AverageNs"]) / 1e3:10.1f} {float(r["MinNs"]) / 1e3:9.1f} {float(r["MaxNs"]) / 1e3:10.1f} {float(r["Percentage"]):6.2f}')
    open(dst, 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out[:30]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 1)
