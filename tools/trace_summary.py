#!/usr/bin/env python
"""Condenses a rocprofv3 `--kernel-trace --output-format csv` run into the summaries kept under profiles/: per-kernel
stats (calls / total / avg / min / max / %) and, when a marker kernel is given, the dispatch breakdown of one steady
state iteration (between two consecutive launches of the marker).

    python tools/trace_summary.py gpurun_out/<run>/trace/bench_kernel_trace.csv profiles/r02_kernel_stats.txt [marker]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    if s.startswith('at::native::'):
        s = 'ATen:' + re.sub(r'<.*', '', s[len('at::native::'):])
        inner = re.search(r'(\w+Functor|\w+_kernel_cuda|sum_functor|MeanOps|NormTwoOps|grid_sampler_\w+|CatArray\w+|FillFunctor)', name)
        if inner:
            s += '[' + inner.group(1) + ']'
        if 'double' in name:
            s += '(f64)'
    return s[:100]


def main(path, out, marker=None):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    rows = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    stats = collections.defaultdict(list)
    for n, s, e in rows:
        stats[short(n)].append((e - s) / 1e3)
    total = sum(sum(v) for v in stats.values())
    lines = [f'# rocprofv3 --kernel-trace summary of {path}', f'# {len(rows)} dispatches, {total / 1e3:.2f} ms GPU busy', '',
             f'{"kernel":100s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}']
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1]))[:45]:
        lines.append(f'{k:100s} {len(v):7d} {sum(v):12.1f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / total:6.2f}')
    if marker:
        marks = [i for i, r in enumerate(rows) if marker in r[0]]
        if len(marks) >= 4:
            a, b = marks[-3], marks[-2]                              # one steady-state iteration
            it = rows[a:b]
            wall = (rows[b][1] - rows[a][1]) / 1e3
            busy = sum(e - s for _, s, e in it) / 1e3
            per = collections.defaultdict(list)
            for n, s, e in it:
                per[short(n)].append((e - s) / 1e3)
            lines += ['', f'# one steady-state iteration (between two launches of {marker}): {len(it)} dispatches, '
                          f'wall {wall / 1e3:.3f} ms, GPU busy {busy / 1e3:.3f} ms',
                      f'{"kernel":100s} {"calls":>7s} {"total_us":>12s} {"pct_busy":>8s}']
            for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:30]:
                lines.append(f'{k:100s} {len(v):7d} {sum(v):12.1f} {100 * sum(v) / busy:8.2f}')
            # the same iteration in launch order: where the idle time between dispatches sits
            lines += ['', '# the same iteration in launch order: start (us from the marker), duration, idle gap before the dispatch',
                      f'{"start_us":>10s} {"dur_us":>9s} {"gap_us":>8s}  kernel']
            t0, prev_end = it[0][1], it[0][1]
            for n, s_, e_ in it:
                lines.append(f'{(s_ - t0) / 1e3:10.1f} {(e_ - s_) / 1e3:9.1f} {max(0, s_ - prev_end) / 1e3:8.1f}  {short(n)}')
                prev_end = max(prev_end, e_)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:40]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
