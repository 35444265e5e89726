#!/usr/bin/env python
"""Condenses a rocprofv3 (rocpd sqlite) kernel trace into the summaries kept under profiles/:
per-kernel stats (calls / total / avg / min / max / %) and the dispatch breakdown of one
steady-state bench iteration.

    python tools/prof_summary.py gpurun_out/prof1/r01_results.db profiles/r01_kernel_stats.txt
"""
import collections
import re
import sqlite3
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


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, start, end from kernels order by start'))
    stats = collections.defaultdict(list)
    for n, s, e in rows:
        stats[short(n)].append((e - s) / 1e3)
    total = sum(sum(v) for v in stats.values())
    lines = [f'# rocprofv3 --kernel-trace summary of {db}', f'# {len(rows)} dispatches, {total / 1e3:.2f} ms GPU busy', '',
             f'{"kernel":100s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}']
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1]))[:40]:
        lines.append(f'{k:100s} {len(v):7d} {sum(v):12.1f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / total:6.2f}')
    marks = [i for i, r in enumerate(rows) if 'resample_fwd_kernel<0' in r[0]]
    if len(marks) >= 3:
        a, b = marks[-3], marks[-2]
        it = rows[a:b]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for n, s, e in it:
            d = agg[short(n)]
            d[0] += 1
            d[1] += (e - s) / 1e3
        busy = sum(d[1] for d in agg.values())
        lines += ['', f'# one steady-state pose iteration (between two O2C launches): {len(it)} dispatches, '
                      f'wall {(rows[b][1] - rows[a][1]) / 1e6:.3f} ms, GPU busy {busy / 1e3:.3f} ms',
                  f'{"kernel":100s} {"calls":>7s} {"total_us":>12s} {"pct_busy":>8s}']
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
            lines.append(f'{k:100s} {c:7d} {t:12.1f} {100 * t / busy:8.2f}')
    if len(marks) >= 3:
        # every launch of the pose loop (first to last O2C launch): the per-kernel averages bench.py's HIP events
        # must agree with (the whole-trace table above also counts the reconstruct phase's smaller launches)
        loop = rows[marks[0]:marks[-1]]
        agg = collections.defaultdict(list)
        for n, s, e in loop:
            agg[short(n)].append((e - s) / 1e3)
        lines += ['', f'# pose loop only ({len(marks) - 1} iterations, first to last O2C launch): {len(loop)} dispatches',
                  f'{"kernel":100s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s}']
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
            lines.append(f'{k:100s} {len(v):7d} {sum(v):12.1f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f}')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:70]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
