#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 `--kernel-trace --output-format csv` run of tools/train_probe.py: the last
step (between the last two launches of the flat Adam kernel), per queue: runs of the same kernel collapsed to
(start offset ms, launches, busy ms), plus per-queue busy time and the wall time of the step.

    python tools/step_timeline.py <..._kernel_trace.csv> [out.txt] [marker substring, default adam]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'at::native::', 'ATen:', name)
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:70]


def main(path, out=None, marker='adam'):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    qk = 'Queue_Id' if 'Queue_Id' in rows[0] else None
    marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name'].lower()]
    a, b = marks[-2] + 1, marks[-1] + 1
    it = rows[a:b]
    t0 = int(it[0]['Start_Timestamp'])
    wall = (int(it[-1]['End_Timestamp']) - t0) / 1e6
    lines = [f'# one step: {len(it)} dispatches, wall {wall:.2f} ms']
    byq = collections.defaultdict(list)
    for r in it:
        byq[r[qk] if qk else '0'].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
        lines.append(f'## queue {q}: {len(rs)} dispatches, busy {busy:.2f} ms')
        run = None
        for r in rs:
            n = short(r['Kernel_Name'])
            s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
            if run is not None and run[0] == n:
                run[2] += 1
                run[3] += e - s
                run[4] = e
            else:
                if run is not None:
                    lines.append(f'{run[1]:8.2f} .. {run[4]:8.2f}  x{run[2]:<4d} busy {run[3]:7.3f}  {run[0]}')
                run = [n, s, 1, e - s, e]
        if run is not None:
            lines.append(f'{run[1]:8.2f} .. {run[4]:8.2f}  x{run[2]:<4d} busy {run[3]:7.3f}  {run[0]}')
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    print(txt[:6000])


if __name__ == '__main__':
    main(*sys.argv[1:4])
