#!/usr/bin/env python
"""Turns the two rocprofv3 PMC passes over tools/hbm_probe.py (FETCH_SIZE, WRITE_SIZE; separate runs,
csv output) into profiles/<round>_conv3d_hbm_bytes.json: per-launch HBM bytes of each conv3d kernel,
with the gfx950 FETCH_SIZE correction calibrated on a copy of known size in the same run
(MI355X_MICROARCH.md, HBM / rocprofv3 section).

    python tools/hbm_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def pick(acc, key, floor=0.0):
    """Mean counter value over the launches of the kernel whose name contains `key` (launches below
    `floor` -- e.g. tiny copies sharing the kernel name -- are ignored)."""
    for k, v in acc.items():
        if key in k:
            v = [x for x in v if x >= floor]
            if v:
                return sum(v) / len(v)
    return None


def main(fcsv, wcsv, out):
    f, w = per_kernel(fcsv, 'FETCH_SIZE'), per_kernel(wcsv, 'WRITE_SIZE')
    GiB = 1024.0 ** 3
    cal_f, cal_w = pick(f, 'copyBuffer', 1e5), pick(w, 'copyBuffer', 1e5)     # x.clone() of 1 GiB
    corr = (GiB / 1024.0) / cal_f
    res = {'shape': 'N=8, C=16, S=128 (SYN(128,16) bench shape), fp32',
           'collected': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and a separate --pmc WRITE_SIZE pass over tools/hbm_probe.py',
           'calibration': {'workload': 'x.clone() of 1 GiB (__amd_rocclr_copyBuffer: 1 GiB read, 1 GiB written)', 'FETCH_SIZE_KB_raw': cal_f,
                           'WRITE_SIZE_KB_raw': cal_w, 'fetch_correction': corr,
                           'note': 'gfx950 FETCH_SIZE counts 128-B requests at 64 B: x2 (calibrated here); WRITE_SIZE reads true'},
           'algorithmic_bytes_per_launch': 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4, 'kernels': {}}
    for key in ('conv3d_c16_persistent_kernel', 'conv3d_c16_wino_kernel'):
        fr, wr = pick(f, key), pick(w, key)
        if fr is None:
            continue
        res['kernels'][key] = {'FETCH_SIZE_KB_raw': fr, 'WRITE_SIZE_KB_raw': wr, 'read_bytes': fr * 1024 * corr,
                               'write_bytes': wr * 1024, 'bytes_per_launch': fr * 1024 * corr + wr * 1024}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['kernels'], indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
