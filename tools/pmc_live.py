#!/usr/bin/env python
"""HBM bytes per launch of the headline's dominant kernel measured LIVE for bench.py: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE:
they cannot share a pass; `--kernel-trace --pmc` only, as MI355X_MICROARCH.md prescribes) over tools/pmc_live_probe.py in a child
process while the bench itself is idle, corrected by the 1 GiB calibration copy of the same pass.  Returns None when rocprofv3 is
missing or a pass fails (bench.py then replays the committed profiles/r06_hbm_bytes.json).   python tools/pmc_live.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def wino_traffic_live(timeout=60, rep=2):
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTX')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None                                                 # (already inside a profiler: no nested passes)
    import pmc_summary
    wd = tempfile.mkdtemp(prefix='lf_pmc_live_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    env.pop('RANK', None), env.pop('LOCAL_RANK', None), env.pop('WORLD_SIZE', None)
    try:
        for name, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
            r = subprocess.run([exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', wd, '-o', name, '--', sys.executable,
                                os.path.join(ROOT, 'tools', 'pmc_live_probe.py'), str(rep)], cwd='/tmp', env=env, timeout=timeout,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None
        data = pmc_summary.load(wd)
        cal = pmc_summary.find(data, 'copyBuffer') or pmc_summary.find(data, 'direct_copy_kernel')
        k = pmc_summary.find(data, 'conv3d_c16_wino_kernel')
        if cal is None or k is None:
            return None
        GiB = 1024.0 ** 3
        calf = [v for v in data[cal].get('FETCH_SIZE', []) if v > 1e5]
        calw = [v for v in data[cal].get('WRITE_SIZE', []) if v > 1e5]
        if not calf or not calw:
            return None
        calf = [v for v in calf if v > 0.9 * max(calf)]
        calw = [v for v in calw if v > 0.9 * max(calw)]
        corr, wcorr = (GiB / 1024.0) / pmc_summary.mean(calf), (GiB / 1024.0) / pmc_summary.mean(calw)
        f, w = data[k].get('FETCH_SIZE', []), data[k].get('WRITE_SIZE', [])
        if len(f) < 2 * rep or len(w) < 2 * rep:
            return None
        hf, hw = len(f) // 2, len(w) // 2
        fwd = pmc_summary.mean(f[:hf]) * 1024 * corr + pmc_summary.mean(w[:hw]) * 1024 * wcorr
        bwd = pmc_summary.mean(f[hf:]) * 1024 * corr + pmc_summary.mean(w[hw:]) * 1024 * wcorr
        return {'forward_form_bytes_per_launch': fwd, 'data_gradient_form_bytes_per_launch': bwd, 'fetch_correction': corr,
                'write_correction': wcorr, 'launches_per_form': rep,
                'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/pmc_live_probe.py inside this bench run, '
                       'scaled by the 1 GiB calibration copy of the same pass'}
    except Exception:                                               # noqa: BLE001
        return None
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def ring_block_traffic_live(timeout=60, rep=2):
    """The same for the cfg 5 block's dominant kernel: the Block step of the bf16 ring convolution on 8 volumes (tools/pmc_live_probe_train.py).
    Returns bytes per launch OF 8 VOLUMES."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTX')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None
    import pmc_summary
    wd = tempfile.mkdtemp(prefix='lf_pmc_live_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    env.pop('RANK', None), env.pop('LOCAL_RANK', None), env.pop('WORLD_SIZE', None)
    try:
        for name, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
            r = subprocess.run([exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', wd, '-o', name, '--', sys.executable,
                                os.path.join(ROOT, 'tools', 'pmc_live_probe_train.py'), str(rep)], cwd='/tmp', env=env, timeout=timeout,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None
        data = pmc_summary.load(wd)
        cal = pmc_summary.find(data, 'copyBuffer') or pmc_summary.find(data, 'direct_copy_kernel')
        k = pmc_summary.find(data, ('ring_multi_kernel<1, true, 4, 6>', 'ring_multi_kernelILi1ELb1ELi4ELi6E'))
        if cal is None or k is None:
            return None
        GiB = 1024.0 ** 3
        calf = [v for v in data[cal].get('FETCH_SIZE', []) if v > 1e5]
        calw = [v for v in data[cal].get('WRITE_SIZE', []) if v > 1e5]
        if not calf or not calw:
            return None
        calf = [v for v in calf if v > 0.9 * max(calf)]
        calw = [v for v in calw if v > 0.9 * max(calw)]
        corr, wcorr = (GiB / 1024.0) / pmc_summary.mean(calf), (GiB / 1024.0) / pmc_summary.mean(calw)
        f, w = data[k].get('FETCH_SIZE', []), data[k].get('WRITE_SIZE', [])
        if not f or not w:
            return None
        return {'bytes_per_launch_of_8_volumes': pmc_summary.mean(f) * 1024 * corr + pmc_summary.mean(w) * 1024 * wcorr,
                'fetch_correction': corr, 'write_correction': wcorr, 'launches': len(f),
                'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/pmc_live_probe_train.py inside this bench '
                       'run, scaled by the 1 GiB calibration copy of the same pass'}
    except Exception:                                               # noqa: BLE001
        return None
    finally:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == '__main__':
    print(json.dumps(wino_traffic_live()))
    print(json.dumps(ring_block_traffic_live()))
