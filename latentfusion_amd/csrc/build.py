"""Builds liblf_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain
C-ABI shared object (include/lf_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['resample.hip', 'conv.hip', 'pointwise.hip', 'image.hip', 'optim.hip', 'resize.hip', 'conv_split.hip', 'conv_wino.hip', 'wgrad.hip', 'sample2d.hip', 'gru.hip', 'wino_gemm.hip', 'reduce.hip', 'wino_fused.hip', 'conv_bf16.hip', 'occlusion.hip', 'conv_gru.hip', 'lift_mfma.hip', 'pw16.hip']
LIB = os.path.join(HERE, 'liblf_hip.so')
# per-file extras (conv_split.hip: see split_piece there; conv_gru.hip: see the note on packed fp32 arithmetic at its top)
EXTRA = {'conv_split.hip': ['-fno-slp-vectorize'], 'conv_gru.hip': ['-fno-slp-vectorize']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps += [os.path.join(HERE, 'lf_common.h'), os.path.join(HERE, 'resample_staged.inc'), os.path.join(HERE, 'ring_tile.h'),
             os.path.join(HERE, '..', '..', 'include', 'lf_hip.h'), os.path.join(HERE, '..', '..', 'include', 'lf_hip_experimental.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc, '-x', 'hip', '-c', path, '-o', obj] + FLAGS + EXTRA.get(src, [])
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + out.decode())
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode())
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
