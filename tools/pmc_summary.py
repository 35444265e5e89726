#!/usr/bin/env python
"""Condenses the passes of tools/pmc_collect.sh into the two files kept under profiles/:

    python tools/pmc_summary.py gpurun_out/<tag> profiles/r02

  profiles/r02_kernels_pmc.txt   per kernel: mean SQ / TCC counters per launch and the derived figures the roofline
                                 discussion uses (VALU per MFMA, MFMA-pipe busy, LDS conflict share, L2 hit rate)
  profiles/r02_hbm_bytes.json    per kernel: HBM bytes per launch = FETCH_SIZE (x2: gfx950 counts 128-B requests at 64 B;
                                 the factor is calibrated on a 1 GiB copy of the same run) + WRITE_SIZE, next to the
                                 algorithmic bytes, stamped with the sha256 of the kernel sources so that bench.py can
                                 refuse a stale file (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = collections.OrderedDict([
    # key -> (substring of the kernel name, algorithmic bytes per launch at N=8, C=16, S=128)
    ('conv3d_c16_wino_kernel', ('conv3d_c16_wino_kernel', 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4)),
    # round 4: conv + factor projection in one launch (forward: x, y, norm + the (8,16,128,128) latent image; backward: the
    # block's activation + norm, the producer's activation + norm, the output gradient; the projection gradient is 8 MB)
    ('conv3d_c16_wino_projfwd_kernel', ('conv3d_c16_wino_projfwd_kernel', 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4 + 8 * 17 * 128 ** 2 * 4)),
    ('conv3d_c16_wino_projbwd_kernel', ('conv3d_c16_wino_projbwd_kernel', 3 * 8 * 16 * 128 ** 3 * 4 + 2 * 8 * 128 ** 3 * 4 + 8 * 16 * 128 ** 2 * 4)),
    ('conv3d_c16_persistent_kernel', ('conv3d_c16_persistent_kernel', 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4)),
    ('conv3d_c16_f16x3_kernel', (('conv3d_c16_f16x3_kernel<false, 3>', 'conv3d_c16_f16x3_kernel<false>', 'conv3d_c16_f16x3_kernelILb0ELi3E'), 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4)),
    ('conv3d_c16_f16x3_kernel_bwd', (('conv3d_c16_f16x3_kernel<true, 3>', 'conv3d_c16_f16x3_kernel<true>', 'conv3d_c16_f16x3_kernelILb1ELi3E'), 3 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4)),
    ('resample_fwd', ('resample_fwd', 8 * 16 * 128 ** 3 * 4 + 16 * 128 ** 3 * 4)),
    ('resample_bwd_coef', ('resample_bwd_coef_', 8 * 16 * 128 ** 3 * 4 + 16 * 128 ** 3 * 4)),
    ('conv1x1_kernel', ('conv1x1_kernel', 8 * 16 * 128 ** 3 * 4)),
    ('column_sum_fwd_kernel', ('column_sum_fwd_kernel', 8 * 16 * 128 ** 3 * 4)),
])
# --train: the kernels of tools/train_kernels_probe.py (training step, 8 x 128^3 x 16)
V = 8 * 16 * 128 ** 3 * 4
TRAIN_KERNELS = collections.OrderedDict([
    # round 5: bf16 STORAGE (32 B per voxel record): Vh = one 8 x 128^3 x 16 volume in bf16
    ('conv3d_c16_ring_bf16', (('conv3d_c16_f16x3_kernel<false, 1, 3>', 'conv3d_c16_f16x3_kernelILb0ELi1ELi3E'), V + V // 16)),
    ('conv3d_c16_ring_bf16_addend', (('conv3d_c16_f16x3_kernel<true, 1, 7>', 'conv3d_c16_f16x3_kernelILb1ELi1ELi7E'), 3 * V // 2)),
    ('wgrad3d_c16_bf16_kernel', (('wgrad3d_c16_bf16_kernel<3>', 'wgrad3d_c16_bf16_kernelILi3E'), V)),
    # round 6: the transpose-load kernel (x + gpre, bf16 records)
    ('wgrad3d_c16_tr_kernel', (('wgrad3d_c16_tr_kernel<3>', 'wgrad3d_c16_tr_kernelILi3E'), V)),
    ('epilogue_bwd_c16_kernel', (('epilogue_bwd_c16_kernel<7>', 'epilogue_bwd_c16_kernelILi7E'), 3 * V // 2 + V // 16)),
    ('resample_fwd_c16_kernel', (('resample_fwd_c16_kernel<1, 3>', 'resample_fwd_c16_kernelILi1ELi3E'), V)),
    ('splat_tile_kernel', (('splat_tile_kernel<1, 3>', 'splat_tile_kernelILi1ELi3E'), V)),
    # round 5, binned form: + the lists (1.6 entries of 4 B per voxel, written by the fill pass and read here)
    ('splat_binned_tile_kernel', (('splat_binned_tile_kernel<1, 3>', 'splat_binned_tile_kernelILi1ELi3E'), V + 8 * 128 ** 3 * 7)),
    ('splat_bin_kernel_fill', (('splat_bin_kernel<1, true>', 'splat_bin_kernelILi1ELb1E'), 8 * 128 ** 3 * 7)),
    ('lift_norm_unfold_kernel', ('lift_norm_unfold_kernel', V + V // 2)),
    ('lift_bwd_fused_kernel', ('lift_bwd_fused_kernel', 2 * V)),
    # round 6: one-group ring kernels with compile-time epilogues (bf16 in / out; in-place bf16 addend), the fused lift
    # (forward: x rows + the bf16 volume; backward: two bf16 volumes in, gx rows out) and the fused 3-D -> 2-D projection
    ('ring_multi_rounded', (('ring_multi_kernel<1, true, 0, 6>', 'ring_multi_kernelILi1ELb1ELi0ELi6E'), V)),
    ('ring_multi_block', (('ring_multi_kernel<1, true, 4, 6>', 'ring_multi_kernelILi1ELb1ELi4ELi6E'), V + V // 16)),
    ('ring_multi_addend_inplace', (('ring_multi_kernel<1, true, 0, 11>', 'ring_multi_kernelILi1ELb1ELi0ELi11E'), 3 * V // 2)),
    ('lift_fwd_mfma_kernel', ('lift_fwd_mfma_kernel', V // 2 + 8 * 128 ** 2 * 17 * 4)),
    ('lift_bwd_mfma_kernel', ('lift_bwd_mfma_kernel', V + 8 * 128 ** 2 * 33 * 4)),
    ('proj16_fwd_kernel', ('proj16_fwd_kernel', V // 2 + 8 * 128 ** 2 * 17 * 4)),
    ('proj16_bwd_kernel', ('proj16_bwd_kernel', V + 8 * 128 ** 2 * 16 * 4)),
    ('pw16_fwd_kernel', ('pw16_fwd_kernel', V)),
    ('pw16_bwd_kernel', ('pw16_bwd_kernel', V)),
])
# hbm_probe.py launches the Winograd kernel REP times in its forward form, then REP times as a data gradient with the
# producer's epilogue backward fused (reads the saved activation and norm as well): reported separately
SPLIT = {'conv3d_c16_wino_kernel': (('forward form', 2 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4),
                                    ('data-gradient form + fused previous-layer backward', 3 * 8 * 16 * 128 ** 3 * 4 + 8 * 128 ** 3 * 4))}
SOURCES = ['conv_wino.hip', 'conv.hip', 'conv_split.hip', 'resample.hip', 'pointwise.hip', 'reduce.hip']
TRAIN_SOURCES = ['conv_split.hip', 'wgrad.hip', 'resample.hip', 'pointwise.hip', 'gru.hip', 'conv_gru.hip', 'lift_mfma.hip', 'ring_tile.h', 'pw16.hip']
# --cfg3: the released architecture's kernels inside the cross_entropy_linemod loop (tools/pmc_collect_cfg3.sh over
# tools/cfg3_probe.py; no calibration copy in that run: FETCH_SIZE x 2, WRITE_SIZE x 1 as calibrated in the other two)
CFG3_KERNELS = collections.OrderedDict([
    # 3-D 256 -> 256 @ 16^3, N = 128: x in + y out (0.537 GB each) + U (17 MB); V (4.3 GB) is an intermediate, not algorithmic
    ('wino_fused_kernel<3>', (('wino_fused_kernel<3, 4, 2, 2, 2>',), 2 * 128 * 256 * 16 ** 3 * 4 + 64 * 256 * 256 * 4)),
    ('wino_fused_kernel<2>', (('wino_fused_kernel<2, 4, 2, 2, 4>',), 1)),
    ('wino_fused_kernel<2> (1 x 8)', (('wino_fused_kernel<2, 1, 8, 4, 2>',), 1)),
    ('wino3d_input_kernel', ('wino3d_input_kernel', 1)),
    ('wino2d_input_kernel', ('wino2d_input_kernel', 1)),
])
CFG3_SOURCES = ['wino_fused.hip', 'wino_gemm.hip']


def source_hashes(sources=None):
    out = {}
    for f in (sources or SOURCES):
        p = os.path.join(ROOT, 'latentfusion_amd', 'csrc', f)
        if os.path.exists(p):
            out[f] = hashlib.sha256(open(p, 'rb').read()).hexdigest()
    return out


def load(dirname):
    """{kernel name: {counter: [values per dispatch]}} over every *_counter_collection.csv of the directory;
    rocprofv3 writes one row per (dispatch, counter[, dimension]): rows of one dispatch are summed."""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in sorted(glob.glob(os.path.join(dirname, '*counter_collection.csv'))):
        tag = os.path.basename(path).split('_')[0]
        for r in csv.DictReader(open(path)):
            per[r['Kernel_Name']][r['Counter_Name']][(tag, r['Dispatch_Id'])] += float(r['Counter_Value'])
    return {k: {c: [v[d] for d in sorted(v, key=lambda td: (td[0], int(td[1])))] for c, v in cs.items()} for k, cs in per.items()}


def mean(v):
    return sum(v) / len(v) if v else None


def find(data, sub, floor_counter=None, floor=0.0):
    subs = sub if isinstance(sub, tuple) else (sub,)                  # (rocprofv3 leaves names with _Float16 parameters mangled)
    hits = [k for k in data if any(x in k for x in subs) and 'coef_reduce' not in k]
    if not hits:
        return None
    # several template instances may match: take the one with the most work
    return max(hits, key=lambda k: mean(data[k].get('SQ_WAVE_CYCLES', data[k].get('FETCH_SIZE', [0]))) or 0)


def main(dirname, prefix, kernels=None, probe='tools/hbm_probe.py', sources=None):
    kernels = kernels or KERNELS
    data = load(dirname)
    GiB = 1024.0 ** 3
    cal = find(data, 'copyBuffer') or find(data, 'direct_copy_kernel')
    calf = [v for v in data.get(cal, {}).get('FETCH_SIZE', []) if v > 1e5] if cal else []
    calw = [v for v in data.get(cal, {}).get('WRITE_SIZE', []) if v > 1e5] if cal else []
    # the calibration copies are the LARGEST dispatches of the copy kernel (the probes also clone smaller tensors with it)
    calf = [v for v in calf if v > 0.9 * max(calf)] if calf else calf
    calw = [v for v in calw if v > 0.9 * max(calw)] if calw else calw
    corr = (GiB / 1024.0) / mean(calf) if calf else 2.0
    wcorr = (GiB / 1024.0) / mean(calw) if calw else 1.0
    lines = [f'# PMC counters per launch (means over the dispatches of {probe}), from {dirname}',
             '# ' + ('released architecture (cfg 3): 128 renders per iteration' if kernels is CFG3_KERNELS else 'bench shape SYN(128,16), N = 8: 8 x 128^3 voxels x 16 channels per volume'),
             f'# FETCH_SIZE correction (1 GiB copy in the same run): x{corr:.4f};  WRITE_SIZE: x{wcorr:.4f}', '']
    shape = ('released architecture, cross_entropy_linemod: N = 128 renders, 3-D 256 -> 256 @ 16^3 (wino_fused_kernel<3>)' if kernels is CFG3_KERNELS
             else 'N=8, C=16, S=128 (SYN(128,16) bench shape), ' + ('bf16 storage (32 B per voxel record)' if sources else 'fp32'))
    hbm = {'shape': shape, 'collected': 'tools/pmc_collect*.sh (separate --pmc passes)',
           'fetch_correction': corr, 'write_correction': wcorr, 'source_sha256': source_hashes(sources), 'kernels': {}}
    for key, (sub, alg) in kernels.items():
        k = find(data, sub)
        if k is None:
            continue
        variants = [(key, alg, {n: mean(v) for n, v in data[k].items()})]
        if key in SPLIT:
            (na, alga), (nb, algb) = SPLIT[key]
            half = {n: len(v) // 2 for n, v in data[k].items()}
            variants = [(f'{key} [{na}]', alga, {n: mean(v[:half[n]]) for n, v in data[k].items() if half[n]}),
                        (f'{key} [{nb}]', algb, {n: mean(v[half[n]:]) for n, v in data[k].items() if half[n]})]
        for vi, (vkey, alg, c) in enumerate(variants):
            summarise(lines, hbm, key if vi == 0 else key + '_bwd', vkey, k, alg, c, corr, wcorr)
    open(prefix + '_kernels_pmc.txt', 'w').write('\n'.join(lines) + '\n')
    json.dump(hbm, open(prefix + '_hbm_bytes.json', 'w'), indent=1)
    print('\n'.join(lines))


def summarise(lines, hbm, key, vkey, k, alg, c, corr, wcorr):
    if True:
        lines.append(f'## {vkey}   ({k[:110]})')
        for n in sorted(c):
            lines.append(f'{n:32s} {c[n]:14.4e}')
        d = []
        if c.get('SQ_INSTS_MFMA') and c.get('SQ_INSTS_VALU'):
            d.append(f"VALU per MFMA = {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.2f}")
        if c.get('SQ_VALU_MFMA_BUSY_CYCLES') and c.get('GRBM_GUI_ACTIVE'):
            d.append('MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = '
                     f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0):.3f}")
        if c.get('SQ_LDS_IDX_ACTIVE'):
            d.append(f"LDS conflict share = {c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
        if c.get('TCC_HIT_sum') is not None and c.get('TCC_MISS_sum') is not None and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            d.append(f"L2 hit rate = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if c.get('FETCH_SIZE') is not None and c.get('WRITE_SIZE') is not None:
            rb, wb = c['FETCH_SIZE'] * 1024 * corr, c['WRITE_SIZE'] * 1024 * wcorr
            known = alg is not None and alg > 1                      # (mean over launches of many shapes: no single algorithmic figure)
            d.append(f'HBM bytes per launch = {rb / 1e9:.3f} GB read + {wb / 1e9:.3f} GB written = {(rb + wb) / 1e9:.3f} GB'
                     + (f' ({(rb + wb) / alg:.2f}x the algorithmic {alg / 1e9:.3f} GB)' if known else ' (mean over launches of several shapes)'))
            hbm['kernels'][key] = {'kernel_name': k[:160], 'read_bytes': rb, 'write_bytes': wb, 'bytes_per_launch': rb + wb,
                                   'algorithmic_bytes_per_launch': alg if known else None}
        lines += ['   ' + x for x in d] + ['']


if __name__ == '__main__':
    if sys.argv[1] == '--train':
        main(sys.argv[2], sys.argv[3], TRAIN_KERNELS, 'tools/train_kernels_probe.py', TRAIN_SOURCES)
    elif sys.argv[1] == '--cfg3':
        main(sys.argv[2], sys.argv[3], CFG3_KERNELS, 'tools/cfg3_probe.py (means over ALL launches of a kernel in the loop)', CFG3_SOURCES)
    else:
        main(sys.argv[1], sys.argv[2])
