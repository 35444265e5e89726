#!/usr/bin/env python
"""Launch times of the ConvGRU recurrence's kernels on ONE 128^3 x 16 volume, isolated (no second stream): the multi-output /
fused-epilogue ring kernels (csrc/conv_gru.hip) next to the one-output ring kernel and the stage kernels they replace.

    python tools/gru_ring_probe.py [out.json]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from latentfusion_amd import _lib, ops, ops_train
    dev = 'cuda'
    S = 128
    g = torch.Generator().manual_seed(0)
    cl3 = torch.channels_last_3d
    NB = 4                                                          # rotate over distinct buffers (each 67 / 134 MB)

    def vol(dtype, scale=1.0):
        return [(torch.randn(1, 16, S, S, S, generator=g) * scale).to(dev).contiguous(memory_format=cl3).to(dtype) for _ in range(NB)]
    x16, x32 = vol(torch.bfloat16), vol(torch.float32)
    a16, a32 = vol(torch.bfloat16), vol(torch.float32)
    b16, b32 = vol(torch.bfloat16), vol(torch.float32)
    o16, o32 = vol(torch.bfloat16), vol(torch.float32)
    p16 = vol(torch.bfloat16)
    w = [torch.randn(16, 16, 3, 3, 3, generator=g).to(dev) for _ in range(2)]
    packs = torch.stack([ops.pack_conv3d_c16_ring_bf16(t) for t in w]).contiguous()
    p0, p1 = packs[:1].contiguous(), packs[1:].contiguous()
    he = 0.068
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    n = x16[0].numel()

    def timeit(fn, reps=24):
        for i in range(4):
            fn(i % NB)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i % NB)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    R = ops_train.ring_multi
    X = _lib
    cases = {
        # name: (fn, algorithmic bytes per voxel)
        'old: conv(h fp32) + bf16 addend -> bf16': (lambda i: ops.conv3d_c16_ring_bf16_io(x32[i], packs[0], None, he, 0, 0, addend=a16[i], out=o16[i]), 64 + 32 + 32),
        'new: (h -> upre) one group': (lambda i: R(x32[i], p0, he, [(o16[i], a16[i], False)]), 64 + 32 + 32),
        'old: conv(x bf16) + bf16 addend -> bf16': (lambda i: ops.conv3d_c16_ring_bf16_io(x16[i], packs[0], None, he, 0, 0, addend=a16[i], out=o16[i]), 96),
        'new: conv(x bf16) + bf16 addend -> bf16, one group': (lambda i: R(x16[i], p0, he, [(o16[i], a16[i], False)]), 96),
        'old: conv(x bf16) rounded -> bf16': (lambda i: ops.conv3d_c16_ring_bf16_io(x16[i], packs[0], None, he, 0, 1, out=o16[i]), 64),
        'new: conv(x bf16) rounded -> bf16, one group': (lambda i: R(x16[i], p0, he, [(o16[i], None, True)]), 64),
        'old: conv(x bf16) + fp32 addend -> fp32': (lambda i: ops.conv3d_c16_ring_bf16_io(x16[i], packs[0], None, he, 0, 0, addend=a32[i], out=o32[i]), 32 + 64 + 64),
        'new: conv(x bf16) + fp32 addend -> fp32, one group': (lambda i: R(x16[i], p0, he, [(o32[i], a32[i], False)]), 160),
        'new: (h -> rpre, r h) one group': (lambda i: R(x32[i], p1, he, [(o16[i], a16[i], False)], extra=X.LF_RING_EX_RH, o2=p16[i]), 64 + 32 + 32 + 32),
        'old: stage_a (r h)': (lambda i: X.check(L.lf_gru_train_stage_a(a16[i].data_ptr(), x32[i].data_ptr(), o16[i].data_ptr(), n, 1, st), 'a'), 32 + 64 + 32),
        'new: (h -> upre, rpre, r h) two groups': (lambda i: R(x32[i], packs, he, [(o16[i], a16[i], False), (b16[i], b16[i], False)], extra=X.LF_RING_EX_RH, o2=p16[i]), 64 + 64 + 64 + 32),
        'new: (r h -> cand, h\') blend': (lambda i: R(x16[i], p0, he, [(o16[i], a16[i], False)], extra=X.LF_RING_EX_BLEND, e0=x32[i], e1=b16[i], o2=o32[i]), 32 + 32 + 32 + 64 + 32 + 64),
        'old: stage_b (blend)': (lambda i: X.check(L.lf_gru_train_stage_b(x32[i].data_ptr(), a16[i].data_ptr(), b16[i].data_ptr(), o32[i].data_ptr(), n, 1, st), 'b'), 64 + 32 + 32 + 64),
        'new: (gc -> grpre, gh12) reset backward, one group': (lambda i: R(x16[i], p0, he, [(b16[i], b16[i], True)], extra=X.LF_RING_EX_ABWD, e0=x32[i], e1=a32[i], o2=o32[i]), 32 + 32 + 64 + 64 + 32 + 64),
        'new: (gc -> grpre, gh12, gz) two groups': (lambda i: R(x16[i], packs, he, [(b16[i], b16[i], True), (o16[i], None, True)], extra=X.LF_RING_EX_ABWD, e0=x32[i], e1=a32[i], o2=o32[i]), 320),
        'old: stage_a_bwd': (lambda i: X.check(L.lf_gru_train_stage_a_bwd(x16[i].data_ptr(), a16[i].data_ptr(), x32[i].data_ptr(), a32[i].data_ptr(), o16[i].data_ptr(), o32[i].data_ptr(), None, n, 1, st), 'ab'), 32 + 32 + 64 + 64 + 32 + 64),
        'stage_b_bwd (in place, no sums)': (lambda i: X.check(L.lf_gru_train_stage_b_bwd(a32[i].data_ptr(), x32[i].data_ptr(), a16[i].data_ptr(), b16[i].data_ptr(), o32[i].data_ptr(), a16[i].data_ptr(), b16[i].data_ptr(), None, None, n, 1, st), 'bb'), 64 + 64 + 32 + 32 + 64 + 32 + 32),
        'new: (x -> y0 += , y1 = a + ) two groups': (lambda i: R(x16[i], packs, he, [(o16[i], o16[i], False), (o32[i], a32[i], False)]), 32 + 64 + 128),
    }
    out = {}
    nvox = S ** 3
    for k, (fn, bpv) in cases.items():
        us = timeit(fn)
        out[k] = {'us': round(us, 1), 'bytes_per_voxel': bpv, 'TBps': round(bpv * nvox / us / 1e6, 2)}
        print(f'{us:8.1f} us  {bpv * nvox / us / 1e6:5.2f} TB/s  {k}')
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
