"""bf16 autocast policy of the training step (BASELINE cfg 5): the bf16-MFMA conv3d kernel against an emulation of what
torch autocast makes of Equalized.forward + LeakyReLU + PixelNorm (modules/equalized.py:57-64 under
recon/models.py:199,405), and one generator step under `use_amp` against the CPU oracle run under
torch.autocast(bfloat16).  Tolerances are bf16's: a value sitting on a rounding boundary may land one bf16 ulp (2^-8
relative) away when the fp32 accumulation order differs."""
import pytest
import torch
import torch.nn.functional as F

import lf_oracle as O
from lf_oracle import nets

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize('shape', [(2, 16, 8, 8, 16), (1, 16, 5, 9, 21), (3, 16, 4, 16, 32)])
def test_conv3d_c16_bf16_kernel(shape):
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g)
    b = torch.randn(16, generator=g) * 0.1
    he = ops.he_constant(w)
    xd = ops.cl(x.to(DEV))
    acc = F.conv3d(bf(x).double(), bf(w).double(), None, 1, 1)              # exact products of the bf16 operands
    # round_out = 0: fp32 epilogue on the accumulator
    y0, n0 = ops.conv3d_c16_bf16(xd, ops.pack_conv3d_c16_bf16(w.to(DEV)), b.to(DEV), he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 0)
    pre = F.leaky_relu(acc * he + b.double().view(1, -1, 1, 1, 1), 0.2)
    nrm = torch.sqrt((pre ** 2).mean(dim=1, keepdim=True) + 1e-8)
    torch.testing.assert_close(y0.cpu().double(), pre / nrm, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(n0.view(shape[0], *shape[2:]).cpu().double(), nrm.squeeze(1), atol=1e-5, rtol=1e-5)
    # round_out = 1: the autocast forward -- bf16(bf16(acc) * he) + bias
    y1, _ = ops.conv3d_c16_bf16(xd, ops.pack_conv3d_c16_bf16(w.to(DEV)), b.to(DEV), he, LF_EPI_LRELU, 1)
    want = F.leaky_relu(bf(bf(acc.float()) * he) + b.view(1, -1, 1, 1, 1), 0.2)
    err = (y1.cpu() - want).abs()
    assert float(err.max()) <= 2 ** -7 * float(want.abs().max()), float(err.max())       # at most bf16 ulps on ties
    assert float((err > 1e-6).float().mean()) < 5e-3                                     # ... and rarely
    # round_out = 2: data gradient (transposed / flipped pack), result rounded to bf16
    gy = torch.randn(shape, generator=g) * 1e-3
    gx, _ = ops.conv3d_c16_bf16(ops.cl(gy.to(DEV)), ops.pack_conv3d_c16_bf16(w.to(DEV), transpose=True), None, he, 0, 2)
    gwant = bf(bf(bf(F.conv_transpose3d(bf(gy).double(), bf(w).double(), None, 1, 1).float()) * he))
    err = (gx.cpu() - gwant).abs()
    assert float(err.max()) <= 2 ** -7 * float(gwant.abs().max())
    assert float((err > 1e-9).float().mean()) < 5e-3


@pytest.mark.parametrize('shape', [(2, 16, 8, 8, 16), (1, 16, 5, 9, 21), (3, 16, 4, 16, 32), (1, 16, 7, 35, 50), (2, 16, 33, 20, 17)])
def test_conv3d_c16_ring_bf16_kernel(shape):
    """lf_conv3d_c16_ring_bf16 (the f16x3 kernel's ring organisation with one bf16 piece: the kernel the autocast step
    uses) against the same emulation of autocast as above -- forward with both epilogues, the data gradient, and the
    addend form of the ConvGRU gates; ragged extents, odd depths, several columns per workgroup; run-to-run identical."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(shape, generator=g)
    w = torch.randn(16, 16, 3, 3, 3, generator=g)
    b = torch.randn(16, generator=g) * 0.1
    he = ops.he_constant(w)
    xd = ops.cl(x.to(DEV))
    wp = ops.pack_conv3d_c16_ring_bf16(w.to(DEV))
    acc = F.conv3d(bf(x).double(), bf(w).double(), None, 1, 1)
    y0, n0 = ops.conv3d_c16_ring_bf16(xd, wp, b.to(DEV), he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 0)
    y0b, _ = ops.conv3d_c16_ring_bf16(xd, wp, b.to(DEV), he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 0)
    assert torch.equal(y0, y0b)
    pre = F.leaky_relu(acc * he + b.double().view(1, -1, 1, 1, 1), 0.2)
    nrm = torch.sqrt((pre ** 2).mean(dim=1, keepdim=True) + 1e-8)
    torch.testing.assert_close(y0.cpu().double(), pre / nrm, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(n0.view(shape[0], *shape[2:]).cpu().double(), nrm.squeeze(1), atol=1e-5, rtol=1e-5)
    y1, _ = ops.conv3d_c16_ring_bf16(xd, wp, b.to(DEV), he, LF_EPI_LRELU, 1)
    want = F.leaky_relu(bf(bf(acc.float()) * he) + b.view(1, -1, 1, 1, 1), 0.2)
    err = (y1.cpu() - want).abs()
    assert float(err.max()) <= 2 ** -7 * float(want.abs().max()), float(err.max())
    assert float((err > 1e-6).float().mean()) < 5e-3
    gy = torch.randn(shape, generator=g) * 1e-3
    gx, _ = ops.conv3d_c16_ring_bf16(ops.cl(gy.to(DEV)), ops.pack_conv3d_c16_ring_bf16(w.to(DEV), transpose=True), None, he, 0, 1)
    gwant = bf(bf(F.conv_transpose3d(bf(gy).double(), bf(w).double(), None, 1, 1).float()) * he)
    err = (gx.cpu() - gwant).abs()
    assert float(err.max()) <= 2 ** -7 * float(gwant.abs().max())
    assert float((err > 1e-9).float().mean()) < 5e-3
    # addend form: y = conv(x) * he + addend, fp32 on the accumulator
    add = torch.randn(shape, generator=g)
    ya, _ = ops.conv3d_c16_ring_bf16(xd, wp, None, he, 0, 0, addend=ops.cl(add.to(DEV)))
    torch.testing.assert_close(ya.cpu().double(), acc * he + add.double(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('shape', [(1, 16, 16, 16, 32), (2, 16, 9, 21, 45), (3, 16, 6, 40, 35), (1, 16, 64, 64, 64)])
def test_weight_gradient_on_the_bf16_mfma(shape):
    """lf_conv_bwd_weight_bf16 (transposing bf16 staging, v_mfma_f32_16x16x32_bf16) against the fp64 contraction of the
    bf16-rounded operands, on un-rounded inputs (the kernel rounds) and ragged extents (partial tiles on every axis); it
    also equals the fp32-MFMA kernel on pre-rounded operands up to summation order, and is run-to-run identical."""
    from latentfusion_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    gp = torch.randn(shape, generator=g) * 1e-3
    xd, gd = ops.cl(x.to(DEV)), ops.cl(gp.to(DEV))
    he = 0.37
    with ops.autocast():
        gw, gb = ops.conv_bwd_weight(xd, gd, 3, 16, he, want_bias=False)           # [27][co][ci]
        gw2, _ = ops.conv_bwd_weight(xd, gd, 3, 16, he, want_bias=False)
    assert gb is None and torch.equal(gw, gw2)
    xp = F.pad(bf(x).double(), (1, 1, 1, 1, 1, 1))
    gq = bf(gp).double()
    D, H, W = shape[2:]
    want = torch.empty(27, 16, 16, dtype=torch.float64)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                want[(kz * 3 + ky) * 3 + kx] = he * torch.einsum('nodhw,nidhw->oi', gq, xp[:, :, kz:kz + D, ky:ky + H, kx:kx + W])
    scale = want.abs().max().item()
    assert (gw.cpu().double() - want).abs().max().item() < 2e-6 * scale
    ref, _ = ops.conv_bwd_weight(ops.round_bf16(xd), ops.round_bf16(gd), 3, 16, he, want_bias=False)   # fp32 MFMA, same products
    assert (gw - ref).abs().max().item() < 2e-6 * scale


def test_round_bf16():
    from latentfusion_amd import ops
    x = torch.randn(3, 5, 7, generator=torch.Generator().manual_seed(0)) * 100
    assert torch.equal(ops.round_bf16(x.to(DEV)).cpu(), bf(x))
    xc = torch.randn(2, 16, 4, 4, 4).to(DEV).contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(ops.round_bf16(xc).cpu(), bf(xc.cpu()))


def test_generator_step_under_bf16_autocast():
    """GeneratorStep(use_amp=True) on SYN(16,16) (GRU fuser: the 16-channel gate convolutions and the 16 -> 16 blocks take
    the bf16 paths): loss terms within bf16 tolerance of the oracle evaluated under torch.autocast(cpu, bfloat16), every
    parameter gradient as close to the fp32 gradient as the oracle's bf16 gradient is (cosine similarity), and a decreasing
    loss over repeated steps."""
    from latentfusion_amd import losses as L, ops, synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.recon import training
    S, C = 16, 16
    model, (sck, fck, pck, dist) = synth.build_model(S, C, 'gru', seed=0, device=DEV, bias_std=0.1)
    d = synth.make_observation_data(3, seed=7)
    obs = model.preprocess_observation(Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'])).to(DEV))
    gen = torch.Generator().manual_seed(5)
    tgt_depth = torch.rand(1, 3, 1, S, S, generator=gen) * 2 - 1
    tgt_mask = (torch.rand(1, 3, 1, S, S, generator=gen) > 0.5).float()
    kw = dict(g_depth_recon_loss_type='l1', g_depth_recon_loss_weight=25.0, g_mask_recon_loss_weight=25.0, generator_lr=1e-3)

    # oracle under CPU autocast(bf16)
    cks = {k: {**ck, 'state_dict': {n: v.clone().requires_grad_(True) for n, v in ck.get('state_dict', {}).items()}}
           for k, ck in (('s', sck), ('f', fck), ('p', pck))}
    cam = obs.camera.to('cpu')
    ocam = O.Cam(cam.intrinsic, cam.log_quaternion, cam.translation, viewport=cam.viewport, z_span=cam.z_span, width=cam.width,
                 height=cam.height)

    def oracle_loss(autocast):
        for ck in cks.values():
            for v in ck['state_dict'].values():
                v.grad = None
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            z = nets.encode(cks['s'], cks['f'], ocam, obs.color.cpu(), obs.depth.cpu(), obs.mask.cpu())
            y, _, _ = nets.decode(cks['p'], z, ocam, apply_mask=False)
        tot = 25.0 * L.reduce_loss(L.get_recon_criterion('l1')(y['depth'].float(), tgt_depth)) + \
            25.0 * L.reduce_loss(L.get_recon_criterion('binary_cross_entropy')(y['mask_logits'].float(), tgt_mask))
        tot.backward()
        return float(tot), {(k, n): v.grad.clone() for k, ck in cks.items() for n, v in ck['state_dict'].items() if v.grad is not None}
    want32, g32 = oracle_loss(False)
    want16, g16 = oracle_loss(True)
    assert abs(want16 - want32) / want32 > 1e-6, 'autocast had no effect on the oracle'

    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, use_amp=True, **kw)
    batch = {'in': {'camera': obs.camera, 'image': obs.color.unsqueeze(0), 'mask': obs.mask.unsqueeze(0)},
             'out_gt': {'camera': obs.camera, 'depth': tgt_depth.to(DEV), 'mask': tgt_mask.to(DEV)}}
    ops.KERNEL_TIMER = []
    try:
        got = step.run_iteration(batch, is_step=False)
        tags = {n for n, _, _ in ops.KERNEL_TIMER}
    finally:
        ops.KERNEL_TIMER = None
    assert 'conv3d_c16_ring_bf16' in tags, tags
    tot = float(got['total'])
    # within bf16 noise of the autocast oracle, and the deviation from fp32 is of the same order as the oracle's own
    assert abs(tot - want16) / want16 < 2e-2, (tot, want16, want32)
    # bf16 rounding noise on the gradients of this small random network is large (the oracle's own bf16 and fp32
    # gradients of the first encoder layers agree only to cos ~0.85), and two bf16 evaluations with different
    # accumulation orders are independent samples of it: the yardstick is the fp32 gradient -- the HIP bf16 step must be
    # as close to it as the oracle's bf16 step is, over the whole parameter vector and (loosely: small tensors make the
    # statistic itself noisy) per parameter
    hip_all, orc_all, ref_all = [], [], []
    for (k, n), gw in g16.items():
        mod = {'s': model.sculptor, 'f': model.fuser, 'p': model.photographer}[k]
        p = dict(mod.named_parameters())[n]
        ref = g32[(k, n)].reshape(1, -1).double()
        hip_all.append(p.grad.reshape(-1).cpu().double()); orc_all.append(gw.reshape(-1).double()); ref_all.append(ref.reshape(-1))
        if gw.abs().max() < 1e-6 * max(v.abs().max() for v in g16.values()):
            continue
        cos_hip = F.cosine_similarity(p.grad.reshape(1, -1).cpu().double(), ref).item()
        cos_orc = F.cosine_similarity(gw.reshape(1, -1).double(), ref).item()
        # (round 5: slack 0.15 -> 0.2.  With the 16-channel volumes STORED in bf16 the per-parameter statistic of this seed moved by
        # up to -0.13 on the first 2-D encoder layers (0.83 -> 0.69 on one bias) while seeds 1-3 moved by +0.00..+0.06 and the
        # whole-vector cosine below stayed within the oracle's own bf16 noise: tools/ac_noise_probe.py, profiles/r05_ac_noise.txt)
        assert cos_hip > min(0.99, cos_orc - 0.2) and cos_hip > 0.5, (k, n, cos_hip, cos_orc)
    hip_all, orc_all, ref_all = torch.cat(hip_all), torch.cat(orc_all), torch.cat(ref_all)
    cos_hip = F.cosine_similarity(hip_all, ref_all, dim=0).item()
    cos_orc = F.cosine_similarity(orc_all, ref_all, dim=0).item()
    print(f'bf16 autocast: cos(grad, fp32 grad) HIP {cos_hip:.4f}, oracle {cos_orc:.4f}; loss HIP {tot:.5f} oracle bf16 {want16:.5f} fp32 {want32:.5f}')
    assert cos_hip > cos_orc - 0.03, (cos_hip, cos_orc)
    losses = [float(step.run_iteration(batch)['total']) for _ in range(6)]
    assert losses[-1] < losses[0]


# ---- round 5: bf16 STORAGE of the 16-channel volumes under the autocast policy ------------------------------------------
def _b16(t):
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)


def test_resampler_bf16_storage_variants():
    """lf_resample3d_fwd_io / lf_resample3d_bwd_vol_det_io: a bf16-stored source holding the same (bf16-representable) values
    gives the fp32 kernel's result bit for bit; a bf16-stored destination is that result rounded once (RNE)."""
    from latentfusion_amd import _lib, ops, synth
    from latentfusion_amd.modules.geometry import Camera, c2o_coefficients, o2c_coefficients
    from latentfusion_amd.pose import utils as pu
    L = _lib.lib()
    S, N = 24, 3
    gen = torch.Generator().manual_seed(11)
    td = synth.make_observation_data(1, seed=2)
    torch.manual_seed(3)
    cams = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic'])).zoom(None, S, 2.85).to(DEV)
    s = torch.cuda.current_stream().cuda_stream
    for kind, coef in ((_lib.LF_MAP_O2C, o2c_coefficients(cams, 1.0)), (_lib.LF_MAP_C2O, c2o_coefficients(cams, 1.0))):
        cf = torch.zeros(N, _lib.LF_MAP_COEFS, device=DEV)
        cf[:, :coef.shape[1]] = coef
        vol = ops.cl(ops.round_bf16(torch.randn(N, 16, S, S, S, generator=gen).to(DEV)))
        g = ops.cl(ops.round_bf16((torch.randn(N, 16, S, S, S, generator=gen) * 1e-2).to(DEV)))
        ref = ops.empty_cl((N, 16, S, S, S), DEV)
        _lib.check(L.lf_resample3d_fwd(vol.data_ptr(), N, cf.data_ptr(), kind, ref.data_ptr(), N, S, S, S, 16, s), 'fwd')
        nb = max(L.lf_resample3d_bwd_vol_det_scratch_bytes(N, S, S, S, 16), L.lf_resample3d_bwd_vol_det_io_scratch_bytes(N, N, S, S, S),
                 L.lf_resample3d_bwd_vol_det_io_scratch_bytes(1, N, S, S, S))
        scr = torch.empty(nb // 8 + 1, device=DEV, dtype=torch.int64)
        gref = ops.empty_cl((N, 16, S, S, S), DEV)
        _lib.check(L.lf_resample3d_bwd_vol_det(g.data_ptr(), cf.data_ptr(), kind, gref.data_ptr(), N, scr.data_ptr(), scr.numel() * 8,
                                               N, S, S, S, 16, s), 'bwd')
        for io in range(4):
            vi = _b16(vol) if io & 1 else vol
            out = ops.empty_cl16((N, 16, S, S, S), DEV, bool(io & 2))
            _lib.check(L.lf_resample3d_fwd_io(vi.data_ptr(), N, cf.data_ptr(), kind, out.data_ptr(), N, S, S, S, io, s), 'fwd io')
            assert torch.equal(out, ref.to(torch.bfloat16) if io & 2 else ref), (kind, io)
            gi = _b16(g) if io & 1 else g
            gv = ops.empty_cl16((N, 16, S, S, S), DEV, bool(io & 2))
            _lib.check(L.lf_resample3d_bwd_vol_det_io(gi.data_ptr(), cf.data_ptr(), kind, gv.data_ptr(), N, scr.data_ptr(), scr.numel() * 8,
                                                      N, S, S, S, io, s), 'bwd io')
            assert torch.equal(gv, gref.to(torch.bfloat16) if io & 2 else gref), (kind, io)     # (binned form: a volume per sample)
            prev = L.lf_set_tuning(4, 3)                                                        # ... and the tile form of the same call
            try:
                gv3 = ops.empty_cl16((N, 16, S, S, S), DEV, bool(io & 2))
                _lib.check(L.lf_resample3d_bwd_vol_det_io(gi.data_ptr(), cf.data_ptr(), kind, gv3.data_ptr(), N, scr.data_ptr(),
                                                          scr.numel() * 8, N, S, S, S, io, s), 'bwd io')
            finally:
                L.lf_set_tuning(4, prev)
            assert torch.equal(gv3, gv), (kind, io)
            prev = L.lf_set_tuning(6, 2)                                                        # ... and the binned form in two passes
            try:
                gv2 = ops.empty_cl16((N, 16, S, S, S), DEV, bool(io & 2))
                _lib.check(L.lf_resample3d_bwd_vol_det_io(gi.data_ptr(), cf.data_ptr(), kind, gv2.data_ptr(), N, scr.data_ptr(),
                                                          scr.numel() * 8, N, S, S, S, io, s), 'bwd io')
            finally:
                L.lf_set_tuning(6, prev)
            assert torch.equal(gv2, gv), (kind, io)
        # one volume shared by the samples (the renderer's transform): binned form (all samples in one pass) == tile form == fp32 entry
        gref1 = ops.empty_cl((1, 16, S, S, S), DEV)
        _lib.check(L.lf_resample3d_bwd_vol_det(g.data_ptr(), cf.data_ptr(), kind, gref1.data_ptr(), 1, scr.data_ptr(), scr.numel() * 8,
                                               N, S, S, S, 16, s), 'bwd')
        for variant in (2, 3):
            prev = L.lf_set_tuning(4, variant)
            try:
                gv1 = ops.empty_cl16((1, 16, S, S, S), DEV, False)
                _lib.check(L.lf_resample3d_bwd_vol_det_io(_b16(g).data_ptr(), cf.data_ptr(), kind, gv1.data_ptr(), 1, scr.data_ptr(),
                                                          scr.numel() * 8, N, S, S, S, 1, s), 'bwd io')
            finally:
                L.lf_set_tuning(4, prev)
            assert torch.equal(gv1, gref1), (kind, variant)


@pytest.mark.parametrize('flags', [0, 1, 3])
def test_epilogue_bwd_c16_with_bias_sums(flags):
    """lf_epilogue_bwd_c16 = lf_epilogue_bwd on 16 channels + the bias gradient (column sums) in the same pass, in every
    storage combination."""
    from latentfusion_amd import ops
    gen = torch.Generator().manual_seed(flags)
    shape = (2, 16, 7, 13, 21)
    gy = ops.cl(ops.round_bf16(torch.randn(shape, generator=gen).to(DEV)))
    y = ops.cl(ops.round_bf16(torch.randn(shape, generator=gen).to(DEV)))
    norm = (torch.rand(shape[0] * shape[2] * shape[3] * shape[4], generator=gen) + 0.5).to(DEV)
    want = ops._epilogue_bwd(gy, y, norm, flags)
    wb = want.double().sum(dim=(0, 2, 3, 4))
    for io in range(8):
        gp, gb = ops.epilogue_bwd_c16(_b16(gy) if io & 1 else gy, _b16(y) if io & 2 else y, norm, flags, True, out_bf16=bool(io & 4))
        assert torch.equal(gp, want.to(torch.bfloat16) if io & 4 else want), io
        assert (gb.double() - wb).abs().max().item() < 1e-5 * max(wb.abs().max().item(), 1.0)


def test_conv16_layers_under_the_storage_policy():
    """A camera block + the pointwise output layer under autocast: bf16 storage (ops.BF16_STORAGE, default) against fp32 storage
    of the same policy -- activations equal up to one bf16 rounding of the stored values, gradients of input / weights /
    biases agree (cosine > 0.999), the output is a bf16 channels-last tensor and two runs agree bit for bit."""
    from latentfusion_amd import ops
    from latentfusion_amd.modules import EqualizedConv3d
    from latentfusion_amd.modules.blocks import Block
    torch.manual_seed(2)
    blk = Block(16, 16, conv_module=EqualizedConv3d).to(DEV)
    out1 = EqualizedConv3d(16, 16, 1).to(DEV)
    with torch.no_grad():
        for c in (blk.conv1, blk.conv2, out1):
            c.bias.normal_(0, 0.2)
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 16, 16, 24, 40, generator=gen).to(DEV)
    gout = torch.randn(2, 16, 16, 24, 40, generator=gen).to(DEV)
    res = {}
    params = list(blk.parameters()) + list(out1.parameters())
    for storage in (False, True, True):
        ops.BF16_STORAGE = storage
        try:
            for p in params:
                p.grad = None
            x = x0.clone().requires_grad_(True)
            with ops.autocast():
                y = out1(blk(x))
            (y.float() * gout).sum().backward()
        finally:
            ops.BF16_STORAGE = True
        res.setdefault(storage, []).append((y.detach(), x.grad.clone(), [p.grad.clone() for p in params]))
    a, b = res[True]
    assert a[0].dtype == torch.bfloat16 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(p, q) for p, q in zip(a[2], b[2]))
    ref = res[False][0]
    assert ref[0].dtype == torch.float32
    err = (a[0].float() - ref[0]).abs()
    assert float(err.max()) <= 2 ** -6 * float(ref[0].abs().max())
    cos = lambda u, v: F.cosine_similarity(u.reshape(1, -1).double(), v.reshape(1, -1).double()).item()   # noqa: E731
    assert cos(a[1], ref[1]) > 0.999
    for p, q in zip(a[2], ref[2]):
        assert cos(p, q) > 0.999, (p.shape, cos(p, q))


@pytest.mark.parametrize('amp', [False, True])
def test_generator_step_vs_the_reference_fixture(golden, amp):
    """GeneratorStep on the HIP path against what the REAL reference computed for the same step (fixture g27, oracle/make_golden.py
    g27_training_gradients: SYN(32,16), GRU fuser, 4 input + 2 output views, hard smooth-L1 depth + BCE mask losses x 25):
      fp32      loss terms to 1e-4, the gradient of every parameter to 1 % rel-L2 (cosine > 0.9999);
      autocast  loss within 2 % of the reference under torch.autocast(cpu, bf16), and the whole gradient vector at least as
                close to the reference's fp32 gradient as the reference's own bf16 gradient is (cosine - 0.03), bf16 storage on."""
    from latentfusion_amd import synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.recon import training
    g = golden('g27_training_gradients')
    S, C, seed = g['S'], g['C'], g['seed']
    model, _ = synth.build_model(S, C, 'gru', seed=seed, device=DEV, bias_std=g['bias_std'])

    def obs(n, sd):
        d = synth.make_observation_data(n, sd)
        return model.preprocess_observation(Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'])).to(DEV))
    oi, oo = obs(g['views_in'], seed + 1), obs(g['views_out'], seed + 2)
    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=amp)
    batch = {'in': {'camera': oi.camera, 'image': oi.color.unsqueeze(0), 'mask': oi.mask.unsqueeze(0)},
             'out_gt': {'camera': oo.camera, 'depth': oo.depth.unsqueeze(0), 'mask': oo.mask.unsqueeze(0)}}
    out = step.run_iteration(batch, is_step=False)
    mods = {'s': model.sculptor, 'f': model.fuser, 'p': model.photographer}
    got = {k + '.' + n: p.grad.detach().cpu() for k, m in mods.items() for n, p in m.named_parameters()}
    ref32 = g['grad_fp32']
    cat = lambda d: torch.cat([d[k].reshape(-1) for k in ref32]).double()      # noqa: E731
    if not amp:
        assert abs(float(out['depth_recon']) - float(g['loss_fp32']['depth_recon'])) < 1e-4 * float(g['loss_fp32']['depth_recon'])
        assert abs(float(out['mask_recon']) - float(g['loss_fp32']['mask_recon'])) < 1e-4 * float(g['loss_fp32']['mask_recon'])
        for key, want in ref32.items():
            rel = float((got[key] - want).norm() / want.norm().clamp_min(1e-30))
            assert rel < 1e-2, (key, rel)
        assert F.cosine_similarity(cat(got), cat(ref32), dim=0).item() > 0.9999
    else:
        want16 = float(g['loss_autocast_bf16']['total'])
        assert abs(float(out['total']) - want16) < 2e-2 * want16, (float(out['total']), want16)
        cos_hip = F.cosine_similarity(cat(got), cat(ref32), dim=0).item()
        cos_ref = F.cosine_similarity(cat(g['grad_autocast_bf16']), cat(ref32), dim=0).item()
        print(f'g27 autocast: cos(HIP bf16 grad, reference fp32 grad) {cos_hip:.4f}; reference bf16 {cos_ref:.4f}')
        assert cos_hip > cos_ref - 0.03, (cos_hip, cos_ref)
