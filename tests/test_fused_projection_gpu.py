"""The factor projection fused into the Winograd kernel of the last camera block (round 4; include/lf_hip.h:
lf_conv3d_c16_wino_projfwd / lf_conv3d_c16_wino_projbwd) against the two-launch forms it replaces
(lf_conv3d_c16_wino + lf_conv1x1_fwd / lf_conv1x1_bwd_data), against fp64, and inside the render-loop engine.
Reference semantics: recon/models.py:417-437 (camera blocks, then FactorProjection3d2d), modules/geometry.py:731-749."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# volumes with partial tiles on every axis (tile = 2 x 8 x 16), one with more columns of tiles than resident workgroups
# (2 x 256 CUs: a workgroup then walks several columns), one tile-aligned, two whose columns are a single tile (D <= 2)
SHAPES = [(2, 16, 16, 16), (1, 18, 12, 20), (3, 5, 9, 33), (5, 6, 128, 128), (1, 32, 32, 32), (2, 2, 16, 16), (3, 1, 9, 17)]


def _problem(shape, seed, bias=True):
    from latentfusion_amd import ops
    N, D, H, W = shape
    g = torch.Generator().manual_seed(seed)
    x = ops.cl(torch.randn(N, 16, D, H, W, generator=g).to(DEV))
    w = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
    b = (torch.randn(16, generator=g) * 0.1).to(DEV) if bias else None
    wp = torch.randn(16, 16 * D, 1, 1, generator=g).to(DEV)            # reference layout: K = c * D + d
    pb = (torch.randn(16, generator=g) * 0.1).to(DEV) if bias else None
    return x, w, b, wp, pb


def _proj_matrices(wp, D):
    """(16, D*16) depth-major matrix (K = d*16 + c), as engine.py builds it for lf_conv1x1_fwd."""
    return wp.reshape(16, 16, D).permute(0, 2, 1).reshape(16, D * 16).contiguous()


@pytest.mark.parametrize('shape', SHAPES)
def test_projfwd_is_the_two_launch_form_bit_for_bit(shape):
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    N, D, H, W = shape
    x, w, b, wp, pb = _problem(shape, 11)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    he, phe = ops.he_constant(w), ops.he_constant(wp)
    up = ops.pack_conv3d_c16_wino(w)
    wdm = _proj_matrices(wp, D)
    # two launches
    y0, n0 = ops.conv3d_c16_wino(x, up, b, he, flags)
    zp0 = ops.empty_cl((N, 16, H, W), DEV)
    pn0 = ops._conv1x1_raw(y0, ops.pack_conv1x1(wdm), pb, N, H * W, 16, D, D * H * W * 16, H * W * 16, 16, zp0, phe, flags)
    # one launch
    y1, n1, zp1, pn1 = ops.conv3d_c16_wino_projfwd(x, up, b, he, flags, ops.pack_wino_proj(wdm), pb, phe, flags)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0) and torch.equal(n1, n0), 'the convolution output itself must not change'
    scale = zp0.abs().max().item()
    assert (zp1 - zp0).abs().max().item() <= 1e-6 * scale, ((zp1 - zp0).abs().max().item(), scale)
    assert (pn1 - pn0).abs().max().item() <= 1e-6 * pn0.abs().max().item()
    # same operands in the same accumulation order: not merely close
    assert torch.equal(zp1, zp0), f'{(zp1 != zp0).sum().item()} of {zp0.numel()} elements differ in the last bits'
    assert torch.equal(pn1, pn0)
    # and against fp64 of the reference formula (view -> 1x1 conv -> LeakyReLU -> PixelNorm)
    yd = y0.double()
    pre = torch.einsum('ok,nkhw->nohw', wp.double().reshape(16, 16 * D), yd.reshape(N, 16 * D, H, W)) * phe
    if pb is not None:
        pre = pre + pb.double().view(1, 16, 1, 1)
    act = torch.nn.functional.leaky_relu(pre, 0.2)
    want = act / torch.sqrt((act ** 2).mean(dim=1, keepdim=True) + 1e-8)
    assert (zp1.double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('with_prev', [True, False])
def test_projbwd_matches_the_two_launch_form(shape, with_prev):
    from latentfusion_amd import _lib, ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, check
    N, D, H, W = shape
    x, w, b, wp, _ = _problem(shape, 23)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    he, phe = ops.he_constant(w), ops.he_constant(wp)
    g = torch.Generator().manual_seed(5)
    # a real chain: act1 = block conv 1 (x), act2 = block conv 2 (act1); gp = gradient at the projection's pre-activation
    w1 = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
    act1, nrm1 = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w1), None, ops.he_constant(w1), flags)
    act2, nrm2 = ops.conv3d_c16_wino(act1, ops.pack_conv3d_c16_wino(w), b, he, flags)
    gp = ops.cl(torch.randn(N, 16, H, W, generator=g).to(DEV))
    wdm = _proj_matrices(wp, D)
    upt = ops.pack_conv3d_c16_wino(w, transpose=True)
    prev = (act1, nrm1, flags) if with_prev else None
    # two launches: projection data gradient with the block's LeakyReLU' / PixelNorm' folded in, then the data-gradient conv
    L = _lib.lib()
    gvol = ops.empty_cl((N, 16, D, H, W), DEV)
    ppack_t = ops.pack_conv1x1(wdm.t().contiguous())
    check(L.lf_conv1x1_bwd_data(gp.data_ptr(), ppack_t.data_ptr(), gvol.data_ptr(), N, H * W, 16, D * 16, D * H * W * 16, 16, 16,
                                H * W * 16, phe, act2.data_ptr(), nrm2.data_ptr(), flags, ops.SLOPE, None,
                                torch.cuda.current_stream().cuda_stream), 'lf_conv1x1_bwd_data')
    want, _ = ops.conv3d_c16_wino(gvol, upt, None, he, 0, prev=prev)
    # one launch
    got = ops.conv3d_c16_wino_projbwd(gp, ops.pack_wino_proj(wdm, transpose=True), phe, act2, nrm2, flags, upt, he, prev=prev)
    torch.cuda.synchronize()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= 2e-6 * scale, (err, scale)                     # differs only by rcp + Newton vs the IEEE division of 1 / norm
    # fp64 of the same chain (autograd of the reference formula on the saved tensors)
    gd = gvol.double()
    wantd = torch.nn.functional.conv_transpose3d(gd, w.double(), padding=1) * he
    if with_prev:
        y1 = act1.double()
        dot = (wantd * y1).mean(dim=1, keepdim=True)
        wantd = (wantd - y1 * dot) / nrm1.double().view(N, 1, D, H, W)
        wantd = torch.where(y1 > 0, wantd, wantd * 0.2)
    assert (got.double() - wantd).abs().max().item() < 2e-5 * max(1.0, wantd.abs().max().item())


def test_projbwd_is_run_to_run_identical_and_refuses_bad_arguments():
    from latentfusion_amd import _lib, ops
    from latentfusion_amd._lib import LF_EPI_ADD, LF_EPI_LRELU, LF_EPI_PIXELNORM
    shape = (2, 16, 24, 40)
    N, D, H, W = shape
    x, w, b, wp, pb = _problem(shape, 31)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    he, phe = ops.he_constant(w), ops.he_constant(wp)
    act2, nrm2 = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w), b, he, flags)
    gp = ops.cl(torch.randn(N, 16, H, W, device=DEV))
    wdm = _proj_matrices(wp, D)
    wt, upt = ops.pack_wino_proj(wdm, transpose=True), ops.pack_conv3d_c16_wino(w, transpose=True)
    first = ops.conv3d_c16_wino_projbwd(gp, wt, phe, act2, nrm2, flags, upt, he)
    for _ in range(5):
        assert torch.equal(ops.conv3d_c16_wino_projbwd(gp, wt, phe, act2, nrm2, flags, upt, he), first)
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    y = torch.empty_like(act2)
    assert L.lf_conv3d_c16_wino_projbwd(None, wt.data_ptr(), phe, act2.data_ptr(), nrm2.data_ptr(), flags, upt.data_ptr(), y.data_ptr(),
                                        N, D, H, W, he, 0.2, None, None, 0, s) == -1
    assert L.lf_conv3d_c16_wino_projbwd(gp.data_ptr(), wt.data_ptr(), phe, act2.data_ptr(), None, flags, upt.data_ptr(), y.data_ptr(),
                                        N, D, H, W, he, 0.2, None, None, 0, s) == -1              # PixelNorm' without the norms
    assert L.lf_conv3d_c16_wino_projbwd(gp.data_ptr(), wt.data_ptr(), phe, act2.data_ptr(), nrm2.data_ptr(), flags, upt.data_ptr(),
                                        y.data_ptr(), N, D, H, W, he, 0.2, act2.data_ptr(), None, LF_EPI_ADD, s) == -1
    assert L.lf_conv3d_c16_wino_projbwd(gp.data_ptr() + 4, wt.data_ptr(), phe, act2.data_ptr(), nrm2.data_ptr(), flags, upt.data_ptr(),
                                        y.data_ptr(), N, D, H, W, he, 0.2, None, None, 0, s) == -2
    zp = ops.empty_cl((N, 16, H, W), DEV)
    up = ops.pack_conv3d_c16_wino(w)
    assert L.lf_conv3d_c16_wino_projfwd(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), None, N, D, H, W, he, flags, 0.2, 1e-8,
                                        None, None, zp.data_ptr(), None, phe, flags, s) == -1
    assert L.lf_conv3d_c16_wino_projfwd(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), None, N, D, H, W, he, flags, 0.2, 1e-8,
                                        wt.data_ptr(), None, zp.data_ptr(), None, phe, 64, s) == -1
    assert int(L.lf_conv3d_c16_wino_proj_pack_floats(D)) == D * 256


@pytest.mark.parametrize('S', [16, 32])
def test_engine_with_fused_projection_equals_the_unfused_engine(S):
    """RenderLoopEngine with the projection fused into the last camera block's kernels (forward: the default; backward:
    opt-in) against the same engine with the separate projection launches: identical losses (the forward is bit-identical),
    camera gradients to the last bits."""
    from latentfusion_amd import synth
    from latentfusion_amd.experimental import RenderLoopEngineX as RenderLoopEngine   # (the fused BACKWARD form is experimental)
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import utils as pu
    model, _ = synth.build_model(S, 16, 'pool:mean', seed=3, device=DEV, bias_std=0.05)
    model.freeze()
    td = synth.make_observation_data(1, seed=7)
    target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(DEV)
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, S, S, S, generator=gen).to(DEV)
    torch.manual_seed(1)
    cam = pu.sample_cameras_with_estimate(5, target.camera.to('cpu')).zoom(None, model.input_size, model.camera_dist).to(DEV)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    plain = RenderLoopEngine(model.photographer, z_obj, target, weights, fuse_projection=False)
    assert plain.fuse_projection == ()
    l0, g0 = plain.forward_backward(cam)
    for sel in (None, True, ('fwd',), ('bwd',)):
        eng = RenderLoopEngine(model.photographer, z_obj, target, weights, fuse_projection=sel)
        assert eng.conv_mode == 'winograd'
        assert eng.fuse_projection == {None: ('fwd',), True: ('fwd', 'bwd')}.get(sel, sel)
        l1, g1 = eng.forward_backward(cam)
        assert torch.equal(l1, l0), sel
        rel = ((g1 - g0).norm(dim=1) / g0.norm(dim=1)).max().item()
        assert rel < 1e-5, (sel, rel)
        if sel in (None, ('fwd',)):
            assert torch.equal(g1, g0)
    with pytest.raises(NotImplementedError):
        RenderLoopEngine(model.photographer, z_obj, target, weights, conv_mode='fp32', fuse_projection=True)


@pytest.mark.parametrize('cin,cout', [(16, 32), (32, 32), (32, 16), (64, 32), (16, 64)])
def test_conv2d_data_gradient_with_fused_epilogue_backward_on_wide_records(cin, cout):
    """lf_conv3x3_bwd_data with the producer's LeakyReLU' / PixelNorm' folded in, on whole records of 16 / 32 / 64 channels
    (round 4: the 2-D decoder's 32-channel layers; before, 16-channel records only) == the plain data gradient followed by
    lf_epilogue_bwd, and fp64."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(cin * 100 + cout)
    N, H, W = 3, 37, 29
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    w0 = torch.randn(cin, 8, 3, 3, generator=g).to(DEV)                  # producer: 8 -> cin, leaves (y, norm)
    x0 = ops.cl(torch.randn(N, 8, H, W, generator=g).to(DEV))
    y, nrm = ops._conv3x3_raw(x0, ops.pack_conv3x3(w0), None, cin, ops.he_constant(w0), flags, True)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(DEV)                # the layer whose data gradient lands on y
    gy = ops.cl(torch.randn(N, cout, H, W, generator=g).to(DEV))
    he = ops.he_constant(w)
    wt = ops.pack_conv3x3(w, transpose=True)
    fused = ops.conv3x3_bwd_data(gy, wt, cin, he, (y, nrm, flags))
    plain = ops._epilogue_bwd(ops.conv3x3_bwd_data(gy, wt, cin, he, None), y, nrm, flags)
    torch.cuda.synchronize()
    assert (fused - plain).abs().max().item() <= 2e-6 * max(1.0, plain.abs().max().item())
    gd = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), padding=1) * he
    yd = y.double()
    want = (gd - yd * (gd * yd).mean(dim=1, keepdim=True)) / nrm.double().view(N, 1, H, W)
    want = torch.where(yd > 0, want, want * 0.2)
    assert (fused.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_fused_projection_kernels_at_the_headline_size():
    """BASELINE cfg 2's shape (8 hypotheses x 128^3 x 16): the fused forward launch is the two-launch form bit for bit, the
    fused backward launch agrees to the reciprocal's last place, and both are run-to-run identical (a workgroup walks two whole
    columns of tiles here: the per-column state is re-armed correctly)."""
    from latentfusion_amd import _lib, ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, check
    N, D, H, W = 8, 128, 128, 128
    x, w, b, wp, pb = _problem((N, D, H, W), 77)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    he, phe = ops.he_constant(w), ops.he_constant(wp)
    up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
    wdm = _proj_matrices(wp, D)
    y0, n0 = ops.conv3d_c16_wino(x, up, b, he, flags)
    zp0 = ops.empty_cl((N, 16, H, W), DEV)
    pn0 = ops._conv1x1_raw(y0, ops.pack_conv1x1(wdm), pb, N, H * W, 16, D, D * H * W * 16, H * W * 16, 16, zp0, phe, flags)
    wA = ops.pack_wino_proj(wdm)
    for _ in range(2):
        y1, n1, zp1, pn1 = ops.conv3d_c16_wino_projfwd(x, up, b, he, flags, wA, pb, phe, flags)
        assert torch.equal(y1, y0) and torch.equal(n1, n0) and torch.equal(zp1, zp0) and torch.equal(pn1, pn0)
    del y1, n1
    gp = ops.cl(torch.randn(N, 16, H, W, generator=torch.Generator().manual_seed(3)).to(DEV))
    gvol = ops.empty_cl((N, 16, D, H, W), DEV)
    check(_lib.lib().lf_conv1x1_bwd_data(gp.data_ptr(), ops.pack_conv1x1(wdm.t().contiguous()).data_ptr(), gvol.data_ptr(), N, H * W, 16,
                                         D * 16, D * H * W * 16, 16, 16, H * W * 16, phe, y0.data_ptr(), n0.data_ptr(), flags, ops.SLOPE,
                                         None, torch.cuda.current_stream().cuda_stream), 'lf_conv1x1_bwd_data')
    want, _ = ops.conv3d_c16_wino(gvol, upt, None, he, 0, prev=(x, n0, flags))      # (x / n0 stand in for a producer's record)
    del gvol
    wtA = ops.pack_wino_proj(wdm, transpose=True)
    got = ops.conv3d_c16_wino_projbwd(gp, wtA, phe, y0, n0, flags, upt, he, prev=(x, n0, flags))
    assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert torch.equal(ops.conv3d_c16_wino_projbwd(gp, wtA, phe, y0, n0, flags, upt, he, prev=(x, n0, flags)), got)
