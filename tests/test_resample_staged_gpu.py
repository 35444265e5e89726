"""LDS-staged forms of the 16-channel resamplers (csrc/resample_staged.inc; lf_set_tuning(1, 4)) through the C ABI:

  * forward: object -> camera map BIT-IDENTICAL to the gather form (variant 3; camera -> object: to rounding), which the golden /
    oracle tests pin to the reference's F.grid_sample (modules/geometry.py:625-690) -- on near-isometric maps (everything is
    staged), on magnified / strongly sheared maps (footprints overflow the window or the buffer: lanes fall back to global
    memory), on clamped-out and NaN maps, on ragged volume sizes, with one shared and with per-sample volumes;
  * coefficient gradient: against the gather form (same value, different summation order) and against autograd through an
    fp64 evaluation of the same map."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
O2C, C2O = 0, 1


def _lib():
    from latentfusion_amd import _lib
    return _lib.lib()


def _rot(gen):
    q = torch.randn(4, generator=gen, dtype=torch.float64)
    w, x, y, z = (q / q.norm()).tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)


def o2c_coefs(N, gen, scales, persp=0.1, shift=0.1):
    """(N,20) blocks: g = t + s R (2 (a,b,k) - 1) + perspective-like a*k, b*k terms."""
    cf = torch.zeros(N, 20, dtype=torch.float64)
    for n in range(N):
        R, s = _rot(gen), scales[n % len(scales)]
        t = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * shift
        M = 2 * s * R
        cf[n, 0:3] = t - s * R.sum(dim=1)
        cf[n, 3:6], cf[n, 6:9], cf[n, 9:12] = M[:, 0], M[:, 1], M[:, 2]
        cf[n, 12:15] = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * persp
        cf[n, 15:18] = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * persp
    return cf


def c2o_coefs(N, gen, scales, persp=0.15):
    cf = torch.zeros(N, 20, dtype=torch.float64)
    for n in range(N):
        R, s = _rot(gen), scales[n % len(scales)]
        A = torch.zeros(4, 4, dtype=torch.float64)
        A[:3, :3] = s * R
        A[:3, 3] = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 0.2
        A[3, :3] = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * persp
        A[3, 3] = 1.0
        cf[n, :16] = A.reshape(-1)
    return cf


def grid64(cf, kind, D, H, W):
    """The sampling grid of include/lf_hip.h in fp64, (N,D,H,W,3), differentiable w.r.t. cf."""
    a = torch.linspace(0, 1, W, dtype=torch.float64) if W > 1 else torch.zeros(1, dtype=torch.float64)
    b = torch.linspace(0, 1, H, dtype=torch.float64) if H > 1 else torch.zeros(1, dtype=torch.float64)
    k = torch.linspace(0, 1, D, dtype=torch.float64) if D > 1 else torch.zeros(1, dtype=torch.float64)
    kk, bb, aa = torch.meshgrid(k, b, a, indexing='ij')
    if kind == O2C:
        basis = torch.stack((torch.ones_like(aa), aa, bb, kk, aa * kk, bb * kk), dim=-1)            # (D,H,W,6)
        c = cf[:, :18].reshape(-1, 6, 3)
        return torch.einsum('dhwj,njc->ndhwc', basis, c)
    l = torch.stack((2 * aa - 1, 2 * bb - 1, 2 * kk - 1, torch.ones_like(aa)), dim=-1)              # (D,H,W,4)
    A = cf[:, :16].reshape(-1, 4, 4)
    n = torch.einsum('dhwj,nrj->ndhwr', l, A)
    return torch.stack((n[..., 0] / n[..., 3], n[..., 1] / n[..., 3], n[..., 2]), dim=-1)


@pytest.fixture(autouse=True)
def _restore_variant():
    L = _lib()
    prev = L.lf_set_tuning(1, 3)
    L.lf_set_tuning(1, prev)
    yield
    L.lf_set_tuning(1, prev)


def run_fwd(L, variant, vol, cf, kind, N, D, H, W):
    from latentfusion_amd import ops
    L.lf_set_tuning(1, variant)
    out = ops.empty_cl((N, 16, D, H, W), DEV)
    rc = L.lf_resample3d_fwd(vol.data_ptr(), vol.shape[0], cf.data_ptr(), kind, out.data_ptr(), N, D, H, W, 16,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return out


def run_bwd(L, variant, gout, vol, cf, N, D, H, W):
    L.lf_set_tuning(1, variant)
    nb = L.lf_resample3d_bwd_coef_scratch_bytes(N, D, H, W)
    scratch = torch.empty(nb // 4 + 1, device=DEV)
    gc = torch.empty(N, 18, device=DEV)
    rc = L.lf_resample3d_bwd_coef(gout.data_ptr(), vol.data_ptr(), vol.shape[0], cf.data_ptr(), gc.data_ptr(), scratch.data_ptr(),
                                  scratch.numel() * 4, N, D, H, W, 16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return gc


CASES = [
    # name, (D,H,W), N, vol_n, scales
    ('isometric', (32, 32, 32), 4, 1, [1.0, 0.8, 0.9, 0.75]),
    ('pose_loop_like', (64, 64, 64), 3, 1, [0.75, 0.7, 0.8]),
    ('magnified', (24, 24, 24), 4, 1, [2.5, 4.0, 1.6, 0.3]),        # footprints beyond the window / buffer -> fall-backs
    ('ragged', (9, 21, 13), 3, 1, [1.0, 0.6, 1.3]),
    ('per_sample_volumes', (16, 20, 24), 3, 3, [1.0, 0.9, 1.2]),
    ('thin', (2, 8, 40), 2, 1, [1.0, 0.5]),
]


@pytest.mark.parametrize('kind', [O2C, C2O])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_staged_gather_is_bit_identical_to_the_gather_form(case, kind):
    from latentfusion_amd import ops
    name, (D, H, W), N, vol_n, scales = case
    L = _lib()
    gen = torch.Generator().manual_seed(sum(map(ord, name)) + kind)
    cf64 = (o2c_coefs if kind == O2C else c2o_coefs)(N, gen, scales)
    cf = cf64.float().to(DEV).contiguous()
    vol = ops.cl(torch.randn(vol_n, 16, D, H, W, generator=gen).to(DEV))
    ref = run_fwd(L, 3, vol, cf, kind, N, D, H, W)
    got = run_fwd(L, 4, vol, cf, kind, N, D, H, W)
    # variant 5: the gather form with the map evaluated once per voxel (lane-per-voxel phase + wave-private LDS hand-over)
    ded = run_fwd(L, 5, vol, cf, kind, N, D, H, W)
    if kind == O2C:
        assert torch.equal(ded, ref), (name, (ded - ref).abs().max().item(), int((ded != ref).sum()))
    else:
        assert (ded - ref).abs().max().item() < 1e-4 * max(1.0, max(scales))
    if kind == O2C:
        assert torch.equal(got, ref), (name, (got - ref).abs().max().item(), int((got != ref).sum()))
    else:
        # the projective map's divisions / multiply-adds contract differently at the two call sites (sampling positions differ
        # in the last bit), so values agree to rounding, not bit for bit; an indexing error would be O(1)
        assert (got - ref).abs().max().item() < 1e-4 * max(1.0, max(scales)), (name, (got - ref).abs().max().item())
    # and both are the reference's grid_sample (fp64 evaluation of the same map; continuous in the position)
    want = F.grid_sample(vol.double().cpu().expand(N, -1, -1, -1, -1) if vol_n == 1 else vol.double().cpu(),
                         grid64(cf.double().cpu(), kind, D, H, W), mode='bilinear', padding_mode='border', align_corners=False)
    assert (got.cpu().double() - want).abs().max().item() < 2e-3 * max(1.0, max(scales))


def test_staged_gather_degenerate_maps():
    """NaN coefficients sample voxel 0 (ATen clip semantics), far-away maps clamp to the border -- identical in both forms."""
    from latentfusion_amd import ops
    L = _lib()
    gen = torch.Generator().manual_seed(5)
    D = H = W = 16
    cf = o2c_coefs(4, gen, [1.0]).float()
    cf[1, 4] = float('nan')
    cf[2, 0:3] += 7.0                                # everything outside: all taps clamp to one corner region
    cf[3, 3:18] = 0.0                                # constant map: the whole tile reads one record
    cf = cf.to(DEV)
    vol = ops.cl(torch.randn(1, 16, D, H, W, generator=gen).to(DEV))
    ref = run_fwd(L, 3, vol, cf, O2C, 4, D, H, W)
    got = run_fwd(L, 4, vol, cf, O2C, 4, D, H, W)
    assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(ref, nan=123.0))


@pytest.mark.parametrize('case', CASES[:5], ids=[c[0] for c in CASES[:5]])
def test_staged_coefficient_gradient(case):
    from latentfusion_amd import ops
    name, (D, H, W), N, vol_n, scales = case
    L = _lib()
    gen = torch.Generator().manual_seed(sum(map(ord, name)) + 17)
    cf64 = o2c_coefs(N, gen, scales)
    cf = cf64.float().to(DEV).contiguous()
    vol = ops.cl(torch.randn(vol_n, 16, D, H, W, generator=gen).to(DEV))
    gout = ops.cl(torch.randn(N, 16, D, H, W, generator=gen).to(DEV))
    ref = run_bwd(L, 3, gout, vol, cf, N, D, H, W)
    got = run_bwd(L, 4, gout, vol, cf, N, D, H, W)
    again = run_bwd(L, 4, gout, vol, cf, N, D, H, W)
    assert torch.equal(got, again)                                    # fixed-order reduction: run-to-run identical
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    assert ((got - ref).abs() / scale).max().item() < 2e-4, ((got - ref).abs() / scale).max().item()
    # autograd through an fp64 evaluation of the same map (the sums cancel heavily: compare per sample against the largest
    # component; the fp32 position of a tap within rounding of a cell boundary takes the other cell's slope)
    c = cf.double().cpu().requires_grad_(True)
    v64 = vol.double().cpu()
    out = F.grid_sample(v64.expand(N, -1, -1, -1, -1) if vol_n == 1 else v64, grid64(c, O2C, D, H, W), mode='bilinear',
                        padding_mode='border', align_corners=False)
    (out * gout.double().cpu()).sum().backward()
    want = c.grad[:, :18]
    s64 = want.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    e4 = ((got.cpu().double() - want).abs() / s64).max().item()
    e3 = ((ref.cpu().double() - want).abs() / s64).max().item()
    assert e4 < max(5e-3, 2.0 * e3), (e4, e3)


def test_staged_forms_at_the_headline_size():
    """N = 8, 128^3 x 16 with the bench's sampled cameras: forward bit-identical to the gather form; coefficient gradient as
    close to autograd through an fp64 evaluation of the map as the gather form's is.  (The two fp32 forms differ from each other
    and from fp64 by ~3e-3 of the largest component at this size: d(trilinear)/d(position) jumps at cell boundaries, a few
    dozen of the 2 M taps per sample sit within rounding of one, and each call site rounds the position in its own way.)"""
    from latentfusion_amd import consts, engine, ops, synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.pose import utils as pu
    from latentfusion_amd.recon.utils import optimal_camera_dist
    L = _lib()
    S, N = 128, 8
    gen = torch.Generator().manual_seed(0)
    z = ops.cl(torch.randn(1, 16, S, S, S, generator=gen).to(DEV))
    gout = ops.cl(torch.randn(N, 16, S, S, S, generator=gen).to(DEV))
    tdata = synth.make_observation_data(1, seed=200)
    torch.manual_seed(300)
    dist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
    cam = pu.sample_cameras_with_estimate(N, Camera(tdata['intrinsic'], tdata['extrinsic'])).zoom(None, S, dist).to(DEV)
    cf = torch.zeros(N, 20, device=DEV)
    cf[:, :18] = engine.camera_coefs(cam, 1.0, S, S).detach()[:, :18]
    ref = run_fwd(L, 3, z, cf, O2C, N, S, S, S)
    got = run_fwd(L, 4, z, cf, O2C, N, S, S, S)
    assert torch.equal(got, ref)
    del ref, got
    g3 = run_bwd(L, 3, gout, z, cf, N, S, S, S)
    g4 = run_bwd(L, 4, gout, z, cf, N, S, S, S)
    assert torch.equal(g4, run_bwd(L, 4, gout, z, cf, N, S, S, S))
    M = 2                                                             # fp64 autograd of the first samples (CPU)
    c = cf[:M].double().cpu().requires_grad_(True)
    out = F.grid_sample(z.double().cpu().expand(M, -1, -1, -1, -1), grid64(c, O2C, S, S, S), mode='bilinear',
                        padding_mode='border', align_corners=False)
    (out * gout[:M].double().cpu()).sum().backward()
    want = c.grad[:, :18]
    sc = want.abs().amax(dim=1, keepdim=True)
    e3 = ((g3[:M].cpu().double() - want).abs() / sc).max().item()
    e4 = ((g4[:M].cpu().double() - want).abs() / sc).max().item()
    assert e4 < max(2e-3, 1.5 * e3) and e4 < 2e-2, (e4, e3)


@pytest.mark.parametrize('variant', [6, 10])
@pytest.mark.parametrize('case', CASES[:5], ids=[c[0] for c in CASES[:5]])
def test_deduplicated_coefficient_gradient(case, variant):
    """lf_set_tuning(2, 6 | 10): the gather form of the coefficient gradient with the per-voxel arithmetic done once per voxel
    (lane-per-voxel phases around the lane-per-quarter gather, wave-private LDS hand-over; 10 = the default: gradient records
    requested one sub-tile ahead) == the round-2 form (2) up to the summation order, and run-to-run identical."""
    from latentfusion_amd import ops
    name, (D, H, W), N, vol_n, scales = case
    L = _lib()
    gen = torch.Generator().manual_seed(sum(map(ord, name)) + 29)
    cf = o2c_coefs(N, gen, scales).float().to(DEV).contiguous()
    vol = ops.cl(torch.randn(vol_n, 16, D, H, W, generator=gen).to(DEV))
    gout = ops.cl(torch.randn(N, 16, D, H, W, generator=gen).to(DEV))
    prev = L.lf_set_tuning(2, 2)
    try:
        ref = run_bwd(L, 3, gout, vol, cf, N, D, H, W)
        assert L.lf_set_tuning(2, variant) == 2
        got = run_bwd(L, 3, gout, vol, cf, N, D, H, W)
        again = run_bwd(L, 3, gout, vol, cf, N, D, H, W)
    finally:
        L.lf_set_tuning(2, prev)
    assert torch.equal(got, again)
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    assert ((got - ref).abs() / scale).max().item() < 2e-4, ((got - ref).abs() / scale).max().item()
