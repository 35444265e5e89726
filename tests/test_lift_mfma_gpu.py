"""lf_lift16_fwd / lf_lift16_bwd (csrc/lift_mfma.hip: the training step's 2-D -> 3-D lift as one MFMA kernel each way, the
16*S-channel row never stored) against the row-materialising path of round 5 (conv1x1 -> lf_lift_norm_unfold; lf_lift_bwd ->
conv1x1 data gradient + generic weight gradient + column sums) and against plain torch (FactorProjection2d3d in fp32 on the
bf16-rounded operands)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(V, S, H, W, mfma, seed=0):
    from latentfusion_amd import ops, ops_train
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(V, 16, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(16 * S, 16, 1, 1, generator=g)).cuda().requires_grad_(True)
    b = (torch.randn(16 * S, generator=g) * 0.2).cuda().requires_grad_(True)
    gout = torch.randn(V, 16, S, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    old = ops_train.LIFT_MFMA
    ops_train.LIFT_MFMA = mfma
    try:
        with ops.autocast(True):
            vol = ops.lift(x, w, b, S)
        assert vol.dtype == torch.bfloat16 and tuple(vol.shape) == (V, 16, S, H, W)
        (vol.float() * gout).sum().backward()
    finally:
        ops_train.LIFT_MFMA = old
    return vol.detach(), x.grad, w.grad, b.grad, (x.detach(), w.detach(), b.detach(), gout)


@pytest.mark.parametrize('V,S,H,W', [(2, 32, 8, 16), (1, 16, 4, 4), (3, 64, 8, 8), (1, 128, 16, 16)])
def test_lift_mfma_equals_the_row_path(V, S, H, W):
    from latentfusion_amd import ops
    new = _run(V, S, H, W, True)
    ref = _run(V, S, H, W, False)
    again = _run(V, S, H, W, True)
    for a, b in zip(new[:4], again[:4]):
        assert torch.equal(a, b)                                      # run-to-run identical
    a, b = new[0].float(), ref[0].float()
    assert ((a - b).abs() <= 1.01 * 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-30).all()   # one bf16 ulp
    cos = lambda p, q: torch.nn.functional.cosine_similarity(p.reshape(1, -1).double(), q.reshape(1, -1).double()).item()  # noqa: E731
    for k in (1, 2, 3):
        assert cos(new[k], ref[k]) > 0.9999, (k, cos(new[k], ref[k]))
        assert float((new[k] - ref[k]).norm() / ref[k].norm()) < 1.5e-2, k
    # plain torch on the rounded operands: conv1x1 * he + b, LeakyReLU, PixelNorm over all channels, view (c, d)
    x, w, bias, gout = new[4]
    xr, wr = ops.round_bf16(x).requires_grad_(True), ops.round_bf16(w).requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    he = (2.0 / 16) ** 0.5
    t = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr, wr) * he + br.view(1, -1, 1, 1), 0.2)
    y = t / torch.sqrt((t * t).mean(dim=1, keepdim=True) + 1e-8)
    yv = y.view(V, 16, S, H, W)
    assert float((new[0].float() - yv.detach()).abs().max()) <= 2.0 ** -7 * float(yv.detach().abs().max()) + 1e-6
    (yv * gout).sum().backward()
    for got, want in ((new[1], xr.grad), (new[2], wr.grad), (new[3], br.grad)):
        assert cos(got, want) > 0.999, cos(got, want)


def _run_proj(N, S, H, W, mfma, seed=0):
    from latentfusion_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 16, S, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d).to(torch.bfloat16).requires_grad_(True)
    w = torch.randn(16, 16 * S, 1, 1, generator=g).cuda().requires_grad_(True)
    b = (torch.randn(16, generator=g) * 0.2).cuda().requires_grad_(True)
    gout = torch.randn(N, 16, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    old = ops.PROJ_MFMA
    ops.PROJ_MFMA = mfma
    try:
        with ops.autocast(True):
            y = ops.factor_project(x, w, b)
        (y * gout).sum().backward()
    finally:
        ops.PROJ_MFMA = old
    return y.detach(), x.grad, w.grad, b.grad, (x.detach(), w.detach(), b.detach(), gout)


@pytest.mark.parametrize('N,S,H,W', [(2, 32, 8, 16), (1, 16, 4, 4), (2, 64, 8, 8), (1, 128, 16, 16)])
def test_proj16_mfma_equals_the_conv1x1_path(N, S, H, W):
    """lf_proj16_fwd / lf_proj16_bwd (FactorProjection3d2d of the training step on the bf16 volume as it lies) against the round-5
    path (fp32 copy + lf_conv1x1_fwd; conv1x1 into an fp32 volume + rounding + row copy + generic weight gradient) and torch."""
    from latentfusion_amd import ops
    new, ref, again = _run_proj(N, S, H, W, True), _run_proj(N, S, H, W, False), _run_proj(N, S, H, W, True)
    for a, b in zip(new[:4], again[:4]):
        assert torch.equal(a, b)
    assert new[1].dtype == torch.bfloat16 and new[1].shape == new[4][0].shape
    torch.testing.assert_close(new[0], ref[0], atol=2e-5, rtol=2e-5)
    cos = lambda p, q: torch.nn.functional.cosine_similarity(p.reshape(1, -1).double(), q.reshape(1, -1).double()).item()  # noqa: E731
    a, b = new[1].float(), ref[1].float()
    assert ((a - b).abs() <= 1.01 * 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-30).all()
    assert cos(new[2], ref[2]) > 0.99999 and float((new[2] - ref[2]).norm() / ref[2].norm()) < 5e-3
    torch.testing.assert_close(new[3], ref[3], atol=1e-5, rtol=1e-4)
    x, w, bias, gout = new[4]
    xr, wr, br = x.float().requires_grad_(True), ops.round_bf16(w).requires_grad_(True), bias.clone().requires_grad_(True)
    he = (2.0 / (16 * S)) ** 0.5
    t = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr.reshape(N, 16 * S, H, W), wr) * he + br.view(1, -1, 1, 1), 0.2)
    y = t / torch.sqrt((t * t).mean(dim=1, keepdim=True) + 1e-8)
    torch.testing.assert_close(new[0], y.detach(), atol=1e-4, rtol=1e-4)
    (y * gout).sum().backward()
    assert cos(new[1].float(), xr.grad) > 0.9995 and cos(new[2], wr.grad) > 0.999 and cos(new[3], br.grad) > 0.9999
