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
