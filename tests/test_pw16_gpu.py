"""lf_pw16_fwd / lf_pw16_bwd (csrc/pw16.hip: 1x1x1 16 -> 16 layers of the training step as one bf16 MFMA per 16 voxels -- the
sculptor's output block, reference recon/models.py:143,222 over modules/blocks.py:108-119) against the path they replace (the
3x3x3 ring kernels with the weights on the centre tap + lf_epilogue_bwd_c16 for the bias gradient) and against plain torch in fp32
on the bf16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(N, D, H, W, lrelu, x_bf16, pw, seed=0):
    from latentfusion_amd import ops, ops_train
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 16, D, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    if x_bf16:
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)
    w = torch.randn(16, 16, 1, 1, 1, generator=g).cuda().requires_grad_(True)
    b = (torch.randn(16, generator=g) * 0.3).cuda().requires_grad_(True)
    gout = torch.randn(N, 16, D, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    old = ops_train.PW16
    ops_train.PW16 = pw
    try:
        with ops.autocast(True):
            y = ops.conv1x1(x, w, b, lrelu=lrelu)
        assert y.dtype == torch.bfloat16
        y.backward(gout)
    finally:
        ops_train.PW16 = old
    return y.detach(), x.grad, w.grad, b.grad, (x.detach(), w.detach(), b.detach(), gout)


def _one_ulp(a, b, frac=2e-3):
    a, b = a.float(), b.float()
    ok = (a - b).abs() <= 1.01 * 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-30
    assert ok.all(), (a - b).abs().max().item()
    assert (a != b).float().mean().item() < frac                   # roundings that fall the other way are rare


@pytest.mark.parametrize('N,D,H,W', [(2, 8, 8, 16), (1, 5, 3, 7), (3, 16, 16, 16)])
@pytest.mark.parametrize('lrelu', [False, True])
@pytest.mark.parametrize('x_bf16', [True, False])
def test_pw16_equals_the_centre_tap_path(N, D, H, W, lrelu, x_bf16):
    new = _run(N, D, H, W, lrelu, x_bf16, True)
    ref = _run(N, D, H, W, lrelu, x_bf16, False)
    again = _run(N, D, H, W, lrelu, x_bf16, True)
    for a, b in zip(new[:4], again[:4]):
        assert torch.equal(a, b)                                      # run-to-run identical
    _one_ulp(new[0], ref[0])
    _one_ulp(new[1], ref[1])
    assert new[1].dtype == ref[1].dtype
    rel = lambda p, q: ((p.double() - q.double()).norm() / q.double().norm().clamp_min(1e-30)).item()  # noqa: E731
    assert rel(new[2], ref[2]) < 2e-3 and rel(new[3], ref[3]) < 1e-4, (rel(new[2], ref[2]), rel(new[3], ref[3]))
    # plain torch on the rounded operands: y = act(bf16(bf16(W x) * he) + b)
    x, w, b, gout = new[4]
    he = (2.0 / 16) ** 0.5
    xr, wr = x.float().to(torch.bfloat16).float(), w.to(torch.bfloat16).float().reshape(16, 16)
    acc = torch.einsum('oc,ncdhw->nodhw', wr, xr)
    want = (acc.to(torch.bfloat16).float() * he).to(torch.bfloat16).float() + b.view(1, 16, 1, 1, 1)
    if lrelu:
        want = torch.nn.functional.leaky_relu(want, 0.2)
    _one_ulp(new[0], want.to(torch.bfloat16), frac=2e-2)
