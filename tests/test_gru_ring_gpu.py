"""lf_conv3d_c16_ring_multi (csrc/conv_gru.hip: the multi-output bf16 ring convolution with the ConvGRU's element-wise stages
in its epilogue) against the one-output ring kernel + the stage kernels it replaces, and against plain torch (fp32 conv3d on the
bf16-rounded operands).  Tolerances: a bf16-stored result may differ by one bf16 ulp (at most 2^-7 relative) where the fused epilogue
rounds once instead of twice; fp32 results 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1, 6, 16, 32), (2, 5, 11, 19), (1, 4, 8, 16)]          # (N, D, H, W): tile-aligned, ragged, a single tile column


def _setup(N, D, H, W, seed=0, x_bf16=True):
    from latentfusion_amd import ops
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda'
    x = torch.randn(N, 16, D, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    if x_bf16:
        x = x.to(torch.bfloat16)
    w = [torch.randn(16, 16, 3, 3, 3, generator=g).to(dev) for _ in range(2)]
    packs = torch.stack([ops.pack_conv3d_c16_ring_bf16(t) for t in w]).contiguous()
    he = float((2.0 / (16 * 27)) ** 0.5)
    return x, w, packs, he, g


def _vol(g, N, D, H, W, dtype, scale=1.0):
    return (torch.randn(N, 16, D, H, W, generator=g) * scale).cuda().contiguous(memory_format=torch.channels_last_3d).to(dtype)


def _close_bf16(a, b, ulps=1.01):
    a, b = a.float(), b.float()
    tol = ulps * 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-30       # one bf16 ulp is 2^-8 .. 2^-7 of the value
    bad = ((a - b).abs() > tol)
    assert not bad.any(), f'{int(bad.sum())} of {a.numel()} beyond one bf16 ulp, max diff {float((a - b).abs().max())}'


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_two_groups_addend_forms_equal_the_one_output_kernel(N, D, H, W):
    from latentfusion_amd import ops, ops_train
    x, w, packs, he, g = _setup(N, D, H, W)
    base = _vol(g, 1, D, H, W, torch.float32)                       # one fp32 addend volume for all samples (the coordinate part)
    addb = _vol(g, N, D, H, W, torch.bfloat16)                      # a per-sample bf16 addend
    y0 = ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    ops_train.ring_multi(x, packs, he, [(y0, base, False), (y0.clone(), None, False)], addend_per_sample=False)
    want0, _ = ops.conv3d_c16_ring_bf16_io(x, packs[0], None, he, 0, 0, addend=base.expand(N, -1, -1, -1, -1).contiguous(memory_format=torch.channels_last_3d), out_bf16=True)
    _close_bf16(y0, want0)
    # group 0: bf16 addend -> bf16, in place; group 1: fp32 addend -> fp32
    acc = addb.clone()
    add32 = _vol(g, N, D, H, W, torch.float32)
    y1 = ops.empty_cl16((N, 16, D, H, W), 'cuda', False)
    ops_train.ring_multi(x, packs, he, [(acc, acc, False), (y1, add32, False)])
    want_a, _ = ops.conv3d_c16_ring_bf16_io(x, packs[0], None, he, 0, 0, addend=addb, out_bf16=True)
    want_b, _ = ops.conv3d_c16_ring_bf16_io(x, packs[1], None, he, 0, 0, addend=add32, out_bf16=False)
    _close_bf16(acc, want_a)
    torch.testing.assert_close(y1, want_b, atol=1e-5, rtol=1e-5)
    # against plain torch: fp32 convolution of the bf16-rounded operands
    ref = torch.nn.functional.conv3d(x.float(), w[1].to(torch.bfloat16).float(), padding=1) * he + add32
    torch.testing.assert_close(y1, ref.contiguous(memory_format=torch.channels_last_3d), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_rounded_forms_and_single_group(N, D, H, W):
    from latentfusion_amd import ops, ops_train
    x, w, packs, he, g = _setup(N, D, H, W, seed=1)
    y0 = ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    ops_train.ring_multi(x, packs[:1].contiguous(), he, [(y0, None, True)])
    want, _ = ops.conv3d_c16_ring_bf16_io(x, packs[0], None, he, 0, 1, out_bf16=True)
    assert torch.equal(y0, want)                                    # the same sums, the same roundings
    base = _vol(g, N, D, H, W, torch.float32)
    ops_train.ring_multi(x, packs[1:].contiguous(), he, [(y0, base, False)])
    want, _ = ops.conv3d_c16_ring_bf16_io(x, packs[1], None, he, 0, 0, addend=base, out_bf16=True)
    _close_bf16(y0, want)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_reset_gate_epilogue(N, D, H, W):
    """(h -> upre, rpre) with r h in the epilogue == two addend convolutions + lf_gru_train_stage_a."""
    from latentfusion_amd import _lib, ops, ops_train
    h, w, packs, he, g = _setup(N, D, H, W, seed=2, x_bf16=False)
    uz, rz = _vol(g, N, D, H, W, torch.bfloat16), _vol(g, N, D, H, W, torch.bfloat16)
    upre, rpre, rh = (ops.empty_cl16((N, 16, D, H, W), 'cuda', True) for _ in range(3))
    ops_train.ring_multi(h, packs, he, [(upre, uz, False), (rpre, rz, False)], extra=_lib.LF_RING_EX_RH, o2=rh)
    wu, _ = ops.conv3d_c16_ring_bf16_io(h, packs[0], None, he, 0, 0, addend=uz, out_bf16=True)
    wr, _ = ops.conv3d_c16_ring_bf16_io(h, packs[1], None, he, 0, 0, addend=rz, out_bf16=True)
    _close_bf16(upre, wu)
    _close_bf16(rpre, wr)
    want = (h * torch.sigmoid(rpre.float())).to(torch.bfloat16)     # from rpre AS STORED by the fused launch
    _close_bf16(rh, want)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_blend_epilogue(N, D, H, W):
    """(r h -> cand) with h' = h (1 - u) + cand u in the epilogue == addend convolution + lf_gru_train_stage_b."""
    from latentfusion_amd import _lib, ops, ops_train
    rh, w, packs, he, g = _setup(N, D, H, W, seed=3)
    oz, upre = _vol(g, N, D, H, W, torch.bfloat16), _vol(g, N, D, H, W, torch.bfloat16, 2.0)
    h = _vol(g, N, D, H, W, torch.float32)
    cand = ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    hn = ops.empty_cl16((N, 16, D, H, W), 'cuda', False)
    ops_train.ring_multi(rh, packs[:1].contiguous(), he, [(cand, oz, False)], extra=_lib.LF_RING_EX_BLEND, e0=h, e1=upre, o2=hn)
    wc, _ = ops.conv3d_c16_ring_bf16_io(rh, packs[0], None, he, 0, 0, addend=oz, out_bf16=True)
    _close_bf16(cand, wc)
    u = torch.sigmoid(upre.float())
    torch.testing.assert_close(hn, h * (1 - u) + cand.float() * u, atol=2e-6, rtol=2e-6)
    L = _lib.lib()                                                  # and the stage kernel it replaces, on the same stored tensors
    hn2 = torch.empty_like(hn)
    _lib.check(L.lf_gru_train_stage_b(h.data_ptr(), upre.data_ptr(), cand.data_ptr(), hn2.data_ptr(), h.numel(), 1,
                                      torch.cuda.current_stream().cuda_stream), 'lf_gru_train_stage_b')
    torch.testing.assert_close(hn, hn2, atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_reset_backward_epilogue(N, D, H, W):
    """(gc -> g_rh, g_x) with grpre / gh12 in the epilogue == two rounded convolutions + lf_gru_train_stage_a_bwd."""
    from latentfusion_amd import _lib, ops, ops_train
    gc, w, packs, he, g = _setup(N, D, H, W, seed=4)
    rpre = _vol(g, N, D, H, W, torch.bfloat16, 2.0)
    h, gh1 = _vol(g, N, D, H, W, torch.float32), _vol(g, N, D, H, W, torch.float32)
    grpre = rpre.clone()                                            # written in place over the saved pre-activation
    gz = ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    gh12 = ops.empty_cl16((N, 16, D, H, W), 'cuda', False)
    ops_train.ring_multi(gc, packs, he, [(grpre, grpre, True), (gz, None, True)], extra=_lib.LF_RING_EX_ABWD, e0=h, e1=gh1, o2=gh12)
    grh, _ = ops.conv3d_c16_ring_bf16_io(gc, packs[0], None, he, 0, 1, out_bf16=True)
    wz, _ = ops.conv3d_c16_ring_bf16_io(gc, packs[1], None, he, 0, 1, out_bf16=True)
    assert torch.equal(gz, wz)
    L = _lib.lib()
    want_p, want_h = torch.empty_like(rpre), torch.empty_like(gh12)
    _lib.check(L.lf_gru_train_stage_a_bwd(grh.data_ptr(), rpre.data_ptr(), h.data_ptr(), gh1.data_ptr(), want_p.data_ptr(),
                                          want_h.data_ptr(), None, h.numel(), 1, torch.cuda.current_stream().cuda_stream),
               'lf_gru_train_stage_a_bwd')
    torch.testing.assert_close(gh12, want_h, atol=2e-6, rtol=2e-5)
    _close_bf16(grpre, want_p)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_one_group_forms_of_the_fused_epilogues(N, D, H, W):
    """The one-output launches the recurrence runs on its sequential chain: (h -> rpre, r h), (gc -> grpre, gh12), (h -> upre)
    equal the corresponding group of the two-output launches bit for bit (same staging, same sums, same epilogue)."""
    from latentfusion_amd import _lib, ops, ops_train
    h, w, packs, he, g = _setup(N, D, H, W, seed=5, x_bf16=False)
    uz, rz = _vol(g, N, D, H, W, torch.bfloat16), _vol(g, N, D, H, W, torch.bfloat16)
    e = lambda: ops.empty_cl16((N, 16, D, H, W), 'cuda', True)     # noqa: E731
    up2, rp2, rh2, up1, rp1, rh1 = e(), e(), e(), e(), e(), e()
    ops_train.ring_multi(h, packs, he, [(up2, uz, False), (rp2, rz, False)], extra=_lib.LF_RING_EX_RH, o2=rh2)
    ops_train.ring_multi(h, packs[:1].contiguous(), he, [(up1, uz, False)])
    ops_train.ring_multi(h, packs[1:].contiguous(), he, [(rp1, rz, False)], extra=_lib.LF_RING_EX_RH, o2=rh1)
    assert torch.equal(up1, up2) and torch.equal(rp1, rp2) and torch.equal(rh1, rh2)
    gc = _vol(g, N, D, H, W, torch.bfloat16)
    rpre = _vol(g, N, D, H, W, torch.bfloat16, 2.0)
    hh, gh1 = _vol(g, N, D, H, W, torch.float32), _vol(g, N, D, H, W, torch.float32)
    gp2, gp1, gz = rpre.clone(), rpre.clone(), e()
    o2, o1 = ops.empty_cl16((N, 16, D, H, W), 'cuda', False), ops.empty_cl16((N, 16, D, H, W), 'cuda', False)
    ops_train.ring_multi(gc, packs, he, [(gp2, gp2, True), (gz, None, True)], extra=_lib.LF_RING_EX_ABWD, e0=hh, e1=gh1, o2=o2)
    ops_train.ring_multi(gc, packs[:1].contiguous(), he, [(gp1, gp1, True)], extra=_lib.LF_RING_EX_ABWD, e0=hh, e1=gh1, o2=o1)
    assert torch.equal(gp1, gp2) and torch.equal(o1, o2)


def test_refused_combinations():
    from latentfusion_amd import _lib, ops, ops_train
    x, w, packs, he, g = _setup(1, 4, 8, 16, x_bf16=False)
    y = ops.empty_cl16((1, 16, 4, 8, 16), 'cuda', True)
    with pytest.raises(_lib.LFHipError):                            # fp32 input without the reset-gate epilogue is not instantiated
        ops_train.ring_multi(x, packs, he, [(y, None, True), (y.clone(), None, True)])
    with pytest.raises(_lib.LFHipError):                            # the blend epilogue needs h / upre / an output
        ops_train.ring_multi(x.to(torch.bfloat16), packs[:1].contiguous(), he, [(y, None, False)], extra=_lib.LF_RING_EX_BLEND)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_previous_layer_backward_in_the_data_gradient_store(N, D, H, W):
    """LF_RING_EX_PREV: the data gradient of a 16 -> 16 layer with the PRODUCER's LeakyReLU' / PixelNorm' and bias sums in its
    store == the rounded data gradient followed by lf_epilogue_bwd_c16 on the producer's activation."""
    from latentfusion_amd import _lib, ops, ops_train
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    gp, w, packs, he, g = _setup(N, D, H, W, seed=7)
    # a producer activation: PixelNorm output with its norms
    t = torch.randn(N, 16, D, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    t = torch.nn.functional.leaky_relu(t, 0.2)
    nrm = torch.sqrt((t * t).mean(dim=1) + 1e-8).reshape(-1).contiguous()
    y = (t / nrm.view(N, 1, D, H, W)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    out = torch.empty_like(gp)
    gbuf = torch.zeros(16 * 1025, device='cuda')
    ops_train.ring_multi(gp, packs[:1].contiguous(), he, [(out, None, True)], extra=_lib.LF_RING_EX_PREV, e0=y, e1=nrm, o2=gbuf)
    gy, _ = ops.conv3d_c16_ring_bf16_io(gp, packs[0], None, he, 0, 1, out_bf16=True)
    want, gb = ops_train.epilogue_bwd_c16(gy, y, nrm, LF_EPI_LRELU | LF_EPI_PIXELNORM, True)
    _close_bf16(out, want)
    torch.testing.assert_close(gbuf[:16], gb, atol=1e-4 * float(gb.abs().max()) + 1e-6, rtol=2e-3)


def test_block_chain_equals_separate_epilogue_backward():
    """A Block of two 16 -> 16 layers under the bf16 policy: conv2's data gradient applies conv1's epilogue backward
    (ops_train.CHAIN_EPILOGUE) -- same output, same gradients as with conv1's own lf_epilogue_bwd_c16 pass."""
    from latentfusion_amd import ops, ops_train
    from latentfusion_amd.modules.blocks import Block
    torch.manual_seed(5)
    blk = Block(16, 16).cuda()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.2)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(2, 16, 12, 16, 32, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d).to(torch.bfloat16)
    gout = torch.randn(2, 16, 12, 16, 32, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    res = []
    for chain in (True, False, True):
        ops_train.CHAIN_EPILOGUE = chain
        try:
            blk.zero_grad()
            x = x0.clone().requires_grad_(True)
            with ops.autocast(True):
                y = blk(x)
            (y.float() * gout).sum().backward()
        finally:
            ops_train.CHAIN_EPILOGUE = True
        res.append((y.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
    a, b, c = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    _close_bf16(a[1], b[1], ulps=2.02)
    for k in a[2]:
        assert torch.equal(a[2][k], c[2][k])
        cos = torch.nn.functional.cosine_similarity(a[2][k].reshape(1, -1).double(), b[2][k].reshape(1, -1).double()).item()
        assert cos > 0.99999, (k, cos)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
@pytest.mark.parametrize('x_bf16', [True, False])
def test_block_epilogue_form_is_bit_identical(N, D, H, W, x_bf16):
    """LF_RING_EX_BLOCK == lf_conv3d_c16_ring_bf16_io with LeakyReLU + PixelNorm and round_out = 1 (activation and norms)."""
    from latentfusion_amd import _lib, ops, ops_train
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    x, w, packs, he, g = _setup(N, D, H, W, seed=8, x_bf16=x_bf16)
    bias = (torch.randn(16, generator=g) * 0.3).cuda()
    y = ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    nrm = torch.empty(N * D * H * W, device='cuda')
    ops_train.ring_multi(x, packs[:1].contiguous(), he, [(y, None, True)], extra=_lib.LF_RING_EX_BLOCK, e0=bias, o2=nrm)
    wy, wn = ops.conv3d_c16_ring_bf16_io(x, packs[0], bias, he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 1, out_bf16=True)
    assert torch.equal(y, wy) and torch.equal(nrm, wn)


@pytest.mark.parametrize('N,D,H,W', SHAPES)
def test_bf16_state_copy_forms(N, D, H, W):
    """lf_conv3d_c16_ring_blend writes the new state also rounded to bf16; the gate convolutions staged from that copy (reset gate:
    r h still from the fp32 state, e0) give bit for bit what they give staged from the fp32 state."""
    from latentfusion_amd import _lib, ops, ops_train
    rh, w, packs, he, g = _setup(N, D, H, W, seed=11)
    oz, upre = _vol(g, N, D, H, W, torch.bfloat16), _vol(g, N, D, H, W, torch.bfloat16, 2.0)
    h = _vol(g, N, D, H, W, torch.float32)
    cand, cand2, h16 = oz.clone(), oz.clone(), ops.empty_cl16((N, 16, D, H, W), 'cuda', True)
    hn, hn2 = ops.empty_cl16((N, 16, D, H, W), 'cuda', False), ops.empty_cl16((N, 16, D, H, W), 'cuda', False)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p0 = packs[:1].contiguous()
    _lib.check(L.lf_conv3d_c16_ring_blend(rh.data_ptr(), p0.data_ptr(), cand.data_ptr(), cand.data_ptr(), h.data_ptr(), upre.data_ptr(),
                                          hn.data_ptr(), h16.data_ptr(), N, D, H, W, he, st), 'blend')
    ops_train.ring_multi(rh, p0, he, [(cand2, cand2, False)], extra=_lib.LF_RING_EX_BLEND, e0=h, e1=upre, o2=hn2)
    assert torch.equal(cand, cand2) and torch.equal(hn, hn2) and torch.equal(h16, hn.to(torch.bfloat16))
    # the next step's gates staged from the bf16 copy
    uz, rz = _vol(g, N, D, H, W, torch.bfloat16), _vol(g, N, D, H, W, torch.bfloat16)
    e = lambda: ops.empty_cl16((N, 16, D, H, W), 'cuda', True)     # noqa: E731
    ua, ub, ra, rb, rha, rhb = e(), e(), e(), e(), e(), e()
    p1 = packs[1:].contiguous()
    ops_train.ring_multi(hn, p0, he, [(ua, uz, False)])
    ops_train.ring_multi(h16, p0, he, [(ub, uz, False)])
    ops_train.ring_multi(hn, p1, he, [(ra, rz, False)], extra=_lib.LF_RING_EX_RH, o2=rha)
    ops_train.ring_multi(h16, p1, he, [(rb, rz, False)], extra=_lib.LF_RING_EX_RH, e0=hn, o2=rhb)
    assert torch.equal(ua, ub) and torch.equal(ra, rb) and torch.equal(rha, rhb)
