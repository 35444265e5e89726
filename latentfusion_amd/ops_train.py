"""Operators of the TRAINING step (BASELINE cfg 5; reference tools/train/train_reconstruct.py:421-535) on top of ops.py:
weight / bias gradients, the 16 -> 16 layers and resampler under the bf16 autocast + storage policy, the ConvGRU fuser as one
autograd node, the fused lift.  Reached through `ops.<name>` (ops.__getattr__) and through ops' dispatchers (conv3x3, conv1x1,
resample_*, lift), which pick these forms when a gradient is wanted / the policy is on."""
import math  # noqa: F401

import torch

from . import _lib, ops
from ._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, LF_MAP_C2O, LF_MAP_COEFS, LF_MAP_O2C, check  # noqa: F401
from .ops import (PN_EPS, SLOPE, _ac_in, _cached, _conv1x1_raw, _dense_views, _f32, _lift_permute, _pk, _ptr, _req, _stream, _timed, _wsrc, autocast, cl, conv3d_c16_ring_bf16, conv3d_c16_ring_bf16_io, conv3d_c16_wino, empty_cl, empty_cl16, he_constant, pack_conv1x1, pack_conv3d_c16_ring_bf16, pack_conv3d_c16_wino, round_bf16, storage_bf16)  # noqa: F401


# Weight-gradient launches of the training step on a second HIP stream (round 5).  The 16 -> 16 kernels of the step are bound by
# latency at two resident workgroups per CU, not by HBM or the matrix pipe, and a CU has room for two weight-gradient
# workgroups (62 KB of LDS each) plus one ring-convolution workgroup (35 KB): the weight gradient of a layer -- nothing in the
# backward chain waits for it -- runs beside the data-gradient convolution of the same layer instead of in front of it.
WGRAD_STREAM = True
# ---- round-6 forms of the training step, each with its A/B switch (module attribute; LF_<NAME>=0/1 in the environment sets the
# default for a fresh process: tools/train_probe.py A/Bs, profiles/r06_*) --------------------------------------------------
import os as _os


def _env(name, default):
    return type(default)(int(_os.environ.get('LF_' + name, int(default))))


# the ConvGRU recurrence on the ring kernels with fused epilogues (csrc/conv_gru.hip); False: round 5's one-output kernels + stage
# kernels (kept: fp32 storage / no autocast take that path anyway)
GRU_RING = _env('GRU_RING', True)
# 1: one-output launches on the sequential chain (two workgroups per CU overlap their phases), everything that does not depend
# on the state batched over the views before / after the loop; 2: two outputs per staged halo (one 8-wave workgroup per CU --
# measured slower: its waves run in lock-step, profiles/r06_gru_ring_probe.json)
GRU_RING_GROUPS = _env('GRU_RING_GROUPS', 1)
# steps per multi-volume weight-gradient launch of the recurrence on the side stream (0: one launch over all steps, after the chain)
GRU_WGRAD_CHUNK = _env('GRU_WGRAD_CHUNK', 16)
# a bf16 copy of the recurrent state for the staged operands: bit-identical, measured -0.2 ms for +1.9 GB: off
GRU_STATE_BF16 = _env('GRU_STATE_BF16', False)
RING_BLOCK_FWD = _env('RING_BLOCK_FWD', True)          # forward Block steps on ring_multi (LF_RING_EX_BLOCK, bit-identical)
RING_DGRAD = _env('RING_DGRAD', True)                  # data gradients of the 16 -> 16 layers on ring_multi (bit-identical)
CHAIN_EPILOGUE = _env('CHAIN_EPILOGUE', True)          # a Block's conv2 data gradient applies conv1's epilogue backward (LF_RING_EX_PREV)
LIFT_MFMA = _env('LIFT_MFMA', True)                    # the 16-channel lift as one MFMA kernel each way (csrc/lift_mfma.hip)
PW16 = _env('PW16', True)                              # 1x1x1 16 -> 16 layers on the pointwise MFMA kernels (csrc/pw16.hip), not on the centre tap of the ring kernels


_SIDE = {}


def side_stream(device):
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st


class _SplitViews(torch.autograd.Function):
    """(1, V, C, D, H, W) -> V tensors (1, C, D, H, W), like `unbind(1)`, for the fusers' walk over the views.  The point is
    the backward: the V gradients are copied ONCE into a (V, C, D, H, W) channels-last block and handed back as its
    (1, V, ...) view -- `unbind` / `z[:, i]` give a standard-layout stack that the view reshape and the producing kernels'
    channels-last conversion then copy two more times (1 GB each at 8 x 128^3 x 16)."""

    @staticmethod
    def forward(ctx, z):
        ctx.shape = tuple(z.shape)
        return tuple(z[:, i] for i in range(z.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        B, V = ctx.shape[0], ctx.shape[1]
        ref = next(g for g in grads if g is not None)
        block = empty_cl((B * V,) + ctx.shape[2:], ref.device) if len(ctx.shape) == 6 else \
            torch.empty((B * V,) + ctx.shape[2:], device=ref.device, dtype=ref.dtype, memory_format=torch.channels_last)
        blk = block.view(B, V, *ctx.shape[2:])
        for i, g in enumerate(grads):
            if g is None:
                blk[:, i].zero_()
            else:
                blk[:, i].copy_(g)
        return blk


def split_views(z):
    """The views of a (B, V, C, [D,] H, W) stack as a tuple (autograd-friendly `unbind(1)`, see _SplitViews)."""
    if z.dim() in (5, 6) and z.is_cuda and z.shape[0] == 1 and torch.is_grad_enabled() and z.requires_grad:
        return _SplitViews.apply(z)
    return z.unbind(1)


class _ResampleAC(torch.autograd.Function):
    """The 16-channel resampler under the bf16 storage policy: source in fp32 or bf16, destination bf16; the volume gradient
    (deterministic fixed-point splat) comes back in the source's storage type.  No camera gradient (training path)."""

    @staticmethod
    def forward(ctx, vol, coef, kind):
        L = _lib.lib()
        n = coef.shape[0]
        vol_n = 1 if (vol.shape[0] == 1 or vol.stride(0) == 0) else vol.shape[0]
        if vol_n not in (1, n):
            raise ValueError('batch dimension of the volume and the cameras must match')
        v = cl(vol[:1] if vol_n == 1 else vol)
        _, C, D, H, W = v.shape
        out = empty_cl16((n, 16, D, H, W), v.device, True)
        cf = torch.zeros(n, LF_MAP_COEFS, device=v.device, dtype=torch.float32)
        cf[:, :coef.shape[1]] = coef.detach().float()
        io = (1 if v.dtype == torch.bfloat16 else 0) | 2
        with _timed('resample_fwd', f'{kind}:{n}:io{io}'):
            check(L.lf_resample3d_fwd_io(_ptr(v), vol_n, _ptr(cf), kind, _ptr(out), n, D, H, W, io, _stream()), 'lf_resample3d_fwd_io')
        ctx.save_for_backward(cf)
        ctx.meta = (kind, vol_n, tuple(vol.shape), vol.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        cf, = ctx.saved_tensors
        kind, vol_n, vshape, vdtype = ctx.meta
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('camera gradients run on the fp32 resampler (the pose loop); not under the training storage policy')
        if not ctx.needs_input_grad[0]:
            return None, None, None
        g = cl(gout)
        n, C, D, H, W = g.shape
        gv = empty_cl16((vol_n, 16, D, H, W), g.device, vdtype == torch.bfloat16)
        nb = L.lf_resample3d_bwd_vol_det_io_scratch_bytes(vol_n, n, D, H, W)
        scr = torch.empty(nb // 8 + 1, device=g.device, dtype=torch.int64)
        io = (1 if g.dtype == torch.bfloat16 else 0) | (2 if gv.dtype == torch.bfloat16 else 0)
        with _timed('resample_bwd_vol', f'{kind}:{n}:io{io}'):
            check(L.lf_resample3d_bwd_vol_det_io(_ptr(g), _ptr(cf), kind, _ptr(gv), vol_n, _ptr(scr, True), scr.numel() * 8, n, D, H, W, io,
                                                 _stream()), 'lf_resample3d_bwd_vol_det_io')
        if gv.shape[0] == vshape[0]:
            return gv, None, None
        gvol = torch.zeros(vshape, device=g.device, dtype=gv.dtype)     # expanded (stride-0) input: see _Resample.backward
        gvol[0] = gv[0]
        return gvol, None, None


def _resample_ac_ok(vol, coef):
    return (vol.dim() == 5 and vol.shape[1] == 16 and vol.is_cuda and not coef.requires_grad and ops.DETERMINISTIC_SPLAT
            and (vol.dtype == torch.bfloat16 or storage_bf16()) and vol.shape[2] * vol.shape[3] * vol.shape[4] * 64 < 0xffffffff
            and max(vol.shape[2:]) < 0x7fff)


def bias_grad(gp, dims):
    """Column sums of the pre-activation gradient = d(loss)/d(bias) (lf_conv_bwd_weight with x == NULL).
    gp: channels-last (N,C,[D,]H,W), or a plain [rows][C] matrix for dims = 0."""
    L = _lib.lib()
    if dims == 0:
        rows, cout = gp.shape[0], gp.shape[1]
        N, D, H, W = 1, 1, 1, rows
    else:
        N, cout = gp.shape[0], gp.shape[1]
        D, H, W = (gp.shape[2:] if dims == 3 else (1,) + tuple(gp.shape[2:]))
    gb = torch.empty(1, cout, 1, device=gp.device, dtype=torch.float32)
    nbytes = L.lf_conv_bwd_weight_scratch_bytes(0, N, D, H, W, 0, cout)
    scratch = torch.empty(nbytes // 4 + 1, device=gp.device, dtype=torch.float32)
    check(L.lf_conv_bwd_weight(None, _ptr(gp), _ptr(gb), _ptr(scratch, True), scratch.numel() * 4, 0, N, D, H, W, 0, cout,
                               1.0, _stream()), 'lf_conv_bwd_weight')
    return gb.reshape(cout)


def _wgrad_bf16_ok(gp, dims, cin, cout):
    """Shapes lf_conv_bwd_weight_bf16 takes (the rest stays on lf_conv_bwd_weight with pre-rounded operands)."""
    if not (dims == 3 and cin == 16 and cout == 16 and gp.numel() // gp.shape[1] >= 8192):
        return False
    D, H, W = gp.shape[2:]
    return D * H * W * 64 < 2 ** 31 and (D + 3) * H * W * 64 <= 0xffffffff


def conv_bwd_weight(x, gp, dims, cin, he, want_bias=True, bf16=None):
    """Weight and bias gradients of y = conv(x, W) * he + b from the pre-activation gradient `gp`
    (lf_conv_bwd_weight).  x, gp: channels-last (N,C,[D,]H,W), or plain [rows][C] matrices for dims = 0.
    Returns (gw [taps][Cout][Cin], gb [Cout] or None when want_bias is False).  bf16: both operands are bf16 values (None:
    the ambient autocast policy) -- 3-D 16 -> 16 layers then run on the bf16 MFMA (lf_conv_bwd_weight_bf16)."""
    L = _lib.lib()
    if bf16 is None:
        bf16 = ops.AUTOCAST is not None
    if dims == 3 and gp.shape[1] == 16 and cin > 16:
        # the LDS-staged 16 -> 16 kernel is ~4x faster than the generic one even with the slice copies:
        # the weight gradient of a wider input is the concatenation of the gradients of its channel chunks
        parts, gb = [], None
        for c0 in range(0, cin, 16):
            c1 = min(c0 + 16, cin)
            if c1 - c0 < 16:
                # ragged last chunk (the 3 coordinate channels of the 35-channel GRU gates): zero-pad it to 16 so it
                # also takes the LDS-staged kernel (0.25 ms instead of 6.6 ms on the generic one at 128^3)
                xc = empty_cl((x.shape[0], 16) + tuple(x.shape[2:]), x.device).zero_()
                xc[:, :c1 - c0] = x[:, c0:c1]
                gw_c, gb_c = conv_bwd_weight(xc, gp, dims, 16, he, want_bias and c0 == 0, bf16)
                gw_c = gw_c[:, :, :c1 - c0]
            else:
                gw_c, gb_c = conv_bwd_weight(cl(x[:, c0:c1]), gp, dims, 16, he, want_bias and c0 == 0, bf16)
            gb = gb_c if c0 == 0 else gb                       # the bias gradient (column sums of gp) once, not per chunk
            parts.append(gw_c)
        return torch.cat(parts, dim=2), gb
    if dims == 0:
        rows, cout = gp.shape[0], gp.shape[1]
        N, D, H, W = 1, 1, 1, rows
    else:
        N, cout = gp.shape[0], gp.shape[1]
        D, H, W = (gp.shape[2:] if dims == 3 else (1,) + tuple(gp.shape[2:]))
    taps = {0: 1, 2: 9, 3: 27}[dims]
    gw = torch.empty(taps, cout, cin, device=gp.device, dtype=torch.float32)
    gb = torch.empty(1, cout, 1, device=gp.device, dtype=torch.float32) if want_bias else None
    nbytes = max(L.lf_conv_bwd_weight_scratch_bytes(dims, N, D, H, W, cin, cout),
                 L.lf_conv_bwd_weight_scratch_bytes(0, N, D, H, W, 0, cout))
    scratch = torch.empty(nbytes // 4 + 1, device=gp.device, dtype=torch.float32)
    if bf16 and _wgrad_bf16_ok(gp, dims, cin, cout):
        # autocast: both operands are bf16 values -- the bf16 MFMA forms the same exact products 8x faster
        check(L.lf_conv_bwd_weight_bf16(_ptr(x), _ptr(gp), _ptr(gw), _ptr(scratch, True), scratch.numel() * 4, dims, N, D, H, W, cin, cout,
                                        he, _stream()), 'lf_conv_bwd_weight_bf16')
    else:
        check(L.lf_conv_bwd_weight(_ptr(x), _ptr(gp), _ptr(gw), _ptr(scratch, True), scratch.numel() * 4, dims, N, D, H, W, cin, cout,
                                   he, _stream()), 'lf_conv_bwd_weight')
    if not want_bias:
        return gw, None
    check(L.lf_conv_bwd_weight(None, _ptr(gp), _ptr(gb), _ptr(scratch, True), scratch.numel() * 4, 0, N, D, H, W, 0, cout,
                               1.0, _stream()), 'lf_conv_bwd_weight')
    return gw, gb.reshape(cout)


def pack_center_tap(weight):
    """(16,16,1,1,1) -> (16,16,3,3,3) with the weights on the centre tap: a pointwise 16 -> 16 layer on the 3x3x3 kernels."""
    w3 = weight.new_zeros(16, 16, 3, 3, 3)
    w3[:, :, 1, 1, 1] = weight.reshape(16, 16)
    return w3


def epilogue_bwd_c16(gy, y, norm, flags, want_bias, out_bf16=True):
    """lf_epilogue_bwd_c16 on channels-last (N,16,D,H,W) tensors in fp32 / bf16 storage: (gp, gbias or None)."""
    L = _lib.lib()
    rows = gy.numel() // 16
    gp = torch.empty_like(gy, dtype=torch.bfloat16 if out_bf16 else torch.float32, memory_format=torch.preserve_format)
    gb = torch.empty(16, device=gy.device, dtype=torch.float32) if want_bias else None
    nb = L.lf_epilogue_bwd_c16_scratch_bytes(rows) if want_bias else 0
    scr = torch.empty(nb // 4 + 1, device=gy.device, dtype=torch.float32) if want_bias else None
    io = (1 if gy.dtype == torch.bfloat16 else 0) | (2 if (y is not None and y.dtype == torch.bfloat16) else 0) | (4 if out_bf16 else 0)
    with _timed('epilogue_bwd_c16', f'{rows}:io{io}'):
        check(L.lf_epilogue_bwd_c16(_ptr(gy), _ptr(y) if y is not None else None, _ptr(norm) if norm is not None else None, _ptr(gp),
                                    _ptr(gb) if gb is not None else None, _ptr(scr, True) if scr is not None else None,
                                    scr.numel() * 4 if scr is not None else 0, rows, flags, SLOPE, io, _stream()), 'lf_epilogue_bwd_c16')
    return gp, gb


class _Conv16AC(torch.autograd.Function):
    """A 3-D 16 -> 16 layer (3x3x3, or 1x1x1 on the centre tap) of the training step under the bf16 autocast + storage policy:
    x (fp32 or bf16 storage) -> epilogue(conv(x, W) * he + b) in bf16 storage on lf_conv3d_c16_ring_bf16_io.  Backward: one
    pass for LeakyReLU' / PixelNorm' and the bias gradient (lf_epilogue_bwd_c16), the data gradient on the ring kernel
    in the input's storage type, the weight gradient on lf_conv_bwd_weight_bf16_io -- every volume moves as bf16.

    chain (round 6): the caller states that x is the output of ANOTHER _Conv16AC layer with LeakyReLU + PixelNorm that nothing else
    consumes (the two convolutions of a Block, modules/blocks.py:152-158).  This layer's data gradient then applies that layer's
    epilogue backward in its store and sums its bias gradient (LF_RING_EX_PREV): the producer receives its PRE-activation
    gradient and skips its own lf_epilogue_bwd_c16 pass (1.28 ms and 4.3 GB per Block at 32 views)."""

    @staticmethod
    def forward(ctx, x, weight, bias, flags, chain=False):
        _req(weight, 'weight')
        link = getattr(x, '_lf_link', None) if (chain and CHAIN_EPILOGUE) else None
        x = cl(x)
        he = he_constant(weight)
        one = weight.shape[2] == 1
        w3 = (lambda t: pack_center_tap(t)) if one else (lambda t: t)
        ctx.pw = bool(one and PW16 and flags in (0, LF_EPI_LRELU))
        if ctx.pw:
            # (round 6) a pointwise layer is one MFMA per 16 voxels, streamed: same products, same roundings as the ring kernel
            L = _lib.lib()
            wq = _pk(weight, 'p1f', lambda t: t.detach().reshape(16, 16).to(torch.bfloat16).contiguous())
            y = empty_cl16(tuple(x.shape), x.device, True)
            norm = None
            b_ = bias.detach() if bias is not None else None
            with _timed('pw16_fwd', f'{x.shape[0]}:{x.dtype == torch.bfloat16}'):
                check(L.lf_pw16_fwd(_ptr(x), 1 if x.dtype == torch.bfloat16 else 0, _ptr(wq), _ptr(b_) if b_ is not None else None, he, flags,
                                    SLOPE, _ptr(y), x.numel() // 16, _stream()), 'lf_pw16_fwd')
        elif RING_BLOCK_FWD and flags == (LF_EPI_LRELU | LF_EPI_PIXELNORM) and x.shape[2] * x.shape[3] * x.shape[4] * 64 < 2 ** 31:
            # the same arithmetic on the one-group ring kernel with its epilogue at compile time (LF_RING_EX_BLOCK): bit-identical
            pack = _pk(weight, 'a3f', lambda t: pack_conv3d_c16_ring_bf16(w3(t)))
            y = empty_cl16(tuple(x.shape), x.device, True)
            norm = torch.empty(x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4], device=x.device, dtype=torch.float32)
            ring_multi(x, pack.reshape(1, 14, 16, 32), he, [(y, None, True)], extra=_lib.LF_RING_EX_BLOCK,
                       e0=bias.detach() if bias is not None else None, o2=norm)
        else:
            pack = _pk(weight, 'a3f', lambda t: pack_conv3d_c16_ring_bf16(w3(t)))
            y, norm = conv3d_c16_ring_bf16_io(x, pack, bias.detach() if bias is not None else None, he, flags, 1, out_bf16=True)
        ctx.flags, ctx.he, ctx.one = flags, he, one
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(y if flags else None, norm, weight, x if need_w else None)
        ctx.xdtype = x.dtype
        # the producer's side of the chain: what a consumer needs to run this layer's epilogue backward (its activation reaches
        # the consumer as that layer's input), and the box through which the consumer reports having done so
        ctx.box = None
        if CHAIN_EPILOGUE and flags == (LF_EPI_LRELU | LF_EPI_PIXELNORM) and not one:
            ctx.box = {'done': False, 'gb': None}
            y._lf_link = (norm, ctx.box)
        ctx.link = None
        if link is not None and x.dtype == torch.bfloat16 and need_w and not one and x.shape[2] * x.shape[3] * x.shape[4] * 64 < 2 ** 31:
            ctx.link = link                                       # (norm of the producer, its box)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm, w, x_saved = ctx.saved_tensors
        gy = cl(gy)
        want_b = ctx.needs_input_grad[2]
        pw_bwd = ctx.pw and ctx.flags == 0 and gy.dtype == torch.bfloat16 and ctx.needs_input_grad[0]
        if pw_bwd:
            gp, gb = gy, None                                     # no activation: the pre-activation gradient IS gy; the bias sums ride in lf_pw16_bwd
        elif ctx.box is not None and ctx.box['done']:
            gp, gb = gy, ctx.box['gb']                            # the consumer's data gradient already applied this layer's epilogue backward
            if gp.dtype != torch.bfloat16:
                gp = gp.to(torch.bfloat16)
        elif ctx.flags == 0 and not want_b and gy.dtype == torch.bfloat16:
            gp, gb = gy, None
        else:
            gp, gb = epilogue_bwd_c16(gy, y, norm, ctx.flags, want_b)
        gx = gw = gwt = None
        w3 = (lambda t: pack_center_tap(t)) if ctx.one else (lambda t: t)
        with autocast(True):
            fast = ctx.needs_input_grad[1] and _wgrad_bf16_ok(gp, 3, 16, 16)
            side = done = None
            if fast:
                # the weight gradient first, on the side stream (WGRAD_STREAM): it then runs beside the data gradient below
                L = _lib.lib()
                N, _, D, H, W = gp.shape
                gwt = torch.empty(27, 16, 16, device=gp.device, dtype=torch.float32)
                nb = L.lf_conv_bwd_weight_scratch_bytes(3, N, D, H, W, 16, 16)
                scr = torch.empty(nb // 4 + 1, device=gp.device, dtype=torch.float32)
                io = (1 if x_saved.dtype == torch.bfloat16 else 0) | 2
                main = torch.cuda.current_stream()
                if WGRAD_STREAM and ops.KERNEL_TIMER is None:
                    side = side_stream(gp.device)
                    side.wait_stream(main)                    # gp (and the allocations above) are ready
                with torch.cuda.stream(side) if side is not None else _timed('wgrad3d_c16_bf16', f'{N}:io{io}'):
                    check(L.lf_conv_bwd_weight_bf16_io(_ptr(x_saved), _ptr(gp), _ptr(gwt), _ptr(scr, True), scr.numel() * 4, 3, N, D, H, W, 16, 16,
                                                       ctx.he, io, _stream()), 'lf_conv_bwd_weight_bf16_io')
                    if side is not None:
                        done = torch.cuda.Event()
                        done.record(side)
                        for t in (x_saved, gp, gwt, scr):
                            t.record_stream(side)
            if ctx.needs_input_grad[0] and ctx.pw and gp.dtype == torch.bfloat16:
                # pointwise data gradient (+ this layer's bias gradient when nothing ran before it): one streamed pass
                L = _lib.lib()
                wtq = _pk(w, 'p1b', lambda t: t.detach().reshape(16, 16).t().to(torch.bfloat16).contiguous())
                rows = gp.numel() // 16
                out16 = ctx.xdtype == torch.bfloat16
                gx = torch.empty_like(gp, dtype=torch.bfloat16 if out16 else torch.float32, memory_format=torch.preserve_format)
                need_gb = pw_bwd and want_b
                if need_gb:
                    gb = torch.empty(16, device=gp.device, dtype=torch.float32)
                    nb = L.lf_pw16_bwd_scratch_bytes(rows)
                    scr = torch.empty(nb // 4 + 4, device=gp.device, dtype=torch.float32)
                with _timed('pw16_bwd', f'{gp.shape[0]}:{out16}'):
                    check(L.lf_pw16_bwd(_ptr(gp), _ptr(wtq), ctx.he, _ptr(gx), 1 if out16 else 0, _ptr(gb) if need_gb else None,
                                        _ptr(scr, True) if need_gb else None, scr.numel() * 4 if need_gb else 0, rows, _stream()), 'lf_pw16_bwd')
            elif ctx.needs_input_grad[0]:
                pack_t = _pk(w, 'a3b', lambda t: pack_conv3d_c16_ring_bf16(w3(t), transpose=True))
                if ctx.link is not None and gp.dtype == torch.bfloat16:
                    # this layer's data gradient + the PRODUCER's epilogue backward and bias sums in one launch: x_saved is the
                    # producer's activation
                    pnorm, pbox = ctx.link
                    gx = torch.empty_like(gp)
                    gbuf = torch.zeros(16 * 1025, device=gp.device, dtype=torch.float32)
                    ring_multi(gp, pack_t.reshape(1, 14, 16, 32), ctx.he, [(gx, None, True)], extra=_lib.LF_RING_EX_PREV, e0=x_saved, e1=pnorm,
                               o2=gbuf)
                    pbox['done'], pbox['gb'] = True, gbuf[:16]
                elif RING_DGRAD and gp.dtype == torch.bfloat16 and ctx.xdtype == torch.bfloat16:
                    # the same sums and roundings on the one-group ring kernel with a compile-time epilogue (csrc/conv_gru.hip):
                    # bit-identical, 32 instead of 53 us per 128^3 volume
                    gx = torch.empty_like(gp)
                    ring_multi(gp, pack_t.reshape(1, 14, 16, 32), ctx.he, [(gx, None, True)])
                else:
                    gx, _ = conv3d_c16_ring_bf16_io(gp, pack_t, None, ctx.he, 0, 1, out_bf16=ctx.xdtype == torch.bfloat16)
            if ctx.needs_input_grad[1]:
                if done is not None:
                    torch.cuda.current_stream().wait_event(done)
                if not fast:                                  # small volumes: the fp32-MFMA kernel on the same bf16 values
                    xs = x_saved if x_saved.dtype == torch.bfloat16 else round_bf16(x_saved)
                    gwt, _ = conv_bwd_weight(cl(xs.float()), cl(gp.float()), 3, 16, ctx.he, want_bias=False, bf16=False)
                if ctx.one:
                    gw = round_bf16(gwt[13].reshape(w.shape).contiguous())
                else:
                    gw = round_bf16(gwt.reshape(3, 3, 3, 16, 16).permute(3, 4, 0, 1, 2).contiguous())
        return gx, gw, gb, None, None


def _conv16_ac_ok(x, weight):
    return (storage_bf16() and x.dim() == 5 and weight.dim() == 5 and tuple(weight.shape[:2]) == (16, 16) and x.shape[1] == 16
            and weight.shape[2] in (1, 3) and x.is_cuda and (x.shape[2] * x.shape[3] * x.shape[4]) * 64 < 2 ** 31)


class _Conv3x3Sum16(torch.autograd.Function):
    """sum_p conv3d(parts[p], W[:, c0_p : c0_p + w_p]) * he (+ b) (+ addend) for a 16-channel output: a convolution over a
    channel concatenation evaluated WITHOUT the concatenation, as a sum of 16 -> 16 convolutions (the addend form of
    lf_conv3d_c16_wino / lf_conv3d_c16_ring_bf16), one per part.  `parts` are channels-last (N,16,D,H,W) tensors; part p
    stands for the input channels cols[p] = (c0_p, w_p) of W (narrower parts -- the 3 coordinate channels of the ConvGRU
    gates -- are zero-padded to 16 by the caller).  The parts need not cover W: a part that is the same in every call (those
    coordinates) is convolved ONCE, bias included, and handed to the other calls as `addend`; autograd then sums its
    gradient over the calls, and its weight / bias gradients are one launch instead of one per call.  Backward: per-part
    data gradients on the same kernel, weight gradients on the LDS-staged 16 -> 16 kernels, no slice copies."""

    @staticmethod
    def forward(ctx, weight, bias, cols, addend, *parts):
        _req(weight, 'weight')
        assert weight.dim() == 5 and weight.shape[0] == 16 and len(cols) == len(parts) >= 1
        assert all(0 <= c0 and 0 < wdt <= 16 and c0 + wdt <= weight.shape[1] for c0, wdt in cols)
        he = he_constant(weight)

        ctx.ac = ops.AUTOCAST is not None
        # autocast: the bf16 ring kernel (and the bf16 weight-gradient kernel) round their operands while staging them; only
        # volumes the latter does not take keep the rounding pass in front
        ctx.pre_round = ctx.ac and not _wgrad_bf16_ok(parts[0], 3, 16, 16)
        parts = tuple((round_bf16(cl(p)) if ctx.pre_round else cl(p)) for p in parts)

        def make():
            wd, packs = _wsrc(weight).detach(), []
            pack = pack_conv3d_c16_ring_bf16 if ctx.ac else pack_conv3d_c16_wino
            for c0, wdt in cols:
                wp = wd.new_zeros(16, 16, 3, 3, 3)
                wp[:, :wdt] = wd[:, c0:c0 + wdt]
                packs.append((pack(wp), pack(wp, transpose=True)))
            return packs
        packs = _cached(weight, 'sum16_' + '_'.join(f'{c0}+{wdt}' for c0, wdt in cols) + ('@ac' if ctx.ac else ''), make)
        y = cl(addend) if addend is not None else None
        for i, (p, (pf, _pt)) in enumerate(zip(parts, packs)):
            _req(p, 'part')
            b = bias.detach() if (bias is not None and i == 0) else None
            if b is not None and y is not None:                  # (the addend forms take no bias: fold it in beforehand)
                y = y + b.view(1, -1, 1, 1, 1)
                b = None
            if ctx.ac:
                y, _ = conv3d_c16_ring_bf16(p, pf, b, he, 0, 0, addend=y)
            else:
                prev = None if y is None else (y, None, _lib.LF_EPI_ADD)
                y, _ = conv3d_c16_wino(p, pf, b, he, 0, prev=prev)
        ctx.he, ctx.cols, ctx.packs = he, cols, packs
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(weight, *(parts if need_w else []))
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = cl(gy)
        gy_full = gy
        w, *saved_parts = ctx.saved_tensors
        if ctx.pre_round:
            gy = round_bf16(gy)
        gparts = []
        for i, (_pf, pt) in enumerate(ctx.packs):
            if not ctx.needs_input_grad[4 + i]:
                gparts.append(None)
            elif ctx.ac:
                gparts.append(conv3d_c16_ring_bf16(gy, pt, None, ctx.he, 0, 1)[0])     # (rounded like autocast's conv backward)
            else:
                gparts.append(conv3d_c16_wino(gy, pt, None, ctx.he, 0)[0])
        gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gwt = torch.zeros(27, 16, w.shape[1], device=gy.device, dtype=torch.float32)      # columns of other calls stay zero
            for i, (p, (c0, wdt)) in enumerate(zip(saved_parts, ctx.cols)):
                g_i, gb_i = conv_bwd_weight(p, gy, 3, 16, ctx.he, want_bias=(i == 0 and ctx.needs_input_grad[1] and not ctx.ac), bf16=ctx.ac)
                gb = gb_i if i == 0 else gb
                gwt[:, :, c0:c0 + wdt] = g_i[:, :, :wdt]
            gw = gwt.reshape(3, 3, 3, 16, w.shape[1]).permute(3, 4, 0, 1, 2).contiguous()
            if ctx.ac:
                gw = round_bf16(gw)
                if ctx.needs_input_grad[1]:
                    gb = bias_grad(gy_full, 3)
        return (gw if ctx.needs_input_grad[0] else None, gb if ctx.needs_input_grad[1] else None, None,
                gy_full if ctx.needs_input_grad[3] else None, *gparts)


def ring_multi(x, wpacks, he, outs, extra=0, e0=None, e1=None, o2=None, addend_per_sample=True):
    """lf_conv3d_c16_ring_multi: ONE staged (N,16,D,H,W) channels-last volume `x` (bf16 or fp32 storage) convolved with 1 or 2
    ring-kernel weight packs.  wpacks: the packs back to back ((ngroups,14,16,32) bf16); outs: per group (y, addend or None, round)
    -- y / addend fp32 or bf16 channels-last volumes (their dtypes select the storage flags), round: bf16(bf16(conv) * he) instead
    of fma(conv, he, addend).  extra / e0 / e1 / o2: the fused element-wise stage (LF_RING_EX_*, include/lf_hip.h)."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    ng = len(outs)
    assert wpacks.dtype == torch.bfloat16 and wpacks.numel() == ng * 14 * 512 and wpacks.is_contiguous()
    args = []
    for y, add, rnd in outs:
        fl = ((_lib.LF_RING_OUT_BF16 if y.dtype == torch.bfloat16 else 0) | (_lib.LF_RING_ROUND if rnd else 0)
              | (_lib.LF_RING_ADD_BF16 if (add is not None and add.dtype == torch.bfloat16) else 0))
        args += [_ptr(y), _ptr(add) if add is not None else None, fl]
    if ng == 1:
        args += [None, None, 0]
    fl0 = args[2] | (8 if outs[0][1] is not None else 0)           # group 0's epilogue form, as the kernel template sees it
    with _timed('conv3d_c16_ring_multi', f'{ng}:{extra}:{N}:{fl0}:{x.element_size()}'):
        check(L.lf_conv3d_c16_ring_multi(_ptr(x), int(x.dtype == torch.bfloat16), _ptr(wpacks), ng, *args, extra,
                                         _ptr(e0) if e0 is not None else None, _ptr(e1) if e1 is not None else None,
                                         _ptr(o2) if o2 is not None else None, N, D, H, W, he, int(bool(addend_per_sample)), _stream()),
              'lf_conv3d_c16_ring_multi')


class _GruGates(torch.autograd.Function):
    """(upre, rpre, h) -> (u = sigmoid(upre), rh = h * sigmoid(rpre)); one kernel each way (lf_gru_stage_a[_bwd])
    instead of five element-wise passes over the volume (modules/gru.py:37-40)."""

    @staticmethod
    def forward(ctx, upre, rpre, h):
        L = _lib.lib()
        upre, rpre, h = cl(upre), cl(rpre), cl(h)
        u, rh = torch.empty_like(h), torch.empty_like(h)
        C = h.shape[1]
        nvox = h.numel() // C
        check(L.lf_gru_stage_a(_ptr(upre), _ptr(rpre), C, _ptr(h), _ptr(u), _ptr(rh), nvox, C, C, 0, _stream()), 'lf_gru_stage_a')
        ctx.save_for_backward(u, rpre, h)
        return u, rh

    @staticmethod
    def backward(ctx, gu, grh):
        L = _lib.lib()
        u, rpre, h = ctx.saved_tensors
        gu, grh = cl(gu), cl(grh)
        gupre, grpre, gh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        check(L.lf_gru_stage_a_bwd(_ptr(gu), _ptr(grh), _ptr(u), _ptr(rpre), _ptr(h), _ptr(gupre), _ptr(grpre), _ptr(gh),
                                   h.numel(), _stream()), 'lf_gru_stage_a_bwd')
        return gupre, grpre, gh


class _GruBlend(torch.autograd.Function):
    """(h, u, cand) -> h (1 - u) + cand u  (modules/gru.py:42; lf_gru_stage_b[_bwd])."""

    @staticmethod
    def forward(ctx, h, u, cand):
        L = _lib.lib()
        h, u, cand = cl(h), cl(u), cl(cand)
        out = torch.empty_like(h)
        C = h.shape[1]
        check(L.lf_gru_stage_b(_ptr(h), _ptr(u), _ptr(cand), _ptr(out), None, h.numel() // C, C, C, 0, _stream()), 'lf_gru_stage_b')
        ctx.save_for_backward(h, u, cand)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        h, u, cand = ctx.saved_tensors
        g = cl(g)
        gh, gu, gc = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        check(L.lf_gru_stage_b_bwd(_ptr(g), _ptr(h), _ptr(u), _ptr(cand), _ptr(gh), _ptr(gu), _ptr(gc), h.numel(), _stream()),
              'lf_gru_stage_b_bwd')
        return gh, gu, gc


class _LstmCell(torch.autograd.Function):
    """(cc (N,4Ch,...), c) -> (h', c') of the ConvLSTM cell (modules/lstm.py:49-56; lf_lstm_cell_fwd/bwd): one kernel each
    way instead of four activations, three products and a split."""

    @staticmethod
    def forward(ctx, cc, c_cur):
        L = _lib.lib()
        cc, c_cur = cl(_req(cc, 'cc')), cl(_req(c_cur, 'c'))
        Ch = c_cur.shape[1]
        if cc.shape[1] != 4 * Ch:
            raise ValueError('lstm_cell: cc must hold 4 x hidden channels')
        h, cn = torch.empty_like(c_cur), torch.empty_like(c_cur)
        check(L.lf_lstm_cell_fwd(_ptr(cc), _ptr(c_cur), _ptr(h), _ptr(cn), c_cur.numel() // Ch, Ch, _stream()), 'lf_lstm_cell_fwd')
        ctx.save_for_backward(cc, c_cur)
        ctx.set_materialize_grads(False)
        return h, cn

    @staticmethod
    def backward(ctx, gh, gcn):
        L = _lib.lib()
        cc, c_cur = ctx.saved_tensors
        if gh is None and gcn is None:
            return None, None
        Ch = c_cur.shape[1]
        gh = cl(gh) if gh is not None else None
        gcn = cl(gcn) if gcn is not None else None
        gcc, gc = torch.empty_like(cc), torch.empty_like(c_cur)
        check(L.lf_lstm_cell_bwd(_ptr(cc), _ptr(c_cur), _ptr(gh) if gh is not None else None, _ptr(gcn) if gcn is not None else None,
                                 _ptr(gcc), _ptr(gc), c_cur.numel() // Ch, Ch, _stream()), 'lf_lstm_cell_bwd')
        return gcc, gc


class _GruFuse(torch.autograd.Function):
    """GRUFuser.forward for 16-channel volumes as ONE autograd node (recon/fusion.py:188-197 over modules/gru.py:30-43): the
    recurrence h_i = cell(cat(z_i, coords), h_{i-1}), h_0 = z_0, forward and backward sequenced explicitly.

    Compared with the per-gate autograd functions (_Conv3x3Sum16 / _GruGates / _GruBlend) this removes every ATen gradient
    accumulation over full volumes (the data gradients of a step chain through the addend form of the convolution, the
    gate gradients' sums over the views are kept by the stage kernels), the sigmoid outputs are recomputed instead of stored,
    and under the bf16 autocast policy the tensors that only half-precision convolutions produce / consume (gate
    pre-activations, h*r, candidate, gate gradients) live in bf16 storage: 37 instead of ~90 volume passes per view.
    z: (1,V,16,D,H,W) with dense channels-last views.  Returns h_V (1,16,D,H,W)."""

    @staticmethod
    def forward(ctx, z, c16, wu, bu, wr, br, wo, bo):
        L = _lib.lib()
        B, V = z.shape[0], z.shape[1]
        assert B == 1 and z.shape[2] == 16 and z.is_cuda and z.dtype in (torch.float32, torch.bfloat16)
        zz = _dense_views(z)                                      # (V,16,D,H,W) channels-last, fp32 or bf16 storage
        D, H, W = zz.shape[2:]
        ac = ops.AUTOCAST is not None
        if not ac:
            zz = _f32(zz)                                         # (a bf16-stored stack fused OUTSIDE the policy: the fp32 kernels read 64 B records)
        T16 = ac                                                  # bf16 storage of the once-per-step tensors
        he = he_constant(wu)
        gates = ((wu, bu), (wr, br), (wo, bo))

        def packs(w):
            def make():
                wd = _wsrc(w).detach()
                wc = wd.new_zeros(16, 16, 3, 3, 3)
                wc[:, :3] = wd[:, 16:19]
                pk = pack_conv3d_c16_ring_bf16 if ac else pack_conv3d_c16_wino
                return tuple((pk(t), pk(t, transpose=True)) for t in (wd[:, :16].contiguous(), wc, wd[:, 19:].contiguous()))
            return _cached(w, 'gru_fuse' + ('@ac' if ac else ''), make)
        pk = [packs(w) for w, _ in gates]                         # [gate][z | coords | state][fwd | transposed]

        def conv(x, pack, addend=None, bias=None, out=None, out16=False, rnd=0):
            if ac:
                return conv3d_c16_ring_bf16_io(x, pack, bias, he, 0, rnd if addend is None else 0, addend=addend, out=out,
                                               out_bf16=out16)[0]
            prev = None if addend is None else (addend, None, _lib.LF_EPI_ADD)
            return conv3d_c16_wino(x, pack, bias, he, 0, prev=prev, out=out)[0]
        n = zz[0].numel()
        s = _stream()
        base = [conv(c16, pk[k][1][0], bias=(b.detach() if b is not None else None)) for k, (_, b) in enumerate(gates)]
        ctx.ring = bool(GRU_RING and ac and V >= 2 and D * H * W * 64 < 2 ** 31)
        if ctx.ring:
            ctx.z32 = zz.dtype != torch.bfloat16
            z_in = zz
            if ctx.z32:                                           # (the convolutions round x to bf16 while staging anyway: same numbers)
                zz = zz.to(torch.bfloat16)
            # ---- round 6: the multi-output ring kernels (csrc/conv_gru.hip).  The per-step tensors are STACKED blocks
            # [V-1]: UP / RP / CA are first filled with the state-independent parts of the three gates (one launch over all
            # views: x feeds three gates), then overwritten in place by the pre-activations / the candidate of their step;
            # the backward overwrites them once more with the gate GRADIENTS, which the weight gradients then read as
            # multi-volume launches -- no per-step allocation, no element-wise stage launch in the forward at all.
            wz = (torch.stack((pk[0][0][0], pk[1][0][0])), pk[2][0][0].reshape(1, 14, 16, 32))     # packs back to back: [u | r], [o]
            wh = (torch.stack((pk[0][2][0], pk[1][2][0])), pk[2][2][0].reshape(1, 14, 16, 32))
            blk = lambda: empty_cl16((V - 1, 16, D, H, W), zz.device, True)      # noqa: E731
            UP, RP, CA, RH = blk(), blk(), blk(), blk()
            HS = empty_cl((V - 1, 16, D, H, W), zz.device)                       # h_0 .. h_{V-2} (fp32 state)
            HS[0:1].copy_(z_in[0:1])                              # (h_0 = view 0, un-rounded when the stack is fp32)
            # a bf16 copy of every state (round 6): what the gate convolutions and the weight gradients STAGE -- they round h to
            # bf16 while staging anyway, so reading the rounded copy is the same arithmetic at half the bytes; the fp32 state
            # stays what the element-wise stages (r h, the blend and their backward) use
            H16 = blk() if GRU_STATE_BF16 else None
            if H16 is not None:
                H16[0:1].copy_(z_in[0:1])
            if GRU_RING_GROUPS == 2:
                ring_multi(zz[1:], wz[0], he, [(UP, base[0], False), (RP, base[1], False)], addend_per_sample=False)
            else:
                for k_, blk_ in enumerate((UP, RP)):
                    ring_multi(zz[1:], pk[k_][0][0].reshape(1, 14, 16, 32), he, [(blk_, base[k_], False)], addend_per_sample=False)
            ring_multi(zz[1:], wz[1], he, [(CA, base[2], False)], addend_per_sample=False)
            out = empty_cl(HS[0:1].shape, zz.device)
            whu, whr = pk[0][2][0].reshape(1, 14, 16, 32), pk[1][2][0].reshape(1, 14, 16, 32)
            L = _lib.lib()
            for i in range(1, V):
                h, j = HS[i - 1:i], slice(i - 1, i)
                hx = H16[j] if H16 is not None else h            # the staged operand
                if GRU_RING_GROUPS == 2:
                    ring_multi(h, wh[0], he, [(UP[j], UP[j], False), (RP[j], RP[j], False)], extra=_lib.LF_RING_EX_RH, o2=RH[j])
                else:                                             # (two one-output launches: two workgroups per CU overlap their phases)
                    ring_multi(hx, whu, he, [(UP[j], UP[j], False)])
                    ring_multi(hx, whr, he, [(RP[j], RP[j], False)], extra=_lib.LF_RING_EX_RH, e0=h if H16 is not None else None, o2=RH[j])
                hn = HS[i:i + 1] if i < V - 1 else out
                with _timed('conv3d_c16_ring_multi', '1:2:1:11:2'):
                    check(L.lf_conv3d_c16_ring_blend(_ptr(RH[j]), _ptr(wh[1]), _ptr(CA[j]), _ptr(CA[j]), _ptr(h), _ptr(UP[j]), _ptr(hn),
                                                     _ptr(H16[i:i + 1]) if (H16 is not None and i < V - 1) else None, 1, D, H, W, he, _stream()),
                          'lf_conv3d_c16_ring_blend')
            ctx.ac, ctx.T16, ctx.he, ctx.pk = ac, T16, he, pk
            ctx.blocks = [UP, RP, CA, RH, HS, H16]
            ctx.zshape = tuple(z.shape)
            ctx.save_for_backward(zz, c16, wu, wr, wo)
            ctx.has_bias = tuple(b is not None for _, b in gates)
            return out
        hs, saved = [_f32(zz[0:1])], []                            # (the state is fp32; view 0 seeds it)
        for i in range(1, V):
            zi, h = zz[i:i + 1], hs[-1]
            pre = []
            for k in (0, 1):
                xk = conv(zi, pk[k][0][0], addend=base[k], out16=T16)
                pre.append(conv(h, pk[k][2][0], addend=xk, out16=T16))
                del xk
            upre, rpre = pre
            rh = empty_cl16(h.shape, h.device, T16)
            check(L.lf_gru_train_stage_a(_ptr(rpre), _ptr(h), _ptr(rh), n, int(T16), s), 'lf_gru_train_stage_a')
            xo = conv(zi, pk[2][0][0], addend=base[2], out16=T16)
            cand = conv(rh, pk[2][2][0], addend=xo, out16=T16)
            del xo
            hn = empty_cl(h.shape, h.device)
            check(L.lf_gru_train_stage_b(_ptr(h), _ptr(upre), _ptr(cand), _ptr(hn), n, int(T16), s), 'lf_gru_train_stage_b')
            saved.append((upre, rpre, rh, cand))
            hs.append(hn)
        ctx.ac, ctx.T16, ctx.he, ctx.pk = ac, T16, he, pk
        ctx.steps = saved
        ctx.hs = hs[1:-1]                                         # h_1 .. h_{V-2}
        ctx.h0 = hs[0] if zz.dtype != torch.float32 else None     # (fp32 storage: h_0 is a view of z, saved below)
        ctx.zshape = tuple(z.shape)
        ctx.save_for_backward(zz, c16, wu, wr, wo)
        ctx.has_bias = tuple(b is not None for _, b in gates)
        return hs[-1].clone() if V == 1 else hs[-1]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        zz, c16, wu, wr, wo = ctx.saved_tensors
        V = zz.shape[0]
        ac, T16, he, pk = ctx.ac, ctx.T16, ctx.he, ctx.pk
        dev = zz.device
        s = _stream()
        n = zz[0].numel()
        shape1 = (1,) + tuple(zz.shape[1:])
        D, H, W = zz.shape[2:]
        need_z = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[i] for i in (2, 3, 4, 5, 6, 7))
        g = cl(g.reshape(shape1))
        if ctx.ring:
            return _GruFuse._backward_ring(ctx, g, zz, c16, (wu, wr, wo), need_z, need_w)

        def conv(x, pack, addend=None, out=None, out16=False, rnd=0):
            if ac:
                return conv3d_c16_ring_bf16_io(x, pack, None, he, 0, rnd if addend is None else 0, addend=addend, out=out,
                                               out_bf16=out16)[0]
            prev = None if addend is None else (addend, None, _lib.LF_EPI_ADD)
            return conv3d_c16_wino(x, pack, None, he, 0, prev=prev, out=out)[0]
        gz = empty_cl16((V, 16, D, H, W), dev, zz.dtype == torch.bfloat16) if need_z else None
        acc = [empty_cl(shape1, dev).zero_() for _ in range(3)] if need_w else [None] * 3
        # weight-gradient blocks [step][gate][z | state][27][16][16], summed over the steps at the end (fixed order)
        gwb = torch.zeros(max(V - 1, 1), 3, 2, 27, 16, 16, device=dev, dtype=torch.float32) if need_w else None
        nbytes = max(L.lf_conv_bwd_weight_scratch_bytes(3, 1, D, H, W, 16, 16), L.lf_conv_bwd_weight_scratch_bytes(0, 1, D, H, W, 0, 16))
        scratch = torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32) if need_w else None
        fast = _wgrad_bf16_ok(zz[0:1], 3, 16, 16)

        def wgrad(x, gp, dst):
            if ac and fast:
                io = (1 if x.dtype == torch.bfloat16 else 0) | (2 if gp.dtype == torch.bfloat16 else 0)
                check(L.lf_conv_bwd_weight_bf16_io(_ptr(x), _ptr(gp), _ptr(dst), _ptr(scratch, True), scratch.numel() * 4, 3, 1, D, H, W,
                                                   16, 16, he, io, _stream()), 'lf_conv_bwd_weight_bf16_io')
            else:
                xf = round_bf16(x.float()) if ac else x
                gf = round_bf16(gp.float()) if ac else gp
                check(L.lf_conv_bwd_weight(_ptr(xf), _ptr(gf), _ptr(dst), _ptr(scratch, True), scratch.numel() * 4, 3, 1, D, H, W, 16, 16,
                                           he, _stream()), 'lf_conv_bwd_weight')
        gh1 = empty_cl(shape1, dev)
        gh12 = empty_cl(shape1, dev)
        grh = empty_cl16(shape1, dev, T16)
        # the six weight gradients of a step run on the side stream (WGRAD_STREAM) beside the data-gradient chain; the gate
        # gradients they read are double-buffered, a buffer is rewritten only after the launches that read it have finished
        main = torch.cuda.current_stream()
        side = side_stream(dev) if (WGRAD_STREAM and need_w and ops.KERNEL_TIMER is None) else None
        gbuf = [tuple(empty_cl16(shape1, dev, T16) for _ in range(3)) for _ in range(2 if side is not None else 1)]
        gdone = [None, None]
        if side is not None:
            side.wait_stream(main)
            for t in (gwb, scratch, zz) + tuple(b for bb in gbuf for b in bb):
                t.record_stream(side)
        steps, hs = ctx.steps, ctx.hs
        for i in range(V - 1, 0, -1):
            gupre, gc, grpre = gbuf[i & 1 if side is not None else 0]
            if gdone[i & 1] is not None:
                main.wait_event(gdone[i & 1])
            if steps[i - 1] is None:
                raise RuntimeError('the fused GRU recurrence frees its activations during backward: a second backward through '
                                   'the same graph is not supported')
            upre, rpre, rh, cand = steps[i - 1]
            h = (ctx.h0 if ctx.h0 is not None else zz[0:1]) if i == 1 else hs[i - 2]
            zi = zz[i:i + 1]
            check(L.lf_gru_train_stage_b_bwd(_ptr(g), _ptr(h), _ptr(upre), _ptr(cand), _ptr(gh1), _ptr(gupre), _ptr(gc),
                                             _ptr(acc[0]) if need_w else None, _ptr(acc[2]) if need_w else None, n, int(T16), s),
                  'lf_gru_train_stage_b_bwd')
            conv(gc, pk[2][2][1], out=grh, rnd=1)
            check(L.lf_gru_train_stage_a_bwd(_ptr(grh), _ptr(rpre), _ptr(h), _ptr(gh1), _ptr(grpre), _ptr(gh12),
                                             _ptr(acc[1]) if need_w else None, n, int(T16), s), 'lf_gru_train_stage_a_bwd')
            if side is not None:
                ready = torch.cuda.Event()
                ready.record(main)                                # the three gate gradients of this step are complete
                side.wait_event(ready)
                h.record_stream(side)
                rh.record_stream(side)
                with torch.cuda.stream(side):
                    for k, (xs, gp) in enumerate((((zi, h), gupre), ((zi, h), grpre), ((zi, rh), gc))):
                        wgrad(xs[0], gp, gwb[i - 1, k, 0])
                        wgrad(xs[1], gp, gwb[i - 1, k, 1])
                    gdone[i & 1] = torch.cuda.Event()
                    gdone[i & 1].record(side)
            if need_z:
                gzi = gz[i:i + 1]
                conv(gc, pk[2][0][1], out=gzi, rnd=1)
                conv(gupre, pk[0][0][1], addend=gzi, out=gzi)
                conv(grpre, pk[1][0][1], addend=gzi, out=gzi)
            gnext = empty_cl(shape1, dev)
            conv(gupre, pk[0][2][1], addend=gh12, out=gnext)
            conv(grpre, pk[1][2][1], addend=gnext, out=gnext)
            if need_w and side is None:
                for k, (xs, gp) in enumerate((((zi, h), gupre), ((zi, h), grpre), ((zi, rh), gc))):
                    wgrad(xs[0], gp, gwb[i - 1, k, 0])
                    wgrad(xs[1], gp, gwb[i - 1, k, 1])
            g = gnext
            steps[i - 1] = None                                   # the step's tensors are dead: free them as the walk goes
            if i >= 2:
                hs[i - 2] = None
        if side is not None:
            main.wait_stream(side)
        if need_z:
            gz[0:1].copy_(g)
        outs = [gz.view(ctx.zshape) if need_z else None, None]
        if need_w:
            gsum = gwb.sum(dim=0)                                 # [gate][z | state][27][16][16]
            for k, w in enumerate((wu, wr, wo)):
                gwt = torch.empty(27, 16, w.shape[1], device=dev, dtype=torch.float32)
                gwt[:, :, :16] = gsum[k, 0]
                gwt[:, :, 19:] = gsum[k, 1]
                gc_, _ = conv_bwd_weight(c16, acc[k], 3, 16, he, want_bias=False, bf16=ac)
                gwt[:, :, 16:19] = gc_[:, :, :3]
                gw = gwt.reshape(3, 3, 3, 16, w.shape[1]).permute(3, 4, 0, 1, 2).contiguous()
                if ac:
                    gw = round_bf16(gw)
                outs.append(gw if ctx.needs_input_grad[2 + 2 * k] else None)
                outs.append(bias_grad(acc[k], 3) if (ctx.has_bias[k] and ctx.needs_input_grad[3 + 2 * k]) else None)
        else:
            outs += [None] * 6
        return tuple(outs)


def _gru_backward_ring(ctx, g, zz, c16, ws, need_z, need_w):
    """Backward of the recurrence on the multi-output ring kernels (see the forward): per step ONE element-wise launch
    (lf_gru_train_stage_b_bwd, gate gradients written over the saved pre-activations) and three convolution launches --
        gc    -> (g_rh with the reset gate's backward in its epilogue, g_x)
        gupre -> (g_x +=, g_h = gh12 +)          grpre -> (g_x +=, g_h +=)
    (was: two stage kernels + six convolutions + six weight-gradient launches).  The weight gradients run afterwards over the
    STORED gate gradients as multi-volume launches, in chunks of GRU_WGRAD_CHUNK steps on the side stream beside the chain;
    the sums over the steps that the bias / coordinate-channel gradients need come from lf_sum_views_bf16 over the same blocks."""
    L = _lib.lib()
    if ctx.blocks is None:
        raise RuntimeError('the fused GRU recurrence overwrites its activations during backward: a second backward through '
                           'the same graph is not supported')
    UP, RP, CA, RH, HS, H16 = ctx.blocks
    HX = H16 if H16 is not None else HS                          # the state as the weight gradients stage it
    ctx.blocks = None
    V = zz.shape[0]
    he, pk = ctx.he, ctx.pk
    dev = zz.device
    D, H, W = zz.shape[2:]
    shape1 = (1,) + tuple(zz.shape[1:])
    n = zz[0].numel()
    s = _stream()
    wu, wr, wo = ws
    # transposed packs: [o: towards r h | towards x], [u: towards x | towards h], [r: towards x | towards h]
    tp = tuple(torch.stack(p) for p in ((pk[2][2][1], pk[2][0][1]), (pk[0][0][1], pk[0][2][1]), (pk[1][0][1], pk[1][2][1])))
    gz = empty_cl16((V, 16, D, H, W), dev, True)
    gh1, gh12 = empty_cl(shape1, dev), empty_cl(shape1, dev)
    gbuf = [empty_cl(shape1, dev), empty_cl(shape1, dev)]
    main = torch.cuda.current_stream()
    side = side_stream(dev) if (WGRAD_STREAM and need_w and ops.KERNEL_TIMER is None) else None
    chunks = []                                                   # (first step, one past the last step) in launch order
    hi = V
    while hi > 1:
        lo = max(1, hi - GRU_WGRAD_CHUNK) if GRU_WGRAD_CHUNK > 0 else 1
        chunks.append((lo, hi))
        hi = lo
    gwb = torch.zeros(len(chunks), 3, 2, 27, 16, 16, device=dev, dtype=torch.float32) if need_w else None
    acc = [empty_cl(shape1, dev) for _ in range(3)] if need_w else None
    nbytes = L.lf_conv_bwd_weight_scratch_bytes(3, V, D, H, W, 16, 16)
    scratch = torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32) if need_w else None     # (one: the launches are in order)
    if side is not None:
        side.wait_stream(main)                                    # (the allocations above)
        for t in [gwb, zz, UP, RP, CA, RH, HS, HX, scratch] + acc:
            t.record_stream(side)

    def weight_grads(ci, lo, hi):
        """The six weight gradients of steps lo .. hi-1 (their gate gradients are final) + this chunk's share of the sums."""
        nv = hi - lo
        j = slice(lo - 1, hi - 1)
        for k, (gp, xh) in enumerate(((UP, HX), (RP, HX), (CA, RH))):
            for q, x in enumerate((zz[lo:hi], xh[j])):
                io = (1 if x.dtype == torch.bfloat16 else 0) | 2
                check(L.lf_conv_bwd_weight_bf16_io(_ptr(x), _ptr(gp[j]), _ptr(gwb[ci, k, q]), _ptr(scratch, True), scratch.numel() * 4, 3, nv, D, H, W,
                                                   16, 16, he, io, _stream()), 'lf_conv_bwd_weight_bf16_io')
            check(L.lf_sum_views_bf16(_ptr(gp[j]), _ptr(acc[k]), n, nv, int(ci > 0), _stream()), 'lf_sum_views_bf16')

    def on_side(fn, *a):
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(main)                                    # the gate gradients the launches read are complete
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn(*a)
        else:
            fn(*a)
    two = GRU_RING_GROUPS == 2
    one = lambda p: p.reshape(1, 14, 16, 32)                      # noqa: E731
    for ci, (lo, hi) in enumerate(chunks):
        for i in range(hi - 1, lo - 1, -1):
            j = slice(i - 1, i)
            h = HS[j]
            gnext = gbuf[i & 1]
            check(L.lf_gru_train_stage_b_bwd(_ptr(g), _ptr(h), _ptr(UP[j]), _ptr(CA[j]), _ptr(gh1), _ptr(UP[j]), _ptr(CA[j]),
                                             None, None, n, 1, s), 'lf_gru_train_stage_b_bwd')
            if two:
                ring_multi(CA[j], tp[0], he, [(RP[j], RP[j], True), (gz[i:i + 1], None, True)], extra=_lib.LF_RING_EX_ABWD,
                           e0=h, e1=gh1, o2=gh12)
                ring_multi(UP[j], tp[1], he, [(gz[i:i + 1], gz[i:i + 1], False), (gnext, gh12, False)])
                ring_multi(RP[j], tp[2], he, [(gz[i:i + 1], gz[i:i + 1], False), (gnext, gnext, False)])
            else:
                # one-output launches on the chain (the state's gradient only); the gradients of the views follow the loop
                ring_multi(CA[j], one(tp[0][0]), he, [(RP[j], RP[j], True)], extra=_lib.LF_RING_EX_ABWD, e0=h, e1=gh1, o2=gh12)
                ring_multi(UP[j], one(tp[1][1]), he, [(gnext, gh12, False)])
                ring_multi(RP[j], one(tp[2][1]), he, [(gnext, gnext, False)])
            g = gnext
        if need_w:
            on_side(weight_grads, ci, lo, hi)
    if not two:
        if need_z:                                                # the views' gradients, from the stored gate gradients of all steps
            gzs = gz[1:]
            ring_multi(CA, one(tp[0][1]), he, [(gzs, None, True)])
            ring_multi(UP, one(tp[1][0]), he, [(gzs, gzs, False)])
            ring_multi(RP, one(tp[2][0]), he, [(gzs, gzs, False)])
    if side is not None:
        main.wait_stream(side)
    if ctx.z32:
        gz = gz.float()
    gz[0:1].copy_(g)
    outs = [gz.view(ctx.zshape) if need_z else None, None]
    if need_w:
        gsum = gwb.sum(dim=0)                                     # [gate][z | state][27][16][16], fixed order
        for k, w in enumerate((wu, wr, wo)):
            gwt = torch.empty(27, 16, w.shape[1], device=dev, dtype=torch.float32)
            gwt[:, :, :16] = gsum[k, 0]
            gwt[:, :, 19:] = gsum[k, 1]
            gc_, _ = conv_bwd_weight(c16, acc[k], 3, 16, he, want_bias=False, bf16=True)
            gwt[:, :, 16:19] = gc_[:, :, :3]
            gw = round_bf16(gwt.reshape(3, 3, 3, 16, w.shape[1]).permute(3, 4, 0, 1, 2).contiguous())
            outs.append(gw if ctx.needs_input_grad[2 + 2 * k] else None)
            outs.append(bias_grad(acc[k], 3) if (ctx.has_bias[k] and ctx.needs_input_grad[3 + 2 * k]) else None)
    else:
        outs += [None] * 6
    return tuple(outs)


_GruFuse._backward_ring = staticmethod(_gru_backward_ring)


def gru_fuse(z, c16, cell):
    """See _GruFuse.  `cell`: the ConvGRUCell (three EqualizedConv3d gates over 16 + 3 + 16 input channels)."""
    gs = (cell.update_gate, cell.reset_gate, cell.out_gate)
    return _GruFuse.apply(z, c16, gs[0].module.weight, gs[0].bias, gs[1].module.weight, gs[1].bias, gs[2].module.weight, gs[2].bias)


def lstm_cell(cc, c_cur):
    return _LstmCell.apply(cc, c_cur)


def gru_gates(upre, rpre, h):
    return _GruGates.apply(upre, rpre, h)


def gru_blend(h, u, cand):
    return _GruBlend.apply(h, u, cand)


def conv3x3_sum16(weight, bias, widths, parts, cols=None, addend=None):
    """See _Conv3x3Sum16.  `widths`: the parts cover W's input channels back to back; or `cols` = ((first, width), ...)."""
    if cols is None:
        cols, c0 = [], 0
        for wdt in widths:
            cols.append((c0, wdt))
            c0 += wdt
        assert c0 == weight.shape[1]
    return _Conv3x3Sum16.apply(weight, bias, tuple(tuple(c) for c in cols), addend, *parts)


class _LiftView(torch.autograd.Function):
    """(V, c0*S, H, W) channels-last, channel = c*S + d  ->  (V, c0, S, H, W) channels-last volume: the `.view` of
    FactorProjection2d3d (modules/geometry.py:728) as ONE permutation kernel each way (lf_lift_permute) instead of a
    strided ATen copy (1.9 ms per GB here; the kernel moves full lines both sides)."""

    @staticmethod
    def forward(ctx, y, c0, S):
        y = cl(y)
        V, cs, H, W = y.shape
        ctx.dims = (V, H, W, c0, S)
        return _lift_permute(y, V, H * W, c0, S, False, empty_cl((V, c0, S, H, W), y.device))

    @staticmethod
    def backward(ctx, g):
        V, H, W, c0, S = ctx.dims
        return _lift_permute(cl(g), V, H * W, c0, S, True, empty_cl((V, c0 * S, H, W), g.device)), None, None


def _lift_fused_ok(c0, S):
    return c0 % 4 == 0 and S % 4 == 0 and (2 * 4 * c0 * (S + 1) + 16) * 4 <= 150 * 1024


class _LiftFused(torch.autograd.Function):
    """FactorProjection2d3d with a gradient (training step), two passes each way: pointwise conv + LeakyReLU as rows, then
    lf_lift_norm_unfold (PixelNorm over all C0*S channels + the layout change into the channels-last volume, bf16 storage under
    the autocast storage policy); backward lf_lift_bwd (PixelNorm' / LeakyReLU' from the volume-layout gradient and the saved
    volume, written as rows) feeding the weight / data gradient products.  Replaces conv1x1 + PixelNorm pass + lift_permute
    (and their three backward passes): the (V, C0*S, H, W) activation is never stored."""

    @staticmethod
    def forward(ctx, x, weight, bias, S):
        L = _lib.lib()
        _req(x, 'x'), _req(weight, 'weight')
        ctx.ac = ops.AUTOCAST is not None
        x = _ac_in(cl(x))
        V, cin, H, W = x.shape
        cs = weight.shape[0]
        c0 = cs // S
        P = H * W
        he = he_constant(weight)
        ctx.mfma = bool(LIFT_MFMA and ctx.ac and storage_bf16() and c0 == 16 and cin == 16 and S in (16, 32, 64, 128) and P % 16 == 0)
        if ctx.mfma:
            # round 6 (csrc/lift_mfma.hip): K = 16 -- the 16*S-channel row is recomputed by MFMA instead of stored as fp32 rows
            wtab = _pk(weight, 'l16f', lambda w: w.reshape(16, S, 16).permute(1, 0, 2).contiguous().to(torch.bfloat16))
            btab = bias.detach().reshape(16, S).t().contiguous() if bias is not None else None
            vol = empty_cl16((V, c0, S, H, W), x.device, True)
            norm = torch.empty(V * P, device=x.device, dtype=torch.float32)
            with _timed('lift16_fwd'):
                check(L.lf_lift16_fwd(_ptr(x), _ptr(wtab), _ptr(btab) if btab is not None else None, _ptr(vol), _ptr(norm), V * P, P, S,
                                      he, SLOPE, PN_EPS, _stream()), 'lf_lift16_fwd')
            ctx.dims, ctx.he = (V, cin, H, W, c0, S), he
            ctx.save_for_backward(vol, norm, weight, x)
            return vol
        wpack = _pk(weight, 'c1f', lambda w: pack_conv1x1(w.reshape(cs, cin)))
        tmp = torch.empty(V * P, cs, device=x.device, dtype=torch.float32)
        _conv1x1_raw(x, wpack, bias.detach() if bias is not None else None, V, P, cin, 1, P * cin, 0, cs, tmp, he, LF_EPI_LRELU)
        out16 = storage_bf16() and c0 == 16
        vol = empty_cl16((V, c0, S, H, W), x.device, out16)
        norm = torch.empty(V * P, device=x.device, dtype=torch.float32)
        with _timed('lift_norm_unfold'):
            check(L.lf_lift_norm_unfold(_ptr(tmp), _ptr(vol), _ptr(norm), V, P, c0, S, PN_EPS, int(out16), _stream()), 'lf_lift_norm_unfold')
        del tmp
        ctx.dims, ctx.he = (V, cin, H, W, c0, S), he
        ctx.save_for_backward(vol, norm, weight, x)
        return vol

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        vol, norm, w, x = ctx.saved_tensors
        V, cin, H, W, c0, S = ctx.dims
        cs, P = c0 * S, H * W
        g = cl(g)
        if ctx.mfma and g.dtype == torch.bfloat16 and vol.dtype == torch.bfloat16:
            wtab_t = _pk(w, 'l16b', lambda t: t.reshape(16, S, 16).permute(1, 2, 0).contiguous().to(torch.bfloat16))
            gx = empty_cl((V, cin, H, W), g.device)
            gw = torch.empty(cs, cin, device=g.device, dtype=torch.float32)
            gb = torch.empty(cs, device=g.device, dtype=torch.float32)
            nb = L.lf_lift16_bwd_scratch_bytes(S)
            scr = torch.empty(nb // 4 + 4, device=g.device, dtype=torch.float32)
            with _timed('lift16_bwd'):
                check(L.lf_lift16_bwd(_ptr(g), _ptr(vol), _ptr(norm), _ptr(x), _ptr(wtab_t), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(scr, True),
                                      scr.numel() * 4, V * P, P, S, ctx.he, SLOPE, 1, _stream()), 'lf_lift16_bwd')
            return (gx if ctx.needs_input_grad[0] else None, _ac_in(gw.reshape(w.shape)) if ctx.needs_input_grad[1] else None,
                    gb if ctx.needs_input_grad[2] else None, None)
        gp = torch.empty(V * P, cs, device=g.device, dtype=torch.float32)
        io = (1 if g.dtype == torch.bfloat16 else 0) | (2 if vol.dtype == torch.bfloat16 else 0)
        with _timed('lift_bwd'):
            check(L.lf_lift_bwd(_ptr(g), _ptr(vol), _ptr(norm), _ptr(gp), V, P, c0, S, SLOPE, int(ctx.ac), io, _stream()), 'lf_lift_bwd')
        gx = gw = gb = None
        with autocast(ctx.ac):
            if ctx.needs_input_grad[0]:
                wpack_t = _pk(w, 'c1b', lambda t: pack_conv1x1(t.reshape(cs, cin).t()))
                gx = empty_cl((V, cin, H, W), g.device)
                _conv1x1_raw(gp, wpack_t, None, V, P, cs, 1, P * cs, 0, cin, gx, ctx.he, 0)
                gx = _ac_in(gx)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                gwt, gb = conv_bwd_weight(x.permute(0, 2, 3, 1).reshape(V * P, cin), gp, 0, cin, ctx.he, want_bias=not ctx.ac)
                if ctx.ac and ctx.needs_input_grad[2]:
                    gb = bias_grad(gp, 0)              # (of the rounded rows: 5e5 roundings of 2^-9 average out far below fp32's own noise)
                gw = _ac_in(gwt.reshape(w.shape))
        return gx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None, None
