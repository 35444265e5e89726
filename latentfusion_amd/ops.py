"""Device operators of the hot path: thin autograd wrappers over the C ABI (include/lf_hip.h).

Tensors keep the reference's LOGICAL shapes (N,C,D,H,W) / (N,C,H,W) but live in channels-last
memory (torch.channels_last_3d / torch.channels_last), which is exactly the [N][D][H][W][C]
layout the kernels use -- so the modules built on these ops are shape-compatible drop-ins for
the reference's nn.Modules without any transposes between layers.

There is no CPU or ATen fallback in this file: every op calls into liblf_hip.so.
"""
import math

import torch

from . import _lib
from ._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, LF_MAP_C2O, LF_MAP_COEFS, LF_MAP_O2C, check

SLOPE = 0.2
PN_EPS = 1e-8
# d(loss)/d(sampled volume) of the 3-D resamplers (training / encoder backward): True = fixed-point accumulation,
# bit-reproducible (lf_resample3d_bwd_vol_det); False = fp32 atomics (lf_resample3d_bwd_vol, order-dependent rounding)
DETERMINISTIC_SPLAT = True

# bf16 autocast policy of the training step (reference: `autocast(enabled=self.training)` around Sculptor / Photographer
# .forward, recon/models.py:199,405; tools/train/train_reconstruct.py:455).  Inside `with ops.autocast():` every
# convolution (3x3(x3), 1x1, factor projections, GRU gates) sees bf16-rounded inputs and weights with fp32 accumulation,
# its data / weight gradients likewise and rounded to bf16, exactly the tensors torch autocast hands to / takes from the
# convolution; everything else (bias, LeakyReLU, PixelNorm, resampling, losses, master weights, Adam) stays fp32.
# The 3-D 16 -> 16 blocks run on the bf16 MFMA (lf_conv3d_c16_bf16, forward and data gradient); the other layers round
# their operands (lf_round_bf16) and keep the fp32-MFMA kernels, which then compute what a bf16 MFMA would (bf16 x bf16
# products are exact in fp32).
AUTOCAST = None
# Storage policy under autocast (round 5): the 16-channel 3-D volumes that half-precision convolutions produce and consume
# -- block activations, their gradients, the resampled volumes between them, the per-view latent volumes -- are kept as
# bf16 channels-last tensors (32 B per voxel).  The convolutions round their inputs to bf16 while staging them anyway, so a
# volume that only convolutions read loses nothing; a volume that an fp32 stage also reads (PixelNorm' / LeakyReLU' of the
# epilogue backward, the trilinear resampler) is read rounded.  Halves the HBM bytes and the saved-activation memory of the
# training step.  False: fp32 storage everywhere (the round-2..4 behaviour), kept for A/B runs and tests.
BF16_STORAGE = True


def storage_bf16():
    return AUTOCAST is not None and BF16_STORAGE


def _f32(t):
    """An fp32 view of a tensor an op without a bf16-storage form was handed (a cast pass; the hot ops take bf16 natively)."""
    return t if t.dtype == torch.float32 else t.float()


class autocast:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global AUTOCAST
        self.prev = AUTOCAST
        AUTOCAST = 'bf16' if self.enabled else None
        return self

    def __exit__(self, *exc):
        global AUTOCAST
        AUTOCAST = self.prev
        return False


def round_bf16(t):
    """bf16(t) in an fp32 container of the same layout (lf_round_bf16)."""
    L = _lib.lib()
    _req(t, 'tensor')
    src = t if (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)
                or (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))) else t.contiguous()
    out = torch.empty_like(src, memory_format=torch.preserve_format)
    check(L.lf_round_bf16(_ptr(src), _ptr(out), src.numel(), _stream()), 'lf_round_bf16')
    return out


def _ac_in(t):
    """An operand as the convolution sees it under the active policy."""
    return round_bf16(t) if AUTOCAST is not None else t

# Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
# roofline of the dominant kernel).  Set to a list to collect (name, start_event, end_event).
KERNEL_TIMER = None
# Restrict the events to these kernel tags (None = every tagged launch).  An event pair costs ~4-5 us of dispatch gap
# around its kernel: bench.py times only the kernels its roofline fields report.
KERNEL_TIMER_TAGS = None


class _Tag(str):
    """A kernel tag that compares like its name and carries a `detail` (form / batch / storage of the launch)."""
    detail = ''


class _timed:
    def __init__(self, name, detail=''):
        self.name = _Tag(name)
        self.name.detail = detail
        self.on = False

    def __enter__(self):
        self.on = KERNEL_TIMER is not None and (KERNEL_TIMER_TAGS is None or self.name in KERNEL_TIMER_TAGS)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()                    # current stream == the stream the kernel is launched on
        return self

    def __exit__(self, *exc):
        if self.on and KERNEL_TIMER is not None:
            self.e1.record()
            KERNEL_TIMER.append((self.name, self.e0, self.e1))
        return False



def _T():
    """The training-step operators (ops_train.py), imported on first use: that module builds on this one."""
    from . import ops_train
    return ops_train


def __getattr__(name):
    """`ops.<name>` for the operators that live in ops_train.py (training step: weight gradients, bf16-storage layers, the
    fused ConvGRU recurrence / lift) and experimental.py (A/B and superseded kernels, include/lf_hip_experimental.h)."""
    import importlib
    for mod in ('ops_train', 'experimental'):
        m = importlib.import_module('.' + mod, __package__)
        if name in m.__dict__:
            return m.__dict__[name]
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')

def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name):
    if not t.is_cuda:
        raise _lib.LFHipError(f'{name} must be a device tensor: the HIP path has no CPU fallback')
    if t.dtype != torch.float32:
        raise TypeError(f'{name} must be float32')
    return t


def cl(t):
    """channels-last physical layout for a logical NC[D]HW tensor (no copy if already so)."""
    if t.dim() == 5:
        return t.contiguous(memory_format=torch.channels_last_3d)
    if t.dim() == 4:
        return t.contiguous(memory_format=torch.channels_last)
    raise ValueError('expected a 4-D or 5-D tensor')


def empty_cl(shape, device):
    fmt = torch.channels_last_3d if len(shape) == 5 else torch.channels_last
    return torch.empty(shape, device=device, dtype=torch.float32, memory_format=fmt)


def _ptr(t, scratch=False):
    if _lib.BYTE_LOG is not None:
        _lib.note_bytes(t, scratch)
    return t.data_ptr()


# ---------------------------------------------------------------------------------------------
# weight packing (host side, cached per parameter version)
# ---------------------------------------------------------------------------------------------
_WEIGHT_EPOCH = 0


def invalidate_weight_packs():
    """Call after writing parameters through a raw pointer (e.g. lf_adam_step on a flat buffer): such writes
    do not bump torch's version counters, so the cached packs must be told."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def _cached(key_tensor, tag, fn):
    """Memoises a packed layout ON the parameter object (so it dies with it and can never be
    confused with another tensor that later reuses the same device address); re-packs when the
    parameter is modified in place (version counter), moved, or when an optimiser that writes through
    raw pointers announced an update (invalidate_weight_packs)."""
    store = key_tensor.__dict__.setdefault('_lf_pack', {})
    stamp = (key_tensor._version, key_tensor.data_ptr(), _WEIGHT_EPOCH)
    hit = store.get(tag)
    if hit is None or hit[0] != stamp:
        hit = (stamp, fn())
        store[tag] = hit
    return hit[1]


def _wsrc(weight):
    """The weight tensor the kernels should see: the parameter itself, or its bf16 rounding under autocast."""
    if AUTOCAST is None:
        return weight
    return _cached(weight, 'ac_round', lambda: round_bf16(weight.detach()))


def _pk(weight, tag, fn):
    """Cached packed layout of `weight` (of its bf16 rounding under autocast) -- fn(tensor) builds it."""
    return _cached(weight, tag + ('@ac' if AUTOCAST is not None else ''), lambda: fn(_wsrc(weight)))


def _pad16(v):
    return (v + 15) // 16 * 16


def pack_conv3x3(weight, transpose=False):
    """[Cout,Cin,k..] -> [tap][CoutP][CinP] (lf_conv3x3_fwd layout).  transpose=True builds the
    data-gradient operator: taps flipped, in/out channels swapped."""
    L = _lib.lib()
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=tuple(range(2, w.dim())))
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    coutp, cinp = L.lf_conv3x3_cout_padded(cout), _pad16(cin)
    out = torch.zeros(taps, coutp, cinp, device=w.device, dtype=torch.float32)
    out[:, :cout, :cin] = w.reshape(cout, cin, taps).permute(2, 0, 1)
    return out.contiguous()


def pack_conv1x1(weight2d):
    """[Cout,K] -> [CoutP][Kp] (lf_conv1x1_fwd layout)."""
    L = _lib.lib()
    cout, k = weight2d.shape
    out = torch.zeros(L.lf_conv1x1_cout_padded(cout), _pad16(k), device=weight2d.device, dtype=torch.float32)
    out[:cout, :k] = weight2d.detach()
    return out.contiguous()


def amax_buffer(value=None, device='cuda', rows=1):
    """Zeroed max-abs side-channel buffer(s) (rows x LF_AMAX_FLOATS); `value` (a tensor) pre-loads slot 0 of
    row 0, e.g. a maximum computed on the host side of the ABI."""
    buf = torch.zeros(rows, _lib.LF_AMAX_FLOATS, device=device, dtype=torch.float32)
    if value is not None:
        buf[0, 0] = value.reshape(()).to(device)
    return buf


_PAIR_TABLE = {}


def _pair_taps(w):
    """[16 cout][16 cin][3,3,3] -> [14 pairs][16 cout][32 = 2 taps x 16 cin] in the tap-pair order of the ring kernels
    (lf_conv3d_c16_split_pairs; a missing second tap = zeros): one gather, not 28 slice copies per pack -- the training
    step repacks every weight every iteration."""
    idx = _PAIR_TABLE.get(w.device)
    if idx is None:
        import ctypes
        table = (ctypes.c_int * 28)()
        _lib.lib().lf_conv3d_c16_split_pairs(table)
        idx = _PAIR_TABLE[w.device] = torch.tensor([t if t >= 0 else 27 for t in table], device=w.device)
    taps = torch.cat((w.reshape(16, 16, 27), w.new_zeros(16, 16, 1)), dim=2)       # tap 27 = zeros
    return taps[:, :, idx].reshape(16, 16, 14, 2).permute(2, 0, 3, 1).reshape(14, 16, 32)


def pack_conv3d_c16_split(weight, transpose=False):
    """[16,16,3,3,3] fp32 -> f16 hi/lo packs [14 pairs][hi,lo][16 cout][32 = 2 taps x 16 cin] for
    lf_conv3d_c16_split (the second tap of the last pair is zero)."""
    w = weight.detach().float()
    if transpose:
        w = w.transpose(0, 1).flip(dims=(2, 3, 4))
    assert tuple(w.shape) == (16, 16, 3, 3, 3)
    k = _pair_taps(w)                                              # [14 pairs][cout][tapsel*16 + cin]
    hi = k.half()
    lo = (k - hi.float()).half()
    return torch.stack((hi, lo), dim=1).contiguous()              # [14][2][16][32]


def conv3d_c16_split(x, wsplit, bias, he, flags, prev=None, amax_in=None, amax_out=None, want_norm=True):
    """Launch lf_conv3d_c16_split on a channels-last (N,16,D,H,W) tensor."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    py, pn, pf = (prev[0], prev[1], prev[2]) if prev is not None else (None, None, 0)
    with _timed('conv3d_c16_split'):
        check(L.lf_conv3d_c16_split(_ptr(x), _ptr(wsplit), _ptr(bias) if bias is not None else None, _ptr(y),
                                    _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                    _ptr(py) if py is not None else None, _ptr(pn) if pn is not None else None, pf,
                                    _ptr(amax_in) if amax_in is not None else None,
                                    _ptr(amax_out) if amax_out is not None else None, _stream()), 'lf_conv3d_c16_split')
    return y, norm


def pack_conv3d_c16_ring_bf16(weight, transpose=False):
    """[16,16,3,3,3] -> bf16 [14 pairs][16 cout][32 = 2 taps x 16 cin] for lf_conv3d_c16_ring_bf16 (the tap pairs of
    lf_conv3d_c16_split_pairs; the second tap of the last pair is zero)."""
    w = weight.detach().float()
    if transpose:
        w = w.transpose(0, 1).flip(dims=(2, 3, 4))
    assert tuple(w.shape) == (16, 16, 3, 3, 3)
    k = _pair_taps(w)
    return k.to(torch.bfloat16).contiguous()


def conv3d_c16_ring_bf16(x, wpack, bias, he, flags, round_out, addend=None):
    """Launch lf_conv3d_c16_ring_bf16 on a channels-last (N,16,D,H,W) tensor (un-rounded input: the kernel rounds)."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    with _timed('conv3d_c16_ring_bf16', f"{'add' if addend is not None else 'fwd'}:{N}:io0"):
        check(L.lf_conv3d_c16_ring_bf16(_ptr(x), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y),
                                        _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                        _ptr(addend) if addend is not None else None, round_out, _stream()), 'lf_conv3d_c16_ring_bf16')
    return y, norm


def empty_cl16(shape, device, bf16):
    """A channels-last (N,16,D,H,W) volume in fp32 or bf16 storage."""
    return torch.empty(shape, device=device, dtype=torch.bfloat16 if bf16 else torch.float32, memory_format=torch.channels_last_3d)


def conv3d_c16_ring_bf16_io(x, wpack, bias, he, flags, round_out, addend=None, out=None, out_bf16=False):
    """lf_conv3d_c16_ring_bf16_io: the bf16 ring convolution on volumes stored as fp32 OR bf16 channels-last records (the
    dtype of `x` / `addend` / `out` says which).  `out`: write into this (N,16,D,H,W) channels-last tensor (e.g. a slice of a
    gradient block) instead of a new one.  Returns (y, norm)."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = out if out is not None else empty_cl16((N, 16, D, H, W), x.device, out_bf16)
    io = ((_lib.LF_IO_IN_BF16 if x.dtype == torch.bfloat16 else 0) | (_lib.LF_IO_OUT_BF16 if y.dtype == torch.bfloat16 else 0)
          | (_lib.LF_IO_ADDEND_BF16 if (addend is not None and addend.dtype == torch.bfloat16) else 0))
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    with _timed('conv3d_c16_ring_bf16', f"{'add' if addend is not None else 'fwd'}:{N}:io{io}"):
        check(L.lf_conv3d_c16_ring_bf16_io(_ptr(x), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y),
                                           _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                           _ptr(addend) if addend is not None else None, round_out, io, _stream()),
              'lf_conv3d_c16_ring_bf16_io')
    return y, norm


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def pack_conv3d_c16_wino(weight, transpose=False):
    """[16,16,3,3,3] fp32 -> Winograd-domain pack [4 a][16 b*4+c][4 i][64 lanes] for lf_conv3d_c16_wino:
    U = (G x G x G) w evaluated in fp64, rounded once to fp32."""
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=(2, 3, 4))
    assert tuple(w.shape) == (16, 16, 3, 3, 3)
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    U = torch.einsum('ai,bj,ck,omijk->abcom', G, G, G, w.double())          # [a][b][c][cout][cin]
    U = U.reshape(4, 16, 16, 4, 4)                                          # [a][bc][cout][kg][i]
    U = U.permute(0, 1, 4, 3, 2).reshape(4, 16, 4, 64)                      # [a][bc][i][lane = kg*16 + cout]
    return U.float().contiguous()


def conv3d_c16_wino(x, upack, bias, he, flags, prev=None, amax_out=None, out=None):
    """Launch lf_conv3d_c16_wino on a channels-last (N,16,D,H,W) tensor (`out`: write into this tensor)."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = out if out is not None else empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    py, pn, pf = (prev[0], prev[1], prev[2]) if prev is not None else (None, None, 0)
    with _timed('conv3d_c16_wino'):
        check(L.lf_conv3d_c16_wino(_ptr(x), _ptr(upack), _ptr(bias) if bias is not None else None, _ptr(y),
                                   _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                   _ptr(py) if py is not None else None, _ptr(pn) if pn is not None else None, pf,
                                   _ptr(amax_out) if amax_out is not None else None, _stream()), 'lf_conv3d_c16_wino')
    return y, norm


def pack_wino_proj(wp_dmajor, transpose=False):
    """Depth slices of the factor projection for the fused forms of lf_conv3d_c16_wino (include/lf_hip.h):
    wp_dmajor = the (16, D*16) matrix lf_conv1x1_fwd takes (K = d*16 + c) -> [D][64 lanes][4].
    transpose=False: A operands of the forward projection, [d][l][i] = Wp[l & 15][d*16 + (l >> 4)*4 + i];
    transpose=True : A operands of its data gradient,      [d][l][i] = Wp[(l >> 4)*4 + i][d*16 + (l & 15)]."""
    w = wp_dmajor.detach()
    cout, K = w.shape
    assert cout == 16 and K % 16 == 0
    D = K // 16
    if not transpose:
        out = w.reshape(16, D, 4, 4).permute(1, 2, 0, 3)          # [m][d][kg][i] -> [d][kg][m][i]
    else:
        out = w.reshape(4, 4, D, 16).permute(2, 0, 3, 1)          # [kg][i][d][m] -> [d][kg][m][i]
    return out.reshape(D, 64, 4).float().contiguous()


def conv3d_c16_wino_projfwd(x, upack, bias, he, flags, proj_wA, proj_bias, proj_he, proj_flags):
    """lf_conv3d_c16_wino_projfwd: the last camera block's convolution AND the factor projection behind it in one launch.
    Returns (y, norm, zp (N,16,H,W) channels-last, pnorm)."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    zp = empty_cl((N, 16, H, W), x.device)
    pnorm = torch.empty(N * H * W, device=x.device, dtype=torch.float32) if (proj_flags & LF_EPI_PIXELNORM) else None
    with _timed('conv3d_c16_wino_projfwd'):
        check(L.lf_conv3d_c16_wino_projfwd(_ptr(x), _ptr(upack), _ptr(bias) if bias is not None else None, _ptr(y),
                                           _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                           _ptr(proj_wA), _ptr(proj_bias) if proj_bias is not None else None, _ptr(zp),
                                           _ptr(pnorm) if pnorm is not None else None, proj_he, proj_flags, _stream()),
              'lf_conv3d_c16_wino_projfwd')
    return y, norm, zp, pnorm


# 'fused' (default): input transform -> lf_wino_fused_gemm (this library's fp32-MFMA GEMM with the output transform and
# the epilogue folded in; the Winograd-domain products never reach memory).  'bmm': the three-stage form with the
# per-frequency products on the library GEMM (torch.bmm -> rocBLAS), kept for A/B measurements (tools/rel_probe.py).
WIDE_CONV_MODE = 'fused'


def pack_conv_wino_fused(weight, transpose=False):
    """[Cout,Cin,3,3(,3)] -> U2 [64 | 16 f][CoutP][Cin] (output-channel major, CoutP = lf_wino_fused_cout_padded(Cout),
    zero padded) for lf_wino_fused_gemm: U2[f][co][ci] = ((G x ..) w)[co][ci][f], evaluated in fp64."""
    L = _lib.lib()
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=tuple(range(2, w.dim())))
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    cout, cin = w.shape[0], w.shape[1]
    if w.dim() == 5:
        U = torch.einsum('ai,bj,ck,omijk->abcom', G, G, G, w.double()).reshape(64, cout, cin)
    else:
        U = torch.einsum('bj,ck,omjk->bcom', G, G, w.double()).reshape(16, cout, cin)
    out = torch.zeros(U.shape[0], L.lf_wino_fused_cout_padded(cout), cin, device=w.device, dtype=torch.float32)
    out[:, :cout] = U.float()
    return out.contiguous()


def conv_wino_fused(x, U2, cout, bias, he, flags, depth_inner=False):
    """Wide 2-D / 3-D conv: Winograd input transform, then ONE launch for the per-frequency fp32-MFMA products, the
    output transform, He scale, bias and LeakyReLU (lf_wino_fused_gemm); PixelNorm as a pass over the (small) output.
    depth_inner (3-D only): y comes back as a plain (N, H, W, D, cout) tensor (LF_OUT_DEPTH_INNER): the layout the factor
    projection that follows the last camera block wants.  Returns (y, norm or None)."""
    L = _lib.lib()
    dims = x.dim() - 2
    N, cin = x.shape[0], x.shape[1]
    D, H, W = (x.shape[2:] if dims == 3 else (1,) + tuple(x.shape[2:]))
    if dims == 3:
        T = L.lf_wino3d_tiles(N, D, H, W)
        V = torch.empty(64, T, cin, device=x.device, dtype=torch.float32)
        with _timed('wino3d_input'):
            check(L.lf_wino3d_input_transform(_ptr(x), _ptr(V), N, D, H, W, cin, _stream()), 'lf_wino3d_input_transform')
    else:
        T = L.lf_wino2d_tiles(N, H, W)
        V = torch.empty(16, T, cin, device=x.device, dtype=torch.float32)
        with _timed('wino2d_input'):
            check(L.lf_wino2d_input_transform(_ptr(x), _ptr(V), N, H, W, cin, _stream()), 'lf_wino2d_input_transform')
    if depth_inner:
        assert dims == 3
        y = torch.empty((N, H, W, D, cout), device=x.device, dtype=torch.float32)
    else:
        y = empty_cl((N, cout) + tuple(x.shape[2:]), x.device)
    nscr = L.lf_wino_fused_scratch_bytes(dims, N, D, H, W, cout)
    scr = torch.empty(nscr // 4, device=x.device, dtype=torch.float32) if nscr else None
    with _timed(f'wino{dims}d_fused'):
        check(L.lf_wino_fused_gemm(_ptr(V), _ptr(U2), _ptr(bias) if bias is not None else None, _ptr(y),
                                   _ptr(scr, True) if scr is not None else None, nscr, dims, N, D, H, W, cin,
                                   cout, he, (flags & LF_EPI_LRELU) | (_lib.LF_OUT_DEPTH_INNER if depth_inner else 0), SLOPE, _stream()),
              'lf_wino_fused_gemm')
    del V
    norm = None
    if flags & LF_EPI_PIXELNORM:
        norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32)
        check(L.lf_pixelnorm_fwd(_ptr(y), _ptr(y), _ptr(norm), N * D * H * W, cout, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
    return y, norm


def wide_conv(x, weight, bias, he, flags, transpose=False, depth_inner=False):
    """Dispatch of a wide (>= 64-channel) 3x3(x3) convolution or its data gradient (transpose=True) by WIDE_CONV_MODE."""
    cout = weight.shape[1] if transpose else weight.shape[0]
    if WIDE_CONV_MODE == 'fused':
        U2 = _pk(weight, 'wfb' if transpose else 'wff', lambda w: pack_conv_wino_fused(w, transpose=transpose))
        return conv_wino_fused(x, U2, cout, bias, he, flags, depth_inner)
    from . import experimental                     # 'bmm': the three-stage form on the library GEMM (A/B reference)
    U = _pk(weight, 'g3b' if transpose else 'g3f', lambda w: experimental.pack_conv3d_wino_gemm(w, transpose=transpose))
    return experimental.conv3d_wino_gemm(x, U, bias, he, flags)


def _wino_gemm_ok(x, weight):
    """Wide 2-D / 3-D 3x3 convolutions where the transforms amortise over the channels."""
    return (x.dim() == weight.dim() and x.dim() in (4, 5) and weight.shape[0] >= 64 and weight.shape[1] >= 64
            and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0)


def he_constant(weight):
    """sqrt(2 / fan_in)  (modules/equalized.py:66-74)."""
    return math.sqrt(2.0 / weight[0].numel())


# ---------------------------------------------------------------------------------------------
# 3-D resampling
# ---------------------------------------------------------------------------------------------
def _resample_fwd(vol, coef, kind):
    L = _lib.lib()
    n = coef.shape[0]
    vol_n = 1 if (vol.shape[0] == 1 or vol.stride(0) == 0) else vol.shape[0]
    if vol_n not in (1, n):
        raise ValueError('batch dimension of the volume and the cameras must match')
    v = cl(vol[:1] if vol_n == 1 else vol)
    _, C, D, H, W = v.shape
    out = empty_cl((n, C, D, H, W), v.device)
    cf = torch.zeros(n, LF_MAP_COEFS, device=v.device, dtype=torch.float32)
    cf[:, :coef.shape[1]] = coef
    check(L.lf_resample3d_fwd(_ptr(v), vol_n, _ptr(cf), kind, _ptr(out), n, D, H, W, C, _stream()),
          'lf_resample3d_fwd')
    return out, v, cf, vol_n


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, coef, kind):
        vol = _f32(vol)
        _req(vol, 'vol'), _req(coef, 'coef')
        out, v, cf, vol_n = _resample_fwd(vol, coef.detach().float(), kind)
        ctx.save_for_backward(v, cf)
        ctx.kind, ctx.vol_n, ctx.vol_shape, ctx.ncoef = kind, vol_n, vol.shape, coef.shape[1]
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        v, cf = ctx.saved_tensors
        g = cl(gout)
        n, C, D, H, W = g.shape
        gvol = gcoef = None
        if ctx.needs_input_grad[1]:
            if ctx.kind != LF_MAP_O2C:
                raise NotImplementedError('the reference C2O transform is not differentiable w.r.t. the camera '
                                          '(in-place division, modules/geometry.py:636)')
            gcoef = torch.empty(n, 18, device=g.device, dtype=torch.float32)
            nbytes = L.lf_resample3d_bwd_coef_scratch_bytes(n, D, H, W)
            scratch = torch.empty(max(nbytes, 4) // 4 + 1, device=g.device, dtype=torch.float32)
            check(L.lf_resample3d_bwd_coef(_ptr(g), _ptr(v), ctx.vol_n, _ptr(cf), _ptr(gcoef), _ptr(scratch, True),
                                           scratch.numel() * 4, n, D, H, W, C, _stream()), 'lf_resample3d_bwd_coef')
        if ctx.needs_input_grad[0]:
            if DETERMINISTIC_SPLAT:
                gv = empty_cl((ctx.vol_n, C, D, H, W), g.device)
                nb = L.lf_resample3d_bwd_vol_det_scratch_bytes(ctx.vol_n, D, H, W, C)
                scr = torch.empty(nb // 8 + 1, device=g.device, dtype=torch.int64)
                check(L.lf_resample3d_bwd_vol_det(_ptr(g), _ptr(cf), ctx.kind, _ptr(gv), ctx.vol_n, _ptr(scr, True), scr.numel() * 8,
                                                  n, D, H, W, C, _stream()), 'lf_resample3d_bwd_vol_det')
            else:
                gv = empty_cl((ctx.vol_n, C, D, H, W), g.device).zero_()
                check(L.lf_resample3d_bwd_vol(_ptr(g), _ptr(cf), ctx.kind, _ptr(gv), ctx.vol_n, n, D, H, W, C, _stream()),
                      'lf_resample3d_bwd_vol')
            if gv.shape[0] == ctx.vol_shape[0]:
                gvol = gv
            else:
                # expanded (stride-0) input: autograd sums over the expanded batch itself, so hand
                # back the accumulated volume once and zeros elsewhere
                gvol = torch.zeros(ctx.vol_shape, device=g.device, dtype=torch.float32)
                gvol[0] = gv[0]
        return gvol, gcoef, None


def resample_o2c(vol, coef):
    """ObjectToCameraTransform as an op: vol (1|N,C,S,S,S), coef (N,18) -> (N,C,S,S,S)."""
    if _T()._resample_ac_ok(vol, coef):
        return _T()._ResampleAC.apply(vol, coef, LF_MAP_O2C)
    return _Resample.apply(vol, coef, LF_MAP_O2C)


def resample_c2o(vol, coef):
    """CameraToObjectTransform as an op: vol (N,C,S,S,S), coef (N,16) -> (N,C,S,S,S)."""
    if _T()._resample_ac_ok(vol, coef):
        return _T()._ResampleAC.apply(vol, coef, LF_MAP_C2O)
    return _Resample.apply(vol, coef, LF_MAP_C2O)


# ---------------------------------------------------------------------------------------------
# convolutions with fused epilogue
# ---------------------------------------------------------------------------------------------
def _conv3x3_raw(x, wpack, bias, cout, he, flags, want_norm):
    L = _lib.lib()
    dims = x.dim() - 2
    if dims == 3:
        N, cin, D, H, W = x.shape
    else:
        N, cin, H, W = x.shape
        D = 1
    fuse_pn = bool(flags & LF_EPI_PIXELNORM) and cout <= 64
    kflags = flags if fuse_pn else (flags & ~LF_EPI_PIXELNORM)
    y = empty_cl((N, cout) + tuple(x.shape[2:]), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    with _timed(f'conv3x3_{dims}d_{cin}x{cout}'):
        check(L.lf_conv3x3_fwd(_ptr(x), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y),
                               _ptr(norm) if (norm is not None and fuse_pn) else None,
                               dims, N, D, H, W, cin, cout, he, kflags, SLOPE, PN_EPS, _stream()), 'lf_conv3x3_fwd')
    if (flags & LF_EPI_PIXELNORM) and not fuse_pn:
        check(L.lf_pixelnorm_fwd(_ptr(y), _ptr(y), _ptr(norm), N * D * H * W, cout, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
    return y, norm


def conv3x3_bwd_data(gy, wpack_t, cin, he, prev=None):
    """gx = conv^T(gy) * he; prev = (y_prev, norm_prev, flags_prev) folds the producing layer's
    epilogue backward into the store (16-channel records only)."""
    L = _lib.lib()
    dims = gy.dim() - 2
    N, cout = gy.shape[0], gy.shape[1]
    D, H, W = (gy.shape[2:] if dims == 3 else (1,) + tuple(gy.shape[2:]))
    gx = empty_cl((N, cin) + tuple(gy.shape[2:]), gy.device)
    py, pn, pf = (prev[0], prev[1], prev[2]) if prev is not None else (None, None, 0)
    with _timed(f'conv3x3_{dims}d_{cout}x{cin}'):
        check(L.lf_conv3x3_bwd_data(_ptr(gy), _ptr(wpack_t), _ptr(gx), dims, N, D, H, W, cout, cin, he,
                                    _ptr(py) if py is not None else None, _ptr(pn) if pn is not None else None, pf,
                                    SLOPE, _stream()), 'lf_conv3x3_bwd_data')
    return gx


def _epilogue_bwd(gy, y, norm, flags):
    L = _lib.lib()
    if flags == 0:
        return gy
    rows = gy.numel() // gy.shape[1]
    gp = torch.empty_like(gy, memory_format=torch.preserve_format)
    check(L.lf_epilogue_bwd(_ptr(gy), _ptr(y), _ptr(norm) if norm is not None else None, _ptr(gp), rows,
                            gy.shape[1], flags, SLOPE, _stream()), 'lf_epilogue_bwd')
    return gp


def _wino_ok(x, weight):
    """The Winograd kernel serves 3-D 16 -> 16 convolutions on volumes of at least one 2x8x16 tile row."""
    return (x.dim() == 5 and tuple(weight.shape[:2]) == (16, 16) and x.shape[1] == 16
            and (x.shape[2] * x.shape[3] * x.shape[4]) * 64 < 2 ** 31)


class _Conv3x3(torch.autograd.Function):
    """x -> epilogue(conv3x3(x, W) * he + b).  The input is kept for the weight gradient only when the
    weight or bias asks for one (the pose loop never does)."""

    @staticmethod
    def forward(ctx, x, weight, bias, flags):
        _req(x, 'x'), _req(weight, 'weight')
        x = cl(x)
        he = he_constant(weight)
        b = bias.detach() if bias is not None else None
        ctx.ac = AUTOCAST is not None
        if ctx.ac and _wino_ok(x, weight):                    # autocast, 3-D 16 -> 16: direct conv on the bf16 MFMA
            y, norm = conv3d_c16_ring_bf16(x, _pk(weight, 'r3f', pack_conv3d_c16_ring_bf16), b, he, flags, 1)
            # (the input is saved un-rounded: the bf16 weight-gradient kernel rounds it while staging, like this one)
            if (weight.requires_grad or (bias is not None and bias.requires_grad)) and not _T()._wgrad_bf16_ok(x, 3, 16, 16):
                x = round_bf16(x)
        else:
            x = _ac_in(x)
            if _wino_ok(x, weight):                           # 3-D 16 -> 16: the all-fp32 Winograd kernel
                y, norm = conv3d_c16_wino(x, _pk(weight, 'w3f', pack_conv3d_c16_wino), b, he, flags)
            elif _wino_gemm_ok(x, weight):                    # wide 2-D / 3-D: Winograd with the fused fp32-MFMA GEMM
                y, norm = wide_conv(x, weight, b, he, flags)
            else:
                wpack = _pk(weight, 'c3f', pack_conv3x3)
                y, norm = _conv3x3_raw(x, wpack, b, weight.shape[0], he, flags, True)
        ctx.flags, ctx.he = flags, he
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        # everything the backward reads goes through save_for_backward (autograd's version check then catches
        # in-place edits between forward and backward); None slots are allowed
        ctx.save_for_backward(y, norm, weight, x if need_w else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm, w, x_saved = ctx.saved_tensors
        gp = _epilogue_bwd(cl(gy), y, norm, ctx.flags)
        gx = gw = gb = None
        with autocast(ctx.ac):                                # the backward runs under the policy of its forward
            if ctx.needs_input_grad[0]:
                if ctx.ac and _wino_ok(gp, w):
                    gx, _ = conv3d_c16_ring_bf16(gp, _pk(w, 'r3b', lambda t: pack_conv3d_c16_ring_bf16(t, transpose=True)), None, ctx.he, 0, 1)
                else:
                    gpc = _ac_in(gp)
                    if _wino_ok(gpc, w):
                        gx, _ = conv3d_c16_wino(gpc, _pk(w, 'w3b', lambda t: pack_conv3d_c16_wino(t, transpose=True)), None, ctx.he, 0)
                    elif _wino_gemm_ok(gpc, w):
                        gx, _ = wide_conv(gpc, w, None, ctx.he, 0, transpose=True)
                    else:
                        wpack_t = _pk(w, 'c3b', lambda t: pack_conv3x3(t, transpose=True))
                        gx, _ = _conv3x3_raw(gpc, wpack_t, None, w.shape[1], ctx.he, 0, False)
                    gx = _ac_in(gx)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dims = w.dim() - 2
                # (autocast: the weight gradient sees the half-precision gradient; the bias is added in fp32, so its
                # gradient is the column sum of the un-rounded one)
                in_kernel = ctx.ac and _T()._wgrad_bf16_ok(gp, dims, w.shape[1], w.shape[0])       # that kernel rounds its operands itself
                gwt, gb = _T().conv_bwd_weight(x_saved, gp if in_kernel else _ac_in(gp), dims, w.shape[1], ctx.he, want_bias=not ctx.ac)
                if ctx.ac and ctx.needs_input_grad[2]:
                    gb = _T().bias_grad(gp, dims)
                k = (3,) * dims
                gw = _ac_in(gwt.reshape(*k, w.shape[0], w.shape[1]).permute(dims, dims + 1, *range(dims)).contiguous())
        return gx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None, None


def conv3x3(x, weight, bias, lrelu=True, pixelnorm=True, chain=False):
    """chain: x is the output of the previous convolution of the same Block and has no other consumer (ops_train._Conv16AC)."""
    flags = (LF_EPI_LRELU if lrelu else 0) | (LF_EPI_PIXELNORM if pixelnorm else 0)
    if _T()._conv16_ac_ok(x, weight):
        return _T()._Conv16AC.apply(x, weight, bias, flags, chain)
    return _Conv3x3.apply(_f32(x), weight, bias, flags)


def _conv1x1_raw(x_ptr_tensor, wpack, bias, N, P, cin, ksl, xbs, xss, cout, y2d, he, flags, yaddr=None, xscale=None):
    """y2d: output buffer; yaddr = (batch_stride, row_stride, slice_channels, slice_stride) or None
    for plain [N*P][cout] rows.  xscale: a factor per (sample, slice, pixel) applied to the operand (lf_conv1x1_fwd_scaled).
    Returns norm or None."""
    L = _lib.lib()
    ybs, yrs, ysc, yss = yaddr if yaddr is not None else (P * cout, cout, 1 << 30, 0)
    fuse_pn = bool(flags & LF_EPI_PIXELNORM) and cout <= 128
    kflags = flags if fuse_pn else (flags & ~LF_EPI_PIXELNORM)
    norm = torch.empty(N * P, device=y2d.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    if xscale is not None:
        assert xscale.numel() == N * ksl * P and xscale.dtype == torch.float32 and xscale.is_contiguous()
        check(L.lf_conv1x1_fwd_scaled(_ptr(x_ptr_tensor), _ptr(xscale), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y2d),
                                      _ptr(norm) if (norm is not None and fuse_pn) else None,
                                      N, P, cin, ksl, xbs, xss, cout, ybs, yrs, ysc, yss, he, kflags, SLOPE, PN_EPS, _stream()),
              'lf_conv1x1_fwd_scaled')
    else:
        check(L.lf_conv1x1_fwd(_ptr(x_ptr_tensor), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y2d),
                               _ptr(norm) if (norm is not None and fuse_pn) else None,
                               N, P, cin, ksl, xbs, xss, cout, ybs, yrs, ysc, yss, he, kflags, SLOPE, PN_EPS, _stream()),
              'lf_conv1x1_fwd')
    if (flags & LF_EPI_PIXELNORM) and not fuse_pn:
        check(L.lf_pixelnorm_fwd(_ptr(y2d), _ptr(y2d), _ptr(norm), N * P, cout, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
    return norm


class _Conv1x1(torch.autograd.Function):
    """Pointwise conv on NC[D]HW (channels-last) tensors; weight [Cout,Cin,1,1[,1]]."""

    @staticmethod
    def forward(ctx, x, weight, bias, flags):
        _req(x, 'x'), _req(weight, 'weight')
        ctx.ac = AUTOCAST is not None
        x = _ac_in(cl(x))
        N, cin = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        cout = weight.shape[0]
        he = he_constant(weight)
        wpack = _pk(weight, 'c1f', lambda w: pack_conv1x1(w.reshape(cout, cin)))
        y = empty_cl((N, cout) + tuple(x.shape[2:]), x.device)
        norm = _conv1x1_raw(x, wpack, bias.detach() if bias is not None else None, N, P, cin, 1, P * cin, 0, cout, y, he, flags)
        ctx.flags, ctx.he = flags, he
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(y, norm, weight, x if need_w else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm, w, x_saved = ctx.saved_tensors
        gp = _epilogue_bwd(cl(gy), y, norm, ctx.flags)
        cout, cin = w.shape[0], w.shape[1]
        gx = gw = gb = None
        with autocast(ctx.ac):
            gp_full, gp = gp, _ac_in(gp)
            if ctx.needs_input_grad[0]:
                wpack_t = _pk(w, 'c1b', lambda t: pack_conv1x1(t.reshape(cout, cin).t()))
                N = gp.shape[0]
                P = gp[0, 0].numel()
                gx = empty_cl((N, cin) + tuple(gp.shape[2:]), gp.device)
                _conv1x1_raw(gp, wpack_t, None, N, P, cout, 1, P * cout, 0, cin, gx, ctx.he, 0)
                gx = _ac_in(gx)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                rows = gp.numel() // cout                                   # channels-last: plain [rows][C] matrices
                gwt, gb = _T().conv_bwd_weight(x_saved.permute(0, *range(2, x_saved.dim()), 1).reshape(rows, cin),
                                          gp.permute(0, *range(2, gp.dim()), 1).reshape(rows, cout), 0, cin, ctx.he,
                                          want_bias=not ctx.ac)
                if ctx.ac and ctx.needs_input_grad[2]:
                    gb = _T().bias_grad(gp_full.permute(0, *range(2, gp_full.dim()), 1).reshape(rows, cout), 0)
                gw = _ac_in(gwt.reshape(w.shape))
        return gx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None, None


def conv1x1(x, weight, bias, lrelu=False, pixelnorm=False):
    flags = (LF_EPI_LRELU if lrelu else 0) | (LF_EPI_PIXELNORM if pixelnorm else 0)
    if _T()._conv16_ac_ok(x, weight):                       # (the encoder's 16 -> 16 output layer: the ring kernel's centre tap)
        return _T()._Conv16AC.apply(x, weight, bias, flags, False)
    return _Conv1x1.apply(_f32(x), weight, bias, flags)


class _FactorProject(torch.autograd.Function):
    """FactorProjection3d2d (modules/geometry.py:731-749): folds the depth axis of a
    channels-last volume into the contraction dimension without copying it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.x_bf16 = x.dtype == torch.bfloat16
        ctx.ac = AUTOCAST is not None
        ctx.mfma = bool(PROJ_MFMA and ctx.ac and ctx.x_bf16 and x.dim() == 5 and x.shape[1] == 16 and weight.shape[0] == 16
                        and x.shape[2] in (16, 32, 64, 128) and (x.shape[3] * x.shape[4]) % 16 == 0 and x.is_cuda)
        if ctx.mfma:
            # round 6 (csrc/lift_mfma.hip): the bf16 volume read as it lies, one MFMA per depth; no fp32 copy of the volume
            L = _lib.lib()
            _req(weight, 'weight')
            x = cl(x)
            N, C, D, H, W = x.shape
            he = he_constant(weight)
            wtab = _pk(weight, 'p16f', lambda w: w.reshape(16, 16, D).permute(2, 0, 1).contiguous().to(torch.bfloat16))
            y = empty_cl((N, 16, H, W), x.device)
            norm = torch.empty(N * H * W, device=x.device, dtype=torch.float32)
            with _timed('proj16_fwd'):
                check(L.lf_proj16_fwd(_ptr(x), _ptr(wtab), _ptr(bias) if bias is not None else None, _ptr(y), _ptr(norm), N * H * W, H * W,
                                      D, he, SLOPE, PN_EPS, _stream()), 'lf_proj16_fwd')
            ctx.flags, ctx.he, ctx.xshape = LF_EPI_LRELU | LF_EPI_PIXELNORM, he, x.shape
            ctx.save_for_backward(y, norm, weight, x)
            return y
        x = _f32(x)
        _req(x, 'x'), _req(weight, 'weight')
        x = cl(x) if ctx.x_bf16 else _ac_in(cl(x))             # (a bf16-stored volume is already rounded)
        N, C, D, H, W = x.shape
        cout = weight.shape[0]
        he = he_constant(weight)                      # fan_in = C*D
        # reference K index = c*D + d; kernel K index = d*C + c
        wpack = _pk(weight, 'fpf', lambda w: pack_conv1x1(w.reshape(cout, C, D).permute(0, 2, 1).reshape(cout, D * C)))
        y = empty_cl((N, cout, H, W), x.device)
        flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
        if C % 4:
            raise NotImplementedError('factor projection needs C % 4 == 0 on the HIP path')
        norm = _conv1x1_raw(x, wpack, bias.detach() if bias is not None else None, N, H * W, C, D,
                            D * H * W * C, H * W * C, cout, y, he, flags)
        ctx.flags, ctx.he, ctx.xshape = flags, he, x.shape
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(y, norm, weight, x if need_w else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm, w, x_saved = ctx.saved_tensors
        gp = _epilogue_bwd(cl(gy), y, norm, ctx.flags)
        N, C, D, H, W = ctx.xshape
        cout = w.shape[0]
        gw = gb = None
        if ctx.mfma:
            L = _lib.lib()
            wtab_t = _pk(w, 'p16b', lambda t: t.reshape(16, 16, D).permute(2, 1, 0).contiguous().to(torch.bfloat16))
            gx = empty_cl16((N, 16, D, H, W), gp.device, True)
            gwt = torch.empty(16, 16 * D, device=gp.device, dtype=torch.float32)
            nb = L.lf_proj16_bwd_scratch_bytes(D)
            scr = torch.empty(nb // 4 + 4, device=gp.device, dtype=torch.float32)
            with _timed('proj16_bwd'):
                check(L.lf_proj16_bwd(_ptr(gp), _ptr(x_saved), _ptr(wtab_t), _ptr(gx), _ptr(gwt), _ptr(scr, True), scr.numel() * 4,
                                      N * H * W, H * W, D, ctx.he, _stream()), 'lf_proj16_bwd')
            with autocast(True):
                if ctx.needs_input_grad[2]:
                    gb = _T().bias_grad(gp.permute(0, 2, 3, 1).reshape(N * H * W, cout), 0)
                gw = _ac_in(gwt.reshape(w.shape))
            return gx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None
        with autocast(ctx.ac):
            gp_full, gp = gp, _ac_in(gp)
            # gx[n,d,p,c] = he * sum_co gp[n,p,co] * W[co, c*D+d]  == pointwise conv with Cout' = D*C
            wt = _pk(w, 'fpb', lambda t: pack_conv1x1(t.reshape(cout, C, D).permute(2, 1, 0).reshape(D * C, cout)))
            # output channel d*C + c of pixel p goes straight to gx[n][d][p][c] (channels-last volume)
            gx = empty_cl((N, C, D, H, W), gp.device)
            _conv1x1_raw(gp, wt, None, N, H * W, cout, 1, H * W * cout, 0, D * C, gx, ctx.he, 0,
                         yaddr=(D * H * W * C, C, C, H * W * C))
            gx = _ac_in(gx)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                # rows = pixels, K = (c, d) in the reference's order c*D + d: one copy of the volume (training only)
                if _lift_ok(C, D):
                    xr = _lift_permute(x_saved, N, H * W, C, D, True, torch.empty(N * H * W, C * D, device=gp.device, dtype=torch.float32))
                else:
                    xr = x_saved.permute(0, 3, 4, 1, 2).reshape(N * H * W, C * D).contiguous()
                gwt, gb = _T().conv_bwd_weight(xr, gp.permute(0, 2, 3, 1).reshape(N * H * W, cout), 0, C * D, ctx.he,
                                          want_bias=not ctx.ac)
                if ctx.ac and ctx.needs_input_grad[2]:
                    gb = _T().bias_grad(gp_full.permute(0, 2, 3, 1).reshape(N * H * W, cout), 0)
                gw = _ac_in(gwt.reshape(w.shape))
        if ctx.x_bf16:
            gx = gx.to(torch.bfloat16)
        return gx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None


def factor_project(x, weight, bias):
    return _FactorProject.apply(x, weight, bias)


import os as _os
PROJ_MFMA = bool(int(_os.environ.get('LF_PROJ_MFMA', '1')))   # the training step's 3-D -> 2-D factor projection as one MFMA kernel each way (csrc/lift_mfma.hip)
LIFT_FUSED = True          # A/B switch of the fused training-path lift (_LiftFused); False: conv1x1 + PixelNorm + permutation


def _lift_permute(src, V, P, c0, S, fold, out):
    check(_lib.lib().lf_lift_permute(_ptr(src), _ptr(out), V, P, c0, S, 1 if fold else 0, _stream()), 'lf_lift_permute')
    return out


def _lift_ok(c0, S):
    return 4 * c0 * (S + 1) * 4 <= 64 * 1024


def lift(x, weight, bias, out_size):
    """FactorProjection2d3d (modules/geometry.py:711-728): 1x1 conv to C0*S channels, LeakyReLU,
    PixelNorm over ALL C0*S channels, viewed as (V,C0,S,H,W).  The fused unfold below is the inference path;
    when a gradient is wanted the same result is composed from the differentiable pointwise conv and a
    layout copy."""
    L = _lib.lib()
    _req(x, 'x')
    x = cl(x)
    V, cin, H, W = x.shape
    cs = weight.shape[0]
    c0 = cs // out_size
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        if LIFT_FUSED and _T()._lift_fused_ok(c0, out_size) and c0 * out_size == cs:
            return _T()._LiftFused.apply(x, weight, bias, out_size)
        y = conv1x1(x, weight, bias, lrelu=True, pixelnorm=True)            # (V, c0*S, H, W), channel = c*S + d
        if _lift_ok(c0, out_size):
            return _T()._LiftView.apply(y, c0, out_size)
        return cl(y.view(V, c0, out_size, H, W))
    he = he_constant(weight)
    wpack = _cached(weight, 'c1f', lambda: pack_conv1x1(weight.reshape(cs, cin)))
    P = H * W
    tmp = torch.empty(V * P, cs, device=x.device, dtype=torch.float32)
    norm = _conv1x1_raw(x, wpack, bias.detach() if bias is not None else None, V, P, cin, 1, P * cin, 0, cs, tmp, he,
                        LF_EPI_LRELU | (LF_EPI_PIXELNORM if cs <= 128 else 0))
    if cs > 128:
        norm = torch.empty(V * P, device=x.device, dtype=torch.float32)
        # norm only: normalisation itself is folded into the unfold pass below
        tmp2 = torch.empty_like(tmp)
        check(L.lf_pixelnorm_fwd(_ptr(tmp), _ptr(tmp2), _ptr(norm), V * P, cs, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
        tmp = tmp2
    out = empty_cl((V, c0, out_size, H, W), x.device)
    check(L.lf_lift_unfold(_ptr(tmp), None, _ptr(out), V, P, c0, out_size, _stream()), 'lf_lift_unfold')
    return out


# ---------------------------------------------------------------------------------------------
# depth-column composites (renderer) and view reductions (fusers)
# ---------------------------------------------------------------------------------------------
class _ColumnSum(torch.autograd.Function):
    """(N,C,D,H,W) -> (N,C,H,W), sum over the depth axis: the 'sum' projection (recon/models.py:436-437)."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        x = cl(_req(_f32(x), 'x'))
        N, C, D, H, W = x.shape
        y = empty_cl((N, C, H, W), x.device)
        with _timed('column_sum'):
            check(L.lf_column_reduce_sum_fwd(_ptr(x), _ptr(y), N, D, H * W, C, _stream()), 'lf_column_reduce_sum_fwd')
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        N, C, D, H, W = ctx.xshape
        g = cl(gy)
        gx = empty_cl(ctx.xshape, g.device)
        check(L.lf_column_reduce_sum_bwd(_ptr(g), _ptr(gx), N, D, H * W, C, _stream()), 'lf_column_reduce_sum_bwd')
        return gx


def column_sum(x):
    return _ColumnSum.apply(x)


class _ColumnSoftmax(torch.autograd.Function):
    """Single-channel logits (N,1,D,H,W) -> (softmax over D (N,1,D,H,W), expected depth sum_d w * linspace(-1,1,D)
    (N,1,H,W)): Photographer._compute_depth_weights / _depth_from_weight (recon/models.py:378-395)."""

    @staticmethod
    def forward(ctx, logits):
        L = _lib.lib()
        lg = _req(logits, 'logits').contiguous()
        N, one, D, H, W = lg.shape
        if one != 1:
            raise ValueError('column_softmax expects single-channel logits')
        w = torch.empty_like(lg)
        zd = torch.empty(N, 1, H, W, device=lg.device, dtype=torch.float32)
        with _timed('column_softmax'):
            check(L.lf_column_softmax_fwd(_ptr(lg), _ptr(w), _ptr(zd), N, D, H * W, _stream()), 'lf_column_softmax_fwd')
        ctx.save_for_backward(w)
        ctx.set_materialize_grads(False)
        return w, zd

    @staticmethod
    def backward(ctx, gw, gzd):
        L = _lib.lib()
        w, = ctx.saved_tensors
        N, _, D, H, W = w.shape
        if gw is None and gzd is None:
            return None
        gw = gw.contiguous() if gw is not None else None
        gzd = gzd.contiguous() if gzd is not None else None
        gl = torch.empty_like(w)
        check(L.lf_column_softmax_bwd(_ptr(w), _ptr(gw) if gw is not None else None, _ptr(gzd) if gzd is not None else None,
                                      _ptr(gl), N, D, H * W, _stream()), 'lf_column_softmax_bwd')
        return gl


def column_softmax(logits):
    return _ColumnSoftmax.apply(logits)


class _ColumnScale(torch.autograd.Function):
    """z (N,C,D,H,W) * w (N,1,D,H,W): the occlusion weighting z * depth_weights_resized (recon/models.py:427-430)."""

    @staticmethod
    def forward(ctx, z, w):
        L = _lib.lib()
        z = cl(_req(_f32(z), 'z'))
        w = _req(w, 'w').contiguous()
        N, C, D, H, W = z.shape
        if tuple(w.shape) != (N, 1, D, H, W) or C % 4:
            raise ValueError('column_scale: w must be (N,1,D,H,W) and C a multiple of 4')
        out = empty_cl(z.shape, z.device)
        check(L.lf_column_scale_fwd(_ptr(z), _ptr(w), _ptr(out), N * D * H * W, C, _stream()), 'lf_column_scale_fwd')
        ctx.save_for_backward(z, w)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        z, w = ctx.saved_tensors
        N, C, D, H, W = z.shape
        g = cl(gout)
        gz = empty_cl(z.shape, z.device) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        if gz is None and gw is None:
            return None, None
        check(L.lf_column_scale_bwd(_ptr(g), _ptr(z), _ptr(w), _ptr(gz) if gz is not None else None,
                                    _ptr(gw) if gw is not None else None, N * D * H * W, C, _stream()), 'lf_column_scale_bwd')
        return gz, gw


def column_scale(z, w):
    return _ColumnScale.apply(z, w)


def _dense_views(z):
    """(B,V,...) -> (B*V,...) with every view dense in ONE common element order (channels-last when the per-view
    tensors are 4-/5-D, which is what the encoder produces: no copy then)."""
    zz = z.reshape(z.shape[0] * z.shape[1], *z.shape[2:])
    return cl(zz) if zz.dim() in (4, 5) else zz.contiguous()


def _dense_views_f32(z):
    return _dense_views(_f32(z))


FUSE_KINDS = {'mean': _lib.LF_FUSE_MEAN, 'max': _lib.LF_FUSE_MAX, 'abs_max': _lib.LF_FUSE_ABSMAX, 'median': _lib.LF_FUSE_MEDIAN}


class _FuseViews(torch.autograd.Function):
    """PoolFuser reductions over the view axis: z (B,V,...) -> (B,1,...)  (recon/fusion.py:45-57)."""

    @staticmethod
    def forward(ctx, z, kind):
        L = _lib.lib()
        z = _f32(z)
        _req(z, 'z')
        B, V = z.shape[0], z.shape[1]
        zz = _dense_views(z)
        n = zz[0].numel()
        out = torch.empty_like(zz[:B])
        need_idx = kind != _lib.LF_FUSE_MEAN and ctx.needs_input_grad[0]
        idx = torch.empty(B, n, device=z.device, dtype=torch.int32) if need_idx else None
        with _timed('fuse_views'):
            for b in range(B):
                check(L.lf_fuse_views_fwd(_ptr(zz[b * V]), _ptr(out[b]), _ptr(idx[b]) if idx is not None else None, kind, V, n, n,
                                          _stream()), 'lf_fuse_views_fwd')
        ctx.meta = (kind, B, V, n, tuple(zz.shape), tuple(z.shape))
        if idx is not None:
            ctx.save_for_backward(idx)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        kind, B, V, n, zshape, orig = ctx.meta
        idx = ctx.saved_tensors[0] if ctx.saved_tensors else None
        g = g.squeeze(1)
        g = cl(g) if g.dim() in (4, 5) else g.contiguous()
        gz = torch.empty(zshape, device=g.device, dtype=torch.float32,
                         memory_format=(torch.channels_last_3d if len(zshape) == 5 else torch.channels_last if len(zshape) == 4
                                        else torch.contiguous_format))
        for b in range(B):
            check(L.lf_fuse_views_bwd(_ptr(g[b]), _ptr(idx[b]) if idx is not None else None, _ptr(gz[b * V]), kind, V, n, n,
                                      _stream()), 'lf_fuse_views_bwd')
        return gz.view(orig), None


def fuse_views(z, pool_type):
    if pool_type not in FUSE_KINDS:
        raise ValueError(f'Unknown pool_type value {pool_type}')
    return _FuseViews.apply(z, FUSE_KINDS[pool_type])


class _FuseBlend(torch.autograd.Function):
    """BlendFuser.forward (recon/fusion.py:139-148): z (B,V,C,D,H,W), logits (B,V,1,D,H,W) ->
    (sum_v z * softmax_v(logits) (B,1,C,D,H,W), weights (B,V,1,D,H,W))."""

    @staticmethod
    def forward(ctx, z, logits):
        L = _lib.lib()
        z = _f32(z)
        _req(z, 'z'), _req(logits, 'logits')
        B, V, C = z.shape[:3]
        zz = _dense_views(z)                                    # (B*V,C,D,H,W) channels-last
        lg = logits.reshape(B * V, -1).contiguous()             # (B*V, rows)
        rows = lg.shape[1]
        if zz[0].numel() != rows * C or C % 4:
            raise ValueError('fuse_blend: shapes of z and logits do not match / C % 4 != 0')
        w = torch.empty_like(lg)
        out = torch.empty_like(zz[:B])
        with _timed('fuse_blend'):
            for b in range(B):
                check(L.lf_fuse_blend_fwd(_ptr(zz[b * V]), _ptr(lg[b * V]), _ptr(w[b * V]), _ptr(out[b]), V, rows, C, rows * C, rows,
                                          _stream()), 'lf_fuse_blend_fwd')
        ctx.save_for_backward(zz, w)
        ctx.meta = (B, V, C, rows, tuple(z.shape), tuple(logits.shape))
        ctx.mark_non_differentiable(w)
        return out.unsqueeze(1), w.view(logits.shape)

    @staticmethod
    def backward(ctx, g, _gw):
        L = _lib.lib()
        zz, w = ctx.saved_tensors
        B, V, C, rows, zshape, lshape = ctx.meta
        g = cl(g.squeeze(1))
        gz = torch.empty_like(zz) if ctx.needs_input_grad[0] else None
        gl = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        if gz is None and gl is None:
            return None, None
        for b in range(B):
            check(L.lf_fuse_blend_bwd(_ptr(g[b]), _ptr(zz[b * V]), _ptr(w[b * V]), _ptr(gz[b * V]) if gz is not None else None,
                                      _ptr(gl[b * V]) if gl is not None else None, V, rows, C, rows * C, rows, _stream()),
                  'lf_fuse_blend_bwd')
        return (gz.view(zshape) if gz is not None else None), (gl.view(lshape) if gl is not None else None)


def fuse_blend(z, logits):
    return _FuseBlend.apply(z, logits)


# ---------------------------------------------------------------------------------------------
# 2-D grid sampling (image crops / uncrops / IBR warps)
# ---------------------------------------------------------------------------------------------
class _GridSample2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, grid, bilinear, border):
        L = _lib.lib()
        _req(img, 'image')
        img, grid = img.float().contiguous(), grid.float().contiguous()
        N, C, H, W = img.shape
        Ho, Wo = grid.shape[1], grid.shape[2]
        if grid.shape[0] != N or grid.shape[3] != 2:
            raise ValueError('grid must be (N, Ho, Wo, 2)')
        out = torch.empty(N, C, Ho, Wo, device=img.device, dtype=torch.float32)
        check(L.lf_grid_sample2d_fwd(_ptr(img), _ptr(grid), _ptr(out), N, C, H, W, Ho, Wo, int(bilinear), int(border), _stream()),
              'lf_grid_sample2d_fwd')
        ctx.save_for_backward(img, grid)
        ctx.mode = (int(bilinear), int(border))
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        img, grid = ctx.saved_tensors
        N, C, H, W = img.shape
        Ho, Wo = grid.shape[1], grid.shape[2]
        gimg = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        ggrid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        if gimg is None and ggrid is None:
            return None, None, None, None
        check(L.lf_grid_sample2d_bwd(_ptr(img), _ptr(grid), _ptr(gout.float().contiguous()),
                                     _ptr(gimg) if gimg is not None else None, _ptr(ggrid) if ggrid is not None else None,
                                     N, C, H, W, Ho, Wo, ctx.mode[0], ctx.mode[1], _stream()), 'lf_grid_sample2d_bwd')
        return gimg, ggrid, None, None


def grid_sample2d(img, grid, mode='bilinear', padding_mode='zeros'):
    """F.grid_sample(img, grid, mode, padding_mode, align_corners=False) for 4-D inputs on the HIP path."""
    if mode not in ('bilinear', 'nearest') or padding_mode not in ('zeros', 'border'):
        raise ValueError(f'unsupported mode {mode!r} / padding {padding_mode!r}')
    return _GridSample2d.apply(img, grid, mode == 'bilinear', padding_mode == 'border')


# ---------------------------------------------------------------------------------------------
# standalone PixelNorm / rescale
# ---------------------------------------------------------------------------------------------
class _PixelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        _req(x, 'x')
        x = cl(x)
        rows = x.numel() // x.shape[1]
        y = torch.empty_like(x, memory_format=torch.preserve_format)
        norm = torch.empty(rows, device=x.device, dtype=torch.float32)
        check(L.lf_pixelnorm_fwd(_ptr(x), _ptr(y), _ptr(norm), rows, x.shape[1], PN_EPS, _stream()), 'lf_pixelnorm_fwd')
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm = ctx.saved_tensors
        return _epilogue_bwd(cl(gy), y, norm, LF_EPI_PIXELNORM)


def pixelnorm(x):
    return _PixelNorm.apply(x)


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, linear, up):
        L = _lib.lib()
        _req(x, 'x')
        x = cl(x)
        dims = x.dim() - 2
        N, C = x.shape[0], x.shape[1]
        D, H, W = (x.shape[2:] if dims == 3 else (1,) + tuple(x.shape[2:]))
        out_sp = tuple((s * 2 if up else s // 2) for s in x.shape[2:])
        y = empty_cl((N, C) + out_sp, x.device)
        check(L.lf_resize_fwd(_ptr(x), _ptr(y), dims, N, D, H, W, C, linear, up, _stream()), 'lf_resize_fwd')
        ctx.meta = (dims, N, C, D, H, W, linear, up, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        dims, N, C, D, H, W, linear, up, xshape = ctx.meta
        g = cl(gy)
        gx = empty_cl(xshape, g.device)
        check(L.lf_resize_bwd(_ptr(g), _ptr(gx), dims, N, D, H, W, C, linear, up, _stream()), 'lf_resize_bwd')
        return gx, None, None


def interpolate(x, scale_factor, mode):
    """Block-end rescale (modules/__init__.py:18-36): x2 / x0.5, nearest or (bi|tri)linear with
    align_corners=False -- the only combinations the block grammar (I/U/D tokens) can produce."""
    if scale_factor not in (2.0, 0.5) or mode not in ('nearest', 'bilinear', 'trilinear'):
        raise NotImplementedError(f'rescale {scale_factor} / {mode} is not produced by the block grammar')
    return _Resize.apply(x, 0 if mode == 'nearest' else 1, 1 if scale_factor == 2.0 else 0)


def resize_nearest_to(x, size):
    """F.interpolate(x, size) (nearest) for cubic / square maps whose size ratio is a power of two -- what the occlusion
    U-Net's logits need when its resolution differs from the volume's (reference recon/models.py:385) -- on lf_resize_*."""
    cur = x.shape[-1]
    if any(d != cur for d in x.shape[2:]):
        raise NotImplementedError('resize_nearest_to expects equal spatial extents')
    while cur < size:
        x, cur = interpolate(x, 2.0, 'nearest'), cur * 2
    while cur > size:
        if cur % 2:
            break
        x, cur = interpolate(x, 0.5, 'nearest'), cur // 2
    if cur != size:
        raise NotImplementedError(f'resize from {x.shape[-1]} to {size} is not a power-of-two ratio')
    return x
