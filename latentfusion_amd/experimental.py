"""Wrappers of the EXPERIMENTAL entry points (include/lf_hip_experimental.h): kernels a measured alternative superseded and the
A/B reference paths, kept for tools/ and the tests that pin them -- nothing on a default path or a documented preset uses them.

  conv3d_c16_bf16          the first bf16 16 -> 16 kernel (the ring form lf_conv3d_c16_ring_bf16 is what the training step runs)
  conv3d_c16_wino_split    Winograd with three-term f16 products (engine conv_mode 'winograd_f16x3')
  conv3d_c16_wino_projbwd  factor projection backward fused into the first data-gradient conv (1.45 vs 0.40 + 0.95 ms)
  conv3d_wino_gemm         wide Winograd convolution in three stages on the library GEMM (ops.WIDE_CONV_MODE = 'bmm')"""
import math  # noqa: F401

import torch

from . import _lib, ops
from ._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, LF_MAP_C2O, LF_MAP_COEFS, LF_MAP_O2C, check  # noqa: F401
from .ops import (PN_EPS, SLOPE, _WINO_G, _ptr, _stream, _timed, empty_cl)  # noqa: F401


def pack_conv3d_c16_bf16(weight, transpose=False):
    """[16,16,3,3,3] -> bf16 [27 taps][64 lanes][4] for lf_conv3d_c16_bf16: lane = (cin // 4) * 16 + cout."""
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=(2, 3, 4))
    assert tuple(w.shape) == (16, 16, 3, 3, 3)
    u = w.reshape(16, 4, 4, 27).permute(3, 1, 0, 2).reshape(27, 64, 4)      # [tap][kg][cout][i] -> [tap][lane][i]
    return u.contiguous().to(torch.bfloat16)


def conv3d_c16_bf16(x, wpack, bias, he, flags, round_out):
    """Launch lf_conv3d_c16_bf16 on a channels-last (N,16,D,H,W) tensor."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    with _timed('conv3d_c16_bf16'):
        check(L.lf_conv3d_c16_bf16(_ptr(x), _ptr(wpack), _ptr(bias) if bias is not None else None, _ptr(y),
                                   _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS, round_out,
                                   _stream()), 'lf_conv3d_c16_bf16')
    return y, norm


def pack_conv3d_c16_wino_split(weight, transpose=False):
    """Winograd-domain weights split into f16 hi + lo: [4 a][16 bc][hi, lo][64 lanes][4 cin] for
    lf_conv3d_c16_wino_split (lane = (cin // 4) * 16 + cout)."""
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=(2, 3, 4))
    assert tuple(w.shape) == (16, 16, 3, 3, 3)
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    U = torch.einsum('ai,bj,ck,omijk->abcom', G, G, G, w.double()).float()  # [a][b][c][cout][cin]
    U = U.reshape(4, 16, 16, 4, 4).permute(0, 1, 3, 2, 4).reshape(4, 16, 64, 4)   # [a][bc][lane = kg*16 + cout][j]
    hi = U.half()
    lo = (U - hi.float()).half()
    return torch.stack((hi, lo), dim=2).contiguous()                        # [4][16][2][64][4]


def conv3d_c16_wino_split(x, upack, bias, he, flags, prev=None, amax_in=None, amax_out=None):
    """Launch lf_conv3d_c16_wino_split on a channels-last (N,16,D,H,W) tensor."""
    L = _lib.lib()
    N, _, D, H, W = x.shape
    y = empty_cl((N, 16, D, H, W), x.device)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if (flags & LF_EPI_PIXELNORM) else None
    py, pn, pf = (prev[0], prev[1], prev[2]) if prev is not None else (None, None, 0)
    with _timed('conv3d_c16_wino_split'):
        check(L.lf_conv3d_c16_wino_split(_ptr(x), _ptr(upack), _ptr(bias) if bias is not None else None, _ptr(y),
                                         _ptr(norm) if norm is not None else None, N, D, H, W, he, flags, SLOPE, PN_EPS,
                                         _ptr(py) if py is not None else None, _ptr(pn) if pn is not None else None, pf,
                                         _ptr(amax_in) if amax_in is not None else None,
                                         _ptr(amax_out) if amax_out is not None else None, _stream()),
              'lf_conv3d_c16_wino_split')
    return y, norm


def conv3d_c16_wino_projbwd(gp, proj_wtA, proj_he, act, act_norm, act_flags, upack_t, he, prev=None):
    """lf_conv3d_c16_wino_projbwd: data gradient of the factor projection AND of the last camera block's convolution in one
    launch (the gradient volume between them is formed on chip).  gp: (N,16,H,W) channels-last gradient w.r.t. the
    projection's pre-activation; act / act_norm: the block's saved output; prev: (y, norm, flags) of the layer feeding the
    block.  Returns the (N,16,D,H,W) gradient w.r.t. that layer's pre-activation (or the block's input if prev is None)."""
    L = _lib.lib()
    N, _, D, H, W = act.shape
    g = empty_cl((N, 16, D, H, W), act.device)
    py, pn, pf = (prev[0], prev[1], prev[2]) if prev is not None else (None, None, 0)
    with _timed('conv3d_c16_wino_projbwd'):
        check(L.lf_conv3d_c16_wino_projbwd(_ptr(gp), _ptr(proj_wtA), proj_he, _ptr(act),
                                           _ptr(act_norm) if act_norm is not None else None, act_flags, _ptr(upack_t), _ptr(g),
                                           N, D, H, W, he, SLOPE, _ptr(py) if py is not None else None,
                                           _ptr(pn) if pn is not None else None, pf, _stream()), 'lf_conv3d_c16_wino_projbwd')
    return g


def pack_conv3d_wino_gemm(weight, transpose=False):
    """[Cout,Cin,3,3(,3)] -> U [64 | 16 f][Cin][Cout], f = (a*4+b)*4+c (3-D) or b*4+c (2-D), for the three-stage
    Winograd path (lf_wino3d_* / lf_wino2d_*): U[f][ci][co] = ((G x ..) w)[co][ci][f], evaluated in fp64."""
    w = weight.detach()
    if transpose:
        w = w.transpose(0, 1).flip(dims=tuple(range(2, w.dim())))
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    if w.dim() == 5:
        U = torch.einsum('ai,bj,ck,omijk->abcmo', G, G, G, w.double())      # [a][b][c][cin][cout]
        return U.reshape(64, w.shape[1], w.shape[0]).float().contiguous()
    U = torch.einsum('bj,ck,omjk->bcmo', G, G, w.double())                  # [b][c][cin][cout]
    return U.reshape(16, w.shape[1], w.shape[0]).float().contiguous()


def conv3d_wino_gemm(x, U, bias, he, flags):
    """Wide 3-D conv as Winograd F(2x2x2,3x3x3): input transform (HIP) -> 64 batched fp32 GEMMs (rocBLAS via
    torch.bmm) -> output transform with the fused epilogue (HIP).  Returns (y, norm or None).  (A/B reference of the
    fused path: ops.WIDE_CONV_MODE = 'bmm'.)"""
    L = _lib.lib()
    if x.dim() == 4:
        return _conv2d_wino_gemm(x, U, bias, he, flags)
    N, cin, D, H, W = x.shape
    cout = U.shape[2]
    T = L.lf_wino3d_tiles(N, D, H, W)
    V = torch.empty(64, T, cin, device=x.device, dtype=torch.float32)
    with _timed('wino3d_input'):
        check(L.lf_wino3d_input_transform(_ptr(x), _ptr(V), N, D, H, W, cin, _stream()), 'lf_wino3d_input_transform')
    with _timed('wino3d_gemm'):
        M = torch.bmm(V, U)
    del V
    y = empty_cl((N, cout, D, H, W), x.device)
    pn = bool(flags & LF_EPI_PIXELNORM)
    norm = torch.empty(N * D * H * W, device=x.device, dtype=torch.float32) if pn else None
    with _timed('wino3d_output'):
        check(L.lf_wino3d_output_transform(_ptr(M), _ptr(bias) if bias is not None else None, _ptr(y),
                                           _ptr(norm) if norm is not None else None, N, D, H, W, cout, he, flags, SLOPE,
                                           PN_EPS, _stream()), 'lf_wino3d_output_transform')
    if pn and cout > 256:
        check(L.lf_pixelnorm_fwd(_ptr(y), _ptr(y), _ptr(norm), N * D * H * W, cout, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
    return y, norm


def _conv2d_wino_gemm(x, U, bias, he, flags):
    L = _lib.lib()
    N, cin, H, W = x.shape
    cout = U.shape[2]
    T = L.lf_wino2d_tiles(N, H, W)
    V = torch.empty(16, T, cin, device=x.device, dtype=torch.float32)
    with _timed('wino2d_input'):
        check(L.lf_wino2d_input_transform(_ptr(x), _ptr(V), N, H, W, cin, _stream()), 'lf_wino2d_input_transform')
    with _timed('wino2d_gemm'):
        M = torch.bmm(V, U)
    del V
    y = empty_cl((N, cout, H, W), x.device)
    pn = bool(flags & LF_EPI_PIXELNORM)
    norm = torch.empty(N * H * W, device=x.device, dtype=torch.float32) if pn else None
    with _timed('wino2d_output'):
        check(L.lf_wino2d_output_transform(_ptr(M), _ptr(bias) if bias is not None else None, _ptr(y),
                                           _ptr(norm) if norm is not None else None, N, H, W, cout, he, flags, SLOPE, PN_EPS,
                                           _stream()), 'lf_wino2d_output_transform')
    if pn and cout > 256:
        check(L.lf_pixelnorm_fwd(_ptr(y), _ptr(y), _ptr(norm), N * H * W, cout, PN_EPS, _stream()), 'lf_pixelnorm_fwd')
    return y, norm


# ---------------------------------------------------------------------------------------------------------------------------
# Render-loop engine variants that were built, measured and NOT adopted (evidence under profiles/): kept so that the A/B
# tools and the tests that pin their correctness keep running.  The estimators select this class when one of its options
# is asked for (conv_mode='winograd_f16x3', fuse_projection containing 'bwd', engine_streams > 1, engine_graph=True).
# ---------------------------------------------------------------------------------------------------------------------------
def _engine_base():
    from .engine import RenderLoopEngine
    return RenderLoopEngine


def make_engine_x():
    Base = _engine_base()

    class RenderLoopEngineX(Base):
        """RenderLoopEngine plus:
          conv_mode 'winograd_f16x3'   Winograd with fp32 transforms and three-term f16 products (22-bit operands: narrower
                                       arithmetic than the reference's fp32 by construction);
          fuse_projection 'bwd'        projection backward inside the first data-gradient conv (1.45 vs 0.40 + 0.95 ms,
                                       profiles/r04_proj_fuse_ab.txt);
          set_streams(k)               hypothesis groups on k HIP streams (no gain, profiles/r03_stream_groups_ab.json);
          forward_backward_graph       hipGraph replay of one evaluation (186.8 vs 187.0 it/s, profiles/r04_hipgraph_ab.json)."""
        CONV_MODES = Base.CONV_MODES + ('winograd_f16x3',)
        FUSE_FORMS = ('fwd', 'bwd')

        def __init__(self, photographer, z_obj, target_obs, loss_weights, conv_mode='auto', fuse_projection=None):
            wino_split = conv_mode == 'winograd_f16x3'
            super().__init__(photographer, z_obj, target_obs, loss_weights, 'f16x3' if wino_split else conv_mode, fuse_projection)
            if wino_split:
                if self.generic_tail:
                    raise NotImplementedError("the split-precision conv modes drive the plain 'factor' renderer only")
                self.conv_mode = 'winograd_f16x3'
                self.split = [(pack_conv3d_c16_wino_split(w), pack_conv3d_c16_wino_split(w, transpose=True)) for w, *_ in self.convs]
                # trilinear resampling is a convex combination: max|x0| <= max|z_obj|
                self.z_amax = ops.amax_buffer(self.z.abs().max(), self.dev)
            self.streams, self._side_streams = 1, []
            self._packs_built = False
            self._graph = None

        def _conv_fwd(self, li, x, flags, depth_inner=False):
            if self.conv_mode == 'winograd_f16x3':
                _w, b, he, _wp, _wt = self.convs[li]
                return conv3d_c16_wino_split(x, self.split[li][0], b, he, flags, amax_in=self.z_amax if li == 0 else None) + (None,)
            return super()._conv_fwd(li, x, flags, depth_inner)

        def _conv_bwd(self, i, g, prev, amax):
            if self.conv_mode == 'winograd_f16x3':
                he = self.convs[i][2]
                return conv3d_c16_wino_split(g, self.split[i][1], None, he, 0, prev=prev, amax_in=amax[i + 1], amax_out=amax[i])[0]
            return super()._conv_bwd(i, g, prev, amax)

        def _proj_bwd_fused(self, gp, acts, norms, flags):
            if 'bwd' not in self.fuse_projection:
                return None
            nconv = len(self.convs)
            i = nconv - 1
            prev = (acts[i], norms[i - 1], flags) if i > 0 else None
            g = conv3d_c16_wino_projbwd(gp, self.proj_fused[1], self.proj[2], acts[nconv], norms[nconv - 1], flags,
                                        self.wino[i][1], self.convs[i][2], prev=prev)
            return g, nconv - 2

        def set_streams(self, k):
            """Number of hypothesis groups evaluated concurrently on separate HIP streams (1 = everything on the current stream)."""
            self.streams = max(1, int(k))
            while len(self._side_streams) < self.streams:
                self._side_streams.append(torch.cuda.Stream(device=self.dev))
            return self

        def _run(self, params, intr, z_span, need_grad, zt, masked_depth):
            """With `streams` = k > 1 the N hypotheses are evaluated as k independent groups on k HIP streams: hypotheses do not
            interact (the reference optimises N separate cameras, estimation.py:580-594); same kernels, same per-hypothesis
            arithmetic: bit-identical to the single-stream evaluation (tests/test_engine_gpu.py)."""
            n = params.shape[0]
            k = min(self.streams, n)
            if k > 1 and not self._packs_built:
                # weight packs are memoised on the parameters when the HOST enqueues the packing kernels (ops._cached), with no
                # stream attached: the first evaluation therefore runs on the current stream alone
                k = 1
            self._packs_built = True
            if k <= 1:
                return self._forward_backward_group(params, intr, z_span, need_grad, 1.0, zt, masked_depth)
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            bounds = [(n * i) // k for i in range(k + 1)]
            outs = []
            for i in range(k):
                b, e = bounds[i], bounds[i + 1]
                st = self._side_streams[i]
                st.wait_event(ready)
                params.record_stream(st)                           # (made on the main stream, read on the side streams)
                intr.record_stream(st)
                if zt is not None:
                    zt.record_stream(st)
                with torch.cuda.stream(st):
                    # d(mean over all N) = (group size / N) x d(mean over the group)
                    lo, gp = self._forward_backward_group(params[b:e], intr[b:e], z_span, need_grad, (e - b) / n,
                                                          zt[b:e] if zt is not None else None, masked_depth)
                    done = torch.cuda.Event()
                    done.record(st)
                outs.append((lo, gp, done))
            for lo, gp, done in outs:
                main.wait_event(done)
                lo.record_stream(main)
                if gp is not None:
                    gp.record_stream(main)
            losses = torch.cat([o[0] for o in outs], dim=0)
            gparams = torch.cat([o[1] for o in outs], dim=0) if need_grad else None
            return losses, gparams

        def forward_backward_graph(self, camera, params):
            """forward_backward(need_grad=True) replayed from a hipGraph (stream capture through torch.cuda.graph of the C-ABI
            launches).  Conditions: `params` is the SAME (N,10) device tensor on every call, the loss weights have not been
            re-set, no latent term, one stream, no kernel timer.  Returns the graph's static output tensors (overwritten by the
            next replay).  A model whose renderer parameters still require grad is frozen for warm-up and capture (replays need
            no autograd state)."""
            from .engine import camera_intrinsics
            if ops.KERNEL_TIMER is not None or self.streams > 1 or self.w_latent != 0.0:
                return self.forward_backward(camera, need_grad=True, params=params)
            pc = params.detach()
            if not pc.is_contiguous():
                raise ValueError('forward_backward_graph needs a contiguous (N,10) parameter block')
            K = camera.intrinsic
            if self._intr is None or self._intr[0] is not K or self._intr[1] != K._version:
                self._intr = (K, K._version, camera_intrinsics(camera))
            intr = self._intr[2]
            # (the captured kernels hold raw pointers: the graph is valid for exactly these tensors -- compared by identity, and
            # kept alive in the tuple below, so a freed buffer's address cannot come back as a false match)
            key = (pc.data_ptr(), tuple(pc.shape), intr.data_ptr(), float(camera.z_span))
            if self._graph is None or self._graph[0] != key or self._graph[6] is not self.weights or self._graph[7] is not params:
                live = [p for p in self._params if p.requires_grad]
                for p in live:
                    p.requires_grad_(False)
                try:
                    cur = torch.cuda.current_stream()
                    side = torch.cuda.Stream(device=self.dev)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side), torch.no_grad():        # warm-up: weight packs, allocator pools, code objects
                        for _ in range(2):
                            self._forward_backward_group(pc, intr, float(camera.z_span), True, 1.0)
                    cur.wait_stream(side)
                    self._packs_built = True
                    g = torch.cuda.CUDAGraph()
                    with torch.no_grad(), torch.cuda.graph(g):
                        lo, gp = self._forward_backward_group(pc, intr, float(camera.z_span), True, 1.0)
                finally:
                    for p in live:
                        p.requires_grad_(True)
                self._graph = (key, g, lo, gp, pc, intr, self.weights, params)        # (keeps the captured inputs alive)
            self._graph[1].replay()
            return self._graph[2], self._graph[3]
    return RenderLoopEngineX


_ENGINE_X = None


def RenderLoopEngineX(*args, **kwargs):
    """Constructs the experimental engine (the class is built on first use: engine.py imports nothing from here)."""
    global _ENGINE_X
    if _ENGINE_X is None:
        _ENGINE_X = make_engine_x()
    return _ENGINE_X(*args, **kwargs)
