"""Fused render-and-score engine of the gradient pose loop.

One call = the forward AND backward of
    camera params -> coefficient blocks -> O2C resample -> camera conv3d blocks -> factor projection
                  -> 2-D decoder + heads -> interpret_logits -> uncrop -> pose loss
for N pose hypotheses of one latent object, returning the per-sample losses and
d(mean weighted loss)/d(log_quaternion, translation, viewport).

Compared with driving the same kernels through the generic autograd modules this removes the
~600 tiny ATen launches per iteration of the camera algebra and of the loss (they become
lf_camera_coefs, lf_pose_loss_fwd/bwd and lf_camera_coefs_bwd), keeps the 3-D activations in
preallocated buffers and sequences the heavy kernels explicitly.  A PLAIN 2-D decoder (blocks without rescaling or skip
concatenations: the SYN family) is sequenced explicitly as well since round 4 -- every data-gradient kernel applies the
LeakyReLU' / PixelNorm' of the layer feeding it, so no separate epilogue-backward pass runs; a generic U-Net (the released
model) runs through the autograd ops on its (N,C,h,w) maps.

Numerically it is the same computation as Photographer.decode + default_pose_loss
(reference recon/models.py:397-505, pose/estimation.py:70-118); tests/test_engine_gpu.py checks it
against the module path and against the reference's golden loop trace.
"""
import torch

from . import _lib, ops
from ._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, LF_MAP_O2C, check

NCOEF = 24
NPAR = 10


def _s():
    return torch.cuda.current_stream().cuda_stream


class _CameraCoefs(torch.autograd.Function):
    """(N,10) camera parameters -> (N,24) coefficient blocks (lf_camera_coefs)."""

    @staticmethod
    def forward(ctx, params, intr, cube, z_span, h, w):
        L = _lib.lib()
        n = params.shape[0]
        coefs = torch.empty(n, NCOEF, device=params.device, dtype=torch.float32)
        jac = torch.empty(n, NCOEF, NPAR, device=params.device, dtype=torch.float32)
        check(L.lf_camera_coefs(params.contiguous().data_ptr(), intr.contiguous().data_ptr(), cube, z_span, h, w,
                                coefs.data_ptr(), jac.data_ptr(), n, _s()), 'lf_camera_coefs')
        ctx.save_for_backward(jac)
        return coefs

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        jac, = ctx.saved_tensors
        n = jac.shape[0]
        gp = torch.empty(n, NPAR, device=jac.device, dtype=torch.float32)
        check(L.lf_camera_coefs_bwd(g.contiguous().data_ptr(), jac.data_ptr(), gp.data_ptr(), n, _s()), 'lf_camera_coefs_bwd')
        return gp, None, None, None, None, None


def camera_params(camera):
    return torch.cat((camera.log_quaternion, camera.translation, camera.viewport), dim=1)


def camera_intrinsics(camera):
    K = camera.intrinsic
    return torch.stack((K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]), dim=1).contiguous()


def camera_coefs(camera, cube_size, crop_h, crop_w):
    """Differentiable (w.r.t. log_quaternion / translation / viewport) coefficient blocks."""
    return _CameraCoefs.apply(camera_params(camera), camera_intrinsics(camera), float(cube_size), float(camera.z_span),
                              int(crop_h), int(crop_w))


class _PoseLoss(torch.autograd.Function):
    """Fused default_pose_loss over head logits.  Returns (total (N,), components (N,4))."""

    @staticmethod
    def forward(ctx, logits, coefs, tdepth, tmask, weights, H, W):
        L = _lib.lib()
        lg = ops.cl(logits)                                   # (N,2,h,w) channels-last == [N][h*w][2]
        n, _, h, w = lg.shape
        nbytes = L.lf_pose_loss_scratch_bytes(n, h, w, H, W)
        scratch = torch.empty(nbytes // 4 + 1, device=lg.device, dtype=torch.float32)
        sums = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        losses = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        gsums = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        cf = coefs.detach().contiguous()
        check(L.lf_pose_loss_fwd(lg.data_ptr(), cf.data_ptr(), tdepth.data_ptr(), tmask.data_ptr(), weights.data_ptr(),
                                 sums.data_ptr(), losses.data_ptr(), gsums.data_ptr(), scratch.data_ptr(),
                                 scratch.numel() * 4, n, h, w, H, W, _s()), 'lf_pose_loss_fwd')
        ctx.save_for_backward(lg, cf, tdepth, tmask, gsums, scratch)
        ctx.dims = (n, h, w, H, W)
        ctx.mark_non_differentiable(losses)
        return losses[:, 4].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        L = _lib.lib()
        lg, cf, tdepth, tmask, gsums, scratch = ctx.saved_tensors
        n, h, w, H, W = ctx.dims
        # gsums were formed for d(mean_n total); rescale to the incoming per-sample gradient
        gs = (gsums * (g_total * n).unsqueeze(1)).contiguous()
        glogits = torch.empty_like(lg)
        gcoefs = torch.zeros(n, NCOEF, device=lg.device, dtype=torch.float32)
        check(L.lf_pose_loss_bwd(lg.data_ptr(), cf.data_ptr(), tdepth.data_ptr(), tmask.data_ptr(), gs.data_ptr(),
                                 glogits.data_ptr(), gcoefs.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                 n, h, w, H, W, _s()), 'lf_pose_loss_bwd')
        return glogits, gcoefs, None, None, None, None, None


def pose_loss(logits, coefs, tdepth, tmask, weights, H, W):
    return _PoseLoss.apply(logits, coefs, tdepth, tmask, weights, H, W)


class RenderLoopEngine:
    """Explicit forward+backward of N pose hypotheses through a Photographer and the fused pose loss.

    An engine is a per-(object, target) INFERENCE object, driven from one host thread: it caches packed weights, the output of
    the object-frame blocks and the resident volume at construction, and it freezes the renderer's trainable parameters for
    the duration of a call (their `requires_grad` flags are restored on return).  Rebuild it after any weight update; do not
    share the model with a concurrently running training step."""

    LOSS_KEYS = ('depth', 'ov_depth', 'iou', 'mask')
    EXPLICIT_DECODER = True          # A/B switch (tools/engine_ab.py): False runs a plain 2-D decoder through autograd like a generic one
    EXPLICIT_OCCLUSION = True        # A/B switch (tools/variant_probe.py): False runs the occlusion module through autograd

    @staticmethod
    def supports(photographer, loss_weights):
        """Renderers the engine sequences: the 'factor' projection (explicit kernels end to end) and -- through the
        depth-column kernels lf_column_softmax_* / lf_column_scale_* / lf_column_reduce_sum_* -- the 'sum' projection and the
        occlusion module (reference recon/models.py:378-395,427-437)."""
        return (photographer.projection_type in ('factor', 'sum')
                and not photographer.skip_connections
                and all(b.interpolate is None for b in photographer.camera_blocks)
                and photographer.predict_depth and photographer.predict_mask and not photographer.predict_color
                and photographer.camera_config[-1] % 4 == 0)

    CONV_MODES = ('auto', 'fp32', 'winograd', 'f16x3')
    FUSE_FORMS = ('fwd',)

    def __init__(self, photographer, z_obj, target_obs, loss_weights, conv_mode='auto', fuse_projection=None):
        """conv_mode selects the kernels of the 16->16 camera-block convolutions:
        'fp32'     direct implicit-GEMM on the fp32 MFMA (works for every channel count);
        'winograd' F(2x2x2,3x3x3) minimal filtering, all-fp32 arithmetic (fp32 MFMA + fp32 transforms);
        'f16x3'    direct, each fp32 product from three f16 MFMAs with fp32 accumulation (documented preset: bench `alt`);
        'auto'     (default) 'winograd' when the blocks are 16->16 or >= 64 channels wide, else 'fp32'.
        All stay within the fp32 kernel's distance of an fp64 reference (tests/test_engine_gpu.py).
        fuse_projection: None = the default (factor projection forward fused into the last block's Winograd launch where the
        shapes allow), False = two launches, ('fwd',) = required.
        (Measured-and-rejected variants -- three-term Winograd products, the fused projection backward, hypothesis groups on
        several streams, hipGraph replay -- live in experimental.RenderLoopEngineX.)"""
        if conv_mode not in self.CONV_MODES:
            raise ValueError(conv_mode)
        self.ph = photographer
        self.cube = photographer.cube_size
        dev = z_obj.device
        self.z = ops.cl(z_obj.reshape(1, *z_obj.shape[-4:]))              # (1,C,S,S,S) channels-last, resident
        if len(photographer.object_blocks):
            # object-frame blocks act on the volume BEFORE the camera transform (reference recon/models.py:410-415): every
            # hypothesis sees the same input, so they are evaluated once per object here, not once per hypothesis per call
            with torch.no_grad():
                zo = self.z
                for blk in photographer.object_blocks:
                    zo = blk(zo)
            self.z = ops.cl(zo.detach())
        self.S = self.z.shape[-1]
        self.C = self.z.shape[1]
        self.crop = photographer.out_size
        # (the loss kernels take raw device pointers: a host-resident target must not reach them)
        self.tdepth = target_obs.depth.reshape(-1).float().to(dev).contiguous()
        self.tmask = target_obs.mask.reshape(-1).float().to(dev).contiguous()
        self.H, self.W = target_obs.depth.shape[-2:]
        self.set_weights(loss_weights)
        self.convs = []
        for blk in photographer.camera_blocks:
            for conv in (blk.conv1, blk.conv2):
                w = conv.module.weight
                self.convs.append((w, conv.bias, ops.he_constant(w), ops.pack_conv3x3(w), ops.pack_conv3x3(w, transpose=True)))
        c16 = self.C == 16 and len(self.convs) > 0 and all(tuple(w.shape[:2]) == (16, 16) for w, *_ in self.convs)
        wide = len(self.convs) > 0 and all(w.shape[0] >= 64 and w.shape[1] >= 64 and w.shape[0] % 4 == 0
                                           and w.shape[1] % 4 == 0 for w, *_ in self.convs)
        if conv_mode == 'auto':
            conv_mode = 'winograd' if (c16 or wide) else 'fp32'
        if conv_mode not in ('fp32', 'winograd') and not c16:
            raise NotImplementedError(f'{conv_mode} mode is implemented for 16->16 camera blocks')
        if conv_mode == 'winograd' and not (c16 or wide):
            raise NotImplementedError('winograd mode needs 16->16 or wide (>= 64 channel) camera blocks')
        self.conv_mode = conv_mode
        self.split = self.wino = self.wgemm = None
        if conv_mode == 'winograd' and not c16:
            # wide blocks (released model: 256 -> 256 on 16^3): Winograd input transform + the fused fp32-MFMA GEMM /
            # output transform / epilogue kernel (lf_wino_fused_gemm)
            self.wgemm = [w for w, *_ in self.convs]            # packs are cached on the parameters (ops.wide_conv)
        if conv_mode == 'f16x3':
            self.split = [(ops.pack_conv3d_c16_split(w), ops.pack_conv3d_c16_split(w, transpose=True)) for w, *_ in self.convs]
        elif conv_mode == 'winograd' and c16:
            self.wino = [(ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)) for w, *_ in self.convs]
        # 'sum' projection / occlusion module: the tail between the camera blocks and the 2-D decoder runs through the
        # depth-column ops (autograd functions over lf_column_*), the rest of the iteration stays explicit
        self.generic_tail = photographer.projection_type != 'factor' or photographer.occlusion_module is not None
        if self.generic_tail and self.split is not None:
            raise NotImplementedError("the split-precision conv modes drive the plain 'factor' renderer only")
        C, D = (self.convs[-1][0].shape[0] if self.convs else self.C), self.S
        # the occlusion module on this library's kernels, sequenced here like the camera blocks (round 5); None: its
        # architecture is not the one planned for, it runs as the module with autograd
        self.occ = None
        if photographer.occlusion_module is not None and self.wino is not None and C == 16 and RenderLoopEngine.EXPLICIT_OCCLUSION:
            self.occ = self._plan_occlusion(photographer.occlusion_module)
        cout, pw = C, None
        self.proj = None
        if photographer.projection_type == 'factor':
            pw = photographer.projection_block.conv.module.weight
            cout = pw.shape[0]
            # [D*C][cout], K = d*C + c: the projection as one row-major GEMM over depth-innermost rows (PROJ_GEMM)
            self.proj_rows_t = pw.detach().reshape(cout, C, D).permute(2, 1, 0).reshape(D * C, cout).contiguous()
            self.proj = (pw, photographer.projection_block.conv.bias, ops.he_constant(pw),
                         ops.pack_conv1x1(pw.reshape(cout, C, D).permute(0, 2, 1).reshape(cout, D * C)),
                         ops.pack_conv1x1(pw.reshape(cout, C, D).permute(2, 1, 0).reshape(D * C, cout)))
        # factor projection fused into the Winograd kernel of the last camera block (lf_conv3d_c16_wino_projfwd; round 4):
        # bit-identical to the two-launch form, measured -0.07 ms per iteration
        can_fuse = self.wino is not None and cout == 16 and C == 16 and not self.generic_tail
        if fuse_projection is None:
            fuse_projection = ('fwd',) if can_fuse else ()
        elif fuse_projection is False:
            fuse_projection = ()
        elif fuse_projection is True:
            fuse_projection = self.FUSE_FORMS
        if any(f not in self.FUSE_FORMS for f in fuse_projection):
            raise ValueError(f'fuse_projection {fuse_projection!r}: this engine fuses {self.FUSE_FORMS}')
        if fuse_projection and not can_fuse:
            raise NotImplementedError("fuse_projection needs conv_mode 'winograd' on 16-channel blocks and a 16-channel projection")
        self.fuse_projection = tuple(fuse_projection)
        self.proj_fused = None
        if self.fuse_projection:
            wdm = pw.reshape(cout, C, D).permute(0, 2, 1).reshape(cout, D * C)        # depth-major K, as ppack
            self.proj_fused = (ops.pack_wino_proj(wdm), ops.pack_wino_proj(wdm, transpose=True))
        self.dev = dev
        self._params = list(photographer.parameters())
        self._intr = None                                            # (K, version, gathered): the intrinsics do not change during a loop
        # the output heads (1x1 convolutions without activation, reference blocks.py:108-119) as ONE pointwise convolution
        # with their weights stacked along the output channels: same arithmetic per channel, one launch each way instead
        # of a launch per head plus a concatenation
        self.heads = None
        obs = list(photographer.output_blocks)
        if len(obs) > 1 and all(getattr(ob, 'activation', None) is None and getattr(ob.conv, 'kernel_size', 0) == 1
                                and ob.conv.bias is not None for ob in obs):
            self.heads = (torch.cat([ob.conv.module.weight.detach() for ob in obs], dim=0).contiguous(),
                          torch.cat([ob.conv.bias.detach() for ob in obs], dim=0).contiguous())
        # a plain 2-D decoder is sequenced by the engine itself (no autograd graph, fused epilogue backward)
        self.dec = None
        dec = photographer.image_decoder
        blocks2d = list(dec.down_blocks) + list(dec.up_blocks)
        plain = (dec.input_block is None and dec.output_block is None and self.heads is not None
                 and (not self.generic_tail or (self.occ is not None and photographer.projection_type == 'factor'))
                 and (len(dec.up_blocks) < 2 or len(dec.down_blocks) < 2)               # no skip concatenation happens
                 and all(getattr(b, 'interpolate', True) is None for b in blocks2d) and len(blocks2d) > 0
                 and all(c.module.weight.dim() == 4 and tuple(c.module.weight.shape[2:]) == (3, 3) and c.module.weight.shape[0] <= 64
                         for b in blocks2d for c in (b.conv1, b.conv2)))
        if plain and RenderLoopEngine.EXPLICIT_DECODER:
            self.dec = []
            for blk in blocks2d:
                for conv in (blk.conv1, blk.conv2):
                    w = conv.module.weight
                    self.dec.append((w, conv.bias, ops.he_constant(w), ops.pack_conv3x3(w), ops.pack_conv3x3(w, transpose=True)))
            hw = self.heads[0]
            self.heads_pack = (ops.he_constant(hw), ops.pack_conv1x1(hw.reshape(hw.shape[0], hw.shape[1])),
                               ops.pack_conv1x1(hw.reshape(hw.shape[0], hw.shape[1]).t()))

    def set_weights(self, loss_weights):
        self.weights = torch.tensor([loss_weights.get(k, 0.0) for k in self.LOSS_KEYS], dtype=torch.float32,
                                    device=self.z.device)
        self.w_latent = float(loss_weights.get('latent', 0.0))

    # -----------------------------------------------------------------------------------------
    def forward_backward(self, camera, need_grad=True, z_target_latent=None, params=None, masked_depth=False):
        """Returns (losses (N,8): depth, ov_depth, iou, mask, weighted total, latent, 0, 0; gparams (N,10) or None).

        z_target_latent (N or 1, C2, h, w): the latent code of the target under every hypothesis
        (LatentFusionModel.compute_latent_code); with a non-zero 'latent' weight the cosine distance between it and the
        renderer's projected latent joins the loss (reference pose/estimation.py:112-116), column 5 of `losses`.

        masked_depth (forward only): the loss as the cross-entropy / Metropolis estimators evaluate it -- the crop depth
        times the crop's sigmoid mask before the uncrop (reference pose/estimation.py:207-216; lf_pose_loss_fwd_masked)."""
        # `params`: the (N,10) block [log_quaternion | translation | viewport] when the caller already holds it (the estimators'
        # cameras are views of one such tensor), else it is gathered from the camera; the intrinsics are gathered once
        # the engine differentiates w.r.t. the cameras only: with trainable parameters (the default of a model that was not
        # frozen: eval() is a pure mode switch, as in the reference) the autograd pieces would save activations for, and
        # launch, weight-gradient kernels nobody reads -- freeze them for the duration of the call
        if any(p.requires_grad for p in self._params):
            live = [p for p in self._params if p.requires_grad]
            for p in live:
                p.requires_grad_(False)
            try:
                return self.forward_backward(camera, need_grad, z_target_latent, params, masked_depth)
            finally:
                for p in live:
                    p.requires_grad_(True)
        if masked_depth and need_grad:
            raise ValueError('masked_depth is the forward-only loss form of the ranking estimators (no gradient is defined here)')
        if self.w_latent != 0.0 and z_target_latent is None:
            raise ValueError("the loss weights carry a 'latent' term: pass z_target_latent (LatentFusionModel.compute_latent_code)")
        params = (camera_params(camera) if params is None else params).detach().contiguous()
        K = camera.intrinsic
        # (keyed on the tensor OBJECT, which the cache keeps alive, and its version counter: an address alone could be
        # reused by another camera's intrinsics)
        if self._intr is None or self._intr[0] is not K or self._intr[1] != K._version:
            self._intr = (K, K._version, camera_intrinsics(camera))
        intr = self._intr[2]
        n = params.shape[0]
        zt = z_target_latent if (z_target_latent is not None and self.w_latent != 0.0) else None
        if zt is not None and zt.shape[0] == 1 and n > 1:
            zt = zt.expand(n, *zt.shape[1:])
        return self._run(params, intr, float(camera.z_span), need_grad, zt, masked_depth)

    def _run(self, params, intr, z_span, need_grad, zt, masked_depth):
        """All hypotheses as one group on the current stream (experimental.RenderLoopEngineX splits them over streams)."""
        return self._forward_backward_group(params, intr, z_span, need_grad, 1.0, zt, masked_depth)

    # ---- per-layer launches of the camera blocks (the experimental subclass adds kernel variants here) ----
    PROJ_GEMM = True        # wide blocks, ranking only: depth-innermost last block + library GEMM for the factor projection
    OCC_FUSE_SCALE = True   # occlusion weights: the factor projection scales its operand / sums the weights' gradient itself (r06)

    def _conv_fwd(self, li, x, flags, depth_inner=False):
        """Forward of camera-block convolution `li`: (y, norm, (zp, pnorm) when the factor projection rode along, else None)."""
        w, b, he, wp, _wt = self.convs[li]
        if depth_inner:
            return ops.wide_conv(x, self.wgemm[li], b, he, flags, depth_inner=True) + (None,)
        if self.split is not None:
            return ops.conv3d_c16_split(x, self.split[li][0], b, he, flags) + (None,)
        if self.wino is not None and li == len(self.convs) - 1 and 'fwd' in self.fuse_projection:
            _pw, pb, phe, _ppack, _ppack_t = self.proj
            y, nrm, zp, pnorm = ops.conv3d_c16_wino_projfwd(x, self.wino[li][0], b, he, flags, self.proj_fused[0], pb, phe, flags)
            return y, nrm, (zp, pnorm)
        if self.wino is not None:
            return ops.conv3d_c16_wino(x, self.wino[li][0], b, he, flags) + (None,)
        if self.wgemm is not None:
            return ops.wide_conv(x, self.wgemm[li], b, he, flags) + (None,)
        return ops._conv3x3_raw(x, wp, b, w.shape[0], he, flags, True) + (None,)

    def _conv_bwd(self, i, g, prev, amax):
        """Data gradient of camera-block convolution `i` with the LeakyReLU' / PixelNorm' of the layer feeding it (`prev`) folded
        into the store (16-channel blocks)."""
        w, _b, he, _wp, wt = self.convs[i]
        if self.split is not None:
            return ops.conv3d_c16_split(g, self.split[i][1], None, he, 0, prev=prev, amax_in=amax[i + 1], amax_out=amax[i])[0]
        if self.wino is not None:
            return ops.conv3d_c16_wino(g, self.wino[i][1], None, he, 0, prev=prev)[0]
        return ops.conv3x3_bwd_data(g, wt, w.shape[1], he, prev)

    def _tail_leaf(self, act_leaf, zs_leaf, zp_leaf):
        """The tensor autograd differentiates the 2-D tail back to: the projected latent (factor projection), the scaled volume
        (explicit occlusion + 'sum'), or the last camera block's output (the tail as modules)."""
        if self.occ is not None:
            return zp_leaf if self.proj is not None else zs_leaf
        return act_leaf if self.generic_tail else zp_leaf

    def _proj_bwd_fused(self, gp, acts, norms, flags):
        """Hook: (gradient after the LAST block's data-gradient convolution, index of the next layer to walk) when the projection
        backward is fused into that launch; None here (experimental.RenderLoopEngineX: lf_conv3d_c16_wino_projbwd)."""
        return None

    # ---- occlusion module on explicit kernels ----
    @staticmethod
    def _plan_occlusion(om):
        """Weights of UNet3d(17, 1, [[17, 16], [16, 16]]) without rescaling (the occlusion module as the reference
        constructs it, recon/models.py:305-306 over modules/unet.py, with a [[17, 16], [16, 16]] block configuration) packed for the explicit sequence, or None for any other shape:
        input block 1x1x1 17 -> 17 (lf_occ_input_*), conv 17 -> 16 as a 16 -> 16 Winograd launch over outputs 0..15 of the input
        block plus the 1 -> 16 convolution of its output 16 (lf_occ_conv17_*), three 16 -> 16 convolutions, output block 1x1x1 16 -> 1 without activation."""
        from .modules.blocks import OutputBlock
        blocks = list(om.down_blocks) + list(om.up_blocks)
        ib, ob = om.input_block, om.output_block
        if (ib is None or not isinstance(ob, OutputBlock) or ob.activation is not None or len(om.down_blocks) != 1
                or len(om.up_blocks) != 1 or any(b.interpolate is not None for b in blocks)):
            return None
        w1, wo = ib.conv.module.weight, ob.conv.module.weight
        convs = [c for b in blocks for c in (b.conv1, b.conv2)]
        shapes = [tuple(c.module.weight.shape) for c in convs]
        if (tuple(w1.shape) != (17, 17, 1, 1, 1) or tuple(wo.shape) != (1, 16, 1, 1, 1) or shapes[0] != (16, 17, 3, 3, 3)
                or any(sh != (16, 16, 3, 3, 3) for sh in shapes[1:]) or ib.conv.bias is None or ob.conv.bias is None):
            return None
        dev = w1.device
        wi = torch.zeros(17, 20, device=dev, dtype=torch.float32)
        wi[:, :17] = w1.detach().reshape(17, 17) * ops.he_constant(w1)
        bi = torch.zeros(20, device=dev, dtype=torch.float32)
        bi[:17] = ib.conv.bias.detach()
        w2 = convs[0].module.weight.detach()
        w2a = w2[:, :16].contiguous()
        he2 = ops.he_constant(w2)
        w27 = (w2[:, 16].reshape(16, 27).t() * he2).float().contiguous()              # [tap][co]
        pk = ops.pack_conv3d_c16_wino
        first = (convs[0].bias, he2, pk(w2a), pk(w2a, transpose=True), w27)
        rest = [(c.bias, ops.he_constant(c.module.weight), pk(c.module.weight), pk(c.module.weight, transpose=True)) for c in convs[1:]]
        wo2 = wo.detach().reshape(1, 16)
        head = (ob.conv.bias, ops.he_constant(wo), ops.pack_conv1x1(wo2), ops.pack_conv1x1(wo2.t().contiguous()))
        return dict(wi=wi.contiguous(), bi=bi, first=first, rest=rest, head=head, head_w16=wo2.reshape(16).float().contiguous())

    def _occlusion_fwd(self, zc, flags):
        """zc: output of the last camera block (n,16,S,S,S).  Returns (zs = zc * softmax_D(occlusion logits), saved)."""
        L = _lib.lib()
        o, s, dev = self.occ, _s(), self.dev
        n, _, D, H, W = zc.shape
        P = H * W
        ta, t16 = ops.empty_cl(zc.shape, dev), torch.empty(n, 1, D, H, W, device=dev, dtype=torch.float32)
        with ops._timed('occ_input_fwd'):
            check(L.lf_occ_input_fwd(zc.data_ptr(), o['wi'].data_ptr(), o['bi'].data_ptr(), ta.data_ptr(), t16.data_ptr(), n, D, P,
                                     ops.SLOPE, s), 'lf_occ_input_fwd')
        b2, he2, pa, _pat, w27 = o['first']
        pre = ops.empty_cl(zc.shape, dev)
        with ops._timed('occ_conv17_fwd'):
            check(L.lf_occ_conv17_fwd(t16.data_ptr(), w27.data_ptr(), pre.data_ptr(), n, D, H, W, s), 'lf_occ_conv17_fwd')
        y, nrm = ops.conv3d_c16_wino(ta, pa, b2, he2, flags, prev=(pre, None, _lib.LF_EPI_ADD), out=pre)
        ys, ns = [y], [nrm]
        for (b, he, pk, _pkt) in o['rest']:
            y, nrm = ops.conv3d_c16_wino(ys[-1], pk, b, he, flags)
            ys.append(y)
            ns.append(nrm)
        hb, hhe, hpk, _hpkt = o['head']
        wocc = torch.empty(n, 1, D, H, W, device=dev, dtype=torch.float32)
        if self.OCC_FUSE_SCALE and D <= 256:
            # (round 6) output block + softmax over the depth column in one pass over the last activation: no logits volume
            with ops._timed('column_softmax_head'):
                check(L.lf_column_softmax_head_fwd(ys[-1].data_ptr(), o['head_w16'].data_ptr(), hb.data_ptr() if hb is not None else None,
                                                   hhe, wocc.data_ptr(), None, n, D, P, s), 'lf_column_softmax_head_fwd')
        else:
            logits = torch.empty(n, 1, D, H, W, device=dev, dtype=torch.float32)
            ops._conv1x1_raw(ys[-1], hpk, hb, n, D * P, 16, 1, D * P * 16, 0, 1, logits, hhe, 0)
            with ops._timed('column_softmax'):
                check(L.lf_column_softmax_fwd(logits.data_ptr(), wocc.data_ptr(), None, n, D, P, s), 'lf_column_softmax_fwd')
        if self.proj is not None and self.OCC_FUSE_SCALE:
            # (round 6) the factor projection scales its operand itself (lf_conv1x1_fwd_scaled): the scaled volume is never written
            return None, (ta, t16, ys, ns, wocc)
        zs = ops.empty_cl(zc.shape, dev)
        check(L.lf_column_scale_fwd(zc.data_ptr(), wocc.data_ptr(), zs.data_ptr(), n * D * P, 16, s), 'lf_column_scale_fwd')
        return zs, (ta, t16, ys, ns, wocc)

    def _occlusion_bwd(self, g_zs, zc, saved, flags, prev=None, gw=None, proj_bwd=None):
        """d/d(zs) -> d/d(zc): the scaling, the softmax, the output block, the four convolutions (each data-gradient launch folds
        in the LeakyReLU' / PixelNorm' of the layer it lands on) and the input block, plus the direct term of the scaling.
        prev = (zc, norm, flags) of the camera-block layer that produced zc: its epilogue backward is applied on the way out."""
        L = _lib.lib()
        o, s, dev = self.occ, _s(), self.dev
        ta, t16, ys, ns, wocc = saved
        n, _, D, H, W = zc.shape
        P = H * W
        # (the direct term g_zs * wocc of the scaling joins the input block's backward below: no volume is written for it)
        gl = torch.empty_like(wocc)
        if proj_bwd is not None and D <= 256:
            # (round 6) the weights' gradient sum_c zc * g_zs, g_zs recomputed from the projection's 2-D gradient, and the softmax
            # backward over each pixel's depth column in one launch: one read of zc; neither g_zs nor gw exist as volumes
            with ops._timed('occ_weight_grad_softmax_bwd'):
                check(L.lf_occ_weight_grad_softmax_bwd(zc.data_ptr(), proj_bwd[0].data_ptr(), proj_bwd[1].data_ptr(), proj_bwd[2], wocc.data_ptr(),
                                                       gl.data_ptr(), n, D, P, s), 'lf_occ_weight_grad_softmax_bwd')
        else:
            if gw is None and proj_bwd is not None:
                gw = torch.empty_like(wocc)
                check(L.lf_occ_weight_grad(zc.data_ptr(), proj_bwd[0].data_ptr(), proj_bwd[1].data_ptr(), proj_bwd[2], gw.data_ptr(), n, D, P, s),
                      'lf_occ_weight_grad')
            if gw is None:                                        # (else: the gradient of the weights is already summed per voxel)
                gw = torch.empty_like(wocc)
                check(L.lf_column_scale_bwd(g_zs.data_ptr(), zc.data_ptr(), wocc.data_ptr(), None, gw.data_ptr(), n * D * P, 16, s),
                      'lf_column_scale_bwd')
            check(L.lf_column_softmax_bwd(wocc.data_ptr(), gw.data_ptr(), None, gl.data_ptr(), n, D, P, s), 'lf_column_softmax_bwd')
        _hb, hhe, _hpk, hpkt = o['head']
        g = ops.empty_cl(zc.shape, dev)
        if self.OCC_FUSE_SCALE:
            with ops._timed('occ_head_bwd'):
                check(L.lf_occ_head_bwd(gl.data_ptr(), o['head_w16'].data_ptr(), hhe, ys[-1].data_ptr(), ns[-1].data_ptr(), flags, ops.SLOPE,
                                        g.data_ptr(), n * D * P, s), 'lf_occ_head_bwd')
        else:
            check(L.lf_conv1x1_bwd_data(gl.data_ptr(), hpkt.data_ptr(), g.data_ptr(), n, D * P, 1, 16, D * P * 16, 16, 16, 0, hhe,
                                        ys[-1].data_ptr(), ns[-1].data_ptr(), flags, ops.SLOPE, None, s), 'lf_conv1x1_bwd_data')
        for i in range(len(o['rest']) - 1, -1, -1):
            _b, he, _pk, pkt = o['rest'][i]
            g = ops.conv3d_c16_wino(g, pkt, None, he, 0, prev=(ys[i], ns[i], flags))[0]
        _b2, he2, _pa, pat, w27 = o['first']
        # (round 6) LeakyReLU' of the input block's outputs 0..15 rides in the store of this data gradient (prev = ta, no norm): the
        # input block's backward then does not read ta
        ta_in_store = self.OCC_FUSE_SCALE or proj_bwd is not None
        gp16 = torch.empty_like(t16)
        # (measured, round 6: this launch on a second stream BESIDE the Winograd data gradient that reads the same g is slower -- the
        # persistent Winograd workgroups hold every CU, the side kernel took 1.33 ms instead of 0.38 and stretched its neighbour:
        # 63.4 against 64.4 it/s)
        gta = ops.conv3d_c16_wino(g, pat, None, he2, 0, prev=(ta, None, LF_EPI_LRELU) if ta_in_store else None)[0]
        with ops._timed('occ_conv17_bwd'):
            check(L.lf_occ_conv17_bwd(g.data_ptr(), t16.data_ptr(), w27.data_ptr(), gp16.data_ptr(), n, D, H, W, ops.SLOPE, s),
                  'lf_occ_conv17_bwd')
        gz = ops.empty_cl(zc.shape, dev)
        if proj_bwd is not None:
            gp2d, ppack_t, phe = proj_bwd
            with ops._timed('occ_input_bwd'):
                check(L.lf_occ_input_bwd_proj(gta.data_ptr(), gp16.data_ptr(), o['wi'].data_ptr(), gp2d.data_ptr(), ppack_t.data_ptr(), phe,
                                              wocc.data_ptr(), gz.data_ptr(), n, D, P, ops.SLOPE, zc.data_ptr() if prev is not None else None,
                                              prev[1].data_ptr() if prev is not None else None, prev[2] if prev is not None else 0, s),
                      'lf_occ_input_bwd_proj')
            return gz
        with ops._timed('occ_input_bwd'):
            check(L.lf_occ_input_bwd(gta.data_ptr(), None if ta_in_store else ta.data_ptr(), gp16.data_ptr(), o['wi'].data_ptr(), g_zs.data_ptr(),
                                     wocc.data_ptr(), gz.data_ptr(), n * D * P, ops.SLOPE, zc.data_ptr() if prev is not None else None,
                                     prev[1].data_ptr() if prev is not None else None, prev[2] if prev is not None else 0, s),
                  'lf_occ_input_bwd')
        return gz

    def _forward_backward_group(self, params, intr, z_span, need_grad, grad_scale, zt=None, masked_depth=False):
        L = _lib.lib()
        dev, S, s = self.dev, self.S, _s()
        params = params.contiguous()
        intr = intr.contiguous()
        n = params.shape[0]
        coefs = torch.empty(n, NCOEF, device=dev, dtype=torch.float32)
        jac = torch.empty(n, NCOEF, NPAR, device=dev, dtype=torch.float32)
        check(L.lf_camera_coefs(params.data_ptr(), intr.data_ptr(), float(self.cube), z_span, self.crop,
                                self.crop, coefs.data_ptr(), jac.data_ptr(), n, s), 'lf_camera_coefs')
        # O2C coefficient block padded to the resampler's stride (LF_MAP_COEFS = 20)
        # (the object->camera map reads 18 of the 20 floats of a block; the two pad floats are never read: no fill launch)
        cf20 = torch.empty(n, 20, device=dev, dtype=torch.float32)
        cf20[:, :18] = coefs[:, :18]

        # ---- 3-D forward ----
        x0 = ops.empty_cl((n, self.C, S, S, S), dev)
        with ops._timed('resample_fwd'):
            check(L.lf_resample3d_fwd(self.z.data_ptr(), 1, cf20.data_ptr(), LF_MAP_O2C, x0.data_ptr(), n, S, S, S, self.C, s),
                  'lf_resample3d_fwd')
        acts, norms = [x0], []
        flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
        # wide blocks, ranking only (the released architecture under the cross-entropy search): the last block writes its output
        # depth-innermost, so that the factor projection K = D * C -> C2 below is ONE row-major GEMM for the library
        # (128 renders: 32768 x 4096 x 256, 0.5 ms on hipBLASLt's fp32 MFMA kernel against 1.27 ms of lf_conv1x1_fwd's K slices)
        proj_gemm = (self.PROJ_GEMM and self.wgemm is not None and not need_grad and self.occ is None and not self.generic_tail
                     and self.proj is not None and 'fwd' not in self.fuse_projection)
        for li_ in range(len(self.convs)):
            y, nrm, rode = self._conv_fwd(li_, acts[-1], flags, depth_inner=proj_gemm and li_ == len(self.convs) - 1)
            if rode is not None:
                zp, pnorm = rode
            acts.append(y)
            norms.append(nrm)
        Cl = acts[-1].shape[-1] if proj_gemm else acts[-1].shape[1]
        act_leaf = occ_saved = zs_leaf = None
        if self.occ is not None:
            # occlusion weights on explicit kernels; the composite behind them: the factor projection as in the plain renderer
            # (its own launch: the fused form reads the block's output, not the scaled one), or the 'sum' as a depth-column op
            zs, occ_saved = self._occlusion_fwd(acts[-1], flags)
            if self.proj is not None:
                pw, pb, phe, ppack, ppack_t = self.proj
                cout = pw.shape[0]
                zp = ops.empty_cl((n, cout, S, S), dev)
                with ops._timed('factor_project_fwd'):
                    if zs is None:
                        pnorm = ops._conv1x1_raw(acts[-1], ppack, pb, n, S * S, Cl, S, S * S * S * Cl, S * S * Cl, cout, zp, phe, flags,
                                                 xscale=occ_saved[4])
                    else:
                        pnorm = ops._conv1x1_raw(zs, ppack, pb, n, S * S, Cl, S, S * S * S * Cl, S * S * Cl, cout, zp, phe, flags)
                zp_leaf = zp.detach().requires_grad_(need_grad)
            else:
                zs_leaf = zs.detach().requires_grad_(need_grad)
                with torch.set_grad_enabled(need_grad):
                    zp_leaf = ops.column_sum(zs_leaf)
                zp = zp_leaf
        elif self.generic_tail:
            # occlusion weights (3-D U-Net logits -> softmax over the depth column -> scale) and / or the 'sum' composite
            # (reference recon/models.py:378-395,427-437), as in Photographer.forward, differentiated by autograd up to the
            # last camera block's output
            from .recon.utils import get_normalized_voxel_depth
            act_leaf = acts[-1].detach().requires_grad_(need_grad)
            with torch.set_grad_enabled(need_grad):
                zt_ = act_leaf
                if self.ph.occlusion_module is not None:
                    occ = self.ph.occlusion_module(torch.cat((zt_, get_normalized_voxel_depth(zt_)), dim=1))
                    wocc = ops.column_softmax(occ)[0]
                    if tuple(occ.shape[-3:]) != tuple(zt_.shape[-3:]):
                        wocc = ops.column_softmax(ops.resize_nearest_to(occ, zt_.size(-1)))[0]
                    zt_ = ops.column_scale(zt_, wocc)
                zp_leaf = ops.column_sum(zt_) if self.ph.projection_type == 'sum' else self.ph.projection_block(zt_)
            zp = zp_leaf
        else:
            pw, pb, phe, ppack, ppack_t = self.proj
            cout = pw.shape[0]
            if proj_gemm:
                with ops._timed('factor_project_fwd'):
                    x2 = acts[-1].view(n * S * S, S * Cl)
                    if pb is not None:
                        rows = torch.addmm(pb.detach(), x2, self.proj_rows_t, alpha=phe)        # he * (x W^T) + b
                    else:
                        rows = torch.mm(x2, self.proj_rows_t).mul_(phe)
                    torch.nn.functional.leaky_relu_(rows, ops.SLOPE)
                    pnorm = torch.empty(n * S * S, device=dev, dtype=torch.float32)
                    check(L.lf_pixelnorm_fwd(rows.data_ptr(), rows.data_ptr(), pnorm.data_ptr(), n * S * S, cout, ops.PN_EPS, s), 'lf_pixelnorm_fwd')
                zp = rows.view(n, S, S, cout).permute(0, 3, 1, 2)                # channels-last (n, cout, S, S)
            elif 'fwd' not in self.fuse_projection:
                zp = ops.empty_cl((n, cout, S, S), dev)
                with ops._timed('factor_project_fwd'):
                    pnorm = ops._conv1x1_raw(acts[-1], ppack, pb, n, S * S, Cl, S, S * S * S * Cl, S * S * Cl, cout, zp, phe, flags)
            zp_leaf = zp.detach().requires_grad_(need_grad)

        # ---- 2-D decoder + heads + fused loss ----
        explicit = self.dec is not None and (zt is None or not need_grad)
        if explicit:
            # plain decoder, sequenced here: conv + He + bias + LeakyReLU + PixelNorm per launch, then the stacked heads
            dacts, dnorms = [zp], [pnorm]
            for (w2, b2, he2, pk2, _pkt2) in self.dec:
                y2, n2 = ops._conv3x3_raw(dacts[-1], pk2, b2, w2.shape[0], he2, flags, True)
                dacts.append(y2)
                dnorms.append(n2)
            hw, hb = self.heads
            hhe, hpk, _hpkt = self.heads_pack
            hh, ww = dacts[-1].shape[-2:]
            logits = ops.empty_cl((n, hw.shape[0], hh, ww), dev)
            ops._conv1x1_raw(dacts[-1], hpk, hb, n, hh * ww, hw.shape[1], 1, hh * ww * hw.shape[1], 0, hw.shape[0], logits, hhe, 0)
        else:
            with torch.set_grad_enabled(need_grad):
                yimg, rescale = self.ph.decode_features(zp_leaf)      # (a final nearest up-sampling is owed to the logits)
                if self.heads is not None:
                    logits = ops.conv1x1(yimg, self.heads[0], self.heads[1])
                else:
                    logits = torch.cat([ob(yimg) for ob in self.ph.output_blocks], dim=1)
                if rescale is not None:
                    logits = rescale(logits)
        if zt is None or not need_grad:
            # the optimised quantity is mean_n(total) (estimation.py:616-617): lf_pose_loss_fwd already leaves the sums'
            # gradients for exactly that, so the loss needs no autograd node -- logits -> loss -> d/d(logits, coefficients)
            # are two kernel calls, and autograd only carries d(logits) back through the decoder
            lg = ops.cl(logits.detach())                          # (n,2,h,w) channels-last == [n][h*w][2]
            h_, w_ = lg.shape[-2:]
            nbytes = L.lf_pose_loss_scratch_bytes(n, h_, w_, self.H, self.W)
            scratch_l = torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32)
            sums = torch.empty(n, 8, device=dev, dtype=torch.float32)
            losses = torch.empty(n, 8, device=dev, dtype=torch.float32)
            gsums = torch.empty(n, 8, device=dev, dtype=torch.float32)
            if masked_depth:
                check(L.lf_pose_loss_fwd_masked(lg.data_ptr(), coefs.data_ptr(), self.tdepth.data_ptr(), self.tmask.data_ptr(),
                                                self.weights.data_ptr(), sums.data_ptr(), losses.data_ptr(), scratch_l.data_ptr(),
                                                scratch_l.numel() * 4, n, h_, w_, self.H, self.W, s), 'lf_pose_loss_fwd_masked')
            else:
                check(L.lf_pose_loss_fwd(lg.data_ptr(), coefs.data_ptr(), self.tdepth.data_ptr(), self.tmask.data_ptr(),
                                         self.weights.data_ptr(), sums.data_ptr(), losses.data_ptr(), gsums.data_ptr(),
                                         scratch_l.data_ptr(), scratch_l.numel() * 4, n, h_, w_, self.H, self.W, s), 'lf_pose_loss_fwd')
            if not need_grad:
                if zt is not None:
                    # latent term of a ranking-only evaluation: cosine distance of the projected latent (estimation.py:112-116)
                    lat = 1.0 - torch.cosine_similarity(zp.reshape(n, -1), zt.reshape(n, -1).to(zp.dtype), 1, 1e-8)
                    losses[:, 5] = lat
                    losses[:, 4] += self.w_latent * lat
                return losses, None
            glogits = torch.empty_like(lg)
            # (lf_pose_loss_bwd writes entries 18..23; 0..17 come from the resampler's coefficient gradient below)
            g_cf = torch.empty(n, NCOEF, device=dev, dtype=torch.float32)
            check(L.lf_pose_loss_bwd(lg.data_ptr(), coefs.data_ptr(), self.tdepth.data_ptr(), self.tmask.data_ptr(), gsums.data_ptr(),
                                     glogits.data_ptr(), g_cf.data_ptr(), scratch_l.data_ptr(), scratch_l.numel() * 4,
                                     n, h_, w_, self.H, self.W, s), 'lf_pose_loss_bwd')
            gp_explicit = None
            if explicit:
                # heads, then the decoder's data gradients in reverse; each launch folds in the LeakyReLU' / PixelNorm' of the
                # activation it lands on (whole records of 16 / 32 / 64 channels; otherwise a separate pass)
                def landed(gx, i):
                    return ops._epilogue_bwd(gx, dacts[i], dnorms[i], flags)
                hw = self.heads[0]
                cin_h = hw.shape[1]
                g2 = ops.empty_cl((n, cin_h, hh, ww), dev)
                fuse_h = cin_h == 16
                check(L.lf_conv1x1_bwd_data(glogits.data_ptr(), self.heads_pack[2].data_ptr(), g2.data_ptr(), n, hh * ww, hw.shape[0],
                                            cin_h, hh * ww * cin_h, cin_h, cin_h if fuse_h else (1 << 30), 0, self.heads_pack[0],
                                            dacts[-1].data_ptr() if fuse_h else None, dnorms[-1].data_ptr() if fuse_h else None,
                                            flags if fuse_h else 0, ops.SLOPE, None, s), 'lf_conv1x1_bwd_data')
                if not fuse_h:
                    g2 = landed(g2, len(self.dec))
                for i in range(len(self.dec) - 1, -1, -1):
                    w2, _b2, he2, _pk2, pkt2 = self.dec[i]
                    cin2 = w2.shape[1]
                    if cin2 in (16, 32, 64):
                        g2 = ops.conv3x3_bwd_data(g2, pkt2, cin2, he2, (dacts[i], dnorms[i], flags))
                    else:
                        g2 = landed(ops.conv3x3_bwd_data(g2, pkt2, cin2, he2, None), i)
                gp_explicit = g2                                  # d/d(projection pre-activation): dacts[0] is zp itself
                g_zp = None
            else:
                g_zp, = torch.autograd.grad(logits, [self._tail_leaf(act_leaf, zs_leaf, zp_leaf)], grad_outputs=[glogits])
        else:
            cf_leaf = coefs.detach().requires_grad_(need_grad)
            with torch.set_grad_enabled(need_grad):
                total, losses = pose_loss(logits, cf_leaf, self.tdepth, self.tmask, self.weights, self.H, self.W)
                # latent term: cosine distance of the projected latent (what Photographer.forward returns as its latent)
                lat = 1.0 - torch.cosine_similarity(zp_leaf.reshape(n, -1), zt.reshape(n, -1).to(zp_leaf.dtype), 1, 1e-8)
                total = total + self.w_latent * lat
                losses = losses.clone()
                losses[:, 5] = lat.detach()
                losses[:, 4] += self.w_latent * lat.detach()
                objective = total.mean()                     # the optimised quantity (estimation.py:616-617)
            if not need_grad:
                return losses, None
            g_zp, g_cf = torch.autograd.grad(objective, [self._tail_leaf(act_leaf, zs_leaf, zp_leaf), cf_leaf])

        # ---- 3-D backward (data gradients only) ----
        fuse = (Cl == 16 and self.C == 16 and all(w.shape[0] == 16 and w.shape[1] == 16 for w, *_ in self.convs))
        nconv = len(self.convs)
        if self.generic_tail:
            if self.occ is not None:
                # d/d(scaled volume): the projection's data gradient (factor) or autograd's column-sum gradient ('sum'),
                # then the occlusion module's explicit backward
                if self.proj is not None:
                    gp = gp_explicit if (explicit and gp_explicit is not None) else ops._epilogue_bwd(ops.cl(g_zp), zp, pnorm, flags)
                    gw_occ = proj_bwd = None
                    with ops._timed('factor_project_bwd'):
                        if self.OCC_FUSE_SCALE and Cl == 16 and cout == 16 and (S * S) % 16 == 0:
                            # the projection's data gradient is NOT stored: lf_occ_weight_grad sums g_zs * zc over the channels of every
                            # voxel -- the gradient of the occlusion weights, one read of zc instead of lf_column_scale_bwd's pass over
                            # two volumes -- and the input block's backward recomputes g_zs where it needs it (lf_occ_input_bwd_proj)
                            g_zs = None
                            proj_bwd = (gp, ppack_t, phe)
                        else:
                            g_zs = ops.empty_cl((n, Cl, S, S, S), dev)
                            ops._conv1x1_raw(gp, ppack_t, None, n, S * S, cout, 1, S * S * cout, 0, S * Cl, g_zs, phe, 0,
                                             yaddr=(S * S * S * Cl, Cl, Cl, S * S * Cl))
                else:
                    g_zs = ops.cl(g_zp)
                    gw_occ = proj_bwd = None
                landed = fuse and nconv > 0
                g_zp = self._occlusion_bwd(g_zs, acts[-1], occ_saved, flags, (acts[nconv], norms[nconv - 1], flags) if landed else None,
                                           gw=gw_occ, proj_bwd=proj_bwd)
            # g_zp is d/d(output of the last camera block); from here the explicit data-gradient chain
            g = ops.cl(g_zp)
            if fuse and nconv:
                if self.occ is None:
                    g = ops._epilogue_bwd(g, acts[nconv], norms[nconv - 1], flags)
                for i in range(nconv - 1, -1, -1):
                    w, b, he, _wp, wt = self.convs[i]
                    prev = (acts[i], norms[i - 1], flags) if i > 0 else None
                    if self.split is not None:
                        raise NotImplementedError('split-precision kernels drive the factor renderer only')
                    g = self._conv_bwd(i, g, prev, None)
            else:
                for i in range(nconv - 1, -1, -1):
                    w, b, he, _wp, wt = self.convs[i]
                    gpre = ops._epilogue_bwd(g, acts[i + 1], norms[i], flags)
                    if self.wgemm is not None:
                        g, _ = ops.wide_conv(gpre, self.wgemm[i], None, he, 0, transpose=True)
                    else:
                        g, _ = ops._conv3x3_raw(gpre, wt, None, w.shape[1], he, 0, False)
            return self._finish_backward(g, g_cf, cf20, jac, n, grad_scale, losses)
        gp = gp_explicit if (explicit and gp_explicit is not None) else ops._epilogue_bwd(ops.cl(g_zp), zp, pnorm, flags)
        fused_pb = self._proj_bwd_fused(gp, acts, norms, flags) if (fuse and nconv) else None
        g = None if fused_pb is not None else ops.empty_cl((n, Cl, S, S, S), dev)
        if fused_pb is not None:
            g, nxt = fused_pb
            for i in range(nxt, -1, -1):
                g = self._conv_bwd(i, g, (acts[i], norms[i - 1], flags) if i > 0 else None, None)
        elif fuse and nconv:
            # every data-gradient kernel also applies the LeakyReLU'/PixelNorm' of the layer feeding it,
            # so no separate epilogue-backward pass touches the (N,16,S,S,S) volumes
            # max-abs of each gradient volume (order-independent atomic max inside the producing kernel):
            # lets the split kernels pre-scale tiny gradients by an exact power of two
            amax = ops.amax_buffer(None, dev, rows=nconv + 1) if self.split is not None else None
            with ops._timed('factor_project_bwd'):
                check(L.lf_conv1x1_bwd_data(gp.data_ptr(), ppack_t.data_ptr(), g.data_ptr(), n, S * S, cout, S * Cl,
                                            S * S * S * Cl, Cl, Cl, S * S * Cl, phe, acts[nconv].data_ptr(),
                                            norms[nconv - 1].data_ptr(), flags, ops.SLOPE,
                                            amax[nconv].data_ptr() if amax is not None else None, s), 'lf_conv1x1_bwd_data')
            for i in range(nconv - 1, -1, -1):
                g = self._conv_bwd(i, g, (acts[i], norms[i - 1], flags) if i > 0 else None, amax)
        else:
            ops._conv1x1_raw(gp, ppack_t, None, n, S * S, cout, 1, S * S * cout, 0, S * Cl, g, phe, 0,
                             yaddr=(S * S * S * Cl, Cl, Cl, S * S * Cl))
            for i in range(nconv - 1, -1, -1):
                w, b, he, _wp, wt = self.convs[i]
                gpre = ops._epilogue_bwd(g, acts[i + 1], norms[i], flags)
                if self.wgemm is not None:
                    g, _ = ops.wide_conv(gpre, self.wgemm[i], None, he, 0, transpose=True)
                else:
                    g, _ = ops._conv3x3_raw(gpre, wt, None, w.shape[1], he, 0, False)
        return self._finish_backward(g, g_cf, cf20, jac, n, grad_scale, losses)

    def _finish_backward(self, g, g_cf, cf20, jac, n, grad_scale, losses):
        """d/d(O2C output) -> coefficient gradient -> camera parameters (lf_resample3d_bwd_coef, lf_camera_coefs_bwd)."""
        L = _lib.lib()
        dev, S, s = self.dev, self.S, _s()
        gcoef18 = torch.empty(n, 18, device=dev, dtype=torch.float32)
        nbytes = L.lf_resample3d_bwd_coef_scratch_bytes(n, S, S, S)
        scratch = torch.empty(nbytes // 4 + 1, device=dev, dtype=torch.float32)
        with ops._timed('resample_bwd_coef'):
            check(L.lf_resample3d_bwd_coef(g.data_ptr(), self.z.data_ptr(), 1, cf20.data_ptr(), gcoef18.data_ptr(),
                                           scratch.data_ptr(), scratch.numel() * 4, n, S, S, S, self.C, s), 'lf_resample3d_bwd_coef')
        gcoefs = g_cf.contiguous()
        gcoefs[:, :18] = gcoef18
        gparams = torch.empty(n, NPAR, device=dev, dtype=torch.float32)
        check(L.lf_camera_coefs_bwd(gcoefs.data_ptr(), jac.data_ptr(), gparams.data_ptr(), n, s), 'lf_camera_coefs_bwd')
        if grad_scale != 1.0:
            gparams *= grad_scale
        return losses, gparams
