"""The generator's training step of tools/train/train_reconstruct.py:421-535 on the HIP path: encode the
input views, decode the reconstruction views, depth / mask reconstruction losses, backward through every
kernel (weight-gradient, data-gradient, volume splat), optimiser step.

What is reproduced: the loss composition and weighting (`loss_g`, :510-516 incl. the division by
`batch_groups`), the criterion and optimiser factories (Adam with betas (0, 0.99)), the depth noise clamp
for `generator_input_depth`, `crop_predicted_mask`.  What is not: the discriminator branch (disabled in the
released recipe, train.sh:59) and dataset / augmentation / logging plumbing.  `use_amp=True` runs the generator under
the bf16 autocast policy (ops.autocast; the 3-D 16 -> 16 blocks on the bf16 MFMA).  Data parallelism replaces `MyDataParallel` by one process per GPU with a bucketed gradient
all-reduce over RCCL (`parallel.allreduce_flat_`).

Parameters live in ONE flat fp32 buffer (the modules' tensors are views into it) and so do the gradients:
the optimiser is a single `lf_adam_step` launch and the all-reduce needs no packing."""
import itertools
import math

import torch
import torch.nn.functional as F

from .. import _lib, losses, ops, parallel


class FlatParameters:
    """Re-homes the parameters of `modules` into one contiguous buffer (views keep their shapes)."""

    def __init__(self, modules):
        self.params = [p for m in modules if m is not None for p in m.parameters()]
        if not self.params:
            raise ValueError('no parameters to train')
        dev = self.params[0].device
        # every parameter starts on a 64-byte boundary: the HIP kernels read weights and biases as 16-byte vectors
        # (the 1- and 3-element biases of the output heads would otherwise shift everything behind them)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 15) // 16 * 16
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.requires_grad_(True)
            p.grad = self.grad[off:off + k].view(p.shape)          # autograd accumulates into the flat buffer

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):                # re-attach views a caller may have dropped
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[off:off + k].data_ptr():
                p.grad = self.grad[off:off + k].view(p.shape)


class FlatAdam:
    """torch.optim.Adam / AdamW (defaults except the betas) over a FlatParameters, one lf_adam_step launch."""

    def __init__(self, flat, lr, betas=(0.0, 0.99), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.t = 0

    def step(self):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        dev = self.flat.data.device
        rows = torch.tensor([self.lr / bc1, self.lr], dtype=torch.float32, device=dev)
        L = _lib.lib()
        n = self.flat.data.numel()
        if n >= 2 ** 31:
            raise ValueError('flat parameter buffer too large for one launch')
        _lib.check(L.lf_adam_step(self.flat.data.data_ptr(), self.flat.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                  rows[0:1].data_ptr(), rows[1:2].data_ptr(), math.sqrt(bc2), b1, b2, self.eps,
                                  self.weight_decay, 1, n, torch.cuda.current_stream().cuda_stream), 'lf_adam_step')
        ops.invalidate_weight_packs()                     # the kernel wrote the weights behind torch's back


class GeneratorStep:
    """One `run_iteration` of the reference trainer without the discriminator (see the module docstring).

    batch = {'in':     {'camera', 'image' (B,V,3,H,W), 'mask' (B,V,1,H,W)[, 'depth']},
             'out_gt': {'camera', 'depth' (B,Vo,1,H,W), 'mask' (B,Vo,1,H,W)}}   -- already normalised / zoomed."""

    def __init__(self, sculptor, fuser, photographer, *, optimizer='adam', generator_lr=0.00075,
                 g_depth_recon_loss_type='hard_smooth_l1', g_depth_recon_loss_weight=25.0, g_depth_recon_loss_k=16384,
                 g_mask_recon_loss_type='binary_cross_entropy', g_mask_recon_loss_weight=25.0, g_mask_recon_loss_k=2000,
                 g_mask_beta_loss_weight=0.0, g_mask_beta_loss_param=0.01, batch_groups=1, generator_input_depth=False,
                 depth_noise_std=0.0, process_group=None, use_amp=False):
        """use_amp: the reference's `--use-amp` (trainutils.py:41,243-246; train_reconstruct.py:455,524-532): the generator's
        forward runs under autocast.  Here the policy is bf16 (ops.autocast): same exponent range as fp32, so the
        reference's GradScaler has nothing to do and is not reproduced; master weights, gradients' accumulation, losses and
        Adam stay fp32."""
        if optimizer not in ('adam', 'adamw'):
            raise ValueError(f'Unknown optimizer {optimizer!r}')
        self.sculptor, self.fuser, self.photographer = sculptor, fuser, photographer
        for m in (sculptor, fuser, photographer):
            if m is not None:
                m.train()
        self.flat = FlatParameters([sculptor, photographer, fuser])           # the reference's parameter order (:376-381)
        self.optim = FlatAdam(self.flat, generator_lr, weight_decay=1e-2 if optimizer == 'adamw' else 0.0)
        self.depth_criterion = losses.get_recon_criterion(g_depth_recon_loss_type, g_depth_recon_loss_k)
        self.mask_criterion = losses.get_recon_criterion(g_mask_recon_loss_type, g_mask_recon_loss_k)
        self.mask_loss_type = g_mask_recon_loss_type
        self.w_depth, self.w_mask, self.w_beta = g_depth_recon_loss_weight, g_mask_recon_loss_weight, g_mask_beta_loss_weight
        self.beta_param, self.batch_groups = g_mask_beta_loss_param, batch_groups
        self.generator_input_depth, self.depth_noise_std = generator_input_depth, depth_noise_std
        self.group = process_group
        self.use_amp = bool(use_amp)
        # data-parallel ranks: gradient buckets are all-reduced while backward is still running (parallel.GradientBuckets)
        self.buckets = None
        if parallel.world()[1] > 1:
            self.buckets = parallel.GradientBuckets(self.flat.grad, self.flat.params, self.flat.offsets, process_group)

    def losses(self, batch):
        """Forward of the generator and the loss terms of train_reconstruct.py:456-516."""
        b_in, b_out = batch['in'], batch['out_gt']
        depth_in = None
        if self.generator_input_depth:
            depth_in = (b_in['depth'] + self.depth_noise_std * torch.randn_like(b_in['depth'])).clamp(-1, 1)
        with ops.autocast(self.use_amp):
            z_obj, _ = self.sculptor.encode(self.fuser, b_in['camera'], b_in['image'], depth_in, b_in['mask'])
            y, _, _ = self.photographer.decode(z_obj, b_out['camera'], return_latent=True, apply_mask=False)
        out = {}
        dev = z_obj.device
        zero = torch.zeros((), device=dev)
        out['depth_recon'] = losses.reduce_loss(self.depth_criterion(y['depth'], b_out['depth'])) \
            if 'depth' in y else zero
        if 'mask' in y:
            y_mask = y['mask_logits'] if self.mask_loss_type == 'binary_cross_entropy' else y['mask']
            out['mask_recon'] = losses.reduce_loss(self.mask_criterion(y_mask, b_out['mask']))
            out['mask_beta'] = losses.beta_prior_loss(y['mask'], alpha=self.beta_param, beta=self.beta_param)
        else:
            out['mask_recon'] = out['mask_beta'] = zero
        out['total'] = (self.w_depth * out['depth_recon'] + self.w_mask * out['mask_recon']
                        + self.w_beta * out['mask_beta']) / self.batch_groups
        return out, y

    def run_iteration(self, batch, train=True, is_step=True, zero_grad=True):
        """Returns the dictionary of (detached) loss terms.  Gradients of successive calls with
        `zero_grad=False` accumulate like the reference's batch groups (note its Q11: `Trainer.run_epoch` zeroes
        them on every stepping micro-batch, so only the last group is applied there)."""
        if train and zero_grad:
            self.flat.zero_grad()
        if self.buckets is not None:
            self.buckets.arm(train and is_step)
        with torch.set_grad_enabled(train):
            out, _ = self.losses(batch)
            if train:
                out['total'].backward()
        if train and is_step:
            if self.buckets is not None:
                self.buckets.finish()                      # most buckets were launched from the backward hooks
            else:
                parallel.allreduce_flat_(self.flat.grad, self.group)
            self.optim.step()

        return {k: v.detach() for k, v in out.items()}
