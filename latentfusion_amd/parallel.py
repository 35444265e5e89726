"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).  Replaces the reference's single-process nn.DataParallel
(latentfusion/torchutils.py:133-170), which re-broadcasts every parameter each forward.

The hot path shards on three independent axes and needs NO collective inside the pose
loop:
  objects      one object per rank (bench.py --gpus N, BASELINE cfg 4): zero communication;
  views        reference views of one object are encoded V/G per rank, then fused with ONE
               collective over the C*S^3 latent volume (all-reduce forms for pool:mean / pool:max / pool:abs_max /
               blend, all-gather + replicated ordered recurrence for the order-dependent GRU/LSTM fusers);
  hypotheses   pose samples are split across ranks; N loss scalars are all-gathered per iteration.

xGMI note: a ring all-reduce is bound by one ~153 GB/s link (134 MB volume at SYN(128,16): ~1.5 ms);
it happens once per object and is amortised over the whole pose search, so the plain RCCL
all-reduce is used (no NVSwitch-style assumptions, no per-layer traffic).
"""
import torch
import torch.distributed as dist


def world(group=None):
    """(rank, size) within `group` (the default group when None); (0, 1) outside torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _global(group, r):
    """Global rank of group-rank r (send / recv / broadcast address peers by GLOBAL rank even when given a group)."""
    return r if group is None else dist.get_global_rank(group, r)


def shard_range(n, rank, size):
    """Contiguous, balanced [begin, end) of n items for `rank` (first n % size ranks get one extra)."""
    base, extra = divmod(n, size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_views(observation, rank=None, size=None):
    """This rank's slice of the reference views (contiguous, so view order is preserved)."""
    r, s = world()
    rank, size = (r if rank is None else rank), (s if size is None else size)
    b, e = shard_range(len(observation), rank, size)
    return observation[b:e], (b, e)


def _pool(z, kind):
    """Pool over the view axis of (1, V, ...) -- lf_fuse_views_fwd for device tensors (recon/fusion.pool_tensor)."""
    from .recon.fusion import pool_tensor
    return pool_tensor(z, kind, dim=1)


def ops_blend_partial(z_local, logits, gmax):
    """This rank's share of the blend: [sum_v z_v e_v | sum_v e_v] with e = exp(logit - global max), (1,1,C+1,S,S,S)."""
    C = z_local.shape[2]
    e = torch.exp(logits - gmax)
    buf = z_local.new_empty((1, 1, C + 1) + tuple(z_local.shape[3:]))
    buf[:, :, :C] = (z_local * e).sum(dim=1, keepdim=True)
    buf[:, :, C:] = e.sum(dim=1, keepdim=True)
    return buf


def fuse_sharded(fuser, z_local, num_views_total, group=None, z_cam_mid_local=None, camera_local=None):
    """Fuses per-view latent volumes that are sharded over ranks.

    z_local: (1, V_local, C, S, S, S) volumes of this rank's views (V_local may be 0).
    Returns the fused (1, 1, C, S, S, S) volume, identical on every rank.

    pool:mean     local sum -> all-reduce(SUM) -> / V          (ONE collective of C*S^3 floats)
    pool:max      local max -> all-reduce(MAX)
    pool:abs_max  local signed abs-max a -> m = all-reduce(MAX, |a|) -> all-reduce(MAX, where(|a| == m, a, -m)):
                  +m if any rank holds +m, else -m (an exact +/- tie across ranks, a measure-zero event, resolves
                  to + instead of to the earlier view)
    blend         per-view logits l (BlendFuser.compute_blend_logits on the local views): g = all-reduce(MAX, max_v l);
                  e = exp(l - g); ONE all-reduce(SUM) of [sum_v z e | sum_v e] ((C+1)*S^3 floats); out = num / den --
                  the same max-subtracted softmax as the single-process fuser (recon/fusion.py:139-148)
    gru / lstm    order-dependent recurrences (reference recon/fusion.py:180-201): the hidden state is handed
                  from rank to rank (send / recv of ONE volume per hop), each rank runs the steps of its own views, the
                  last rank broadcasts the result (_fuse_recurrent_pipelined)
    median        all-gather of the per-view volumes in rank (= view) order (the median needs every view), then the
                  fuser runs replicated on the full, ordered view list.
    """
    rank, size = world(group)
    kind = type(fuser).__name__
    pool = getattr(fuser, 'pool_type', None)
    if size == 1:
        mids = [z_cam_mid_local] if z_cam_mid_local is not None else None
        return fuser(z_local, mids, None, camera_local)[0]
    shape = (1, 1) + tuple(z_local.shape[2:])
    have = z_local.shape[1] > 0
    if kind == 'PoolFuser' and pool == 'mean':
        acc = (_pool(z_local, 'mean') * float(z_local.shape[1]) if have else z_local.new_zeros(shape)).contiguous()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        return acc / float(num_views_total)
    if kind == 'PoolFuser' and pool == 'max':
        acc = (_pool(z_local, 'max') if have else z_local.new_full(shape, float('-inf'))).contiguous()
        dist.all_reduce(acc, op=dist.ReduceOp.MAX, group=group)
        return acc
    if kind == 'PoolFuser' and pool == 'abs_max':
        a = (_pool(z_local, 'abs_max') if have else z_local.new_zeros(shape)).contiguous()
        m = a.abs()
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        cand = torch.where(a.abs() == m, a, -m).contiguous()
        dist.all_reduce(cand, op=dist.ReduceOp.MAX, group=group)
        return cand
    if kind == 'BlendFuser':
        # the branch is chosen from rank-independent information only (a rank WITHOUT local views must still take part in
        # the same three collectives); a rank that has views but not the mid volumes the logits need is a caller error,
        # which every rank learns about through the first collective instead of some ranks hanging in it
        C = z_local.shape[2]
        bad = have and z_cam_mid_local is None
        flag = z_local.new_tensor([1.0 if bad else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if flag.item() > 0:
            raise ValueError('view-sharded BlendFuser needs the camera-block mid volumes (z_cam_mid_local) on every rank '
                             'that holds views')
        if have:
            logits = fuser.compute_blend_logits(z_cam_mid_local, camera_local)          # (1, V_local, 1, S, S, S)
            gmax = logits.max(dim=1, keepdim=True)[0].contiguous()
        else:
            gmax = z_local.new_full((1, 1, 1) + tuple(z_local.shape[3:]), float('-inf'))
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
        if have:
            buf = ops_blend_partial(z_local, logits, gmax)                             # (1, 1, C+1, S, S, S)
        else:
            buf = z_local.new_zeros((1, 1, C + 1) + tuple(z_local.shape[3:]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return (buf[:, :, :C] / buf[:, :, C:]).contiguous()
    counts = [shard_range(num_views_total, r, size) for r in range(size)]
    recurrence = getattr(fuser, 'recurrence', None)
    if recurrence in ('gru', 'lstm'):
        return _fuse_recurrent_pipelined(fuser, z_local, counts, rank, size, group, recurrence)
    # ordered all-gather (ragged: ranks may hold different numbers of views): the median needs every view
    vmax = max(e - b for b, e in counts)

    def gather_views(t):
        pad = t.new_zeros((1, vmax) + tuple(t.shape[2:]))
        pad[:, :t.shape[1]] = t
        parts = [torch.empty_like(pad) for _ in range(size)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        return torch.cat([p[:, :e - b] for p, (b, e) in zip(parts, counts)], dim=1)
    z_all = gather_views(z_local)
    return fuser(z_all, None, None, None)[0]


def _send(t, dst, group):
    """Point-to-point send; gloo moves host memory only (its send / recv take the raw data pointer), so device tensors
    are staged through the host there -- the test-only configuration (two ranks on one GPU); RCCL sends device memory."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        t = t.cpu()
    dist.send(t.contiguous(), dst=dst, group=group)


def _recv(t, src, group):
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.recv(t, src=src, group=group)
    return t


def _fuse_recurrent_pipelined(fuser, z_local, counts, rank, size, group, recurrence):
    """GRU / LSTM fusion of view-sharded volumes as a PIPELINE of the hidden state: the recurrence
    (reference recon/fusion.py:180-201, 226-246) is h <- cell(view_i, h) in view order with h_0 = view 0, so rank r
    continues it over its own contiguous views from the state rank r-1 hands over -- one point-to-point message of the
    C*S^3 state per hop (134 MB at SYN(128,16); two volumes for the LSTM's (h, c)) and one broadcast of the result,
    instead of an all-gather of all V per-view volumes into every rank (V x 134 MB) followed by V-1 replicated steps.
    The arithmetic is the single-process recurrence in the same order on the same kernels: bit-identical.

    GRU: the incoming state enters as "view 0" of the local stack (h_0 = view 0 is un-gated, so this IS the
    continuation).  LSTM: the fuser takes / returns its (h, c) through `initial_state` / the 'state' entry."""
    holders = [r for r, (b, e) in enumerate(counts) if e > b]                 # contiguous from rank 0 (shard_range)
    last = holders[-1]
    have = z_local.shape[1] > 0
    vol = (1, 1) + tuple(z_local.shape[2:])
    nstate = 2 if recurrence == 'lstm' else 1
    # (rank / size / counts are GROUP ranks; the point-to-point calls and the broadcast source take global ranks.  Inference
    # only: the result is broadcast in place, so nothing here may be an autograd output)
    with torch.no_grad():
        out = z_local.new_empty(vol)
        if have:
            state = None
            if rank > 0:
                state = z_local.new_empty((nstate,) + vol)
                _recv(state, _global(group, rank - 1), group)
            if recurrence == 'gru':
                stack = z_local if state is None else torch.cat((state[0], z_local), dim=1)
                out = fuser(stack, None, None, None)[0].contiguous()
                nxt = out.unsqueeze(0)
            else:
                init = None if state is None else (state[0][:, 0], state[1][:, 0])
                out, extra = fuser(z_local, None, None, None, initial_state=init)
                out = out.contiguous()
                nxt = torch.stack((out, extra['state'][1].unsqueeze(1)), dim=0)
            if rank < last:
                _send(nxt, _global(group, rank + 1), group)
        dist.broadcast(out, src=_global(group, last), group=group)
    return out


def build_latent_object_sharded(model, observation, group=None):
    """LatentFusionModel.build_latent_object with the reference views sharded over the ranks of
    `group`.  Every rank passes the SAME full observation; each encodes only its slice."""
    rank, size = world(group)
    obs = model.preprocess_observation(observation.to(model.device))
    total = len(obs)
    local, (b, e) = shard_views(obs, rank, size)
    with torch.no_grad():
        cap = _Capture()
        if e > b:
            z_local, _ = model.sculptor.encode(cap, camera=local.camera, color=local.color.unsqueeze(0),
                                               depth=local.depth.unsqueeze(0), mask=local.mask.unsqueeze(0))
        else:
            c, s = model.sculptor.out_channels, model.sculptor.out_size
            z_local = torch.zeros(1, 0, c, s, s, s, device=model.device)
        return fuse_sharded(model.fuser, z_local, total, group, z_cam_mid_local=cap.z_cam_mid, camera_local=cap.camera)


class _Capture:
    """Stand-in fuser: returns the un-fused per-view volumes and keeps what a BlendFuser needs for its logits."""

    z_cam_mid = camera = None

    def __call__(self, z_obj, z_cam_mid, z_obj_mid, camera):
        self.z_cam_mid = z_cam_mid[-1] if z_cam_mid else None
        self.camera = camera
        return z_obj, {}


_Identity = _Capture


def shard_hypotheses(camera, rank=None, size=None, group=None):
    r, s = world(group)
    rank, size = (r if rank is None else rank), (s if size is None else size)
    b, e = shard_range(len(camera), rank, size)
    return camera[b:e], (b, e)


def gather_rows(local_rows, n_total, group=None):
    """All-gather of per-hypothesis rows (ragged over ranks) -> (n_total, ...) on every rank, in hypothesis order.
    One small collective per pose iteration: N loss scalars (+ 10 camera parameters when the ranking needs them)."""
    rank, size = world(group)
    if size == 1:
        return local_rows
    counts = [shard_range(n_total, r, size) for r in range(size)]
    nmax = max(e - b for b, e in counts)
    pad = local_rows.new_zeros((nmax,) + tuple(local_rows.shape[1:]))
    pad[:local_rows.shape[0]] = local_rows
    parts = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    return torch.cat([p[:e - b] for p, (b, e) in zip(parts, counts)])


def gather_losses(local_losses, n_total, group=None):
    """All-gather of per-hypothesis loss scalars (ragged) -> (n_total,) on every rank, in order."""
    return gather_rows(local_losses, n_total, group)


def broadcast_(tensor, src=0, group=None):
    """In-place broadcast from rank `src` OF `group` (no-op in a single process).  Used for the few host-random quantities of
    the estimators (GMM samples, initial hypotheses) so that every rank ranks the SAME hypotheses."""
    rank, size = world(group)
    if size > 1:
        dist.broadcast(tensor, src=_global(group, src), group=group)     # (c10d addresses the source by GLOBAL rank)
    return tensor


# ---------------------------------------------------------------------------------------------
# data-parallel training (BASELINE cfg 5): gradient all-reduce over flat buckets
# ---------------------------------------------------------------------------------------------
def allreduce_flat_(flat, group=None, bucket_bytes=64 << 20):
    """In-place mean of a flat fp32 gradient buffer over the ranks, in `bucket_bytes` pieces: a handful of
    large RCCL all-reduces (ring: bound by one xGMI link, so fewer/larger beats per-parameter traffic)
    instead of the reference's per-forward parameter broadcast (torchutils.py:133-170).  Returns `flat`."""
    rank, size = world(group)
    if size == 1:
        return flat
    step = max(1, bucket_bytes // flat.element_size())
    handles = [dist.all_reduce(flat[i:i + step], op=dist.ReduceOp.SUM, group=group, async_op=True)
               for i in range(0, flat.numel(), step)]
    for h in handles:
        h.wait()
    flat.div_(size)
    return flat


class GradientBuckets:
    """Bucketed gradient all-reduce OVERLAPPED with the backward pass (data-parallel training, BASELINE cfg 5).

    The trainer keeps every gradient in ONE flat buffer (recon/training.FlatParameters); it is cut into `bucket_bytes`
    pieces and a piece is all-reduced (asynchronously) as soon as every parameter that overlaps it has received its
    gradient, while autograd is still working on the earlier layers -- the large, few RCCL calls of `allreduce_flat_`
    (ring all-reduce over xGMI is bound by one ~153 GB/s link: fewer and larger beats per-parameter traffic), started early
    instead of after `backward()` returns.  `finish()` launches whatever did not complete (parameters that got no gradient
    this step), waits, and divides by the world size: the result equals `allreduce_flat_(flat.grad)` bit for bit (same
    buckets, same reduction per bucket).  Replaces the reference's single-process DataParallel, which re-broadcasts
    all parameters every forward and gathers gradients onto one device (latentfusion/torchutils.py:133-170)."""

    def __init__(self, flat_grad, params, offsets, group=None, bucket_bytes=64 << 20):
        self.grad, self.group = flat_grad, group
        self.step = max(1, bucket_bytes // flat_grad.element_size())
        self.nb = (flat_grad.numel() + self.step - 1) // self.step
        self.members = [0] * self.nb                      # parameters overlapping each bucket
        self.of_param = []
        for p, off in zip(params, offsets):
            b0, b1 = off // self.step, (off + max(p.numel(), 1) - 1) // self.step
            self.of_param.append((b0, b1))
            for b in range(b0, b1 + 1):
                self.members[b] += 1
        self.handles = []
        self.armed = False                                 # hooks reduce only during the backward of a STEPPING iteration
        self.reset()
        self.hooks = [p.register_post_accumulate_grad_hook(self._hook(i)) for i, p in enumerate(params)]

    def reset(self):
        self.left = list(self.members)
        self.launched = [False] * self.nb
        self.next = self.nb - 1                            # buckets leave in ONE order on every rank: nb-1, nb-2, ..., 0
        self.handles = []

    def _launch(self, b):
        if not self.launched[b]:
            self.launched[b] = True
            piece = self.grad[b * self.step:(b + 1) * self.step]
            self.handles.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _drain(self, force=False):
        """Launches the ready buckets in descending index order (gradients arrive last layer first, i.e. from the end of the flat
        buffer): c10d needs every rank to issue its collectives in the same order, and the order in which hooks happen to
        complete buckets may differ between ranks (a parameter unused on one rank, data-dependent branches) -- a bucket
        therefore waits for every bucket above it, as torch's DDP does."""
        while self.next >= 0 and (force or self.left[self.next] <= 0):
            self._launch(self.next)
            self.next -= 1

    def _hook(self, i):
        def fire(_param):
            if not self.armed:                             # gradient-accumulation micro-batch: nothing leaves the rank yet
                return
            b0, b1 = self.of_param[i]
            for b in range(b0, b1 + 1):
                self.left[b] -= 1
            self._drain()
        return fire

    def arm(self, on=True):
        """Call before the backward of an iteration: on = this iteration ends in an optimiser step."""
        self.reset()
        self.armed = bool(on)

    def finish(self):
        """Call after backward(): reduces the buckets that are still open, waits for all, averages.  Returns the flat buffer."""
        self.armed = False
        rank, size = world(self.group)
        if size > 1:
            self._drain(force=True)
            for h in self.handles:
                h.wait()
            self.grad.div_(size)
        self.reset()
        return self.grad

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []
