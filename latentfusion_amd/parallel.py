"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).  Replaces the reference's single-process nn.DataParallel
(latentfusion/torchutils.py:133-170), which re-broadcasts every parameter each forward.

The hot path shards on three independent axes and needs NO collective inside the pose
loop:
  objects      one object per rank (bench.py --gpus N, BASELINE cfg 4): zero communication;
  views        reference views of one object are encoded V/G per rank, then fused with ONE
               collective over the C*S^3 latent volume (all-reduce for pool:mean / pool:max,
               all-gather + replicated ordered recurrence for the order-dependent GRU/LSTM fusers);
  hypotheses   pose samples are split across ranks; N loss scalars are all-gathered per iteration.

xGMI note: a ring all-reduce is bound by one ~153 GB/s link (134 MB volume at SYN(128,16): ~1.5 ms);
it happens once per object and is amortised over the whole pose search, so the plain RCCL
all-reduce is used (no NVSwitch-style assumptions, no per-layer traffic).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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


def fuse_sharded(fuser, z_local, num_views_total, group=None):
    """Fuses per-view latent volumes that are sharded over ranks.

    z_local: (1, V_local, C, S, S, S) volumes of this rank's views (V_local may be 0).
    Returns the fused (1, 1, C, S, S, S) volume, identical on every rank.

    pool:mean   local sum -> all-reduce(SUM) -> / V          (one collective of C*S^3 floats)
    pool:max    local max -> all-reduce(MAX)
    others      all-gather of the per-view volumes in rank (= view) order, then the fuser runs
                replicated on the full, ordered view list: GRU/LSTM fusion is an order-dependent
                recurrence (reference recon/fusion.py:180-201) and abs_max/median are
                not reducible with a single all-reduce.
    """
    rank, size = world()
    kind = type(fuser).__name__
    pool = getattr(fuser, 'pool_type', None)
    if size == 1:
        return fuser(z_local, None, None, None)[0]
    if kind == 'PoolFuser' and pool in ('mean', 'max'):
        shape = (1, 1) + tuple(z_local.shape[2:])
        if pool == 'mean':
            acc = z_local.sum(dim=1, keepdim=True) if z_local.shape[1] else z_local.new_zeros(shape)
            acc = acc.contiguous()
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
            return acc / float(num_views_total)
        acc = (z_local.max(dim=1, keepdim=True)[0] if z_local.shape[1]
               else z_local.new_full(shape, float('-inf'))).contiguous()
        dist.all_reduce(acc, op=dist.ReduceOp.MAX, group=group)
        return acc
    # ordered all-gather (ragged: ranks may hold different numbers of views)
    counts = [shard_range(num_views_total, r, size) for r in range(size)]
    vmax = max(e - b for b, e in counts)
    pad = z_local.new_zeros((1, vmax) + tuple(z_local.shape[2:]))
    pad[:, :z_local.shape[1]] = z_local
    parts = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    z_all = torch.cat([p[:, :e - b] for p, (b, e) in zip(parts, counts)], dim=1)
    return fuser(z_all, None, None, None)[0]


def build_latent_object_sharded(model, observation, group=None):
    """LatentFusionModel.build_latent_object with the reference views sharded over the ranks of
    `group`.  Every rank passes the SAME full observation; each encodes only its slice."""
    rank, size = world()
    obs = model.preprocess_observation(observation.to(model.device))
    total = len(obs)
    local, (b, e) = shard_views(obs, rank, size)
    with torch.no_grad():
        if e > b:
            z_local, _ = model.sculptor.encode(_Identity(), camera=local.camera, color=local.color.unsqueeze(0),
                                               depth=local.depth.unsqueeze(0), mask=local.mask.unsqueeze(0))
        else:
            c, s = model.sculptor.out_channels, model.sculptor.out_size
            z_local = torch.zeros(1, 0, c, s, s, s, device=model.device)
        return fuse_sharded(model.fuser, z_local, total, group)


class _Identity:
    """Stand-in fuser that returns the un-fused per-view volumes."""

    def __call__(self, z_obj, z_cam_mid, z_obj_mid, camera):
        return z_obj, {}


def shard_hypotheses(camera, rank=None, size=None):
    r, s = world()
    rank, size = (r if rank is None else rank), (s if size is None else size)
    b, e = shard_range(len(camera), rank, size)
    return camera[b:e], (b, e)


def gather_losses(local_losses, n_total, group=None):
    """All-gather of per-hypothesis loss scalars (ragged) -> (n_total,) on every rank, in order."""
    rank, size = world()
    if size == 1:
        return local_losses
    counts = [shard_range(n_total, r, size) for r in range(size)]
    nmax = max(e - b for b, e in counts)
    pad = local_losses.new_zeros(nmax)
    pad[:local_losses.shape[0]] = local_losses
    parts = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:e - b] for p, (b, e) in zip(parts, counts)])


# ---------------------------------------------------------------------------------------------
# data-parallel training (BASELINE cfg 5): gradient all-reduce over flat buckets
# ---------------------------------------------------------------------------------------------
def allreduce_flat_(flat, group=None, bucket_bytes=64 << 20):
    """In-place mean of a flat fp32 gradient buffer over the ranks, in `bucket_bytes` pieces: a handful of
    large RCCL all-reduces (ring: bound by one xGMI link, so fewer/larger beats per-parameter traffic)
    instead of the reference's per-forward parameter broadcast (torchutils.py:133-170).  Returns `flat`."""
    rank, size = world()
    if size == 1:
        return flat
    step = max(1, bucket_bytes // flat.element_size())
    handles = [dist.all_reduce(flat[i:i + step], op=dist.ReduceOp.SUM, group=group, async_op=True)
               for i in range(0, flat.numel(), step)]
    for h in handles:
        h.wait()
    flat.div_(size)
    return flat
