"""The multi-rank paths on REAL kernels, on a one-GPU box: two processes share cuda:0 and talk through gloo (RCCL refuses
two ranks on one device; the collectives' payloads and the code around them are the same).  SURVEY 8(e):
  * hypothesis sharding of the gradient estimator (engine path) -- the ranking of the sharded run equals the one-rank run;
  * view-sharded reconstruction with pool:mean + all-reduce -- the fused volume equals the local build."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, size, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        from latentfusion_amd import parallel, synth
        from latentfusion_amd.modules.geometry import Camera
        from latentfusion_amd.observation import Observation
        from latentfusion_amd.pose import estimation, utils as pu
        dev = 'cuda:0'
        model, _ = synth.build_model(32, 16, 'pool:mean', seed=0, device=dev)      # same object on both ranks
        V = 6
        ref = synth.make_observation(V, seed=100, device=dev)
        z_local = model.build_latent_object(ref)
        # view-sharded build: rank r encodes its slice of the views, ONE all-reduce of the fused volume
        z_sh = parallel.build_latent_object_sharded(model, ref)
        out = {'build_diff': (z_sh - z_local).abs().max().reshape(1).cpu().numpy(),
               'build_max': z_local.abs().max().reshape(1).cpu().numpy()}
        # GRU fuser (the released recipe): the hidden state is handed from rank 0 to rank 1 (send / recv), rank 1 continues
        # the recurrence over its own views and broadcasts the result -- bit-identical to the local build; 5 views: 3 + 2
        gru_model, _ = synth.build_model(32, 16, 'gru', seed=1, device=dev)
        ref5 = synth.make_observation(5, seed=101, device=dev)
        out['gru_diff'] = (parallel.build_latent_object_sharded(gru_model, ref5)
                           - gru_model.build_latent_object(ref5)).abs().max().reshape(1).cpu().numpy()
        del gru_model, ref5
        td = synth.make_observation_data(1, seed=200)
        target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)
        torch.manual_seed(3)
        init = pu.sample_cameras_with_estimate(5, target.camera.to('cpu')).to(dev)  # 5 hypotheses over 2 ranks: 3 + 2
        w = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.1, 'mask': 0.2}
        for sharded in (False, True):
            g = estimation.GradientPoseEstimator(model=model, learning_rate=0.01, num_samples=5, num_iters=4, ranking_size=4,
                                                 converge_threshold=1e-9, converge_patience=100, optimizer='adam', loss_weights=w,
                                                 shard_hypotheses=sharded, return_camera_history=True)
            best, hist = g.estimate(z_local, target, camera=init)
            out[('grad', sharded)] = (torch.cat((best.log_quaternion, best.translation), dim=1).cpu().numpy(),
                                      torch.stack([h[0] for h in hist]).cpu().numpy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_estimator_and_build_on_hip_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]['gru_diff'][0] == 0.0, res[r]['gru_diff']
        assert res[r]['build_diff'][0] <= 2e-6 * max(1.0, res[r]['build_max'][0]), res[r]['build_diff']
        best1, hist1 = res[r][('grad', False)]
        best2, hist2 = res[r][('grad', True)]
        # per-iteration losses of all 5 hypotheses and the final ranking: sharded == one rank (same kernels, same inputs per
        # hypothesis; the slices only change which rank evaluates them)
        torch.testing.assert_close(torch.from_numpy(hist2), torch.from_numpy(hist1), atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(torch.from_numpy(best2), torch.from_numpy(best1), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(torch.from_numpy(res[1][('grad', True)][0]), torch.from_numpy(res[0][('grad', True)][0]), atol=0, rtol=0)


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from latentfusion_amd import parallel, synth
        dev = 'cuda:0'
        out = {'backend': dist.get_backend(), 'version': '.'.join(str(v) for v in torch.cuda.nccl.version())}
        x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
        y = x.clone()
        dist.all_reduce(y)                                                     # RCCL kernel on a device tensor
        parts = [torch.empty_like(x)]
        dist.all_gather(parts, x)
        dist.broadcast(y, src=0)
        torch.cuda.synchronize()
        out['collectives_ok'] = bool(torch.equal(y, x) and torch.equal(parts[0], x))
        # the library's own wrappers, on device tensors, inside an initialised RCCL group
        flat = torch.ones(3_000_000, device=dev)
        parallel.allreduce_flat_(flat, bucket_bytes=1 << 20)
        rows = parallel.gather_rows(torch.arange(8.0, device=dev).view(4, 2), 4)
        parallel.broadcast_(flat)
        model, _ = synth.build_model(16, 16, 'pool:mean', seed=0, device=dev)
        ref = synth.make_observation(3, seed=100, device=dev)
        z_sh = parallel.build_latent_object_sharded(model, ref)
        torch.cuda.synchronize()
        out['wrappers_ok'] = bool(flat.sum().item() == 3_000_000 and rows.shape == (4, 2)
                                  and torch.equal(z_sh, model.build_latent_object(ref)))
        dist.barrier()
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_initialises_and_runs_collectives_on_device_tensors():
    """`backend='nccl'` IS RCCL on ROCm.  A one-GPU box cannot host two RCCL ranks, but it can prove that RCCL loads and
    initialises under this environment (HSA_ENABLE_IPC_MODE_LEGACY=0), that its all-reduce / all-gather / broadcast kernels
    run on device tensors, and that the library's collective wrappers work inside an initialised RCCL group -- so the
    driver's multi-GPU run is not the first time this code meets RCCL (replaces the reference's DataParallel layer,
    latentfusion/torchutils.py:133-170)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out['backend'] == 'nccl' and out['collectives_ok'] and out['wrappers_ok'], out
