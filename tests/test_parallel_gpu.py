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
        assert res[r]['build_diff'][0] <= 2e-6 * max(1.0, res[r]['build_max'][0]), res[r]['build_diff']
        best1, hist1 = res[r][('grad', False)]
        best2, hist2 = res[r][('grad', True)]
        # per-iteration losses of all 5 hypotheses and the final ranking: sharded == one rank (same kernels, same inputs per
        # hypothesis; the slices only change which rank evaluates them)
        torch.testing.assert_close(torch.from_numpy(hist2), torch.from_numpy(hist1), atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(torch.from_numpy(best2), torch.from_numpy(best1), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(torch.from_numpy(res[1][('grad', True)][0]), torch.from_numpy(res[0][('grad', True)][0]), atol=0, rtol=0)
