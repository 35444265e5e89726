#!/usr/bin/env python
"""Benchmark of the reconstruct-and-render hot loop (BASELINE.json metric):

    pose-optim iters/sec (reconstruct + render + backward), 16 views, 128^3 voxels

One "step" = one pose-optimisation iteration of GradientPoseEstimator on the adam_quick preset:
render N=8 pose samples of the fused latent volume (O2C resample, 2 fused conv3d blocks, factor
projection, 2-D decoder), pose loss against the target frame, backward to the 10 camera
parameters of every sample, batched Adam + plateau step, ranking.  Workload: SYN(128,16)
(SURVEY 8d), V=16 reference views, fp32, synthetic data, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                      # bench.py starts the N ranks itself (launch_ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` IS the number of ranks: without a launcher environment bench.py spawns N processes (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, RCCL process group); under a launcher it checks WORLD_SIZE == N and fails
loudly otherwise.  This replaces the reference's single-command multi-GPU path (latentfusion/torchutils.py:133-170
MyDataParallel, `--data-parallel` in tools/train/train.sh:66).

Multi-GPU (weak scaling): every rank owns one object (its own latent volume, target and pose
samples; BASELINE cfg 4) -- the pose loop has no data-path collective; value = total
iterations/s over all ranks, timed between barriers with the max over ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# RCCL / device-tensor sharing between the ranks needs dmabuf IPC on this platform (hipIpcGetMemHandle fails with the legacy
# mode); the driver exports this already -- keep it when bench.py is started from a bare environment
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=128, help='latent volume side S of SYN(S,C)')
    ap.add_argument('--channels', type=int, default=16)
    ap.add_argument('--views', type=int, default=16)
    ap.add_argument('--samples', type=int, default=8)
    ap.add_argument('--fuser', default='gru')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=4,
                    help='timed oracle iterations of the CPU baseline (after 1 warm-up); their per-iteration trace is compared with '
                         'the HIP loop from the same start (trace_parity_at_full_size)')
    ap.add_argument('--no-alt', action='store_true', help='skip the secondary f16x3 measurement')
    ap.add_argument('--no-graph', action='store_true', help='(accepted, no effect: the hipGraph block moved to tools/engine_ab.py)')
    ap.add_argument('--repeats', type=int, default=30,
                    help='the timed region (exactly --steps iterations between barriers) is run this many times back to back; '
                         '`value` is steps / MEDIAN block time, every block time is reported (box-to-box and run-to-run '
                         'spread of a 0.1 s region is a few per cent)')
    ap.add_argument('--no-build-parity', action='store_true',
                    help='skip the CPU oracle reconstruction of the 16-view object (build_parity_at_full_size); the CPU pose '
                         'loop then runs on the GPU-built volume')
    ap.add_argument('--sharded-build', action='store_true',
                    help='reconstruct ONE pool:mean object with its reference views sharded over the ranks and fused by the '
                         'RCCL all-reduce of the latent volume (parallel.build_latent_object_sharded); reported as '
                         '`sharded_build`.  ON by default when N > 1 (the north-star collective); this flag forces it at N = 1')
    ap.add_argument('--no-sharded-build', action='store_true', help='skip the view-sharded build at N > 1')
    ap.add_argument('--fuse-projection', default='default', choices=['default', 'none', 'fwd'],
                    help="factor projection fused into the last camera block's Winograd kernels (engine.py): 'default' = forward form")
    ap.add_argument('--conv-mode', default='winograd', choices=['fp32', 'winograd', 'f16x3'],
                    help="conv3d kernels of the engine: 'winograd' (default; F(2^3,3^3) minimal filtering, all-fp32 "
                         "arithmetic), 'fp32' (direct implicit GEMM on the fp32 MFMA) or 'f16x3' (split precision)")
    ap.add_argument('--no-cfg3', action='store_true',
                    help='skip the secondary cfg 3 block (released architecture, 8 views, cross_entropy_linemod: 128 renders per '
                         'iteration on the fused engine)')
    ap.add_argument('--cfg3-iters', type=int, default=10)
    ap.add_argument('--no-cfg5', action='store_true',
                    help='skip the secondary cfg 5 block (one bf16-autocast generator training step, 32 + 8 views, SYN(128,16))')
    ap.add_argument('--cfg5-steps', type=int, default=5)
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='do not run the two rocprofv3 counter passes for `roofline.traffic` (tools/pmc_live.py, ~10 s at N = 1); the '
                         'committed profiles/r*_hbm_bytes.json is replayed instead')
    ap.add_argument('--no-variants', action='store_true',
                    help="skip the secondary block with the 'sum' and occlusion renderer variants on the engine")
    ap.add_argument('--launcher-selftest', action='store_true',
                    help='run only the rank plumbing (spawn / process group / one all-reduce on HOST tensors over gloo) and '
                         'print a JSON line with n_gpus and ranks_seen: the CPU test of the N-rank path (tests/test_parallel.py)')
    ap.add_argument('--no-hypothesis-sharding', action='store_true',
                    help='skip the strong-scaling section at N > 1 (ONE object, its pose hypotheses sharded over the ranks)')
    ap.add_argument('--no-pipelined-build', action='store_true',
                    help='skip the GRU view-sharded build (hidden state handed rank to rank) at N > 1')
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks from here -- one process per
    GPU, the environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on
    127.0.0.1), the same command line.  Rank 0's stdout (the ONE JSON line) is this process's stdout; the other ranks'
    output goes to stderr.  A rank that fails takes the run down: the others are terminated (by their own PIDs) and the exit
    code is the failing rank's -- a silent N = 1 run under an `--gpus N` label cannot happen."""
    n = a.gpus
    port = _free_port()
    script = os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), LF_BENCH_SPAWNED='1')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def forward():
        # rank 0's JSON line(s) -> stdout; anything a library prints on stdout (gloo's connection banner) -> stderr
        for line in procs[0].stdout:
            (sys.stdout if line.lstrip().startswith('{') else sys.stderr).write(line)
            sys.stdout.flush()
    import threading
    fwd = threading.Thread(target=forward, daemon=True)
    fwd.start()
    code = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                sys.stderr.write(f'bench.py: rank {r} of {n} exited with {rc}; stopping the other ranks\n')
                for o in alive:
                    procs[o].terminate()
        if alive:
            time.sleep(0.05)
    fwd.join(timeout=10)
    return code


def launcher_selftest(a, rank, world):
    """The rank plumbing alone, on host tensors over gloo (no GPU): what tests/test_parallel.py runs here."""
    import torch.distributed as dist
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):
        dist.init_process_group('gloo')
    seen = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.tensor([float(rank + 1)])
    if dist.is_initialized():
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'gpus_flag': a.gpus, 'ranks_seen': seen,
                          'allreduce_sum': t.item(), 'spawned_by_bench': os.environ.get('LF_BENCH_SPAWNED') == '1'}))


def cpu_baseline(cks, z_obj_gpu_cpu, ref_data, target_data, init, cfg, iters, build):
    """The oracle (CPU restatement of the reference, pinned by tests/golden) on the SAME workload: reconstruction of the
    object from the same reference views (`build`), then `iters` timed iterations of the same pose loop after one warm-up
    iteration -- on the ORACLE's own volume, so that nothing on this leg comes from the GPU."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import lf_oracle as O
    from lf_oracle import pose as opose
    sck, fck, pck, dist = cks
    # the oracle's ATen CPU ops peak at ~32 threads on the GPU box's host (measured: 16/32/64/128
    # threads -> 1.52/1.33/1.57/2.80 s per SYN(64,16) iteration); more threads only oversubscribe
    torch.set_num_threads(min(32, os.cpu_count()))
    model = opose.Model(sck, fck, pck, dist)
    build_info, z_obj_cpu = None, z_obj_gpu_cpu
    if build:
        ref = opose.Obs(ref_data['color'], ref_data['depth'], ref_data['mask'],
                        O.Cam.from_extrinsic(ref_data['intrinsic'], ref_data['extrinsic']))
        t0 = time.perf_counter()
        z_obj_cpu = model.build_latent_object(ref)
        t_b = time.perf_counter() - t0
        d = z_obj_gpu_cpu - z_obj_cpu
        build_info = {'t_oracle_build_s': t_b, 'max_abs_diff': d.abs().max().item(), 'max_abs': z_obj_cpu.abs().max().item(),
                      'rel_l2': (d.norm() / z_obj_cpu.norm()).item(),
                      'what': 'fused latent volume of the HIP reconstruction vs the CPU oracle reconstruction of the same '
                              'reference views (per-view encode, camera->object resampling, ConvGRU recurrence)'}
    target = opose.Obs(None, target_data['depth'], target_data['mask'],
                       O.Cam.from_extrinsic(target_data['intrinsic'], target_data['extrinsic']))
    cam0 = O.Cam(init['K'], init['log_q'], init['t'])
    c = dict(cfg)
    c['args'] = dict(cfg['args'])
    c['args']['num_iters'] = 1
    c['args']['converge_patience'] = 10 ** 6
    _, first = opose.gradient_estimate(model, z_obj_cpu, target, cam0, c)   # warm-up (page-in, thread pools)
    c['args']['num_iters'] = iters
    t0 = time.perf_counter()
    _, trace = opose.gradient_estimate(model, z_obj_cpu, target, cam0, c)
    dt = time.perf_counter() - t0
    # iteration 0 of the oracle on ITS volume / the same target / initial cameras: the full-size parity check;
    # `trace`: the timed iterations 0 .. iters-1 from the same start (per-iteration rank losses, parameters)
    ref0 = {'rank_loss': first['rank_loss'][0],
            'grad': torch.cat((first['grad_log_q'][0], first['grad_t'][0], first['grad_viewport'][0]), dim=1),
            'trace_rank_loss': trace['rank_loss'], 'trace_log_q': trace['log_q'], 'trace_t': trace['t']}
    return iters / dt, torch.get_num_threads(), ref0, dt, build_info, z_obj_cpu


def sharded_build_report(a, S, C, V, dev, world, barrier):
    """The north-star collective (reported as `sharded_build`): every rank builds the SAME pool:mean model and observation;
    the reference views are split over the ranks and the per-view volumes meet in ONE RCCL all-reduce of the C*S^3 volume."""
    import torch.distributed as dist
    from latentfusion_amd import synth
    # the north-star collective: every rank builds the SAME pool:mean model and observation; the reference views are
    # split over the ranks and the per-view volumes meet in ONE RCCL all-reduce of the C*S^3 latent volume
    from latentfusion_amd import parallel
    sh_fuser = 'pool:mean'
    model0, _ = synth.build_model(S, C, sh_fuser, seed=12345, device=dev)
    obs0 = synth.make_observation(V, seed=54321, device=dev)
    for rep in range(2):                                       # second pass = warm
        barrier()
        t0 = time.perf_counter()
        z_sh = parallel.build_latent_object_sharded(model0, obs0)
        barrier()
        t_sh = time.perf_counter() - t0
    z_full = model0.build_latent_object(obs0)                  # local full build for comparison
    err = (z_sh - z_full).abs().max().item()
    # the collective alone: all-reduce of one latent volume, timed with barriers on both sides
    ar_ms = None
    if world > 1:
        buf = torch.empty_like(z_full)
        for rep in range(3):
            barrier()
            t0 = time.perf_counter()
            dist.all_reduce(buf)
            barrier()
            ar_ms = (time.perf_counter() - t0) * 1e3
    sharded = {'t_s': t_sh, 'fuser': sh_fuser, 'views': V, 'ranks': world, 'max_abs_diff_vs_local_build': err,
               'volume_MB': z_full.numel() * 4 / 1e6, 'allreduce_ms': ar_ms,
               'allreduce_algbw_GBps': (z_full.numel() * 4 / 1e9) / (ar_ms * 1e-3) if ar_ms else None}

    return sharded


def cfg3_report(a, dev):
    """BASELINE cfg 3 (secondary block, never `value`): the RELEASED architecture (tools/train/train.sh:28-66: 256^2 inputs, 16^3 x
    256-channel volume, 512-channel U-Net levels, GRU fuser; 68 M seeded random parameters -- the public checkpoint and LINEMOD
    are not obtainable offline), 8 reference views, the cross_entropy_linemod preset: 32 samples x 4 flips = 128 renders per
    iteration, ranked without gradients (reference pose/estimation.py:298-497).  The hypotheses are scored on the fused
    engine (forward only, lf_pose_loss_fwd_masked).  Reported: iterations/s and renders/s of the warm loop (host GMM fits
    included, as in the reference), and the launch time / fp32-MFMA fraction of the dominant kernel of this model,
    wino_fused_kernel<3> (the 256 -> 256 camera-block convolutions as Winograd GEMMs with the output transform folded in)."""
    import numpy as np
    from latentfusion_amd import ops, synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    t0 = time.perf_counter()
    model, _ = synth.build_released_model(dev, seed=0)
    model.freeze()
    t_model = time.perf_counter() - t0
    V3 = 8
    ref = synth.make_observation(V3, seed=100, device=dev)
    td = synth.make_observation_data(1, seed=200)
    target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)
    builds = []
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z_obj = model.build_latent_object(ref)
        torch.cuda.synchronize()
        builds.append(time.perf_counter() - t0)
    cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'cross_entropy_linemod.toml'))
    cfg['args']['num_iters'] = a.cfg3_iters
    n_r = cfg['args']['num_samples']
    est = estimation.load_from_config(cfg, model)
    runs, timer = [], []
    for rep in range(3):
        torch.manual_seed(300)
        np.random.seed(300)
        if rep == 2:
            ops.KERNEL_TIMER_TAGS = {'wino3d_fused'}
            ops.KERNEL_TIMER = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        best = est.estimate(z_obj, target, camera=target.camera)
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
        if rep == 2:
            timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    on_engine = bool(getattr(est, 'last_scored_on_engine', False))   # (estimate() drops its engine on return)
    el = min(runs[1:2])                                            # the un-instrumented warm run
    d3 = [e0.elapsed_time(e1) for n_, e0, e1 in timer if n_ == 'wino3d_fused']
    Cw, Sw = 256, 16
    tiles = n_r * (Sw // 2) ** 3
    fl = 2.0 * 64 * tiles * Cw * Cw                                # executed: 64 Winograd frequencies x tiles x Cin x Cout
    out = {'workload': f'released architecture (68 M seeded random parameters), {V3} reference views, cross_entropy_linemod: '
                       f'{n_r} renders per iteration, no gradient; synthetic observation',
           'value': a.cfg3_iters / el, 'unit': 'iters/s', 'renders_per_s': a.cfg3_iters * n_r / el, 'iterations': a.cfg3_iters,
           'run_s': runs, 't_build_s': builds[0], 't_build_warm_s': builds[1], 't_model_setup_s': t_model,
           'scored_on_fused_engine': bool(on_engine), 'ranking_size_returned': len(best)}
    if d3:
        ms = sum(d3) / len(d3)
        out['roofline'] = {'bound': 'mfma', 'kernel': 'wino_fused_kernel<3> (3-D 256 -> 256 @ 16^3, N = %d: per-frequency fp32-MFMA GEMM '
                                                      '+ output transform + He + bias + LeakyReLU)' % n_r,
                           'achieved': fl / (ms * 1e-3) / 1e12, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 'avg_launch_ms': ms, 'launches_timed': len(d3),
                           'executed_flops_per_launch': fl,
                           'algorithmic_flops_per_launch': 2.0 * 27 * Cw * Cw * n_r * Sw ** 3, 'traffic': None}
        # measured HBM bytes per launch (PMC passes over tools/cfg3_probe.py), refused when the kernel source changed since
        import glob
        import hashlib
        for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_released_arch_hbm_bytes.json')), reverse=True):
            try:
                tj = json.load(open(tpath))
                if n_r == 128 and all(hashlib.sha256(open(os.path.join(ROOT, 'latentfusion_amd', 'csrc', f), 'rb').read()).hexdigest() == h
                                      for f, h in tj['source_sha256'].items()):
                    out['roofline']['traffic'] = tj['kernels']['wino_fused_kernel<3>']['bytes_per_launch']
                    out['roofline']['traffic_source'] = os.path.basename(tpath)
                    break
            except Exception:                                       # noqa: BLE001
                continue
    return out


def variants_report(a, dev):
    """Secondary block (never `value`): the reference's other two renderer variants on the render-loop engine, same loop shape as the
    headline (SYN(128,16), N = 8, adam_quick; seeded random renderer) -- the 'sum' projection and the occlusion module
    (reference recon/models.py:378-395,427-437; round 5: the occlusion U-Net on explicit kernels, engine._plan_occlusion)."""
    from latentfusion_amd import consts, synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation, utils as pu
    from latentfusion_amd.recon.utils import optimal_camera_dist
    from latentfusion_amd.recon.models import Photographer
    S, C, N, K = 128, 16, 8, 10
    td = synth.make_observation_data(1, seed=200)
    target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)
    z_obj = torch.randn(1, 1, C, S, S, S, generator=torch.Generator().manual_seed(0)).to(dev)
    cdist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
    cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
    cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
    torch.manual_seed(300)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    out = {'workload': f'SYN({S},{C}) renderer with seeded random weights, N = {N}, adam_quick, median of 3 blocks of {K} timed iterations after 3 warm-up'}
    for name, kw in (('sum', dict(projection_type='sum', object_config=[], occlusion_config=False)),
                     ('occlusion', dict(projection_type='factor', object_config=[16, 16], occlusion_config=[[17, 16], [16, 16]]))):
        torch.manual_seed(1)
        ph = Photographer(in_size=S, camera_config=[C, C], predict_color=False, predict_depth=True, predict_mask=True,
                          scale_mode='nearest', cube_size=1.0, image_config=[[16, 32], [32, 16]], **kw).to(dev)
        for p in ph.parameters():
            p.requires_grad_(False)

        class M:                                                      # the facade surface the estimator touches
            photographer, device, input_size, camera_dist = ph, torch.device(dev), S, cdist
        est = estimation.load_from_config(cfg, M, converge_patience=10 ** 6)
        st = est.start(z_obj, target, init.zoom(None, S, cdist).to(dev))
        for _ in range(3):
            est.iterate(st)
        els = []
        for _blk in range(3):                                         # median of three blocks of K iterations
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                est.iterate(st)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        out[name] = {'iters_per_s': K / el, 'ms_per_iteration': el / K * 1e3, 'blocks_ms': [e / K * 1e3 for e in els], 'on_engine': 'engine' in st,
                     'explicit_occlusion_kernels': bool(getattr(st.get('engine'), 'occ', None) is not None)}
        del est, st, ph
        torch.cuda.empty_cache()
    return out


def cfg5_report(a, dev):
    """BASELINE cfg 5 (secondary block, never `value`): ONE generator training step of the reference trainer
    (tools/train/train_reconstruct.py:421-535, recon/utils.py:68-127, losses.py:33-57) on SYN(128,16) with the GRU fuser:
    32 input views encoded and fused, 8 output views rendered, hard smooth-L1 depth + BCE mask losses, backward through every
    kernel (data, weight and bias gradients, deterministic volume splat), flat Adam -- under the bf16 autocast policy
    (ops.autocast; `--use-amp` of the reference).  One GPU: the data-parallel half (bucketed RCCL gradient all-reduce) needs
    more devices.  Reported: wall time per step (MEDIAN of `--cfg5-steps` steps after three warm-up steps; every step time is listed), peak device memory, the
    launch time and HBM fraction of the dominant training kernel from HIP events in one extra step, and whether two fresh
    runs of the same steps agree bit for bit."""
    from latentfusion_amd import ops, synth
    from latentfusion_amd.recon import training
    S, C, Vi, Vo = a.size, a.channels, 32, 8

    def fresh():
        model, _ = synth.build_model(S, C, 'gru', seed=0, device=dev)
        obs_in = model.preprocess_observation(synth.make_observation(Vi, seed=1, device=dev))
        obs_out = model.preprocess_observation(synth.make_observation(Vo, seed=2, device=dev))
        step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=True)
        batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
                 'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}
        return step, batch

    def run(step, batch, k):
        times, losses = [], []
        for _ in range(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            l_ = step.run_iteration(batch)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            losses.append(float(l_['total']))
        return times, losses
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    resident_before = torch.cuda.memory_allocated() / 2 ** 30      # what earlier blocks of this process still hold (headline volume, ...)
    step, batch = fresh()
    K = max(1, a.cfg5_steps)
    WU = 3                                                        # warm-up steps: the caching allocator re-grows its 30 GB pool over the first steps
    times, losses = run(step, batch, WU + K)                      # (one full bench run of round 6 showed 117 / 112 / 103 / 87 / 86 ms after a single warm-up step)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    params_after = step.flat.data.clone()
    # HIP events around the bf16 ring convolution (forward / data-gradient / ConvGRU addend forms) in one extra, untimed step
    ops.KERNEL_TIMER_TAGS = {'conv3d_c16_ring_bf16', 'conv3d_c16_ring_multi'}
    ops.KERNEL_TIMER = []
    step.run_iteration(batch)
    torch.cuda.synchronize()
    timer, ops.KERNEL_TIMER, ops.KERNEL_TIMER_TAGS = ops.KERNEL_TIMER, None, None
    # algorithmic bytes of the STEP: one more untimed step with the library's byte accounting on (_lib.BYTE_LOG: per launch,
    # the sizes of its input / output tensors, each once, scratch excluded)
    from latentfusion_amd import _lib
    _lib.BYTE_LOG = {}
    step.run_iteration(batch)
    torch.cuda.synchronize()
    byte_log, _lib.BYTE_LOG = _lib.BYTE_LOG, None
    n_par = sum(q.numel() for q in step.flat.params)
    del step, batch
    torch.cuda.empty_cache()
    # the same WU + K steps from a fresh model: identical losses and identical parameters, bit for bit
    step2, batch2 = fresh()
    _, losses2 = run(step2, batch2, WU + K)
    identical = losses == losses2 and bool(torch.equal(params_after, step2.flat.data))
    del step2, batch2, params_after
    torch.cuda.empty_cache()
    ms = sorted(times[WU:])[(K - 1) // 2] * 1e3                    # median step (a single step can catch an allocator / clock hiccup)
    out = {'workload': f'one generator training step (reference tools/train/train_reconstruct.py:421-535) on SYN({S},{C}), GRU fuser: '
                       f'{Vi} input views + {Vo} output views, bf16 autocast policy, flat Adam; 1 GPU (no data-parallel all-reduce)',
           'ms_per_step': ms, 'steps_per_s': 1e3 / ms, 'steps_timed': K, 'warmup_steps': WU, 'step_ms': [t * 1e3 for t in times[WU:]],
           'warmup_step_ms': [t * 1e3 for t in times[:WU]], 'mean_step_ms': sum(times[WU:]) / K * 1e3,
           'first_step_ms': times[0] * 1e3, 'peak_mem_GB': peak, 'resident_before_GB': resident_before,
           'peak_mem_step_GB': peak - resident_before, 'params': n_par, 'loss': losses,
           'run_to_run_identical': bool(identical), 'dtype': 'bf16 MFMA operands (autocast policy), fp32 accumulation / master weights / Adam'}
    step_bytes = float(sum(v[1] for v in byte_log.values()))
    top = sorted(byte_log.items(), key=lambda kv: -kv[1][1])[:12]
    out['roofline_step'] = {
        'bound': 'hbm', 'algorithmic_bytes_per_step': step_bytes, 'library_launches_per_step': int(sum(v[0] for v in byte_log.values())),
        'hbm_floor_ms': step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'achieved': step_bytes / (ms * 1e-3) / 1e9, 'frac': step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'how': 'sum over the library launches of one step of the sizes of each launch\'s input and output tensors (each once; scratch and '
               'the few ATen element-wise launches excluded), divided by the median step time and 8 TB/s',
        'largest_entry_points_GB': {k: {'launches': v[0], 'GB': v[1] / 1e9} for k, v in top}}
    # roofline of the dominant training kernel: the bf16 ring convolution on ONE 128^3 x 16 volume (the ConvGRU recurrence
    # launches it ~370 times per step); algorithmic bytes = input + output (+ the addend of the gates' sum form), fp32
    vol = C * S ** 3 * 4
    forms = {}
    # (with the timer on the weight gradients stay on the main stream: launch times without a neighbour on the side stream)
    EXTRA = {0: 'plain', 1: 'reset gate + r h', 2: 'candidate + blend', 3: 'reset-gate backward', 4: 'Block step (bias, LeakyReLU, PixelNorm)',
             5: 'data gradient + producer epilogue backward'}
    EXTRA_BYTES = {0: 0, 1: 64 + 32, 2: 64 + 32 + 64, 3: 64 + 64 + 64, 4: 4, 5: 32 + 4}     # per voxel, beyond x / y / addend
    for n_, e0, e1 in timer:
        if n_ == 'conv3d_c16_ring_multi':
            ng, ex, nb, fl, xb = (int(v) for v in n_.detail.split(':'))
            if ng != 1:
                continue
            per_vox = 16 * xb + 16 * (2 if fl & 2 else 4) + (16 * (2 if fl & 1 else 4) if fl & 8 else 0) + EXTRA_BYTES[ex]
            forms.setdefault((f'ring_multi<1,{"bf16" if xb == 2 else "fp32"} x,{ex},{fl}> {EXTRA[ex]}', nb, per_vox), []).append(e0.elapsed_time(e1))
        else:
            form, nb, io = n_.detail.split(':')
            io = int(io[2:])
            per_vox = 64 * ((0.5 if io & 1 else 1.0) + (0.5 if io & 2 else 1.0) + ((0.5 if io & 4 else 1.0) if form == 'add' else 0.0))
            forms.setdefault((f'conv3d_c16_f16x3<{"true" if form == "add" else "false"},1,{io}> {form}', int(nb), per_vox), []).append(e0.elapsed_time(e1))
    table = {}
    for (form, nb, per_vox), d in sorted(forms.items()):
        alg = nb * (S ** 3) * per_vox                               # input + output (+ addend / epilogue operands) records, each once
        m_ = sum(d) / len(d)
        table[f'{form}:N={nb}'] = {'launches': len(d), 'avg_launch_ms': m_, 'total_ms': sum(d), 'algorithmic_bytes_per_launch': alg,
                                   'hbm_frac': alg / (m_ * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out['ring_conv_launches'] = table
    dom = max(table.items(), key=lambda kv: kv[1]['total_ms']) if table else None
    if dom is not None:
        key, t_ = dom
        tr, trs = None, None
        import glob
        import hashlib
        pmc_key = ('ring_multi_block' if ',4,6>' in key else 'ring_multi_rounded' if ',0,6>' in key else 'ring_multi_addend_inplace' if ',0,11>' in key
                   else 'conv3d_c16_ring_bf16' if 'false,1,3>' in key else 'conv3d_c16_ring_bf16_addend' if 'true,1,7>' in key else None)
        for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_train_hbm_bytes.json')), reverse=True):
            try:
                tj = json.load(open(tpath))
                fresh_ = all(hashlib.sha256(open(os.path.join(ROOT, 'latentfusion_amd', 'csrc', f), 'rb').read()).hexdigest() == h
                             for f, h in tj['source_sha256'].items())
                if fresh_ and pmc_key in tj['kernels']:
                    # the PMC probe launches the kernel on 8 volumes: per-volume bytes x the volumes of this launch
                    tr = tj['kernels'][pmc_key]['bytes_per_launch'] / 8.0 * int(key.split('N=')[1])
                    trs = os.path.basename(tpath)
                    break
            except Exception:                                       # noqa: BLE001
                continue
        tr_live = None
        if pmc_key == 'ring_multi_block' and not getattr(a, 'no_live_pmc', False) and int(os.environ.get('WORLD_SIZE', '1')) == 1:
            try:                                                      # measured now, on this box (tools/pmc_live.py); the file above is the fallback
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import pmc_live
                torch.cuda.synchronize()
                tr_live = pmc_live.ring_block_traffic_live()
            except Exception:                                       # noqa: BLE001
                tr_live = None
            if tr_live is not None:
                tr = tr_live['bytes_per_launch_of_8_volumes'] / 8.0 * int(key.split('N=')[1])
                trs = 'live: rocprofv3 --pmc passes inside this run (tools/pmc_live.py), 8-volume launch scaled to this launch'
        out['roofline'] = {'bound': 'hbm', 'kernel': f'bf16 ring convolution 16 -> 16, dominant form of the step: {key}',
                           'achieved': t_['algorithmic_bytes_per_launch'] / (t_['avg_launch_ms'] * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                           'unit': 'GB/s', 'frac': t_['hbm_frac'], 'traffic': tr, 'traffic_source': trs, 'traffic_live': tr_live,
                           'avg_launch_ms': t_['avg_launch_ms'], 'launches_timed': t_['launches'],
                           'algorithmic_bytes_per_launch': t_['algorithmic_bytes_per_launch'],
                           'share_of_step': t_['total_ms'] / ms}
    return out


def pipelined_build_report(a, S, C, V, dev, world, barrier):
    """The released recipe's fuser is a ConvGRU over the views: an order-dependent recurrence, so a view-sharded build hands
    the hidden state from rank to rank (parallel._fuse_recurrent_pipelined: one send/recv of the C*S^3 state per hop, one
    broadcast of the result).  Every rank builds the SAME model and observation; reported: the time of the sharded build,
    the difference to the local build (the arithmetic is the same in the same order: 0.0 expected) and the time of ONE hop
    (point-to-point transfer of one latent volume, rank 0 -> rank 1) measured alone.  Returns (report, the fused volume,
    the model) -- the volume is the ONE object the hypothesis-sharded loop below runs on."""
    import torch.distributed as dist
    from latentfusion_amd import parallel, synth
    model0, _ = synth.build_model(S, C, 'gru', seed=777, device=dev)
    model0.freeze()
    obs0 = synth.make_observation(V, seed=778, device=dev)
    for rep in range(2):                                       # second pass = warm
        barrier()
        t0 = time.perf_counter()
        z_sh = parallel.build_latent_object_sharded(model0, obs0)
        barrier()
        t_sh = time.perf_counter() - t0
    barrier()
    t0 = time.perf_counter()
    z_full = model0.build_latent_object(obs0)
    barrier()
    t_local = time.perf_counter() - t0
    err = (z_sh - z_full).abs().max().item()
    hop_ms = None
    if world > 1:
        buf = torch.empty_like(z_full)
        for rep in range(3):
            barrier()
            t0 = time.perf_counter()
            if dist.get_rank() == 0:
                parallel._send(buf, 1, None)
            elif dist.get_rank() == 1:
                parallel._recv(buf, 0, None)
            barrier()
            hop_ms = (time.perf_counter() - t0) * 1e3
    rep = {'fuser': 'gru', 'views': V, 'ranks': world, 't_s': t_sh, 't_local_build_s': t_local,
           'max_abs_diff_vs_local_build': err, 'state_MB': z_full.numel() * 4 / 1e6, 'hops': max(0, min(world, V) - 1),
           'hop_ms': hop_ms, 'hop_GBps': (z_full.numel() * 4 / 1e9) / (hop_ms * 1e-3) if hop_ms else None,
           'what': 'ConvGRU recurrence over the views continued rank to rank: one send/recv of the hidden state per hop + one '
                   'broadcast of the result (reference recon/fusion.py:180-201 is a single-process loop)'}
    return rep, z_full, model0


def hypothesis_sharding_report(a, S, C, N, dev, world, rank, barrier, z_one, model_one, cfg):
    """STRONG scaling of ONE object ("shard N samples, 8 GPUs x 1 sample for adam_quick"): every rank holds the
    same latent volume, renders / scores / differentiates its contiguous slice of the N pose hypotheses and the per-hypothesis
    rows (loss terms + camera parameters) meet in ONE all-gather per iteration (parallel.gather_rows); every rank ranks the
    full set.  Replaces the reference's single-process loop over N optimisers (pose/estimation.py:580-594, 602-617).
    Reported: iterations/s of the sharded loop (max over ranks, barriers on both sides), the same loop on ONE rank's GPU
    with all N hypotheses for comparison (rank 0's headline number is per OBJECT, this one is for ONE object), and the time
    of the all-gather alone."""
    import torch.distributed as dist
    from latentfusion_amd import parallel
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation, utils as pu
    from latentfusion_amd import synth
    if N < world:
        return {'skipped': f'{N} hypotheses cannot be split over {world} ranks'}
    tdata = synth.make_observation_data(1, seed=779)
    target = Observation(tdata['color'], tdata['depth'], tdata['mask'], Camera(tdata['intrinsic'], tdata['extrinsic'])).to(dev)
    torch.manual_seed(780)                                      # same draws on every rank (and broadcast from rank 0 anyway)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    cams = init.zoom(None, model_one.input_size, model_one.camera_dist).to(dev)
    est = estimation.load_from_config(cfg, model_one, converge_patience=10 ** 6, conv_mode=a.conv_mode, shard_hypotheses=True)
    cams = est._sync_cameras_from_rank0(cams)
    st = est.start(z_one, target, cams)
    for _ in range(a.warmup):
        est.iterate(st)
    blocks = []
    for _rep in range(max(1, min(a.repeats, 3))):
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            est.iterate(st)
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        blocks.append(tt.item())
    el = sorted(blocks)[len(blocks) // 2]
    # ranking agreement: every rank holds the same top entry (rows were all-gathered)
    top = torch.tensor([st['ranking'][0][1]], device=dev, dtype=torch.float64)
    lo, hi = top.clone(), top.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    # the collective alone: (N_local,15) rows -> (N,15) on every rank
    b, e = parallel.shard_range(N, rank, world)
    rows = torch.zeros(e - b, 15, device=dev)
    ag = []
    for rep in range(20):
        barrier()
        t0 = time.perf_counter()
        parallel.gather_rows(rows, N)
        torch.cuda.synchronize()
        ag.append((time.perf_counter() - t0) * 1e3)
    return {'value': a.steps / el, 'unit': 'iters/s of ONE object', 'ms_per_step': el / a.steps * 1e3,
            'ms_per_step_blocks': [x / a.steps * 1e3 for x in blocks], 'hypotheses': N, 'ranks': world,
            'hypotheses_per_rank': [parallel.shard_range(N, r, world)[1] - parallel.shard_range(N, r, world)[0] for r in range(world)],
            'allgather_rows_ms': sorted(ag)[len(ag) // 2], 'allgather_payload_bytes': N * 15 * 4,
            'best_loss_identical_on_all_ranks': bool(lo.item() == hi.item()), 'scaling': 'strong'}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: bench.py IS the launcher
        sys.exit(launch_ranks(a, argv))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to print a '
                         f'line whose n_gpus is not what was asked for')
    if a.launcher_selftest:
        return launcher_selftest(a, rank, world)
    # Test hook for boxes with ONE GPU (the multi-rank code path cannot be exercised there with RCCL, which refuses two
    # ranks on one device): LF_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and uses gloo for the collectives.  The
    # numbers of such a run mean nothing (the ranks share the GPU); it only proves that the N > 1 path runs and agrees.
    single = os.environ.get('LF_BENCH_SINGLE_DEVICE') == '1'
    if single:
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f'bench.py: --gpus {a.gpus} needs {world} visible devices, found {torch.cuda.device_count()} '
                         '(LF_BENCH_SINGLE_DEVICE=1 runs the N-rank code path on one device over gloo, for testing only)')
    torch.cuda.set_device(local)
    dev = f'cuda:{local}'
    if world > 1:                                              # the ranks share the host: no 8 x all-cores thread pools
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    import torch.distributed as dist
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ          # torchrun / torch.distributed.run
    if world > 1 or launched:
        import datetime
        # "nccl" IS RCCL on ROCm; a collective that does not complete within 5 minutes aborts the rank (and, through
        # launch_ranks, the run) instead of hanging the node
        dist.init_process_group('gloo' if single else 'nccl', timeout=datetime.timedelta(seconds=300))

    from latentfusion_amd import ops, synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.pose import estimation, utils as pu

    S, C, V, N = a.size, a.channels, a.views, a.samples
    if C != 16 and a.conv_mode != 'fp32':
        a.conv_mode = 'fp32'                       # the Winograd / split kernels are written for the 16-channel blocks
    # every rank owns its own object: different weights seed -> different volume, same shapes
    model, cks = synth.build_model(S, C, a.fuser, seed=rank, device=dev)
    rdata = synth.make_observation_data(V, seed=100 + rank)
    from latentfusion_amd.observation import Observation as _Obs
    ref_obs = _Obs(rdata['color'], rdata['depth'], rdata['mask'],
                   Camera(rdata['intrinsic'], rdata['extrinsic'], width=rdata['width'], height=rdata['height'])).to(dev)
    model.freeze()            # inference process: no weight-gradient kernels anywhere (the estimators would freeze per call)
    tdata = synth.make_observation_data(1, seed=200 + rank)
    from latentfusion_amd.observation import Observation
    target = Observation(tdata['color'], tdata['depth'], tdata['mask'],
                         Camera(tdata['intrinsic'], tdata['extrinsic'])).to(dev)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z_obj = model.build_latent_object(ref_obs)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    # second build of the same object: what a warm process (allocator pools, weight packs, code objects
    # loaded) pays per object; the first includes ~0.4 s of one-time hipMalloc / module-load cost
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z_obj = model.build_latent_object(ref_obs)
    torch.cuda.synchronize()
    t_build_warm = time.perf_counter() - t0
    del ref_obs
    torch.cuda.empty_cache()

    cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
    cfg['args']['num_samples'] = N
    cfg['args']['ranking_size'] = N
    fuse_sel = {'default': None, 'none': False, 'fwd': ('fwd',)}[a.fuse_projection]
    est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, conv_mode=a.conv_mode, fuse_projection=fuse_sel)
    torch.manual_seed(300 + rank)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    init_rec = {'K': init.intrinsic.clone(), 'log_q': init.log_quaternion.clone(), 't': init.translation.clone()}
    st = est.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))
    # losses and camera gradients of iteration 0 (before any optimiser step), for the parity line below
    with torch.no_grad():
        l0, g0 = st['engine'].forward_backward(st['cam'], need_grad=True)
    hip0 = {'rank_loss': l0[:, 4].cpu(), 'grad': g0.cpu()}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_loop(est_, st_, events=True):
        """W warm-up iterations, then `--repeats` blocks of EXACTLY K iterations, each bracketed by barrier + synchronize
        and reduced with MAX over the ranks.  Returns (median block time, all block times, the kernel events of all blocks)."""
        for _ in range(a.warmup):
            est_.iterate(st_)
        blocks, timer_ = [], []
        # HIP events inside the timed region only around the DOMINANT kernel, the one `roofline` reports (an event pair costs
        # ~4.6 us of dispatch gap: seven pairs per iteration were 0.6 % of it); the other heavy kernels of `roofline_iter`
        # are timed in one extra, untimed block of the same K iterations behind the timed blocks
        dominant = {'conv3d_c16_wino', 'conv3d_c16_split', 'conv3d_c16_wino_split', f'conv3x3_3d_{C}x{C}'}
        ops.KERNEL_TIMER_TAGS = dominant
        for _rep in range(max(1, a.repeats)):
            barrier()
            ops.KERNEL_TIMER = [] if events else None
            t0_ = time.perf_counter()
            for _ in range(a.steps):
                est_.iterate(st_)
            barrier()
            el = time.perf_counter() - t0_
            timer_ += ops.KERNEL_TIMER or []
            ops.KERNEL_TIMER = None
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            blocks.append(el)
        if not events:
            return sorted(blocks)[len(blocks) // 2], blocks, timer_
        ops.KERNEL_TIMER_TAGS = {'conv3d_c16_wino_projfwd', 'conv3d_c16_wino_projbwd', 'resample_fwd', 'resample_bwd_coef',
                                 'factor_project_fwd', 'factor_project_bwd'}
        ops.KERNEL_TIMER = []
        for _ in range(a.steps):
            est_.iterate(st_)
        torch.cuda.synchronize()
        timer_ += ops.KERNEL_TIMER
        ops.KERNEL_TIMER = None
        return sorted(blocks)[len(blocks) // 2], blocks, timer_

    def est_for_parity(z_):
        e_ = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, conv_mode=a.conv_mode)
        return e_.start(z_, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))

    elapsed, blocks, timer = timed_loop(est, st)
    alt = None
    if a.conv_mode in ('fp32', 'winograd') and not a.no_alt and C == 16:
        # secondary line (never `value`): the same loop with the split-precision conv3d kernels
        st = est = None
        torch.cuda.empty_cache()
        est2 = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, conv_mode='f16x3')
        st2 = est2.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))
        with torch.no_grad():
            l2, g2 = st2['engine'].forward_backward(st2['cam'], need_grad=True)
        alt0 = {'rank_loss': l2[:, 4].cpu(), 'grad': g2.cpu()}
        el2, blocks2, tm2 = timed_loop(est2, st2)
        d2 = [e0.elapsed_time(e1) for n_, e0, e1 in tm2 if n_ == 'conv3d_c16_split']
        alt = {'conv_mode': 'f16x3 (direct convolution, every fp32 product as 3 f16 MFMAs on hi/lo splits, fp32 accumulate; '
                            'error vs fp64 within the fp32 kernels\', '
                            'tests/test_engine_gpu.py::test_split_precision_conv_matches_fp32_and_fp64)',
               'value': world * a.steps / el2, 'unit': 'iters/s', 'ms_per_step': el2 / a.steps * 1e3,
               'ms_per_step_blocks': [b / a.steps * 1e3 for b in blocks2],
               'conv_avg_launch_ms': sum(d2) / max(len(d2), 1)}
        if d2:
            ms2 = alt['conv_avg_launch_ms']
            ab2 = (2 * C + 1) * N * S ** 3 * 4                       # read x, write y and one norm float per voxel
            fl2 = 3 * 2.0 * 27 * C * C * N * S ** 3                  # three f16 products per fp32 product
            alt['roofline'] = {'bound': 'hbm', 'kernel': 'conv3d_c16_f16x3_kernel', 'achieved': ab2 / (ms2 * 1e-3) / 1e9,
                               'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ab2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               'f16_mfma_frac': fl2 / (ms2 * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 'launches_timed': len(d2)}

    graph = None     # (round 4 measured hipGraph replay of the evaluation: 186.8 vs 187.0 it/s eager -- profiles/r04_hipgraph_ab.json;
                     #  the variant lives in latentfusion_amd/experimental.py, its A/B in tools/engine_ab.py)

    # ---- roofline of the dominant kernel (fused conv3d 16->16 block step; 2 forward + 2 data-gradient launches per
    # iteration), from HIP events recorded on the launch stream inside the timed region -------------------------------
    name = {'fp32': f'conv3x3_3d_{C}x{C}', 'winograd': 'conv3d_c16_wino', 'f16x3': 'conv3d_c16_split'}[a.conv_mode]
    kname = {'fp32': 'conv3d_c16_persistent_kernel', 'winograd': 'conv3d_c16_wino_kernel', 'f16x3': 'conv3d_c16_f16x3_kernel'}[a.conv_mode]

    def avg_ms(tag):
        d = [e0.elapsed_time(e1) for n_, e0, e1 in timer if n_ == tag]
        return (sum(d) / len(d) if d else 0.0), len(d)
    conv_ms, conv_launches = avg_ms(name)
    nvox = N * S ** 3
    alg_flops = 2.0 * 27 * C * C * nvox                            # direct-convolution count
    alg_bytes = (2 * C + 1) * nvox * 4                             # read x, write y and one norm float per voxel
    wino = a.conv_mode == 'winograd'
    exec_flops = alg_flops * (64.0 / 216.0 if wino else 1.0) * (1.0 if a.conv_mode in ('fp32', 'winograd') else 3.0)
    # price against the pipe the products run on: fp32 MFMA for the all-fp32 kernels, dense f16 MFMA for the split ones
    peak = FP32_MFMA_PEAK_TFLOPS if a.conv_mode in ('fp32', 'winograd') else F16_MFMA_PEAK_TFLOPS
    sec = conv_ms * 1e-3
    exec_tflops = exec_flops / sec / 1e12 if sec > 0 else 0.0
    hbm_gbs = alg_bytes / sec / 1e9 if sec > 0 else 0.0
    floor_ms = max(exec_flops / (peak * 1e12), alg_bytes / (HBM_PEAK_GBS * 1e9)) * 1e3
    # measured HBM bytes per launch of this kernel (PMC passes, tools/pmc_collect.sh + tools/pmc_summary.py); refused when
    # the kernel sources changed since the counters were collected
    import glob
    import hashlib

    def pmc_traffic(key):
        for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_bytes.json')), reverse=True):
            try:
                tj = json.load(open(tpath))
                stamp = tj.get('source_sha256', {})
                fresh = bool(stamp) and all(
                    hashlib.sha256(open(os.path.join(ROOT, 'latentfusion_amd', 'csrc', f), 'rb').read()).hexdigest() == h
                    for f, h in stamp.items())
                if fresh and S == 128 and C == 16 and N == 8 and key in tj.get('kernels', {}):
                    return tj['kernels'][key]['bytes_per_launch'], os.path.basename(tpath)
            except Exception:                                       # noqa: BLE001
                continue
        return None, None
    traffic, traffic_src = pmc_traffic(kname)
    traffic_live = None
    if world == 1 and not a.no_live_pmc and wino and S == 128 and C == 16 and N == 8:
        # measured now, on this box: two counter passes over the same kernel at the same shape in a child process (the bench is idle
        # meanwhile); the committed file is the fallback
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import pmc_live
            torch.cuda.synchronize()
            traffic_live = pmc_live.wino_traffic_live()
        except Exception:                                           # noqa: BLE001
            traffic_live = None
        if traffic_live is not None:
            traffic, traffic_src = traffic_live['forward_form_bytes_per_launch'], 'live: rocprofv3 --pmc passes inside this run (tools/pmc_live.py)'
    if alt is not None and 'roofline' in alt:
        alt['roofline']['traffic'], alt['roofline']['traffic_source'] = pmc_traffic('conv3d_c16_f16x3_kernel')
    roofline = {
        'bound': 'mfma', 'kernel': kname + ' (fused conv3d 16->16 + He + bias + LeakyReLU + PixelNorm; forward and data-gradient forms)',
        'achieved': exec_tflops, 'peak': peak, 'unit': 'TFLOP/s', 'frac': exec_tflops / peak,
        'traffic': traffic, 'traffic_source': traffic_src, 'traffic_live': traffic_live,
        'avg_launch_ms': conv_ms, 'launches_timed': conv_launches, 'floor_ms': floor_ms, 'floor_over_measured': floor_ms / conv_ms if conv_ms else 0.0,
        'executed_flops_per_launch': exec_flops, 'algorithmic_flops_per_launch': alg_flops,
        'algorithmic_bytes_per_launch': alg_bytes, 'hbm_achieved_GBps': hbm_gbs, 'hbm_frac': hbm_gbs / HBM_PEAK_GBS,
        'note': ('achieved = flops EXECUTED on the matrix pipe / launch time (Winograd F(2x2x2,3x3x3) executes 64/216 of the '
                 'direct-convolution multiplies: algorithmic rate = %.1f TFLOP/s, a 3.375x algorithmic speed-up that is not '
                 'counted in frac); hbm_frac = algorithmic bytes / time / 8 TB/s; floor_ms = max(MFMA, HBM) time of this launch'
                 % (alg_flops / sec / 1e12 if sec > 0 else 0.0))}
    # the other heavy kernels of the iteration (HBM-bound): algorithmic bytes / measured time vs the 8 TB/s peak
    others = {}
    vol_b = C * nvox * 4
    for tag, nbytes in (('resample_fwd', vol_b + C * S ** 3 * 4), ('resample_bwd_coef', vol_b + C * S ** 3 * 4),
                        ('factor_project_fwd', vol_b), ('factor_project_bwd', 2 * vol_b)):
        ms_, cnt_ = avg_ms(tag)
        if cnt_:
            others[tag] = {'avg_launch_ms': ms_, 'launches_timed': cnt_, 'algorithmic_bytes': nbytes,
                           'hbm_frac': nbytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # round 4: the last camera block's convolution with the factor projection fused in (one launch instead of two): its own
    # executed flops (Winograd products + the projection's 2 * 16 * 16 per voxel) and algorithmic bytes
    for tag, extra_fl, nbytes, what in (
            ('conv3d_c16_wino_projfwd', 2.0 * 16 * 16 * nvox, alg_bytes + N * 17 * S * S * 4,
             'conv3d_c16_wino_projfwd_kernel: camera-block conv 2 + factor projection forward'),
            ('conv3d_c16_wino_projbwd', 2.0 * 16 * 16 * nvox * (10 * 18) / (8 * 16), alg_bytes + (C + 1) * nvox * 4 + N * 16 * S * S * 4,
             'conv3d_c16_wino_projbwd_kernel: factor projection backward + first data-gradient conv')):
        ms_, cnt_ = avg_ms(tag)
        if cnt_ and wino:
            fl_ = alg_flops * 64.0 / 216.0 + extra_fl
            tr_, trs_ = pmc_traffic(tag + '_kernel')
            others[tag] = {'kernel': what, 'avg_launch_ms': ms_, 'launches_timed': cnt_, 'executed_flops': fl_,
                           'mfma_frac': fl_ / (ms_ * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 'algorithmic_bytes': nbytes,
                           'hbm_frac': nbytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': tr_, 'traffic_source': trs_}
    # whole iteration vs its HBM floor (~2.2 GB per pose sample forward + backward)
    iter_bytes = 2.2e9 * N * (S / 128.0) ** 3 * (C / 16.0)
    iter_floor_ms = iter_bytes / (HBM_PEAK_GBS * 1e9) * 1e3

    sharded = None
    if a.sharded_build or (world > 1 and not a.no_sharded_build):
        # auxiliary measurement: whatever happens here must not lose the headline line of an N-GPU run
        try:
            sharded = sharded_build_report(a, S, C, V, dev, world, barrier)
        except Exception as e:                                       # noqa: BLE001
            sharded = {'error': f'{type(e).__name__}: {e}'[:300]}

    # N > 1: the other two sharding axes of SURVEY 8(e), on ONE object (auxiliary: never allowed to lose the headline line)
    pipelined, hyp = None, None
    if world > 1 and not (a.no_pipelined_build and a.no_hypothesis_sharding):
        try:
            st = est = st2 = est2 = None                             # release the per-rank loop states
            torch.cuda.empty_cache()
            pipelined, z_one, model_one = pipelined_build_report(a, S, C, V, dev, world, barrier)
            if a.no_pipelined_build:
                pipelined = None
            if not a.no_hypothesis_sharding:
                try:
                    hyp = hypothesis_sharding_report(a, S, C, N, dev, world, rank, barrier, z_one, model_one, cfg)
                except Exception as e:                               # noqa: BLE001
                    hyp = {'error': f'{type(e).__name__}: {e}'[:300]}
        except Exception as e:                                       # noqa: BLE001
            pipelined = {'error': f'{type(e).__name__}: {e}'[:300]}

    cfg3 = None
    if world == 1 and not a.no_cfg3:
        try:
            st = est = st2 = est2 = None
            torch.cuda.empty_cache()
            cfg3 = cfg3_report(a, dev)
        except Exception as e:                                       # noqa: BLE001  (auxiliary: never loses the headline)
            cfg3 = {'error': f'{type(e).__name__}: {e}'[:300]}

    # the same workload against what the REFERENCE ITSELF produced for it (tests/golden/g26_headline_trace.pt, generated in the
    # build container by oracle/make_golden_headline.py from the imported reference): volume, iteration-0 renders and the whole
    # 100-iteration adam_quick trace (GPU only, ~1 s; the fixture is data, the reference never runs here)
    ref_trace = None
    if world == 1 and (S, C, V, N, a.fuser) == (128, 16, 16, 8, 'gru'):
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location('headline_trace_probe', os.path.join(ROOT, 'tools', 'headline_trace_probe.py'))
            probe = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(probe)
            r_ = probe.compare(dev)
            ref_trace = {'fixture': r_['fixture'], 'volume': r_['volume'], 'iteration0_renders': r_['iteration0_renders'],
                         'trace': {k_: v_ for k_, v_ in r_['trace'].items() if k_ != 'per_iteration'},
                         'what': 'HIP end to end vs the fixture the imported reference wrote for this exact workload (g26): first '
                                 'iteration whose argmin / ranking differs, loss deviation over all iterations, final top-1'}
        except Exception as e:                                       # noqa: BLE001  (auxiliary: never loses the headline)
            ref_trace = {'error': f'{type(e).__name__}: {e}'[:300]}

    cfg5 = None
    if world == 1 and not a.no_cfg5 and C == 16:
        try:
            st = est = st2 = est2 = None
            torch.cuda.empty_cache()
            cfg5 = cfg5_report(a, dev)
        except Exception as e:                                       # noqa: BLE001  (auxiliary: never loses the headline)
            cfg5 = {'error': f'{type(e).__name__}: {e}'[:300]}

    variants = None
    if world == 1 and not a.no_variants:
        try:
            variants = variants_report(a, dev)
        except Exception as e:                                       # noqa: BLE001  (auxiliary: never loses the headline)
            variants = {'error': f'{type(e).__name__}: {e}'[:300]}

    # RCCL on this box: under a launcher the process group above IS an RCCL communicator; a plain `python bench.py` run
    # initialises a world-size-1 group here (after the timed regions, so it cannot touch the number) and runs one
    # all-reduce of a latent-volume-sized tensor through it
    rccl = {'backend': None, 'ranks_seen': dist.get_world_size() if dist.is_initialized() else 1}
    try:
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
            dist.init_process_group('nccl', rank=0, world_size=1)
        rccl['backend'] = dist.get_backend()
        rccl['ranks_seen'] = dist.get_world_size()
        if rccl['backend'] == 'nccl':
            rccl['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        probe = torch.ones(C * S ** 3, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        rccl['allreduce_ok'] = bool(probe[0].item() == dist.get_world_size() and probe[-1].item() == dist.get_world_size())
        del probe
    except Exception as e:                                           # noqa: BLE001  (reported, never fatal for the bench line)
        rccl['error'] = f'{type(e).__name__}: {e}'[:300]
    if dist.is_initialized():
        barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    value = world * a.steps / elapsed
    out = {
        'metric': f'pose-optim iters/sec (reconstruct+render+backward), {V} views, {S}^3 voxels',
        'value': value, 'unit': 'iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'timing': {'what': f'{len(blocks)} back-to-back blocks of exactly {a.steps} iterations, each between barrier + '
                           'synchronize, max over ranks; value = steps / MEDIAN block time',
                   'ms_per_step_blocks': [b / a.steps * 1e3 for b in blocks],
                   'gpu_timed_total_s': sum(blocks),
                   'spread_pct': (max(blocks) - min(blocks)) / elapsed * 100.0},
        'rccl': rccl,
        'dtype': 'f32' if a.conv_mode in ('fp32', 'winograd') else 'f32 (conv3d products split into 3 f16 MFMAs, fp32 accumulate)', 'data': 'synthetic (SYN(S,C) random-init weights, synthetic observations)',
        'config': {'workload': f'SYN({S},{C}) latent volume, {V} reference views, adam_quick pose loop, '
                               f'{N} pose samples per iteration, one object per GPU',
                   'fuser': a.fuser, 'pose_samples': N, 'ref_views': V, 'volume': S, 'channels': C,
                   'parallelism': f'objects x{world} (no data-path collective in the loop)'},
        't_build_s': t_build, 't_build_warm_s': t_build_warm,
        'e2e_100_iters_per_s': 100.0 / (t_build_warm + 100.0 * elapsed / a.steps),
        'roofline': roofline,
        'roofline_iter': {'hbm_floor_ms': iter_floor_ms, 'algorithmic_bytes_per_iteration': iter_bytes,
                          'frac': iter_floor_ms / (elapsed / a.steps * 1e3), 'kernels': others},
    }
    if world == 1 and not a.no_cpu_baseline:
        v, cores, ref0, cpu_dt, build_info, z_ora = cpu_baseline(cks[:3] + (cks[3],), z_obj.cpu(), rdata, tdata, init_rec, cfg,
                                                                  a.cpu_iters, build=not a.no_build_parity)
        # the HIP render + loss + camera gradients on the ORACLE-built volume (iteration 0): render parity that does not
        # rest on the GPU reconstruction
        hip_on_oracle = None
        if build_info is not None:
            st3 = est_for_parity(z_ora.to(dev))
            with torch.no_grad():
                l3, g3 = st3['engine'].forward_backward(st3['cam'], need_grad=True)
            hip_on_oracle = {'rank_loss': l3[:, 4].cpu(), 'grad': g3.cpu()}

        def hip_trace(z_):
            """`cpu_iters` iterations of the HIP loop from the same start, per-iteration rank losses and parameters."""
            e_ = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, conv_mode=a.conv_mode, return_camera_history=True)
            s_ = e_.start(z_, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))
            s_['max_steps'] = a.cpu_iters
            for _ in range(a.cpu_iters):
                e_.iterate(s_)
            rl = torch.stack([h_[0] for h_ in s_['camera_history']]).cpu()
            lq = torch.stack([h_[1].log_quaternion for h_ in s_['camera_history']]).cpu()
            return rl, lq

        def trace_parity(z_):
            rl, lq = hip_trace(z_)
            ref = ref0['trace_rank_loss']
            k = min(len(rl), len(ref))
            rows = []
            for i in range(k):
                rows.append({'iteration': i,
                             'rank_loss_max_rel_diff': ((rl[i] - ref[i]).abs() / ref[i].abs().clamp_min(1e-30)).max().item(),
                             'argmin_equal': bool(torch.argmin(rl[i]) == torch.argmin(ref[i])),
                             'ranking_equal': bool(torch.equal(torch.argsort(rl[i]), torch.argsort(ref[i])))})
            return {'iterations': k, 'per_iteration': rows,
                    'argmin_equal_all': all(r['argmin_equal'] for r in rows),
                    'ranking_equal_all': all(r['ranking_equal'] for r in rows),
                    'what': 'the HIP adam_quick loop vs the oracle loop from the same initial cameras, iteration by iteration '
                            '(rank loss of every hypothesis, index of the best, order of all N)'}

        def parity(h):
            gerr = (h['grad'] - ref0['grad']).norm(dim=1) / ref0['grad'].norm(dim=1).clamp_min(1e-30)
            return {'rank_loss_max_abs_diff': (h['rank_loss'] - ref0['rank_loss']).abs().max().item(),
                    'rank_loss_max_rel_diff': ((h['rank_loss'] - ref0['rank_loss']).abs()
                                               / ref0['rank_loss'].abs().clamp_min(1e-30)).max().item(),
                    'camera_grad_max_rel_l2_err': gerr.max().item(),
                    'argmin_equal': bool(torch.argmin(h['rank_loss']) == torch.argmin(ref0['rank_loss'])),
                    'ranking_equal': bool(torch.equal(torch.argsort(h['rank_loss']), torch.argsort(ref0['rank_loss'])))}
        out['cpu_baseline'] = {'value': v, 'unit': 'iters/s', 'cores': cores, 'host_cores': os.cpu_count(), 'kind': 'port',
                               'timed_s': cpu_dt,
                               'sample': f'{a.cpu_iters} timed iteration(s) of the same SYN({S},{C}) N={N} pose loop (oracle = CPU restatement '
                                         f'of the reference, after 1 warm-up iteration; latent volume '
                                         + ('reconstructed by the oracle itself from the same reference views'
                                            if build_info is not None else 'taken from the GPU build')
                                         + f'; {cores} ATen threads of the {os.cpu_count()} host cores: more threads measured slower)',
                               # END TO END: HIP reconstruction + HIP iteration 0 vs oracle reconstruction + oracle iteration 0
                               # (same reference views, target, initial cameras): per-hypothesis ranking loss and
                               # camera-parameter gradients
                               'parity_at_full_size': parity(hip0)}
        # iterations 0 .. cpu_iters-1: end to end (HIP build + HIP loop vs oracle build + oracle loop) and, when the oracle
        # built its own volume, the HIP loop on THAT volume (render / loss / optimiser parity alone)
        out['cpu_baseline']['trace_parity_at_full_size'] = trace_parity(z_obj)
        if build_info is not None:
            out['cpu_baseline']['build_parity_at_full_size'] = build_info
            out['cpu_baseline']['render_parity_on_oracle_built_volume'] = parity(hip_on_oracle)
            out['cpu_baseline']['trace_parity_on_oracle_built_volume'] = trace_parity(z_ora.to(dev))
            out['cpu_baseline']['e2e_100_iters_per_s'] = 100.0 / (build_info['t_oracle_build_s'] + 100.0 / v)
        if alt is not None:
            alt['parity_at_full_size'] = parity(alt0)
    if alt is not None:
        out['alt'] = alt
    if graph is not None:
        out['graph'] = graph
    if sharded is not None:
        out['sharded_build'] = sharded
    if cfg3 is not None:
        out['cfg3'] = cfg3
    if cfg5 is not None:
        out['cfg5'] = cfg5
    if ref_trace is not None:
        out['reference_trace_parity'] = ref_trace
    if variants is not None:
        out['renderer_variants'] = variants
    if pipelined is not None:
        out['pipelined_gru_build'] = pipelined
    if hyp is not None:
        out['hypothesis_sharded_one_object'] = hyp
    # LAST in the line (the driver keeps the line's tail): every secondary claim of DESIGN.md in one compact object
    def dig(d, *ks):
        for k_ in ks:
            if not isinstance(d, dict) or k_ not in d:
                return None
            d = d[k_]
        return d
    tr = dig(ref_trace, 'trace') or {}
    out['summary'] = {
        'headline_iters_per_s': out.get('value'), 'headline_roofline_frac': dig(out, 'roofline', 'frac'),
        'cpu_baseline_iters_per_s': dig(out, 'cpu_baseline', 'value'),
        'cfg3_iters_per_s': dig(cfg3, 'value'), 'cfg3_roofline_frac': dig(cfg3, 'roofline', 'frac'),
        'cfg3_scored_on_fused_engine': dig(cfg3, 'scored_on_fused_engine'),
        'cfg5_ms_per_step': dig(cfg5, 'ms_per_step'), 'cfg5_peak_mem_GB': dig(cfg5, 'peak_mem_GB'), 'cfg5_peak_mem_step_GB': dig(cfg5, 'peak_mem_step_GB'),
        'cfg5_roofline_step_frac': dig(cfg5, 'roofline_step', 'frac'), 'cfg5_step_GB': (dig(cfg5, 'roofline_step', 'algorithmic_bytes_per_step') or 0) / 1e9 or None,
        'cfg5_run_to_run_identical': dig(cfg5, 'run_to_run_identical'),
        'renderer_variants_iters_per_s': {k: dig(variants, k, 'iters_per_s') for k in ('sum', 'occlusion')} if variants else None,
        'reference_trace': {k: tr.get(k) for k in ('first_iteration_argmin_differs', 'max_rel_diff_all', 'final_top1_equal',
                                                     'argmin_mismatch_at_clear_gap')} if tr else None,
        'reference_self_deviation': {k: dig(tr, 'reference_self_deviation', k) for k in (
            'control_threads', 'first_iteration_references_disagree_on_argmin', 'hip_argmin_mismatch_where_references_agree',
            'max_hip_dev_over_windowed_ref_dev', 'hip_dev_over_ref_dev_at_9')} if dig(tr, 'reference_self_deviation') else None}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
