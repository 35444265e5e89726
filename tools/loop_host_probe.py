#!/usr/bin/env python
"""Where the HOST spends one headline pose iteration (GradientPoseEstimator on the fused engine, SYN(128,16), N = 8): wall time of
the phases of `_iterate_engine` (no device synchronisation added), next to the iteration time.  `wait` = the time the host sits in
`event.synchronize()`.  Measured (round 6, one MI355X): 5.32 ms per iteration, of which the host enqueues for 0.33 ms, steps the
optimiser for 0.03 ms, does its ranking / scheduler work for 0.16 ms and WAITS 4.75 ms: the queue always holds the next iteration, the
loop is GPU-bound (the ~0.3 ms of idle time before every adam_step in a rocprofv3 trace -- `profiles/r06_kernel_stats.txt`, launch-order
section -- is the profiler's own handling of the copy commands: the un-profiled iteration is shorter than the traced GPU-busy time;
replacing the copy commands by kernels that read / write pinned memory directly changed nothing, at N = 8 or at N = 1).
    python tools/loop_host_probe.py [iters=60]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
from latentfusion_amd.pose import estimation, utils as pu  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S, C, V, N, dev = 128, 16, 16, 8, 'cuda:0'
model, _ = synth.build_model(S, C, 'gru', seed=0, device=dev)
rd = synth.make_observation_data(V, seed=100)
ref = Observation(rd['color'], rd['depth'], rd['mask'], Camera(rd['intrinsic'], rd['extrinsic'], width=rd['width'], height=rd['height'])).to(dev)
model.freeze()
td = synth.make_observation_data(1, seed=200)
target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)
z_obj = model.build_latent_object(ref)
cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6)
torch.manual_seed(300)
init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
st = est.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))

T = {'launch': 0.0, 'opt': 0.0, 'wait': 0.0}
_launch, _opt_step = est._engine_launch, st['opt'].step


def launch(*a, **k):
    t = time.perf_counter()
    r = _launch(*a, **k)
    T['launch'] += time.perf_counter() - t
    ev = r['event']

    class Ev:                                                      # times the host's wait for this iteration's read-back
        def synchronize(self):
            t_ = time.perf_counter()
            ev.synchronize()
            T['wait'] += time.perf_counter() - t_
    r['event'] = Ev()
    return r


def opt_step(*a, **k):
    t = time.perf_counter()
    r = _opt_step(*a, **k)
    T['opt'] += time.perf_counter() - t
    return r


est._engine_launch = launch
st['opt'].step = opt_step
for _ in range(10):
    est.iterate(st)
torch.cuda.synchronize()
for k in T:
    T[k] = 0.0
t0 = time.perf_counter()
for _ in range(iters):
    est.iterate(st)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
rest = host - sum(T.values())
print(f'{iters} iterations: {1e3 * total / iters:.3f} ms per iteration ({iters / total:.1f} it/s); host per iteration: '
      f'enqueue forward/backward + read-back {1e3 * T["launch"] / iters:.3f} ms, optimiser step {1e3 * T["opt"] / iters:.3f} ms, '
      f'waiting for the read-back {1e3 * T["wait"] / iters:.3f} ms, ranking / scheduler / bookkeeping {1e3 * rest / iters:.3f} ms')
