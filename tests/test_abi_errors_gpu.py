"""Error behaviour of the boundary: the C ABI rejects bad arguments with a negative LF_E* code and launches nothing
(include/lf_hip.h "Contract"), the Python mirror raises what the reference raises (recon/models.py:398-403,
pose/estimation.py:177-178)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LF_EINVAL, LF_EALIGN, LF_ENOSPC = -1, -2, -3


def test_c_abi_rejects_bad_arguments():
    from latentfusion_amd import _lib, ops
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    S = 8
    vol = ops.cl(torch.randn(1, 16, S, S, S, device=DEV))
    out = ops.empty_cl((2, 16, S, S, S), DEV)
    cf = torch.zeros(2, 20, device=DEV)
    before = out.clone()
    # sizes, map kind, batch relation
    assert L.lf_resample3d_fwd(vol.data_ptr(), 1, cf.data_ptr(), 0, out.data_ptr(), 0, S, S, S, 16, s) == LF_EINVAL
    assert L.lf_resample3d_fwd(vol.data_ptr(), 1, cf.data_ptr(), 7, out.data_ptr(), 2, S, S, S, 16, s) == LF_EINVAL
    assert L.lf_resample3d_fwd(vol.data_ptr(), 3, cf.data_ptr(), 0, out.data_ptr(), 2, S, S, S, 16, s) == LF_EINVAL
    # scratch too small
    gc = torch.empty(2, 18, device=DEV)
    need = L.lf_resample3d_bwd_coef_scratch_bytes(2, S, S, S)
    scratch = torch.empty(need // 4 + 1, device=DEV)
    assert L.lf_resample3d_bwd_coef(out.data_ptr(), vol.data_ptr(), 1, cf.data_ptr(), gc.data_ptr(), scratch.data_ptr(), need - 4,
                                    2, S, S, S, 16, s) == LF_ENOSPC
    # misaligned activation pointer for a kernel that needs 16-byte records
    w = torch.randn(16, 16, 3, 3, 3, device=DEV)
    up = ops.pack_conv3d_c16_wino(w)
    x = ops.cl(torch.randn(1, 16, S, S, S, device=DEV))
    y = torch.empty_like(x)
    assert L.lf_conv3d_c16_wino(x.data_ptr() + 4, up.data_ptr(), None, y.data_ptr(), None, 1, S, S, S, 1.0, 0, 0.2, 1e-8, None, None, 0,
                                None, s) == LF_EALIGN
    assert L.lf_conv3d_c16_wino(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), None, 1, S, S, S, 1.0, 0, 1.5, 1e-8, None, None, 0,
                                None, s) == LF_EINVAL          # slope outside (0, 1)
    # reductions: unknown kind, median beyond its view limit
    z = torch.randn(4, 64, device=DEV)
    o = torch.empty(64, device=DEV)
    assert L.lf_fuse_views_fwd(z.data_ptr(), o.data_ptr(), None, 99, 4, 64, 64, s) == LF_EINVAL
    assert L.lf_set_tuning(42, 1) == LF_EINVAL
    torch.cuda.synchronize()
    assert torch.equal(out, before)                                  # nothing was launched on the rejected calls
    # the Python layer turns codes into exceptions
    with pytest.raises(_lib.LFHipError):
        _lib.check(LF_ENOSPC, 'probe')


def test_python_mirror_raises_like_the_reference(golden):
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    d = g['init']
    cam = Camera(d['K'].to(DEV), None, d['z_span'], d['viewport'].to(DEV), width=d['width'], height=d['height'],
                 log_quaternion=d['log_q'].to(DEV), translation=d['t'].to(DEV))
    z2 = g['z_obj'].to(DEV)[0].expand(3, -1, -1, -1, -1)            # 3 volumes for 8 cameras: batch mismatch
    with pytest.raises(ValueError):
        model.photographer(z2, cam)
    tg = g['target']
    tcam = Camera(tg['cam']['K'], None, tg['cam']['z_span'], tg['cam']['viewport'], width=tg['cam']['width'], height=tg['cam']['height'],
                  log_quaternion=tg['cam']['log_q'], translation=tg['cam']['t'])
    two = Observation(None, tg['depth'].expand(2, -1, -1, -1), tg['mask'].float().expand(2, -1, -1, -1),
                      Camera.cat([tcam, tcam]))
    est = estimation.load_from_config({'type': 'gradient', 'args': dict(learning_rate=0.01, num_samples=8, num_iters=1,
                                                                        converge_threshold=1e-6, converge_patience=5, ranking_size=8,
                                                                        optimizer='adam'),
                                       'loss_weights': {'depth': 1.0}}, model)
    with pytest.raises(ValueError):
        est.estimate(g['z_obj'].to(DEV), two, camera=cam)           # one observation at a time (estimation.py:177-178)
    with pytest.raises(ValueError):
        estimation.load_from_config({'type': 'simulated_annealing', 'args': {}, 'loss_weights': {}}, model)
    with pytest.raises(ValueError):
        fusion.get_fuser('attention', 16, 1.0)
