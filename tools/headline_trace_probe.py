#!/usr/bin/env python
"""HIP path versus the REFERENCE's own run of BASELINE cfg 2 (tests/golden/g26_headline_trace.pt, written by
oracle/make_golden_headline.py from the imported reference): 16-view reconstruction at 128^3, the renders of iteration 0, and
the adam_quick loop iteration by iteration over the fixture's length (the preset's 100 iterations).

    python tools/headline_trace_probe.py [out.json [fixture name, default g26_headline_trace]]

Reports, per iteration: max relative difference of the N rank losses, whether the argmin / the full ranking agree, the
reference's top-2 gap; and the summary figures: first iteration whose ranking differs, first whose argmin differs, the
iterations whose argmin differs although the reference separates its best two by more than 1e-3 (relative), final top-1."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(dev='cuda', name='g26_headline_trace'):
    from latentfusion_amd import synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    g = torch.load(os.path.join(ROOT, 'tests', 'golden', name + '.pt'), weights_only=False)
    S, C, V, N, T = g['S'], g['C'], g['V'], g['N'], g['T']
    sd = g['seeds']
    model, cks = synth.build_model(S, C, 'gru', seed=sd['model'], device=dev)
    model.freeze()

    def obs(n, seed):
        d = synth.make_observation_data(n, seed)
        return Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'], width=d['width'], height=d['height'])).to(dev)
    z = model.build_latent_object(obs(V, sd['ref']))
    target = obs(1, sd['target'])
    out = {'fixture': {'S': S, 'C': C, 'V': V, 'N': N, 'T': T, 'reference_s_per_iteration': g['reference_seconds_per_iteration'],
                       'reference_threads': g['reference_threads']}}
    if 'z_obj_sub' in g:                                           # (the further-seed fixtures share g26's object)
        zs = z[..., ::4, ::4, ::4].cpu()
        scale = float(g['z_obj_absmax'])
        out['volume'] = {'max_abs_diff_over_absmax': float((zs - g['z_obj_sub']).abs().max()) / scale,
                         'rel_l2': float((zs - g['z_obj_sub']).norm() / g['z_obj_sub'].norm())}
    c = g['init']
    init = Camera(c['K'].to(dev), None, c['z_span'], c['viewport'].to(dev), width=c['width'], height=c['height'],
                  log_quaternion=c['log_q'].to(dev), translation=c['t'].to(dev))
    zoomed = init.zoom(None, model.input_size, model.camera_dist)
    with torch.no_grad():
        y0, _ = model.render_latent_object(z, zoomed, return_latent=True)
    r0 = {}
    for k in ('depth_logits', 'mask_logits', 'depth', 'mask'):
        a, b = y0[k].squeeze(0).cpu(), g['iter0'][k]
        r0[k] = {'max_abs_diff': float((a - b).abs().max()), 'max_abs_ref': float(b.abs().max()),
                 'rel_l2': float((a - b).norm() / b.norm().clamp_min(1e-30))}
    out['iteration0_renders'] = r0
    cfg = dict(g['cfg'])
    cfg['args'] = dict(cfg['args'])
    est = estimation.load_from_config(cfg, model, track_stats=True)
    _, stats = est.estimate(z, target, camera=init.to('cpu'))
    got, ref = stats['rank_loss'].cpu(), g['rank_loss']
    k = min(len(got), len(ref))
    rows, rank_first, arg_first, clear_mismatch = [], None, None, []
    for i in range(k):
        srt = torch.sort(ref[i]).values
        gap = float((srt[1] - srt[0]) / srt[0].abs().clamp_min(1e-30))
        rel = float(((got[i] - ref[i]).abs() / ref[i].abs().clamp_min(1e-30)).max())
        a_eq = bool(torch.argmin(got[i]) == torch.argmin(ref[i]))
        r_eq = bool(torch.equal(torch.argsort(got[i]), torch.argsort(ref[i])))
        if not r_eq and rank_first is None:
            rank_first = i
        if not a_eq and arg_first is None:
            arg_first = i
        if not a_eq and gap > 1e-3:
            clear_mismatch.append(i)
        rows.append({'iteration': i, 'rank_loss_max_rel_diff': rel, 'argmin_equal': a_eq, 'ranking_equal': r_eq, 'reference_top2_rel_gap': gap,
                     'best_loss_hip': float(got[i].min()), 'best_loss_reference': float(ref[i].min())})
    out['trace'] = {'iterations': k, 'first_iteration_ranking_differs': rank_first, 'first_iteration_argmin_differs': arg_first,
                    'argmin_equal_count': sum(r['argmin_equal'] for r in rows),
                    'iterations_with_clear_gap': sum(r['reference_top2_rel_gap'] > 1e-3 for r in rows),
                    'argmin_mismatch_at_clear_gap': clear_mismatch,
                    'final_top1_equal': rows[-1]['argmin_equal'], 'final_best_loss_hip': rows[-1]['best_loss_hip'],
                    'final_best_loss_reference': rows[-1]['best_loss_reference'],
                    'max_rel_diff_first_5': max(r['rank_loss_max_rel_diff'] for r in rows[:5]),
                    'max_rel_diff_all': max(r['rank_loss_max_rel_diff'] for r in rows), 'per_iteration': rows}
    return out


if __name__ == '__main__':
    res = compare(name=sys.argv[2] if len(sys.argv) > 2 else 'g26_headline_trace')
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt)
    t = res['trace']
    print(json.dumps({k: v for k, v in t.items() if k != 'per_iteration'}))
    print(json.dumps(res.get('volume')), json.dumps(res['iteration0_renders']))
    for r in t['per_iteration'][:12] + t['per_iteration'][-3:]:
        print(r)
