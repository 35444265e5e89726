#!/usr/bin/env python
"""HIP path versus the REFERENCE's own run of BASELINE cfg 2 (tests/golden/g26_headline_trace.pt, written by
oracle/make_golden_headline.py from the imported reference): 16-view reconstruction at 128^3, the renders of iteration 0, and
the adam_quick loop iteration by iteration over the fixture's length (the preset's 100 iterations).

    python tools/headline_trace_probe.py [out.json [fixture name, default g26_headline_trace]] [--conv-mode fp32|f16x3|winograd]

Reports, per iteration: max relative difference of the N rank losses, whether the argmin / the full ranking agree, the
reference's top-2 gap; and the summary figures: first iteration whose ranking differs, first whose argmin differs, the
iterations whose argmin differs although the reference separates its best two by more than 1e-3 (relative), final top-1."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(dev='cuda', name='g26_headline_trace', conv_mode=None, control='g26n_headline_trace_threads3'):
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
    est = estimation.load_from_config(cfg, model, track_stats=True, **({'conv_mode': conv_mode} if conv_mode else {}))
    _, stats = est.estimate(z, target, camera=init.to('cpu'))
    got, ref = stats['rank_loss'].cpu(), g['rank_loss']
    k = min(len(got), len(ref))
    # the noise-floor control: the reference against ITSELF on the same seeds with another thread count (g26n, written by
    # oracle/make_golden_headline.py --threads 3): per iteration, the reference's own deviation and HIP's as a multiple of it
    ctl_path = os.path.join(ROOT, 'tests', 'golden', control + '.pt')
    ctl = torch.load(ctl_path, weights_only=False) if (name == 'g26_headline_trace' and os.path.exists(ctl_path)) else None
    ref_dev = None
    if ctl is not None:
        kc = min(k, len(ctl['rank_loss']))
        ref_dev = ((ctl['rank_loss'][:kc] - ref[:kc]).abs() / ref[:kc].abs().clamp_min(1e-30)).max(dim=1).values
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
        if ref_dev is not None and i < len(ref_dev):
            rows[-1]['reference_self_deviation'] = float(ref_dev[i])
            rows[-1]['hip_dev_over_ref_dev'] = rel / max(float(ref_dev[i]), 1e-30)
            rows[-1]['references_agree_on_argmin'] = bool(torch.argmin(ctl['rank_loss'][i]) == torch.argmin(ref[i]))
    out['trace'] = {'iterations': k, 'first_iteration_ranking_differs': rank_first, 'first_iteration_argmin_differs': arg_first,
                    'argmin_equal_count': sum(r['argmin_equal'] for r in rows),
                    'iterations_with_clear_gap': sum(r['reference_top2_rel_gap'] > 1e-3 for r in rows),
                    'argmin_mismatch_at_clear_gap': clear_mismatch,
                    'final_top1_equal': rows[-1]['argmin_equal'], 'final_best_loss_hip': rows[-1]['best_loss_hip'],
                    'final_best_loss_reference': rows[-1]['best_loss_reference'],
                    'max_rel_diff_first_5': max(r['rank_loss_max_rel_diff'] for r in rows[:5]),
                    'max_rel_diff_all': max(r['rank_loss_max_rel_diff'] for r in rows), 'per_iteration': rows}
    if ref_dev is not None:
        kc = len(ref_dev)
        both = [r for r in rows[:kc] if r['references_agree_on_argmin']]
        out['trace']['reference_self_deviation'] = {
            'control': control, 'control_threads': int(ctl['reference_threads']), 'iterations': kc,
            'per_iteration': [float(v) for v in ref_dev],
            'first_iteration_references_disagree_on_argmin': next((r['iteration'] for r in rows[:kc] if not r['references_agree_on_argmin']), None),
            'iterations_references_agree_on_argmin': len(both),
            'hip_argmin_mismatch_where_references_agree': [r['iteration'] for r in both if not r['argmin_equal']],
            # windowed: Adam amplifies a rounding difference exponentially over the first iterations, so one iteration's
            # deviation is compared with the control's largest over iterations i-2 .. i+2
            'max_hip_dev_over_windowed_ref_dev': max(
                rows[i]['rank_loss_max_rel_diff'] / max(float(ref_dev[max(0, i - 2):i + 3].max()), 1e-30) for i in range(kc)),
            'hip_dev_over_ref_dev_at_9': rows[9].get('hip_dev_over_ref_dev') if kc > 9 else None}
    if conv_mode:
        out['conv_mode'] = conv_mode
    return out


if __name__ == '__main__':
    cm = None
    if '--conv-mode' in sys.argv:
        i = sys.argv.index('--conv-mode')
        cm = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    res = compare(name=sys.argv[2] if len(sys.argv) > 2 else 'g26_headline_trace', conv_mode=cm)
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(txt)
    t = res['trace']
    print(json.dumps({k: v for k, v in t.items() if k != 'per_iteration'}))
    print(json.dumps(res.get('volume')), json.dumps(res['iteration0_renders']))
    for r in t['per_iteration'][:12] + t['per_iteration'][-3:]:
        print(r)
