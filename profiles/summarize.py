"""Prints the figures README / DESIGN / profiles/README quote from a bench line: python profiles/summarize.py profiles/r4/bench_default.json"""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else 'profiles/r4/bench_default.json'))
r, s = d['roofline'], d['roofline_sfa_stage']
print('step ms %.3f  samples/s %.0f  min/max %.3f/%.3f  bf16x6 %.3f' % (d['ms_per_step'], d['value'], d['ms_per_step_min'], d['ms_per_step_max'],
                                                                      d.get('ms_per_step_bf16x6', float('nan'))))
print('fresh', ['%.3f' % f['ms_per_step']['median'] for f in d.get('fresh_processes', []) if 'ms_per_step' in f])
print('writer us %.1f  frac %.3f  of_fill %.3f  of_memset %.3f  fill us %.1f  memset us %.1f  read us %.1f  traffic %s' % (
    r['launch_ms'] * 1e3, r['frac'], r['frac_of_fill'], r['frac_of_memset'], r['fill_ms'] * 1e3, r['memset_ms'] * 1e3, r['read_ms'] * 1e3, r['traffic']))
b = d['roofline_bwd']
print('mghs backward us %.1f frac_effective %.3f frac_hbm %s traffic %s' % (b['launch_ms'] * 1e3, b.get('frac_effective', b.get('frac')), b.get('frac_hbm'), b['traffic']))
print('sfa fwd %.3f bwd %.3f ms  traffic %s' % (s['launch_ms'], s['backward_ms'], s['traffic']))
print('prepare', {k: round(v, 1) for k, v in d.get('prepare', {}).items() if k.endswith('_us')})
a = d.get('hotpath_amp', {})
for k in ('f32_nodes_plus_casts', 'half_io'):
    if k in a:
        print('amp', k, {q: round(v, 3) for q, v in a[k].items()})
o = d.get('roofline_operator', {})
print('operator', {k: round(o[k], 4) for k in o if k.startswith('python') or k in ('launch_ms', 'frac')}, 'traffic', o.get('traffic'))
l = d.get('roofline_dhdl', {})
if 'ms_per_step' in l:
    print('dhdl step', {k: round(v, 4) for k, v in l['ms_per_step'].items() if isinstance(v, float)}, 'writer us %.1f frac %.3f' % (l['launch_ms'] * 1e3, l['frac']))
e = d.get('e2e', {})
if 'fp32' in e:
    print('e2e fp32 %.1f fp16 %.1f samples/s' % (e['fp32']['samples_per_s'], e['fp16']['samples_per_s']))
c = d.get('cpu_baseline', {})
print('cpu', c.get('value'), c.get('cores'), c.get('kind'))
