"""The key numbers of a round's profile set (profiles/<prefix>_*), as quoted in README.md / DESIGN.md / profiles/README.md.
    python tools/doc_numbers.py [prefix=r06]"""
import csv, json, os, sys
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
pt = sys.argv[1] if len(sys.argv) > 1 else 'r06'
last = lambda n: json.loads(open(os.path.join(R, n)).read().strip().splitlines()[-1])
d = last(pt + '_bench.json')
r = d['roofline']
print('416 bs 32: %.0f img/s (median %.0f), net_frac %.3f / %.3f; dominant %s  %d x %.1f us = %.3f; traffic %.1f MB vs %.1f MB; plan %s / launch list %s'
      % (d['value'], d['value_median'], d['net_frac'], d['net_frac_median'], r['kernel'][22:60], r['launches_per_step'], r['avg_launch_us'], r['frac'],
         (r['traffic'] or 0) / 1e6, r['algorithmic_bytes'] / 1e6, d['plan']['md5'], d['plan_md5']))
for k in ('northstar_608_forward', 'northstar_608', 'northstar_608_nms'):
    v = d[k]; print('%-22s %.0f img/s = %.1f %%  %.0f MHz %.0f W' % (k, v['value'], 100 * v['frac_of_peak'], v['sclk_mhz'] or 0, v['power_w'] or 0))
p = d['parity_path']
print('parity_path bf16x3 %.0f img/s (executed %.3f of bf16 peak, %.2fx f32 path %.0f); 608 nms %.0f; f16x3 %s; f16_path %.0f'
      % (p['value'], p['frac_of_bf16_peak_executed'], p['vs_f32_path'], d['f32_path']['value'], p['northstar_608_nms']['value'],
         p.get('f16x3', {}).get('value'), d['f16_path']['value']))
t = d['train_416_bs64']
print('train %.2f ms (%.3f) %.0f MHz; roofline %s' % (t['ms_per_step'], t['net_frac'], t['sclk_mhz'] or 0, json.dumps({k: t.get('roofline', {}).get(k) for k in ('achieved', 'ms_per_step', 'traffic', 'algorithmic_bytes_per_step')})))
c = d['cpu_baseline']
print('cpu %.2f img/s on %d cores (quota %s, %.0f GFLOP/s): %s' % (c['value'], c['cores'], c.get('cpu_quota'), c['gflops'], json.dumps({k: (v['img_s'] if isinstance(v, dict) else v) for k, v in c['tried'].items()})))
print('box_parity', json.dumps({k: '%.2g' % v['box_max'] for k, v in c['box_parity'].items() if isinstance(v, dict)}))
b = last(pt + '_bench_608.json'); r = b['roofline']
print('608 alone: %.0f img/s %.3f; dominant %d x %.1f us = %.3f; traffic %.0f vs %.0f MB' % (b['value'], b['net_frac'], r['launches_per_step'], r['avg_launch_us'], r['frac'], (r['traffic'] or 0) / 1e6, r['algorithmic_bytes'] / 1e6))
x = last(pt + '_bench_x3.json'); r = x['roofline']
print('x3 line: %.0f img/s %.3f; dominant %d x %.1f us = %.3f' % (x['value'], x['net_frac'], r['launches_per_step'], r['avg_launch_us'], r['frac']))
tb = last(pt + '_train_bench.json'); print('train alone %.2f ms' % tb['ms_per_step'], json.dumps({k: tb['roofline'][k] for k in ('achieved', 'ms_per_step')}))
for name, key in ((pt + '_kernel_stats.csv', '3, 4, 2, 2, 3, 768, 1, 0, 1, 0, 0'), (pt + '_608_kernel_stats.csv', '3, 2, 4, 2, 4, 512, 1, 0, 1, 0, 0')):
    for row in csv.DictReader(open(os.path.join(R, name))):
        if key in row['Name'] and 'bf16_t' in row['Name']:
            print('%s: %s calls x %.1f us' % (name, row['Calls'], float(row['AverageNs']) / 1e3)); break
