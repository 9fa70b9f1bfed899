#!/usr/bin/env python
"""Summarise a rocprofv3 PMC pass with the SQ/GRBM counters of bench.py into profiles/<tag>_pmc_mfma.json:
per kernel (mean per launch) MFMA-busy cycles, GRBM_GUI_ACTIVE, the wave-cycle split, and from them
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)   (busy counts cycles summed over SIMDs)
  clock_ghz  = (GRBM_GUI_ACTIVE / 8) / kernel duration                             (effective shader clock)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
            --kernel-trace --output-format csv -d DIR -o p -- python bench.py ...
  python tools/pmc_mfma.py DIR OUT.json"""
import collections, csv, json, sys

d, out = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(d + '/p_kernel_trace.csv')):
    dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(d + '/p_counter_collection.csv')):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    t_ns = sum(dur[k]) / max(len(dur[k]), 1)
    gui = m.get('GRBM_GUI_ACTIVE', 0.0)
    e = dict(launches=len(next(iter(c.values()))), duration_us=t_ns / 1e3, **{n: v for n, v in m.items()})
    # GRBM_GUI_ACTIVE counts from the command processor's point of view: for a kernel of a few tens of microseconds it holds
    # launch / drain time that is not shader time, so GUI / duration is not a clock there ("3.1-3.6 GHz", 13 GHz for a copy)
    # and the utilisation derived from it is meaningless.  Both are only reported for kernels of >= 50 us; below that the
    # MFMA-busy fraction is given against the waves' own cycles (SQ_WAVE_CYCLES, no clock involved).
    if gui > 0 and t_ns >= 50e3:
        e['mfma_util'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8.0 * 1024.0)
        e['clock_ghz'] = (gui / 8.0) / t_ns
    elif gui > 0:
        e['short_kernel'] = 'under 50 us: no clock / mfma_util from GRBM_GUI_ACTIVE'
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    if wc > 0:
        e['mfma_busy_per_wave_cycle'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wc
    res[k] = e
json.dump(dict(note='means per launch; counters are sums over XCDs / SEs / SIMDs as rocprofv3 reports them', kernels=res),
          open(out, 'w'), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1]['launches'] * kv[1]['duration_us'])[:8]:
    print('%-70s %4d x %8.1f us  mfma_util %s  clock %s' % (k[:70], e['launches'], e['duration_us'],
          '%.3f' % e['mfma_util'] if 'mfma_util' in e else '  n/a', '%.2f GHz' % e['clock_ghz'] if 'clock_ghz' in e else 'n/a (short kernel)'))
