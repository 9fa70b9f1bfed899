#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as MI355X_MICROARCH.md
prescribes) of bench.py into profiles/<tag>_pmc_traffic.json: HBM bytes per launch per kernel.
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -o p -- python bench.py --tune-cache T ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -o p -- python bench.py --tune-cache T ...
  python tools/pmc_traffic.py DIR_F DIR_W BATCH SIZE OUT.json [PLAN_MD5 STEPS_IN_THE_TRACE]
PLAN_MD5 = the `plan_md5` of the bench line of the same command (md5 over the [(op, kernel)] launch list); with STEPS (warm-up +
timed steps of the traced command) every kernel also gets launches_per_step.  bench.py only quotes `roofline.traffic` from a
summary whose plan_md5 and launches_per_step match the run it prints (a kernel name alone does not say
which layers it ran).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(all our global->LDS traffic is 16 B/lane), so it is doubled."""
import collections, csv, json, sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d + '/p_counter_collection.csv')):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


if __name__ == '__main__':
    df, dw, batch, size, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    plan_md5 = sys.argv[6] if len(sys.argv) > 6 else None
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else None
    f, w = load(df, 'FETCH_SIZE'), load(dw, 'WRITE_SIZE')
    kernels = {}
    for k in f:
        if k not in w:
            continue
        fb = 2.0 * 1024 * sum(f[k]) / len(f[k])
        wb = 1024.0 * sum(w[k]) / len(w[k])
        kernels[k] = dict(launches=len(f[k]), fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                          hbm_bytes_per_launch=fb + wb)
        if steps and len(f[k]) % steps == 0:
            kernels[k]['launches_per_step'] = len(f[k]) // steps
    json.dump(dict(workload=[batch, size, size], plan_md5=plan_md5, steps_in_trace=steps, note='FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes, mean per launch',
                   kernels=kernels), open(out, 'w'), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['launches'] * kv[1]['hbm_bytes_per_launch'])[:6]:
        print('%-80s %4d launches %8.1f MB/launch' % (k[:80], v['launches'], v['hbm_bytes_per_launch'] / 1e6))
