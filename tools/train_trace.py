#!/usr/bin/env python
"""Per-launch view of one training step from a rocprofv3 --kernel-trace CSV: the launches of the LAST complete step in
order (short name, duration, gap to the previous kernel's end), so that slow instances of a kernel that runs the same
shape many times can be tied to their position in the step.
    python tools/train_trace.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv, glob, os, re, sys

def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('_kernel', '').replace('bf16_t', 'b')
    return n[:44]

def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a step ends with the adam kernel
    ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
    if len(ends) < 2:
        print('need two optimizer launches in the trace'); return
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    t0 = int(seg[0]['Start_Timestamp']); prev_end = t0
    tot = 0
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.write('%9.1f %-44s %8.1f us  gap %7.1f  grid %s wg %s\n' % ((s - t0) / 1e3, short(r['Kernel_Name']), (e - s) / 1e3,
                  (s - prev_end) / 1e3, r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?')))
        prev_end = max(prev_end, e); tot += e - s
    out.write('step span %.2f ms, kernel sum %.2f ms, %d launches\n' % ((prev_end - t0) / 1e6, tot / 1e6, len(seg)))

if __name__ == '__main__':
    main()
