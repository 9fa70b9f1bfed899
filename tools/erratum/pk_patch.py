#!/usr/bin/env python
"""Instruction-level patches of the packed BatchNorm backward: which CLASS of packed instruction is it?

    python tools/erratum/pk_patch.py <train_pk.s> <patch> <out.s>

Rewrites the body of bn_apply_kernel<bf16_t, 1, 1> in the device assembly hipcc emitted for train.hip (packed build) and leaves
every other instruction, register and the schedule as they are:
  none      nothing (the reassembled kernel must still be corrupted: the control)
  mul       every v_pk_mul_f32 -> two v_mul_f32          add   every v_pk_add_f32 -> two v_add_f32 / v_sub_f32
  mov       every v_pk_mov_b32 ... op_sel:[1,0] -> v_swap_b32 / two v_mov_b32
  all       the three together (no packed instruction left)
  mul-neg / mul-sel / mul-plain / mul-sgpr / mul-inplace / mul-fresh: only that sub-class of the packed multiplies (see patch_line)
  loop-mul / loop-add / prologue-mul ...: the same restricted to the pixel loop (after the first global_store) or to what precedes it
  nop       s_nop 0 after every packed instruction (timing only)
tools/erratum/pk_patch.sh assembles the result into a code object and tools/erratum/pk_patch_run.hip runs it beside the synthetic trigger.
"""
import re
import sys

KERNEL = '_Z15bn_apply_kernelI6bf16_tLi1ELi1EEvPKT_S3_PKfS5_S5_S5_S5_S5_fPS1_ixif7BnFused'
PAIR = r'([vs])\[(\d+):(\d+)\]'


def halves(m):
    k, a, b = m
    return '%s%d' % (k, int(a)), '%s%d' % (k, int(b))


def mods_of(text):
    """op_sel / op_sel_hi / neg_lo / neg_hi lists of a VOP3P instruction's modifier text (defaults: [0,0] [1,1] [0,0] [0,0])."""
    out = {'op_sel': [0, 0], 'op_sel_hi': [1, 1], 'neg_lo': [0, 0], 'neg_hi': [0, 0]}
    for m in re.finditer(r'(op_sel_hi|op_sel|neg_lo|neg_hi):\[([01]),([01])\]', text):
        out[m.group(1)] = [int(m.group(2)), int(m.group(3))]
    rest = re.sub(r'(op_sel_hi|op_sel|neg_lo|neg_hi):\[[01],[01]\]', '', text).strip()
    assert not rest, text
    return out


TMP = 'v254'                                   # the kernel owns v0..v253 of its 256 allocated registers


def scalar_pair(op, d, a, b, md, t):
    """Two scalar instructions for one packed mul / add: d = (lo, hi) destination registers, a / b = (lo, hi) source registers."""
    lo = (a[md['op_sel'][0]], b[md['op_sel'][1]], md['neg_lo'])
    hi = (a[md['op_sel_hi'][0]], b[md['op_sel_hi'][1]], md['neg_hi'])

    def ins(dst, src):
        return '\tv_%s_f32_e64 %s, %s%s, %s%s' % (op, dst, '-' if src[2][0] else '', src[0], '-' if src[2][1] else '', src[1])
    if d[0] not in (hi[0], hi[1]):
        return [ins(d[0], lo), ins(d[1], hi)]
    if d[1] not in (lo[0], lo[1]):
        return [ins(d[1], hi), ins(d[0], lo)]
    return [ins(TMP, lo), ins(d[1], hi), '\tv_mov_b32_e32 %s, %s' % (d[0], TMP)]


def patch_line(line, what):
    t = line.strip()
    m = re.match(r'v_pk_(mul|add)_f32 v\[(\d+):(\d+)\], %s, %s(.*)$' % (PAIR, PAIR), t)
    if m and m.group(1) == 'mul' and any(w.startswith('mul-') for w in what):
        # sub-classes of the packed multiply: mul-neg (neg_lo / neg_hi modifiers), mul-sel (op_sel / op_sel_hi), mul-plain (none),
        # mul-sgpr (an SGPR-pair source), mul-inplace (the destination pair is also a source), mul-fresh (neither)
        mods, srcs = m.group(10), (m.group(4), m.group(7))
        d, a_, b_ = (m.group(2), m.group(3)), (m.group(5), m.group(6)), (m.group(8), m.group(9))
        cls = {'mul-neg': 'neg_' in mods, 'mul-sel': 'op_sel' in mods, 'mul-plain': not mods.strip(), 'mul-sgpr': 's' in srcs,
               'mul-inplace': d in (a_, b_), 'mul-fresh': d not in (a_, b_)}
        if any(cls.get(w) for w in what):
            what = tuple(what) + ('mul',)
    if m and m.group(1) in what:
        op, d = m.group(1), ('v%d' % int(m.group(2)), 'v%d' % int(m.group(3)))
        a = halves(m.groups()[3:6]); b = halves(m.groups()[6:9])
        return scalar_pair(op, d, a, b, mods_of(m.group(10)), t)
    m = re.match(r'v_pk_mov_b32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)$', t)
    if m and 'mov' in what:
        d0, d1, a0, a1, b0, b1 = (int(x) for x in m.groups()[:6])
        md = mods_of(m.group(7))
        assert md['neg_lo'] == [0, 0] and md['neg_hi'] == [0, 0], t
        s0 = (a0, a1)[md['op_sel'][0]]                      # v_pk_mov_b32: D.lo = src0[op_sel[0]], D.hi = src1[op_sel[1]]
        s1 = (b0, b1)[md['op_sel'][1]]
        if d0 == s1 and d1 == s0:
            return ['\tv_swap_b32 v%d, v%d' % (d0, d1)]
        if d0 != s1:
            return ['\tv_mov_b32_e32 v%d, v%d' % (d0, s0), '\tv_mov_b32_e32 v%d, v%d' % (d1, s1)]
        assert d1 != s0, t
        return ['\tv_mov_b32_e32 v%d, v%d' % (d1, s1), '\tv_mov_b32_e32 v%d, v%d' % (d0, s0)]
    if 'nop' in what and re.match(r'v_pk_(mul|add|fma)_f32|v_pk_mov_b32', t):
        return [line.rstrip('\n'), '\ts_nop 0']
    return None


def main():
    src, patch, dst = sys.argv[1:4]
    region = 'all'
    what = patch
    for r in ('loop-', 'prologue-'):
        if patch.startswith(r):
            region, what = r[:-1], patch[len(r):]
    what = {'all': ('mul', 'add', 'mov'), 'none': ()}.get(what, (what,))
    lines = open(src).read().split('\n')
    start = lines.index(KERNEL + ': ; @' + KERNEL) if (KERNEL + ': ; @' + KERNEL) in lines else next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ':'))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    first_store = next(i for i in range(start, end) if lines[i].strip().startswith('global_store_dwordx4'))
    # "prologue" = the set-up of the per-channel factors: everything up to the last v_pk_mov_b32 before the first 16-byte store
    loop_start = 1 + max([i for i in range(start, first_store) if lines[i].strip().startswith('v_pk_mov_b32')] or [start])
    out, n = [], {}
    for i, l in enumerate(lines):
        if start < i < end:
            inreg = region == 'all' or (region == 'loop') == (i >= loop_start)
            r = patch_line(l, what) if inreg else None
            if r is not None:
                key = l.strip().split()[0]
                n[key] = n.get(key, 0) + 1
                out.extend(r)
                continue
        out.append(l)
    open(dst, 'w').write('\n'.join(out))
    print('patch %-14s region %-8s replaced: %s (pixel loop from line %d of the kernel\'s %d)' % (patch, region, n, loop_start - start, end - start))


if __name__ == '__main__':
    main()
