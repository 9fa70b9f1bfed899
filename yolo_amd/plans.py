"""Committed launch plans: the measured per-shape kernel choices of CarNet / Trainer (`tuning_state()`) as a JSON file.

`tune='measure'` times the kernel variants of every layer shape on the box it runs on, so the same commit launches different
kernels on different boxes and a bench line, a rocprofv3 kernel trace and a PMC pass taken in three processes need not describe
the same launches.  A plan file pins them: `load()` + `load_tuning_state()` before the first forward makes
every shape a cache hit -- nothing is timed, every process launches the plan's kernels; a shape the plan does not hold is still
measured (and counted by `new_keys`).  The reference has no counterpart (MXNet's cudnn autotune is per process,
`MXNET_CUDNN_AUTOTUNE_DEFAULT`); this is measurement infrastructure of the build."""
import hashlib
import json
import os

DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'plan.json')
SECTIONS = ('algo', 'dgrad', 'wgrad')


def _freeze(o):
    return tuple(_freeze(v) for v in o) if isinstance(o, (list, tuple)) else o


def _thaw(o):
    return [_thaw(v) for v in o] if isinstance(o, (list, tuple)) else (int(o) if isinstance(o, bool) else o)


def to_json(state):
    """{section: sorted [[key, value], ...]} -- keys are (nested) tuples of ints / strings; bools are written as 0 / 1 (equal as
    dictionary keys in Python) so that the text, and its md5, do not depend on which of the two a key was built with."""
    out = {}
    for sec in SECTIONS:
        rows = [[_thaw(k), int(v)] for k, v in state.get(sec, {}).items()]
        out[sec] = sorted(rows, key=lambda r: json.dumps(r[0]))
    return out


def from_json(obj):
    return {sec: {_freeze(k): v for k, v in obj.get(sec, [])} for sec in SECTIONS}


def md5(state):
    return hashlib.md5(json.dumps(to_json(state), sort_keys=True).encode()).hexdigest()[:12]


def merge(*states):
    out = {sec: {} for sec in SECTIONS}
    for st in states:
        for sec in SECTIONS:
            out[sec].update(st.get(sec, {}))
    return out


def new_keys(state, base):
    """Number of choices in `state` that `base` did not hold (= shapes measured live although a plan was loaded)."""
    return sum(1 for sec in SECTIONS for k in state.get(sec, {}) if k not in base.get(sec, {}))


def save(path, state, meta=None):
    with open(path, 'w') as f:
        json.dump({'meta': dict(meta or {}, md5=md5(state)), 'plan': to_json(state)}, f, indent=0, separators=(',', ':'))
        f.write('\n')


def load(path):
    """-> (state, meta).  Raises if the file carries no md5 or it does not match its contents (a hand-edited plan)."""
    with open(path) as f:
        obj = json.load(f)
    state = from_json(obj['plan'])
    meta = obj.get('meta', {})
    if not meta.get('md5'):
        raise ValueError('%s: no meta.md5 (write plans with plans.save())' % path)
    if meta['md5'] != md5(state):
        raise ValueError('%s: md5 %s does not match its contents (%s)' % (path, meta['md5'], md5(state)))
    return state, meta
