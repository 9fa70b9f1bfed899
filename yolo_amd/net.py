"""CarNet: host-side mirror of the reference's network object.

Same call contract as `CarNet(spec, num_sync_bn_devices=-1)` + `net(x)` (car/utils.py:64-95,
car/YOLO.py:96,381): input (B,3,H,W) float32 NCHW on the device, output a list of 3 tensors
fine->coarse, each (B, H_i*W_i, A, C) float32.  Every arithmetic op runs in libyolo_amd.so
(hand-written HIP for gfx950); torch only owns device memory and the stream.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import lib as L
from .spec import NetGraph, BN_EPS, LEAKY_SLOPE, xavier_bound

_TORCH_DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32, 'bf16x3': torch.bfloat16, 'f16x3': torch.float16}
_LIB_DT = {'bf16': L.BF16, 'f16': L.F16, 'f32': L.F32, 'bf16x3': L.BF16X3, 'f16x3': L.F16X3}
# 'bf16x3' (round 6): SPLIT bf16 -- every activation and weight is a (hi, lo) pair of bf16 numbers, every product three bf16 MFMAs
# (include/yolo_amd.h: YOLO_BF16X3).  The path on which the north-star tolerance and the MFMA rate meet: decoded boxes within
# 1e-3 of the fp32 oracle (~3e-4 on the random-BN D53 nets, tests/test_gpu_boxes.py) at ~1/3 of the bf16 path's speed, where
# the exact-fp32 MFMA of dtype 'f32' runs at ~1/8.  A split activation is a (N, H, W, 2, C) bf16 tensor: plane 0 = hi, 1 = lo.
# 'f16x3': the same with IEEE-half pairs (22 significant bits): boxes as close to the fp32 oracle as the fp32 path's own (6e-5), for nets whose
# activations stay inside half's range (as dtype 'f16' needs); same speed minus the ~4 % half multiplies cost at the power cap.
_SPLIT = ('bf16x3', 'f16x3')


class _Plan(object):
    """Launch list for one input shape: prebuilt descriptors over cached activation buffers."""
    def __init__(self):
        self.ops = []        # (kind, payload)
        self.buffers = []    # keeps tensors alive
        self.x_nhwc = None
        self.merged = None
        self.lp = None       # CarLPNet: (B, h, w, LP channels) float32
        self.offsets = None
        self.act = {}        # name -> (tensor, (N,H,W,C)) for parity taps
        self.side = set()    # indices of ops that run on the side stream


class CarNet(object):
    ALGOS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 31, 32, 33, 35, 36, 37, 38, 39, 40, 41, 42)     # yolo_conv_desc.algo ids tried by tune='measure' (40-42: split type only)

    def __init__(self, spec, num_sync_bn_devices=-1, dtype='bf16', device='cuda:0', tune='auto', tune_cache=None,
                 fuse_stem=True, side_stream=True, fuse_res=True, fuse_concat=True, fuse_tail=True):
        # num_sync_bn_devices is accepted for signature parity; the reference always passes -1
        # (no SyncBN, car/YOLO.py:94-96).
        if dtype not in _TORCH_DT:
            raise ValueError('dtype must be bf16, f16, f32, bf16x3 or f16x3')
        self.graph = NetGraph(spec)
        # 'f16' = the reference's own reduced precision (use_fp16 -> net.cast('float16'), car/YOLO.py:98-100; the executor's fp16
        # flag, yolo_gluon.py:204-214): fp16 activations and weights on v_mfma_f32_32x32x16_f16 -- the bf16 MFMA rate with three more
        # mantissa bits.  Inference only (Trainer refuses it).
        self.dtype = dtype
        self.device = L.resolve_device(device)
        # tune: 'auto' = the library's heuristic picks the conv tile variant; 'measure' = time every
        # variant once per distinct layer shape when a plan is built (HIP events) and pin the fastest
        # (round 6) 'plan' = launch the kernels of a SHIPPED plan file (yolo_amd/plans.py; default profiles/plan.json -- the set
        # bench.py runs and the parity tests compare with the oracle): every layer shape the plan holds gets its variant, a shape it
        # does not hold gets the heuristic's; nothing is ever timed.  What a deployment wants: the same kernels on every box.
        if tune not in ('auto', 'measure', 'plan'):
            raise ValueError("tune must be 'auto', 'measure' or 'plan'")
        self.measure_live = tune == 'measure'      # a shape without a cached choice is timed ('measure') or left to the heuristic ('plan')
        self._plan_state = None
        if tune == 'plan':
            from . import plans
            self._plan_state, self.plan_meta = plans.load(tune_cache or plans.DEFAULT)
            tune, tune_cache = 'measure', None       # (from here on: the cached-choice code paths of 'measure', minus the timing)
        self.tune = tune
        # fuse_stem: run the stem and the first down-sampling conv as one kernel where yolo_stem_down_fwd takes the
        # shape (32 -> 64, bf16); the stem's own output is then not materialised (no 'stem' parity tap)
        self.fuse_stem = bool(fuse_stem)
        # side_stream: a detection block's tip + output convolutions run on a second HIP stream, concurrently with the
        # transition -> up-sample/concat -> first 1x1 chain of the next scale (both only read the block's route): the
        # small kernels of that chain fill the CUs the 13x13 / 26x26 tip kernels leave idle
        self.side_stream = bool(side_stream)
        self._side = None
        # fuse_res: run a residual block of the first stages (C = 64 / 128, bf16) as one kernel (yolo_res_block_fwd): the
        # half-width map between its two convolutions stays in LDS and x is read once.  With tune='measure' the fused
        # kernel is timed against the two separate layers and used where it wins.
        self.fuse_res = bool(fuse_res)
        # fuse_concat: `concat(upsample(transition(x)), route)` (car/utils.py:91-93) without a copy kernel: the stage that
        # produces a route writes it straight into its half of the concat buffer (strided output), the transition conv
        # stores every output pixel to its 2x2 patch of the other half (yolo_conv_desc.upsample2x), and the route's other
        # reader -- the next stage's down-sampling conv -- reads it with an input pixel stride
        self.fuse_concat = bool(fuse_concat)
        # fuse_tail: a 3x3 convolution whose 256-cout tile holds every channel of a pixel also computes the 1x1 convolution that
        # follows it (yolo_conv_desc.tail_*: the next residual block's first conv, a detection block's 1x1, YOLOOutput after the
        # tip) from the output tile it has just stored -- one launch instead of two, no HBM read of the 3x3's output by the 1x1.
        # bf16, Cout <= 256, 1x1 Cout <= 128; with tune='measure' the fused launch is timed against the two separate ones.
        # (measured, round 4: -3..-16 us per pair in isolation, nothing in the whole pass -- so it is only considered where the
        #  tuner can check it, i.e. with tune='measure')
        self._force_tail = fuse_tail == 'force'
        self.fuse_tail = bool(fuse_tail) and (tune == 'measure' or fuse_tail == 'force')      # ('force': tests -- every eligible pair)
        if fuse_tail is True and tune != 'measure' and fuse_tail != 'force':
            self.fuse_tail_note = "fuse_tail=True has no effect under tune='auto' (pairs are only fused where the tuner measured a gain)"
        else:
            self.fuse_tail_note = None
        self._algo_cache = dict(self._plan_state['algo']) if self._plan_state is not None else {}
        self.stale_choices = 0      # adopted choices (plan file / rank 0) this library no longer takes: dropped and measured again
        # optional JSON file remembering measured choices (so a profiled run launches only the chosen kernels)
        self._tune_cache = tune_cache
        if tune_cache and os.path.exists(tune_cache):
            with open(tune_cache) as f:
                self._algo_cache = {tuple(json.loads(k)): v for k, v in json.load(f).items()}
        self.params = {}
        self._prepared = {}
        self._plans = {}
        # parameter version: bumped by everything that changes a parameter or a running statistic (initialize,
        # load_params, every Trainer.train_step); forward() re-folds / re-packs when the prepared images are older
        self._version = 0
        self._prepared_version = -1
        self._lib = L.load()

    # ---- parameters (gluon collect_params().save/load seam, car/YOLO.py:549, yolo_gluon.py:190) --
    def initialize(self, seed=0):
        """init_NN's Xavier branch (yolo_gluon.py:198): weights U(+-bound), gamma 1, beta/mean 0, var 1."""
        gen = torch.Generator(device='cpu').manual_seed(seed)
        for c in self.graph.convs():
            a = xavier_bound(c.cin, c.cout, c.k)
            w = (torch.rand((c.cout, c.cin, c.k, c.k), generator=gen) * 2 - 1) * a
            self._set_param(c.name + '.weight', w.to(self.device))
            if c.bn:
                self._set_param(c.name + '.gamma', torch.ones(c.cout, device=self.device))
                self._set_param(c.name + '.beta', torch.zeros(c.cout, device=self.device))
                self._set_param(c.name + '.running_mean', torch.zeros(c.cout, device=self.device))
                self._set_param(c.name + '.running_var', torch.ones(c.cout, device=self.device))
            else:
                self._set_param(c.name + '.bias', torch.zeros(c.cout, device=self.device))
        self._version += 1
        return self

    def _set_param(self, name, t):
        """Store a parameter.  An existing tensor of the same shape is overwritten IN PLACE: a Trainer's flat
        weight buffer and the launch plans hold views of / pointers into it."""
        old = self.params.get(name)
        if old is not None and old.shape == t.shape:
            old.copy_(t)
        else:
            self.params[name] = t

    def load_params(self, params):
        """params: dict name -> ndarray / tensor, gluon-style names (see ConvSpec.param_names)."""
        for c in self.graph.convs():
            for n in c.param_names():
                if n not in params:
                    raise KeyError('missing parameter %s' % n)
                v = params[n]
                t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
                self._set_param(n, t.detach().to(self.device, torch.float32).contiguous())
        self._version += 1
        return self

    def collect_params(self):
        return self.params

    def averaged_params(self):
        """COLLECTIVE (every rank calls it): the parameter dict with `.running_mean` / `.running_var` averaged over the
        ranks -- what gluon's Parameter._reduce() hands collect_params().save() (car/YOLO.py:549); the live statistics
        stay local to a GPU (no SyncBN).  Hand the result to save_state / save_gluon_params, which never communicate:
            params = net.averaged_params()            # all ranks
            if rank == 0: net.save_state(path, params)"""
        from . import parallel
        return parallel.checkpoint_params(self.params)

    def save_state(self, path, params=None):
        """Writes `params` (averaged_params(), see there) or, by default, this rank's own parameters.  No collective."""
        with open(path, 'wb') as f:                      # (a file object: np.savez(path) would append ".npz" to the name given)
            np.savez(f, **{k: v.detach().cpu().numpy() for k, v in (self.params if params is None else params).items()})

    def load_state(self, path):
        if not os.path.exists(path) and os.path.exists(path + '.npz'):
            path = path + '.npz'                         # (files written before round 3 carry numpy's suffix)
        with np.load(path) as z:
            return self.load_params({k: z[k] for k in z.files})

    def load_gluon_params(self, path, order='auto'):
        """Load a gluon `.params` file written by `net.collect_params().save(path)` (car/YOLO.py:549) or
        `net.export` (yolo_gluon.py:257) -- the counterpart of `collect_params().load(weight, ctx)`
        (yolo_gluon.py:190).  See yolo_amd/mxparams.py for the container and the order mapping."""
        from . import mxparams
        return self.load_params(mxparams.from_gluon(self.graph, mxparams.read_params(path), order))

    def save_gluon_params(self, path, prefix='carnet0_', params=None):
        """`collect_params().save(path)` (car/YOLO.py:549).  params: averaged_params() under N > 1 (see there)."""
        from . import mxparams
        mxparams.write_params(path, mxparams.to_gluon(self.graph, self.params if params is None else params, prefix))

    # ---- one-off preparation: BN folding + weight packing (all on device, HIP kernels) ------------
    def prepare(self):
        L.require_current_device(self.device, 'this CarNet')
        lib, st, dt = self._lib, L.stream_ptr(), _LIB_DT[self.dtype]
        for c in self.graph.convs():
            w = self.params[c.name + '.weight']
            # (split types: the stem runs as a direct convolution on its OIHW weights -- yolo_stem_conv_fwd -- and has no packed image)
            raw_stem = self.dtype in _SPLIT and c is self.graph.stem and c.cin == 3
            nbytes = 0 if raw_stem else lib.yolo_packed_weight_bytes(c.cout, c.cin, c.k, dt)
            if nbytes < 0:
                raise L.YoloError('unsupported conv %s%s' % (c.name, " (the split dtypes need channel counts in multiples of 8)"
                                                             if self.dtype in _SPLIT else ''))
            cp = lib.yolo_padded_channels(c.cout)
            if c.name in self._prepared:
                wp, scale, bias = self._prepared[c.name]      # refreshed in place: the launch plans point at them
            else:
                wp = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                scale = torch.empty(cp, dtype=torch.float32, device=self.device)
                bias = torch.empty(cp, dtype=torch.float32, device=self.device)
            if not raw_stem:
                L.check(lib.yolo_pack_conv_weights(L.ptr(w), L.ptr(wp), c.cout, c.cin, c.k, dt, st), 'pack ' + c.name)
            if c.bn:
                p = lambda s: L.ptr(self.params[c.name + s])
                L.check(lib.yolo_fold_bn(p('.gamma'), p('.beta'), p('.running_mean'), p('.running_var'),
                                         BN_EPS, L.ptr(scale), L.ptr(bias), c.cout, st), 'fold ' + c.name)
            else:
                L.check(lib.yolo_fold_bn(None, L.ptr(self.params[c.name + '.bias']), None, None, BN_EPS,
                                         L.ptr(scale), L.ptr(bias), c.cout, st), 'bias ' + c.name)
            self._prepared[c.name] = (wp, scale, bias)
        self._prepared_version = self._version
        return self

    def _ensure_prepared(self):
        if not self._prepared or self._prepared_version != self._version:
            self.prepare()

    # ---- plan construction --------------------------------------------------------------------------
    def _act(self, N, H, W, Cc):
        """An activation buffer of logical shape (N, H, W, C): NHWC in the net's element type; a split type (bf16x3) carries the
        hi and lo planes of a pixel side by side -- (N, H, W, 2, C)."""
        if self.dtype not in _SPLIT:
            return torch.empty((N, H, W, Cc), dtype=_TORCH_DT[self.dtype], device=self.device)
        # a split plane holds whole 32-channel K-chunks (include/yolo_amd.h): C real channels + zero pad the kernels never write
        Cp = -(-Cc // 32) * 32
        if Cp == Cc:
            return torch.empty((N, H, W, 2, Cc), dtype=_TORCH_DT[self.dtype], device=self.device)
        return torch.zeros((N, H, W, 2, Cp), dtype=_TORCH_DT[self.dtype], device=self.device)[..., :Cc]

    def _conv_op(self, plan, c, x, xshape, residual=None, out=None, out_f32=False, y_bs=0, y_ps=0, cin=None, x_ps=0, up2=False,
                 tail=None):
        """tail: (1x1 conv spec, its output tensor or pointer, out_f32, batch stride, pixel stride) computed by the same launch
        (yolo_conv_desc.tail_*); the caller has checked _tail_eligible and decided (_use_tail)."""
        N, H, W, _ = xshape
        ho, wo = c.out_hw(H, W)
        if out is None:
            out = self._act(N, ho, wo, c.cout)
            plan.buffers.append(out)
        d = self._conv_desc(c, x, xshape, out, residual, out_f32, y_bs, y_ps, cin, x_ps, up2)
        if tail is not None:
            self._set_tail(d, *tail)
            d.algo = self._tail_algo(d)
            plan.ops.append(('conv', d, c.name + '+' + tail[0].name))
        else:
            if self.tune == 'measure':
                d.algo = self._measure_algo(d)
            plan.ops.append(('conv', d, c.name))
        oshape = (N, 2 * ho, 2 * wo, c.cout) if up2 else (N, ho, wo, c.cout)
        if not isinstance(out, int):
            plan.act[c.name] = (out, oshape)
        if tail is not None and not isinstance(tail[1], int):
            plan.act[tail[0].name] = (tail[1], (N, ho, wo, tail[0].cout))
        return out, oshape

    # ---- fused tail 1x1 ---------------------------------------------------------------------------------------------
    TAIL_ALGOS = {1: (7, 6, 2), 2: (17, 9, 18, 16, 10)}      # 8-wave 3x3 variants whose tile can hold a pixel's every channel, by stride

    def _tail_eligible(self, c3, c1):
        return (self.fuse_tail and self.dtype in ('bf16', 'f16') and c3.k == 3 and c3.bn and c3.cout <= 256 and c3.cout % 32 == 0
                and c1.k == 1 and c1.stride == 1 and c1.cin == c3.cout and c1.cout <= 128 and (c1.bn or c1.cout % 2 == 0))

    def _set_tail(self, d, c1, out1, out1_f32=False, t_bs=0, t_ps=0):
        wp1, s1, b1 = self._prepared[c1.name]
        d.tail_w_packed, d.tail_scale, d.tail_bias = L.ptr(wp1), L.ptr(s1), L.ptr(b1)
        d.tail_y = out1 if isinstance(out1, int) else L.ptr(out1)
        d.tail_cout, d.tail_out_f32 = c1.cout, 1 if out1_f32 else 0
        d.tail_slope = LEAKY_SLOPE if c1.bn else 1.0
        d.tail_y_batch_stride, d.tail_y_pixel_stride = t_bs, t_ps

    def _tail_key(self, d):
        # (everything the fused-or-not decision was measured under: the element type, the input view and both outputs' strides --
        #  a decision taken for one layout must not be reused for another through the tune cache / share_tuning)
        return ('tail', d.N, d.H, d.W, d.Cin, d.Cout, d.stride, bool(d.residual), d.tail_cout, d.tail_out_f32, int(d.y_pixel_stride),
                int(d.dtype), int(d.x_pixel_stride), int(d.tail_y_pixel_stride), int(d.tail_y_batch_stride))

    def _tail_algo(self, d):
        """The 256-cout tile variant of a fused launch: the first the library takes, or (tune='measure') the fastest."""
        cands = self.TAIL_ALGOS[d.stride]
        if self.tune != 'measure':
            return 0
        return self._measure_algo(d, algos=cands, key_extra=('tail', d.tail_cout, d.tail_out_f32))

    def _use_tail(self, c3, c1, x, xshape, residual, out, out1, out1_f32=False, t_bs=0, t_ps=0, y_bs=0, y_ps=0):
        """Whether (3x3 c3, then 1x1 c1 on its output) runs as ONE fused launch.  fuse_tail='force' (tests): whenever the library
        takes the pair.  fuse_tail=True acts only under tune='measure' -- the pair is fused when the fused launch is faster than the
        two separate ones with their own best variants (the constructor switches it off under tune='auto' and says so)."""
        if not self._tail_eligible(c3, c1):
            return False
        lib, st = self._lib, L.stream_ptr()
        N, H, W, _ = xshape
        ho, wo = c3.out_hw(H, W)
        d = self._conv_desc(c3, x, xshape, out, residual, False, y_bs, y_ps)
        self._set_tail(d, c1, out1, out1_f32, t_bs, t_ps)
        buf = C.create_string_buffer(256)
        if lib.yolo_conv_kernel_name(C.byref(d), buf, 256) != 0:
            return False
        if self.tune != 'measure' or self._force_tail:
            return True
        key = self._tail_key(d)
        if key in self._algo_cache:
            return bool(self._algo_cache[key])
        if not self.measure_live:
            return False                                  # (tune='plan': pairs are fused only where the plan says a measurement found a gain)
        d3 = self._conv_desc(c3, x, xshape, out, residual, False, y_bs, y_ps)
        d3.algo = self._measure_algo(d3)
        d1 = self._conv_desc(c1, out, (N, ho, wo, c3.cout), out1, None, out1_f32, t_bs, t_ps)
        d1.algo = self._measure_algo(d1)
        d.algo = self._tail_algo(d)

        def timed(fn, n=20):
            best = float('inf')
            for _ in range(3):
                fn(); fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / n)
            return best

        def separate():
            L.check(lib.yolo_conv_fwd(C.byref(d3), st), 'conv')
            L.check(lib.yolo_conv_fwd(C.byref(d1), st), 'conv')

        use = lib.yolo_conv_fwd(C.byref(d), st) == 0 and timed(lambda: lib.yolo_conv_fwd(C.byref(d), st)) < timed(separate)
        self._algo_cache[key] = int(use)
        self._save_tune_cache()
        return use

    def _res_block_payload(self, c1, c2, x, out, shp):
        wp1, s1, b1 = self._prepared[c1.name]
        wp2, s2, b2 = self._prepared[c2.name]
        return (L.ptr(x), L.ptr(wp1), L.ptr(s1), L.ptr(b1), L.ptr(wp2), L.ptr(s2), L.ptr(b2), L.ptr(out), shp[0], shp[1], shp[2], shp[3])

    def _res_block_eligible(self, c1, c2):
        C_ = c1.cin
        return (self.fuse_res and self.dtype in ('bf16', 'f16') and C_ in (64, 128) and c1.bn and c2.bn
                and (c1.k, c1.stride, c1.cin, c1.cout) == (1, 1, C_, C_ // 2) and (c2.k, c2.stride, c2.cin, c2.cout) == (3, 1, C_ // 2, C_))

    def _use_res_block(self, c1, c2, x, shp):
        """Whether the residual block (c1: 1x1 C -> C/2, c2: 3x3 C/2 -> C, + x) runs as the fused kernel."""
        C_ = shp[3]
        if not self._res_block_eligible(c1, c2):
            return False
        if self.tune != 'measure':
            return True
        key = ('res', shp[0], shp[1], shp[2], C_, _LIB_DT[self.dtype])
        if key in self._algo_cache:
            return bool(self._algo_cache[key])
        if not self.measure_live:
            return True                                   # (tune='plan', unknown shape: the heuristic's answer, as under tune='auto')
        lib, st, dt = self._lib, L.stream_ptr(), _LIB_DT[self.dtype]
        tdt = _TORCH_DT[self.dtype]
        mid = torch.empty(shp[:3] + (C_ // 2,), dtype=tdt, device=self.device)
        out = torch.empty(shp, dtype=tdt, device=self.device)
        d1 = self._conv_desc(c1, x, shp, mid)
        d2 = self._conv_desc(c2, mid, shp[:3] + (C_ // 2,), out, residual=x)
        d1.algo, d2.algo = self._measure_algo(d1), self._measure_algo(d2)
        pay = self._res_block_payload(c1, c2, x, out, shp)

        def timed(fn, n=20):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        def separate():
            L.check(lib.yolo_conv_fwd(C.byref(d1), st), 'conv')
            L.check(lib.yolo_conv_fwd(C.byref(d2), st), 'conv')

        fused_ok = lib.yolo_res_block_fwd(*pay, dt, LEAKY_SLOPE, st) == 0
        use = fused_ok and timed(lambda: lib.yolo_res_block_fwd(*pay, dt, LEAKY_SLOPE, st)) < timed(separate)
        self._algo_cache[key] = int(use)
        self._save_tune_cache()
        return use

    # ---- measured kernel choices as a value (N > 1: rank 0 measures, every rank runs rank 0's plan) ----------------
    def tuning_state(self):
        """The measured per-shape kernel choices (tune='measure') as a picklable dict."""
        return {'algo': dict(self._algo_cache)}

    def load_tuning_state(self, state):
        """Adopt another rank's choices: shapes found here are not measured again, so a plan built afterwards launches the
        same kernel instantiations as on the rank the state came from (parallel.share_tuning)."""
        self._algo_cache.update(state['algo'])
        return self

    def plan_signature(self, B, H, W):
        """[(op, kernel instantiation)] of the launch plan for one input shape -- what must agree across ranks."""
        return [(n, k) for n, k, _ in self.plan_kernels(B, H, W)]

    def _save_tune_cache(self):
        if self._tune_cache:
            with open(self._tune_cache, 'w') as f:
                json.dump({json.dumps(list(k)): v for k, v in self._algo_cache.items()}, f)

    def _conv_desc(self, c, x, xshape, out, residual=None, out_f32=False, y_bs=0, y_ps=0, cin=None, x_ps=0, up2=False):
        N, H, W, _ = xshape
        wp, scale, bias = self._prepared[c.name]
        x_lo = y_lo = 0
        if isinstance(x, torch.Tensor) and x.dim() >= 4 and not x.is_contiguous():
            # a channel slice of a wider NHWC buffer (the route half of a concat buffer); split types: (N, H, W, 2, C) views
            if x.stride(-1) != 1 or x.stride(1) != W * x.stride(2) or x.stride(0) != H * W * x.stride(2):
                raise L.YoloError('unsupported input view for %s' % c.name)
            x_ps = x.stride(2)
            if x.dim() == 5:
                x_lo = x.stride(3)
        if isinstance(out, torch.Tensor) and out.dim() >= 4 and not out.is_contiguous():
            if out.stride(-1) != 1 or out.stride(1) != out.shape[2] * out.stride(2):
                raise L.YoloError('unsupported output view for %s' % c.name)
            y_ps, y_bs = out.stride(2), out.stride(0)
            if out.dim() == 5:
                y_lo = out.stride(3)
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias = L.ptr(x), L.ptr(wp), L.ptr(scale), L.ptr(bias)
        d.residual = L.ptr(residual)
        d.y = out if isinstance(out, int) else L.ptr(out)
        d.N, d.H, d.W, d.Cin, d.Cout = N, H, W, (cin or c.cin), c.cout
        d.ksize, d.stride, d.dtype = c.k, c.stride, _LIB_DT[self.dtype]
        d.out_f32 = 1 if out_f32 else 0
        d.slope = LEAKY_SLOPE if c.bn else 1.0
        d.y_batch_stride, d.y_pixel_stride = y_bs, y_ps
        d.x_pixel_stride, d.upsample2x = x_ps, 1 if up2 else 0
        d.x_lo_offset, d.y_lo_offset = x_lo, y_lo
        return d

    def _measure_algo(self, d, iters=5, fn=None, algos=None, key_extra=()):
        """Fastest conv variant for this layer shape (cached).  Outputs are overwritten while timing,
        which is harmless: the plan has not run yet.  fn / algos: another entry point taking the same descriptor
        (yolo_conv_dgrad_s2, with d.ksize = 2 as the cache key's mark) and its variant ids; 1 = none ran."""
        key = (d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.out_f32, bool(d.residual), d.dtype)
        if d.x_pixel_stride or d.upsample2x or (d.y_pixel_stride and not d.out_f32):
            key = key + (int(d.x_pixel_stride), int(d.upsample2x), int(d.y_pixel_stride))
        key = key + tuple(key_extra)
        lib, st = self._lib, L.stream_ptr()
        if key in self._algo_cache:
            # a choice adopted from a plan file / another rank: dry-run it (host only, no launch) -- a library built after the
            # plan was made may no longer take that variant for the shape (a tile's halo budget changed, an id was retired); such
            # an entry is dropped and the shape measured again, which plans.new_keys() then counts as measured live
            cached = self._algo_cache[key]
            if fn is None and cached != 1:
                d.algo = cached
                ok = lib.yolo_conv_kernel_name(C.byref(d), C.create_string_buffer(256), 256) == 0
                d.algo = 0
                if not ok:
                    del self._algo_cache[key]
                    self.stale_choices += 1
            if key in self._algo_cache:
                return cached
        if not self.measure_live:
            return 0                                      # (tune='plan', a shape the plan does not hold: the library's heuristic)
        fn = fn or lib.yolo_conv_fwd

        def time_algo(algo, n):
            d.algo = algo
            if fn(C.byref(d), st) != 0:
                return None
            fn(C.byref(d), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(C.byref(d), st)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        # two passes: a short one over every variant, then the three fastest again with 4x the launches -- a single
        # short timing is noisy enough (DVFS, neighbours' tails) to pick a variant that is 5 % slower
        # (round 3: the second pass INTERLEAVES its candidates over three rounds and keeps each one's fastest round -- timed one
        #  after the other, a clock / power drift of a few per cent between two candidates' windows picked the slower one: the
        #  64 -> 128 stride-2 layer at 608x608 ran the generic kernel, 446 us, where the streaming one takes 417)
        top, mult, rounds = 3, 2, 3
        first = [(t, a) for a in (algos or self.ALGOS) for t in [time_algo(a, iters)] if t is not None]
        first.sort()
        cands = [a for _, a in first[:top]]
        fastest = {a: float('inf') for a in cands}
        for _ in range(rounds):
            for algo in cands:
                t = time_algo(algo, mult * iters)
                if t is not None:
                    fastest[algo] = min(fastest[algo], t)
        best = min(cands, key=lambda a_: fastest[a_]) if cands else 1
        d.algo = 0
        self._algo_cache[key] = best
        self._save_tune_cache()
        return best

    def _stage_ops(self, plan, down, res, x, shp, cat_view, down_done, tdt):
        """The launch list of one backbone stage (down-sampling conv + residual blocks) appended to `plan`: the fused
        residual-block kernel wherever it is eligible and chosen; elsewhere a conv carries the next block's 1x1 as a fused tail
        when the pair is measured faster (_use_tail)."""
        pre_mid = None

        def tail_for(c3, xin, xshp, resid, j_next, out=None):
            """(tail tuple, mid tensor) when conv c3 should also compute res[j_next]'s 1x1, else (None, None)."""
            if j_next >= len(res) or self._res_block_eligible(res[j_next][0], res[j_next][1]):
                return None, None
            c1n = res[j_next][0]
            if not self._tail_eligible(c3, c1n):
                return None, None
            ho_, wo_ = c3.out_hw(xshp[1], xshp[2])
            o3 = out if out is not None else torch.empty((xshp[0], ho_, wo_, c3.cout), dtype=tdt, device=self.device)
            m = torch.empty((xshp[0], ho_, wo_, c1n.cout), dtype=tdt, device=self.device)
            if not self._use_tail(c3, c1n, xin, xshp, resid, o3, m):
                return None, None
            plan.buffers.append(m)
            return (c1n, m, False, 0, 0), m

        if not down_done:
            dout = cat_view if not res else None
            tl, pre_mid = tail_for(down, x, shp, None, 0) if res else (None, None)
            x, shp = self._conv_op(plan, down, x, shp, out=dout, tail=tl)
        for j, (c1, c2) in enumerate(res):
            last = cat_view is not None and j == len(res) - 1
            if not last and pre_mid is None and self._use_res_block(c1, c2, x, shp):
                out = torch.empty(shp, dtype=tdt, device=self.device)
                plan.buffers.append(out)
                plan.ops.append(('res_block', self._res_block_payload(c1, c2, x, out, shp), c2.name))
                plan.act[c2.name] = (out, shp)
                x = out
                continue
            if pre_mid is not None:
                mid, mshp = pre_mid, shp[:3] + (c1.cout,)
            else:
                mid, mshp = self._conv_op(plan, c1, x, shp)
            tl, pre_mid = tail_for(c2, mid, mshp, x, j + 1, out=cat_view if last else None)
            x, shp = self._conv_op(plan, c2, mid, mshp, residual=x, out=cat_view if last else None, tail=tl)
        return x, shp

    def _build_stage(self, plan, down, res, x, shp, cat_view, down_done, tdt):
        """One backbone stage.  (Round 4 also measured, for the stage where both fusions apply -- D53's stage 1 --, a chain of
        fused tails INSTEAD of the residual-block kernel: 620-730 us against 448 us per 3x3 + 1x1 pair at 152x152 bs 64, because
        the 8-wave tiles that can carry a tail are the slow ones for K = 576.  Not offered to the tuner.)"""
        return self._stage_ops(plan, down, res, x, shp, cat_view, down_done, tdt)

    def _build_plan(self, B, H, W):
        g = self.graph
        plan = _Plan()
        tdt = _TORCH_DT[self.dtype]
        fused_down = None
        d0 = g.stages[0][0] if g.stages else None
        if (self.fuse_stem and self.dtype in ('bf16', 'f16') and d0 is not None and g.stem.cin == 3 and g.stem.k == 3
                and g.stem.stride == 1 and g.stem.bn and (g.stem.cout, d0.cin, d0.cout, d0.k, d0.stride, d0.bn) == (32, 32, 64, 3, 2, True)):
            # stem + first down-sampling conv in one kernel (yolo_stem_down_fwd): the 32-channel full-resolution map
            # between them never reaches HBM
            _, s1, b1 = self._prepared[g.stem.name]
            wp2, s2, b2 = self._prepared[d0.name]
            ho, wo = d0.out_hw(H, W)
            x = torch.empty((B, ho, wo, d0.cout), dtype=tdt, device=self.device)
            shp = (B, ho, wo, d0.cout)
            plan.buffers.append(x)
            plan.ops.append(('stem_down', (L.ptr(self.params[g.stem.name + '.weight']), L.ptr(s1), L.ptr(b1), L.ptr(wp2),
                                           L.ptr(s2), L.ptr(b2), L.ptr(x), B, H, W, g.stem.cout, d0.cout), d0.name))
            plan.act[d0.name] = (x, shp)
            fused_down = d0
        elif g.stem.cin == 3 and g.stem.k == 3 and g.stem.stride == 1 and g.stem.bn and (
                (g.stem.cout % 4 == 0 and g.stem.cout <= 64 and self.dtype in ('bf16', 'f16')) or
                (g.stem.cout in (8, 16, 32, 64) and self.dtype in _SPLIT)):
            # fused image-layout change + first conv (yolo_stem_conv_fwd): reads the NCHW image directly
            _, sscale, sbias = self._prepared[g.stem.name]
            x = self._act(B, H, W, g.stem.cout)
            shp = (B, H, W, g.stem.cout)
            plan.buffers.append(x)
            plan.ops.append(('stem', (L.ptr(self.params[g.stem.name + '.weight']), L.ptr(sscale), L.ptr(sbias),
                                      L.ptr(x), B, H, W, 3, g.stem.cout), g.stem.name))
            plan.act[g.stem.name] = (x, shp)
        elif self.dtype in _SPLIT:
            raise L.YoloError("dtype '%s' needs a 3 -> 8 / 16 / 32 / 64-channel 3x3 stem (use dtype='f32' for this spec)" % self.dtype)
        else:
            plan.x_nhwc = torch.empty((B, H, W, 8), dtype=tdt, device=self.device)
            x, shp = self._conv_op(plan, g.stem, plan.x_nhwc, (B, H, W, 8), cin=8)
        routes = []
        nst = len(g.stages)
        cats = {}                                            # stage index -> (concat buffer, channels of the up-sampled half)
        for i, (down, res) in enumerate(g.stages):
            last_of_stage = res[-1][1] if res else down
            cat_view = None
            t_idx = nst - 2 - i                              # the transition whose output is concatenated with this stage's
            if (self.fuse_concat and nst - g.num_pyramid <= i <= nst - 2 and 0 <= t_idx < len(g.transitions)
                    and last_of_stage is not fused_down and not (res and self._res_block_eligible(res[-1][0], res[-1][1]))):
                ho, wo = down.out_hw(shp[1], shp[2])
                up_ch = g.transitions[t_idx].cout
                cat = self._act(B, ho, wo, up_ch + last_of_stage.cout)
                plan.buffers.append(cat)
                cats[i] = (cat, up_ch)
                cat_view = cat[..., up_ch:]
            x, shp = self._build_stage(plan, down, res, x, shp, cat_view, down is fused_down, tdt)
            if i >= nst - g.num_pyramid:
                routes.append((x, shp, cats.get(i)))
        # merged head buffer (B, sum HW, A*C) float32, scales fine->coarse (car/utils.py:95, car/YOLO.py:841)
        hw = [r[1][1] * r[1][2] for r in routes]            # fine -> coarse
        per = [h[3] * g.per_anchor for h in g.heads][::-1]
        if len(set(per)) != 1:
            raise L.YoloError('all scales must have the same number of anchors')
        AC = per[0]
        tot = sum(hw)
        plan.merged = torch.empty((B, tot, AC), dtype=torch.float32, device=self.device)
        offs = [sum(hw[:k]) for k in range(len(hw))]
        plan.offsets = list(zip(offs, hw))
        for i, (body, tip, outc, nA) in enumerate(g.heads):
            if g.lp_out is not None and i >= len(g.heads) - 1:
                # CarLPNet.hybrid_forward (car_and_LP/YOLO.py:72-79): the LP branch reads the finest block's input
                t, tshp = x, shp
                for lbody, ltip in g.lp_blocks:
                    for c in lbody + [ltip]:
                        t, tshp = self._conv_op(plan, c, t, tshp)
                plan.lp = torch.empty((tshp[0], tshp[1], tshp[2], g.lp_out.cout), dtype=torch.float32, device=self.device)
                self._conv_op(plan, g.lp_out, t, tshp, out=plan.lp.data_ptr(), out_f32=True)
            bi_ = 0
            while bi_ < len(body):
                c = body[bi_]
                nxt = body[bi_ + 1] if bi_ + 1 < len(body) else None
                if nxt is not None and self._tail_eligible(c, nxt):
                    ho_, wo_ = c.out_hw(shp[1], shp[2])
                    o3 = torch.empty((shp[0], ho_, wo_, c.cout), dtype=tdt, device=self.device)
                    m = torch.empty((shp[0], ho_, wo_, nxt.cout), dtype=tdt, device=self.device)
                    if self._use_tail(c, nxt, x, shp, None, o3, m):
                        plan.buffers += [o3, m]
                        _, s3 = self._conv_op(plan, c, x, shp, out=o3, tail=(nxt, m, False, 0, 0))
                        x, shp = m, s3[:3] + (nxt.cout,)
                        bi_ += 2
                        continue
                x, shp = self._conv_op(plan, c, x, shp)
                bi_ += 1
            route, rshp = x, shp
            first_side = len(plan.ops)
            k = len(g.heads) - 1 - i                        # position of this scale in fine->coarse order
            yptr = plan.merged.data_ptr() + offs[k] * AC * 4
            tip_fused = False
            if self._tail_eligible(tip, outc):
                ho_, wo_ = tip.out_hw(rshp[1], rshp[2])
                o3 = torch.empty((rshp[0], ho_, wo_, tip.cout), dtype=tdt, device=self.device)
                if self._use_tail(tip, outc, route, rshp, None, o3, yptr, True, tot * AC, AC):
                    plan.buffers.append(o3)
                    self._conv_op(plan, tip, route, rshp, out=o3, tail=(outc, yptr, True, tot * AC, AC))
                    tip_fused = True
            if not tip_fused:
                t, tshp = self._conv_op(plan, tip, route, rshp)
                self._conv_op(plan, outc, t, tshp, out=yptr, out_f32=True, y_bs=tot * AC, y_ps=AC)
            if i >= len(g.heads) - 1:
                break
            if self.side_stream:
                plan.side.update(range(first_side, len(plan.ops)))
            r, rs, rcat = routes[::-1][i + 1]
            if rcat is not None:
                # the route already sits in its half of the concat buffer: the transition conv writes the other half,
                # every pixel to its 2x2 patch (nearest 2x up-sampling)
                cat, up_ch = rcat
                self._conv_op(plan, g.transitions[i], route, rshp, out=cat[..., :up_ch], up2=True)
                x, shp = cat, tuple(cat.shape[:3]) + (cat.shape[-1],)
                continue
            if self.dtype in _SPLIT:
                raise L.YoloError("dtype '%s' needs fuse_concat=True (the up-sample + concat copy kernel takes single-plane types)" % self.dtype)
            x, shp = self._conv_op(plan, g.transitions[i], route, rshp)
            cat = torch.empty((rs[0], rs[1], rs[2], shp[3] + rs[3]), dtype=tdt, device=self.device)
            plan.buffers.append(cat)
            plan.ops.append(('upcat', (L.ptr(x), L.ptr(r), L.ptr(cat), rs[0], rs[1], rs[2], shp[3], rs[3]),
                             'upcat.%d' % i))
            x, shp = cat, (rs[0], rs[1], rs[2], shp[3] + rs[3])
        return plan

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, training=False):
        """-> the three (B, HiWi, A, C) fp32 logit tensors, fine -> coarse (CarLPNet: (outs, [LP])).  They are VIEWS of one merged
        buffer owned by the plan of this input shape -- an executor's output arrays, as in the reference's deployed path
        (`net.forward(is_train=False, data=...)`, yolo_gluon.py:204-242) -- so the next forward of the same shape overwrites them:
        `.clone()` what must outlive it.  (A gluon block called under `autograd` returns fresh arrays; the Trainer keeps its own.)"""
        if training:
            # `self.net(bx)` under autograd.record (car/YOLO.py:381): batch-statistics BatchNorm, everything backward()
            # needs is kept.  Runs through the Trainer that owns the flat parameter buffers (built on first use with the
            # reference's defaults; `Trainer(net, size, ...)` beforehand to choose the hyper-parameters).
            tr = self.trainer((int(x.shape[2]), int(x.shape[3])))
            return tr.forward(x)
        L.require_current_device(self.device, 'this CarNet')
        self._ensure_prepared()
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError('expected a (B,3,H,W) float32 CUDA tensor')
        x = x.contiguous()
        B, _, H, W = x.shape
        down = 2 ** len(self.graph.stages)
        if H % down or W % down:
            # (the reference fails in F.concat for such sizes: the up-sampled map no longer matches its route)
            raise ValueError('image size %dx%d is not a multiple of the total stride %d' % (H, W, down))
        key = (B, H, W)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._build_plan(B, H, W)
        lib, st, dt = self._lib, L.stream_ptr(), _LIB_DT[self.dtype]
        if plan.x_nhwc is not None:
            L.check(lib.yolo_nchw_to_nhwc(x.data_ptr(), L.ptr(plan.x_nhwc), B, 3, H, W, 8, dt, st), 'nchw_to_nhwc')
        if plan.side:
            self._run_two_streams(plan, x, dt)
        else:
            for kind, payload, name in plan.ops:
                rc = self._launch(kind, payload, x, st, dt)
                if rc:
                    raise L.YoloError('%s (%s) failed with status %d' % (kind, name, rc))
        self._last_plan = plan
        A = self.graph.heads[0][3]
        outs = [plan.merged[:, o:o + n].view(B, n, A, self.graph.per_anchor) for o, n in plan.offsets]
        if self.graph.lp_out is not None:
            return outs, [plan.lp]                 # CarLPNet: (all_output[::-1], [LP_output]), car_and_LP/YOLO.py:95
        return outs

    __call__ = forward

    def trainer(self, size=None):
        """The Trainer attached to this net (gluon.Trainer(net.collect_params(), 'adam', ...), car/YOLO.py:106); built with
        the reference's defaults when none exists yet (or when the image size -- the anchor grid -- changed)."""
        tr = getattr(self, '_trainer', None)
        if tr is None or (size is not None and tuple(size) != tr.size):
            if size is None:
                raise L.YoloError('no Trainer is attached to this net yet: pass the image size')
            from .train import Trainer
            # a new image size needs a new anchor grid and activation plan, NOT a new optimiser: hyper-parameters, Adam moments
            # and the update count move over (gluon's Trainer is independent of the input size)
            tr = Trainer(self, size) if tr is None else tr.resized(size)
        return tr

    def backward(self, grads, lp_grads=None):
        """`.backward()` of a loss the caller computed on the training-mode logits (car/YOLO.py:392-394): grads = list of 3
        d(loss)/d(logits) shaped like forward(x, training=True)'s outputs.  Gradients land in `self.grads()`; the update is
        `self.trainer().step(batch_size)` (car/YOLO.py:396)."""
        self.trainer().backward(grads, lp_grads=lp_grads)

    def grads(self):
        """name -> gradient tensor (views of the flat fp32 gradient buffer) after backward()."""
        return self.trainer().grads()

    def _run_two_streams(self, plan, x, dt):
        """The launch list with the ops of plan.side on the side stream.  A run of side ops starts after everything
        launched so far on the main stream (it reads the route the main stream has just produced) and the main stream
        joins the side stream once, at the end (only the caller reads the head logits)."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side, on_side, used = self._side, False, False
        for i, (kind, payload, name) in enumerate(plan.ops):
            want = i in plan.side
            if want and not on_side:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                used = True
            on_side = want
            rc = self._launch(kind, payload, x, (side if want else main).cuda_stream, dt)
            if rc:
                raise L.YoloError('%s (%s) failed with status %d' % (kind, name, rc))
        if used:
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)

    def _launch(self, kind, payload, x, st, dt):
        lib = self._lib
        if kind == 'conv':
            return lib.yolo_conv_fwd(C.byref(payload), st)
        if kind == 'stem':
            w, sc, bi, y, B, H, W, cin, cout = payload
            return lib.yolo_stem_conv_fwd(x.data_ptr(), w, sc, bi, y, B, H, W, cin, cout, dt, LEAKY_SLOPE, st)
        if kind == 'stem_down':
            w1, s1, b1, wp2, s2, b2, y, B, H, W, c1, c2 = payload
            return lib.yolo_stem_down_fwd(x.data_ptr(), w1, s1, b1, wp2, s2, b2, y, B, H, W, c1, c2, dt, LEAKY_SLOPE, st)
        if kind == 'res_block':
            return lib.yolo_res_block_fwd(*payload, dt, LEAKY_SLOPE, st)
        return lib.yolo_upsample2x_concat(*payload, dt, st)

    def plan_kernels(self, B, H, W):
        """[(op name, kernel instantiation name, algorithmic FLOPs)] for the launch list of one input
        shape (FLOPs = 2*Cin*k^2*Cout*Ho*Wo*B per conv; 0 for non-conv ops)."""
        self._ensure_prepared()
        plan = self._plans.get((B, H, W))
        if plan is None:
            plan = self._plans[(B, H, W)] = self._build_plan(B, H, W)
        by_name = {c.name: c for c in self.graph.convs()}
        out = []
        buf = C.create_string_buffer(256)
        for kind, payload, name in plan.ops:
            if kind == 'stem':
                c = by_name[name]
                out.append((name, 'stem_conv_kernel', 2 * c.cin * 9 * c.cout * payload[4] * payload[5] * payload[6]))
                continue
            if kind == 'stem_down':
                B_, H_, W_ = payload[7], payload[8], payload[9]
                c = by_name[name]
                ho, wo = c.out_hw(H_, W_)
                out.append((name, 'stem_down_kernel', 2 * 3 * 9 * payload[10] * B_ * H_ * W_ + 2 * c.cin * 9 * c.cout * ho * wo * B_))
                continue
            if kind == 'res_block':
                N_, H_, W_, C_ = payload[8:12]
                out.append((name, ('res_block2_kernel<%d>' if C_ == 128 else 'res_block_kernel<%d>') % C_, 2 * N_ * H_ * W_ * (C_ * (C_ // 2) + 9 * (C_ // 2) * C_)))
                continue
            if kind != 'conv':
                out.append((name, kind, 0))
                continue
            L.check(self._lib.yolo_conv_kernel_name(C.byref(payload), buf, 256), 'kernel_name')
            fl = 0
            for part in name.split('+'):                     # ('a+b': conv a with conv b as its fused tail 1x1, same output map)
                c = by_name[part]
                if part == name.split('+')[0]:
                    ho, wo = c.out_hw(payload.H, payload.W)
                fl += 2 * c.cin * c.k * c.k * c.cout * ho * wo * payload.N
            out.append((name, buf.value.decode(), fl))
        return out

    def plan_bytes(self, B, H, W):
        """{op name: ALGORITHMIC HBM bytes of that launch} for the conv launches of one input shape: input read once + output
        written once (+ the residual read once) + weights once (SURVEY section 8d's per-layer roofline convention; a fused tail adds
        its own output and weights, its input never leaves the chip) -- the figure `roofline.traffic` is compared with."""
        self._ensure_prepared()
        self.plan_kernels(B, H, W)
        plan = self._plans[(B, H, W)]
        by_name = {c.name: c for c in self.graph.convs()}
        es = 2 if self.dtype in ('bf16', 'f16') else 4       # (split types: two 2-byte planes per value, weights as a hi + lo pair)
        out = {}
        for kind, d, name in plan.ops:
            if kind != 'conv':
                continue
            parts = name.split('+')
            c = by_name[parts[0]]
            ho, wo = c.out_hw(d.H, d.W)
            px = d.N * ho * wo
            nb = d.N * d.H * d.W * d.Cin * es + d.Cout * d.Cin * c.k * c.k * es
            nb += px * d.Cout * (4 if d.out_f32 else es) * (4 if d.upsample2x else 1)
            if d.residual:
                nb += px * d.Cout * es
            for part in parts[1:]:
                t = by_name[part]
                nb += px * t.cout * (4 if d.tail_out_f32 else es) + t.cout * t.cin * es
            out[name] = nb
        return out

    def forward_timed(self, x, events):
        """forward() that brackets every launch with a pair of torch CUDA events (recorded on the
        stream the kernels run on).  events: list that receives (op name, start, end)."""
        B, _, H, W = x.shape
        plan = self._plans[(B, H, W)]
        lib, st, dt = self._lib, L.stream_ptr(), _LIB_DT[self.dtype]
        if plan.x_nhwc is not None:
            L.check(lib.yolo_nchw_to_nhwc(x.data_ptr(), L.ptr(plan.x_nhwc), B, 3, H, W, 8, dt, st), 'nchw_to_nhwc')
        for kind, payload, name in plan.ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self._launch(kind, payload, x, st, dt)
            e1.record()
            if rc:
                raise L.YoloError('%s (%s) failed with status %d' % (kind, name, rc))
            events.append((name, e0, e1))
        self._last_plan = plan

    def merged_output(self):
        """(B, sum HW, A, C) view of the last forward: the concat of merge_and_slice (car/YOLO.py:841-849)."""
        m = self._last_plan.merged
        A = self.graph.heads[0][3]
        return m.view(m.shape[0], m.shape[1], A, self.graph.per_anchor)

    def activation_nchw(self, name):
        """float32 NCHW copy of a named conv output of the last forward (parity taps)."""
        t, (N, H, W, Cc) = self._last_plan.act[name]
        if self.dtype in _SPLIT:                              # (parity tap: value = hi + lo, exact in fp32)
            return (t[..., 0, :].float() + t[..., 1, :].float()).permute(0, 3, 1, 2).contiguous()
        t = t.contiguous()                                   # (a channel slice of a concat buffer is a strided view)
        out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=self.device)
        L.check(self._lib.yolo_nhwc_to_nchw(L.ptr(t), L.ptr(out), N, Cc, H, W, _LIB_DT[self.dtype], L.stream_ptr()),
                'nhwc_to_nchw')
        return out


class CarLPNet(CarNet):
    """car_and_LP/YOLO.py:47-95: CarNet + the licence-plate branch.  forward(x) -> (outs, [LP_output]) with
    LP_output (B, h, w, LP_slice_point[-1]) float32 at the finest scale."""

    def __init__(self, spec, *args, **kwargs):
        if 'LP_slice_point' not in spec:
            raise ValueError("CarLPNet needs spec['LP_slice_point'] (car_and_LP/v1/spec.yaml)")
        super(CarLPNet, self).__init__(spec, *args, **kwargs)

