"""Training step: host-side mirror of YOLO._train_batch (car/YOLO.py:350-399) -- forward with train-mode
BatchNorm, target assignment (_find_best/_loss_mask), the five losses (_get_loss), backward of
sum(losses), gradient all-reduce over ranks and the MXNet Adam update of trainer.step(batch_size).

fp32 parity path: every op is a HIP kernel from libyolo_amd.so (train.hip, loss.hip and the forward conv
kernels re-used for the data gradient on flipped weights); torch owns memory, the stream and the
process group only.  One process per GPU; BN statistics stay local to the GPU (no SyncBN,
car/YOLO.py:94-96).
"""
import os
import ctypes as C

import numpy as np
import torch

from . import lib as L
from . import parallel
from .detect import make_grid, default_ltrb
from .spec import BN_EPS, BN_MOMENTUM, LEAKY_SLOPE

DEFAULT_SCALE = {'score': 0.1, 'box_yx': 0.01, 'box_hw': 10.0, 'rotate': 0.0, 'class': 0.3}   # car/v1/spec.yaml:31-35
LP_DEFAULT_SCALE = {'LP_score': 0.1, 'LP_xy': 10.0, 'LP_z': 1.0, 'LP_r': 0.1, 'LP_class': 0.0}   # car_and_LP/v1/spec.yaml


class _T(object):
    """An activation of the training graph: forward value + (lazily) its gradient."""
    __slots__ = ('val', 'shape', 'grad', 'ready', 'ngot')

    def __init__(self, val, shape):
        self.val, self.shape, self.grad, self.ready, self.ngot = val, shape, None, False, 0


class Trainer(object):
    def __init__(self, net, size, scale=None, learning_rate=1e-3, positive_weight=1.0, negative_weight=0.1,
                 car_rotate=False, beta1=0.9, beta2=0.999, eps=1e-8, lp_scale=None, lp_r_max=(45, 60, 45),
                 lp_positive_weight=1.0, lp_negative_weight=0.1, grad_exchange='f32', grad_buckets=4):
        # grad_exchange / grad_buckets (N > 1): the dtype the gradient buckets travel in ('f32' = the reference's KVStore sum;
        # 'bf16' halves the bytes per xGMI link, parallel.GradBuckets) and how many buckets the 492 MB buffer is cut into
        # dtype of activations and activation gradients: 'f32' (parity path) or 'bf16' (MFMA bf16 convolutions,
        # transposing-read weight gradient); master weights, weight gradients, BN statistics and Adam are fp32.
        self.net, self.size = net, (int(size[0]), int(size[1]))
        self.tdt = torch.float32 if net.dtype == 'f32' else torch.bfloat16
        if net.dtype not in ('f32', 'bf16'):
            raise L.YoloError("Trainer: dtype %r is inference only (the training kernels take 'f32' | 'bf16')" % (net.dtype,))
        self.ldt = L.F32 if net.dtype == 'f32' else L.BF16
        self.lib, self.dev = net._lib, net.device
        self.scale = dict(DEFAULT_SCALE if scale is None else scale)
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, eps
        self.pos_w, self.neg_w, self.car_rotate = positive_weight, negative_weight, car_rotate
        # CarLPNet (car_and_LP/v1/spec.yaml): scales of the five LP losses, LP_r_max, LP score weights
        self.lp_scale = dict(LP_DEFAULT_SCALE if lp_scale is None else lp_scale)
        self.lp_r_max = tuple(float(v) for v in lp_r_max)
        self.lp_pos_w, self.lp_neg_w = lp_positive_weight, lp_negative_weight
        self.t = 0
        g = net.graph
        self.grid, self.nbox = make_grid(g.anchors, self.size, g.steps())
        self.anchors_ltrb = torch.from_numpy(default_ltrb(g.anchors, self.size, g.steps())).to(self.dev).contiguous()
        # ---- flat parameter / gradient / Adam-state buffers; net.params become views of wflat ------------
        names = []
        for c in g.convs():
            names += [c.name + '.weight'] + ([c.name + '.gamma', c.name + '.beta'] if c.bn else [c.name + '.bias'])
        self.names = names
        sizes = [net.params[n].numel() for n in names]
        # 16-byte aligned views
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += (s + 3) // 4 * 4
        # + one 16-byte slot behind the last parameter: the rank's shard size, SUM-reduced with the last gradient bucket
        # (the batch_size of trainer.step(batch_size), car/YOLO.py:396, when the shards are uneven: yolo_gluon.py:100-124)
        self.nparam = o
        self.wflat = torch.zeros(o + 4, dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.wflat)
        self.mflat = torch.zeros_like(self.wflat)
        self.vflat = torch.zeros_like(self.wflat)
        self.pview, self.gview = {}, {}
        for n, s, of in zip(names, sizes, offs):
            shp = net.params[n].shape
            self.wflat[of:of + s].copy_(net.params[n].reshape(-1))
            net.params[n] = self.wflat[of:of + s].view(shp)
            self.pview[n] = net.params[n]
            self.gview[n] = self.gflat[of:of + s].view(shp)
        self._gb_slot = self.gflat[o:o + 1]
        self.buckets = parallel.GradBuckets(self.gflat, names, offs, sizes, nbuckets=grad_buckets, dtype=grad_exchange, exact_tail=4)
        net._trainer = self                     # CarNet.forward(x, training=True) / CarNet.backward(grads) run through it
        # the parameters moved into the flat buffer: the net's launch plans hold pointers to the old tensors (stem
        # weights), and its folded / packed images are re-made on the next inference forward (net._version)
        net._plans = {}
        net._version += 1
        self._prep = {}
        self._prep_s2 = {}          # conv name -> (sub-pixel dgrad weight image, ones, zeros); see _dgrad
        self._plans = {}
        self._dgrad_algo = {}
        self._wgrad_algo = {}
        if getattr(net, '_plan_state', None) is not None:      # CarNet(tune='plan'): the plan's data- and weight-gradient choices too
            self._dgrad_algo.update(net._plan_state['dgrad'])
            self._wgrad_algo.update(net._plan_state['wgrad'])
        self._fwd_B = None
        cmax = max(c.cout for c in g.convs())
        # two BatchNorm workspaces used alternately (yolo_bn_train_*_pp: a call leaves its own dirty and zeroes the next one's)
        self.ws2 = [torch.zeros(2 * cmax, dtype=torch.float64, device=self.dev) for _ in range(2)]
        self._ws_i = 0
        self.probe = None         # a list: _backward appends (family, layer, elements, start event, end event) per BatchNorm backward call
        self._bn3 = bool(L.lab_knob('YOLO_TRAIN_BN3'))      # (the knob: separate finalize launches, for A/B runs)
        self.ws = torch.zeros(2 * cmax, dtype=torch.float64, device=self.dev)
        wsb = max(self.lib.yolo_conv_wgrad_workspace_bytes(max(c.cin, 8), c.cout, c.k, self.ldt) for c in g.convs())
        self.wg_ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=self.dev)       # (kept zeroed by the library)
        # weight gradients run on a side stream: they are off the backward pass's critical path (dy -> data gradient ->
        # previous layer's BN backward) and MFMA-bound, while the BN passes they overlap are HBM-bound
        self._side = torch.cuda.Stream(device=self.dev)
        self._overlap = not L.lab_knob('YOLO_TRAIN_SERIAL_WGRAD')          # (the knob keeps the serial order for A/B runs)
        # BatchNorm sums taken in the producing convolution's epilogue (yolo_conv_desc.stats) instead of in a reduction pass
        # of their own; (the knob: the separate passes, for A/B runs)
        self._fuse_stats = self.ldt == L.BF16 and not L.lab_knob('YOLO_TRAIN_NO_STATS_FUSION')
        self._stats_b = None            # partial rows of the data gradients' statistics epilogues (grown on demand)
        modes = L.lab_knob('YOLO_TRAIN_STATS_MODES', '1')    # '1' forward sums, '2' backward sums, '12' both
        self._fuse_fwd, self._fuse_bwd = self._fuse_stats and '1' in modes, self._fuse_stats and '2' in modes
        self._identity = not L.lab_knob('YOLO_TRAIN_UNIT_EPILOGUE')        # (the knob: scale 1 / bias 0 arrays instead of the identity epilogue)
        self._repack()
        self._packed_version = net._version

    def resized(self, size):
        """A Trainer for another image size with THIS one's hyper-parameters and optimiser state (Adam moments, update count):
        the anchor grid and the activation plan depend on the size, the optimiser does not."""
        new = Trainer(self.net, size, scale=self.scale, learning_rate=self.lr, positive_weight=self.pos_w, negative_weight=self.neg_w,
                      car_rotate=self.car_rotate, beta1=self.b1, beta2=self.b2, eps=self.eps, lp_scale=self.lp_scale,
                      lp_r_max=self.lp_r_max, lp_positive_weight=self.lp_pos_w, lp_negative_weight=self.lp_neg_w)
        new.mflat.copy_(self.mflat)
        new.vflat.copy_(self.vflat)
        new.t = self.t
        return new

    # ---- weight images for the forward and data-gradient convolutions (re-packed after every update) ----
    def _repack(self):
        """Forward and data-gradient weight images of every conv, re-packed after each update in ONE launch
        (yolo_pack_conv_weights_batch) over a device-resident table built on first use."""
        lib, st = self.lib, L.stream_ptr()
        if not self._prep:
            items = np.zeros(0, dtype=[('w', '<u8'), ('packed', '<u8'), ('cout', '<i4'), ('cin', '<i4'), ('k', '<i4'), ('dgrad', '<i4')])
            recs, first = [], [0]
            pairs, pfirst = [], [0]
            pair_dt = np.dtype([('w', '<u8'), ('fwd', '<u8'), ('dgrad', '<u8'), ('cout', '<i4'), ('cin', '<i4'), ('k', '<i4'), ('r', '<i4')])
            for c in self.net.graph.convs():
                w = self.pview[c.name + '.weight']
                # (zero-filled: the pair kernel never writes the images' padding rows)
                wp = torch.zeros(lib.yolo_packed_weight_bytes(c.cout, c.cin, c.k, self.ldt), dtype=torch.uint8, device=self.dev)
                wd = torch.zeros(lib.yolo_packed_weight_bytes(c.cin, c.cout, c.k, self.ldt), dtype=torch.uint8, device=self.dev)
                cp = lib.yolo_padded_channels(max(c.cout, c.cin))
                ones = torch.zeros(cp, dtype=torch.float32, device=self.dev); ones[:max(c.cout, c.cin)] = 1.0
                bias = torch.zeros(cp, dtype=torch.float32, device=self.dev)
                self._prep[c.name] = (wp, wd, ones, bias, torch.zeros(cp, dtype=torch.float32, device=self.dev))
                nb = lib.yolo_pack_pair_blocks(c.cout, c.cin, c.k) if (self.ldt == L.BF16 and not L.lab_knob('YOLO_TRAIN_OLD_PACK')) else -1
                if nb > 0:
                    # both images from one read of the weights (yolo_pack_conv_weights_pairs)
                    pairs.append((w.data_ptr(), wp.data_ptr(), wd.data_ptr(), c.cout, c.cin, c.k, 0))
                    pfirst.append(pfirst[-1] + nb)
                else:
                    # (the dgrad record carries the arguments of yolo_pack_conv_weights_dgrad after its swap: rows = Cin_f)
                    recs.append((w.data_ptr(), wp.data_ptr(), c.cout, c.cin, c.k, 0))
                    first.append(first[-1] + lib.yolo_pack_batch_blocks(c.cout, c.cin, c.k, self.ldt))
                    recs.append((w.data_ptr(), wd.data_ptr(), c.cin, c.cout, c.k, 1))
                    first.append(first[-1] + lib.yolo_pack_batch_blocks(c.cin, c.cout, c.k, self.ldt))
                if (c.k == 3 and c.stride == 2 and self.ldt == L.BF16 and c.cin % 8 == 0 and c.cout % 32 == 0
                        and not L.lab_knob('YOLO_TRAIN_DILATED_DGRAD')):      # (the knob keeps the old form for A/B runs)
                    # sub-pixel data gradient (yolo_conv_dgrad_s2): 2x2-window image with 4 x Cin_f output channels
                    w2 = torch.empty(lib.yolo_packed_weight_bytes(4 * c.cin, c.cout, 2, self.ldt), dtype=torch.uint8, device=self.dev)
                    cp4 = lib.yolo_padded_channels(4 * c.cin)
                    ones4 = torch.zeros(cp4, dtype=torch.float32, device=self.dev); ones4[:4 * c.cin] = 1.0
                    self._prep_s2[c.name] = (w2, ones4, torch.zeros(cp4, dtype=torch.float32, device=self.dev))
                    recs.append((w.data_ptr(), w2.data_ptr(), 4 * c.cin, c.cout, 2, 2))
                    first.append(first[-1] + lib.yolo_pack_batch_blocks(4 * c.cin, c.cout, 2, self.ldt))
            items = np.array(recs, dtype=items.dtype)
            self._pack_items = torch.from_numpy(items.view(np.uint8).copy()).to(self.dev)
            self._pack_first = torch.tensor(first, dtype=torch.int64, device=self.dev)
            self._pack_n, self._pack_blocks = len(recs), first[-1]
            self._pair_n, self._pair_blocks = len(pairs), pfirst[-1]
            if pairs:
                self._pair_items = torch.from_numpy(np.array(pairs, dtype=pair_dt).view(np.uint8).copy()).to(self.dev)
                self._pair_first = torch.tensor(pfirst, dtype=torch.int64, device=self.dev)
        if self._pack_n:
            L.check(lib.yolo_pack_conv_weights_batch(L.ptr(self._pack_items), L.ptr(self._pack_first), self._pack_n,
                                                     self._pack_blocks, self.ldt, st), 'pack batch')
        if self._pair_n:
            L.check(lib.yolo_pack_conv_weights_pairs(L.ptr(self._pair_items), L.ptr(self._pair_first), self._pair_n,
                                                     self._pair_blocks, st), 'pack pairs')
        for c in self.net.graph.convs():
            if not c.bn:
                self._prep[c.name][3][:c.cout].copy_(self.pview[c.name + '.bias'])

    # ---- plan ---------------------------------------------------------------------------------------------
    def _new(self, shape):
        return _T(torch.empty(shape, dtype=self.tdt, device=self.dev), tuple(shape))

    def _conv_desc(self, x, xshape, wp, scale, bias, y, cin, cout, k, stride, residual=None, out_f32=0, y_bs=0, y_ps=0):
        d = L.ConvDesc()
        d.x, d.w_packed, d.scale, d.bias = L.ptr(x), L.ptr(wp), L.ptr(scale), L.ptr(bias)
        d.residual = L.ptr(residual) if residual is not None else None
        d.y = y if isinstance(y, int) else L.ptr(y)
        d.N, d.H, d.W, d.Cin, d.Cout = xshape[0], xshape[1], xshape[2], cin, cout
        d.ksize, d.stride, d.dtype, d.out_f32, d.slope = k, stride, self.ldt, out_f32, 1.0
        d.y_batch_stride, d.y_pixel_stride, d.algo = y_bs, y_ps, 0
        return d

    def _tune(self, d):
        """Pick the conv variant by measurement when the net was built with tune='measure' (the timing runs only
        overwrite d.y, which nothing has consumed yet)."""
        if getattr(self.net, 'tune', None) == 'measure':
            d.algo = self.net._measure_algo(d)

    def _build(self, B, H, W):
        g = self.net.graph
        P = type('Plan', (), {})()
        P.fwd, P.tensors = [], []
        P.stats_floats = 0
        P.x8 = self._new((B, H, W, 8))

        def conv_bn(c, xin, residual=None):
            N, Hh, Ww, Cc = xin.shape
            ho, wo = c.out_hw(Hh, Ww)
            yraw, z = self._new((N, ho, wo, c.cout)), self._new((N, ho, wo, c.cout))
            mean = torch.empty(c.cout, dtype=torch.float32, device=self.dev)
            invstd = torch.empty_like(mean)
            wp, wd, ones, bias, zeros = self._prep[c.name]
            ident = (None, None) if self._identity else (ones, zeros)     # raw convolution: identity epilogue
            d = self._conv_desc(xin.val, xin.shape, wp, ident[0], ident[1], yraw.val, Cc, c.cout, c.k, c.stride)
            self._tune(d)
            op = dict(kind='conv_bn', c=c, x=xin, yraw=yraw, z=z, mean=mean, invstd=invstd, res=residual, desc=d, srows=0, brows=0)
            # (small maps keep the reduction pass of their own: it costs nothing there, and its sums are taken around a value of the
            #  channel -- the epilogue's plain fp32 partial sums lose digits when a few nearly equal values make mean^2 >> variance)
            if self._fuse_fwd and not (c is g.stem) and N * ho * wo >= 4096 and self._pipe_kernel(d):
                d.stats, d.stats_mode = 1, 1                      # (any non-NULL pointer for the query)
                rows = self.lib.yolo_conv_stats_rows(C.byref(d))
                if rows > 0:
                    op['srows'] = rows
                    P.stats_floats = max(P.stats_floats, rows * 2 * self.lib.yolo_padded_channels(c.cout))
                else:
                    d.stats, d.stats_mode = None, 0
            P.fwd.append(op)
            return z

        x = conv_bn(g.stem, P.x8)
        routes = []
        nst = len(g.stages)
        for i, (down, res) in enumerate(g.stages):
            x = conv_bn(down, x)
            for c1, c2 in res:
                x = conv_bn(c2, conv_bn(c1, x), residual=x)
            if i >= nst - g.num_pyramid:
                routes.append(x)
        hw = [r.shape[1] * r.shape[2] for r in routes]
        A = g.heads[0][3]
        AC = A * g.per_anchor
        tot = sum(hw)
        offs = [sum(hw[:k]) for k in range(len(hw))]
        P.merged = torch.empty((B, tot, AC), dtype=torch.float32, device=self.dev)
        P.dmerged = torch.empty_like(P.merged)
        P.tot, P.AC, P.A = tot, AC, A
        P.lp = P.dlp = None
        for i, (body, tip, outc, nA) in enumerate(g.heads):
            if g.lp_out is not None and i >= len(g.heads) - 1:
                # CarLPNet's LP branch (car_and_LP/YOLO.py:72-79): reads the input of the finest detection block
                t = x
                for lbody, ltip in g.lp_blocks:
                    for c in lbody + [ltip]:
                        t = conv_bn(c, t)
                lc = g.lp_out
                hw_lp = t.shape[1] * t.shape[2]
                P.lp = torch.empty((B, hw_lp, lc.cout), dtype=torch.float32, device=self.dev)
                P.dlp = torch.empty_like(P.lp)
                wp, wd, ones, bias, zeros = self._prep[lc.name]
                d = self._conv_desc(t.val, t.shape, wp, ones, bias, P.lp.data_ptr(), lc.cin, lc.cout, 1, 1, out_f32=1)
                self._tune(d)
                cpad = (lc.cout + 7) // 8 * 8
                P.fwd.append(dict(kind='out', c=lc, x=t, desc=d, hw=hw_lp, cpad=cpad,
                                  src=(P.dlp.data_ptr(), hw_lp * lc.cout, lc.cout),
                                  dyp=torch.empty((B * hw_lp, cpad), dtype=self.tdt, device=self.dev)))
                P.lp_hw = (t.shape[1], t.shape[2])
            for c in body:
                x = conv_bn(c, x)
            route = x
            t = conv_bn(tip, route)
            k = len(g.heads) - 1 - i
            wp, wd, ones, bias, zeros = self._prep[outc.name]
            yptr = P.merged.data_ptr() + offs[k] * AC * 4
            d = self._conv_desc(t.val, t.shape, wp, ones, bias, yptr, outc.cin, outc.cout, 1, 1, out_f32=1, y_bs=tot * AC, y_ps=AC)
            self._tune(d)
            cpad = (outc.cout + 7) // 8 * 8
            P.fwd.append(dict(kind='out', c=outc, x=t, desc=d, hw=hw[k], cpad=cpad,
                              src=(P.dmerged.data_ptr() + offs[k] * AC * 4, tot * AC, AC),
                              dyp=torch.empty((B * hw[k], cpad), dtype=self.tdt, device=self.dev)))
            if i >= len(g.heads) - 1:
                break
            x = conv_bn(g.transitions[i], route)
            r = routes[::-1][i + 1]
            cat = self._new((r.shape[0], r.shape[1], r.shape[2], x.shape[3] + r.shape[3]))
            P.fwd.append(dict(kind='upcat', up=x, route=r, cat=cat))
            x = cat
        # forward statistics partials: one buffer, consumed by the BatchNorm call right behind each convolution
        P.stats_f = torch.empty(max(P.stats_floats, 4), dtype=torch.float32, device=self.dev)
        for op in P.fwd:
            if op['kind'] == 'conv_bn' and op['srows']:
                op['desc'].stats = L.ptr(P.stats_f)
        # which layer produced a tensor, and how many gradient contributions it will receive (its consumers)
        P.prod, P.nuse = {}, {}
        for op in P.fwd:
            if op['kind'] == 'conv_bn':
                P.prod[id(op['z'])] = op
            for k in ('x', 'res', 'up', 'route'):
                t = op.get(k)
                if t is not None:
                    P.nuse[id(t)] = P.nuse.get(id(t), 0) + 1
        return P

    def _pipe_kernel(self, d):
        """True when yolo_conv_fwd serves d with a kernel that has a statistics epilogue (pipelined, generic and streaming ones for the forward sums)."""
        buf = C.create_string_buffer(256)
        return self.lib.yolo_conv_kernel_name(C.byref(d), buf, 256) == 0 and any(
            k in buf.value for k in (b'conv_pipe_kernel', b'conv_igemm_kernel', b'conv_stream_kernel'))

    def _next_ws(self):
        """(this call's BatchNorm workspace -- zero --, the one it zeroes for the next call)."""
        a, b = self.ws2[self._ws_i], self.ws2[self._ws_i ^ 1]
        self._ws_i ^= 1
        return L.ptr(a), L.ptr(b)

    # ---- forward (train mode) -----------------------------------------------------------------------------
    def _forward(self, P, images):
        lib, st = self.lib, L.stream_ptr()
        B, _, H, W = images.shape
        L.check(lib.yolo_nchw_to_nhwc(images.data_ptr(), L.ptr(P.x8.val), B, 3, H, W, 8, self.ldt, st), 'nchw_to_nhwc')
        for op in P.fwd:
            if op['kind'] == 'conv_bn':
                c = op['c']
                y, z = op['yraw'], op['z']
                g = self.net.graph
                if c is g.stem and self.ldt == L.BF16 and c.cin == 3 and c.cout % 4 == 0 and c.cout <= 64:
                    # the fused NCHW-image stem kernel of the inference path (identity scale/bias, linear): raw y
                    wp, wd, ones, bias, zeros = self._prep[c.name]
                    rows = lib.yolo_stem_stats_rows(B, H, W, c.cout) if self._fuse_fwd else -1
                    if rows > 0:
                        # the stem's batch sums are taken in the kernel too (the largest reduction pass of the step)
                        if getattr(P, 'stem_part', None) is None:
                            P.stem_part = torch.empty(rows * 2 * c.cout, dtype=torch.float32, device=self.dev)
                        L.check(lib.yolo_stem_conv_fwd_stats(images.data_ptr(), L.ptr(self.pview[c.name + '.weight']), L.ptr(ones),
                                                             L.ptr(zeros), L.ptr(y.val), B, H, W, 3, c.cout, self.ldt, 1.0,
                                                             L.ptr(P.stem_part), st), 'stem')
                        npix, p = B * H * W, self.net.params
                        ws, wn = self._next_ws()
                        L.check(lib.yolo_bn_train_fwd_partials(L.ptr(P.stem_part), rows, c.cout, L.ptr(y.val),
                                                               L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']), None,
                                                               L.ptr(z.val), L.ptr(op['mean']), L.ptr(op['invstd']),
                                                               L.ptr(p[c.name + '.running_mean']), L.ptr(p[c.name + '.running_var']),
                                                               ws, wn, self.ws2[0].numel(), npix, c.cout, BN_EPS, BN_MOMENTUM,
                                                               LEAKY_SLOPE, self.ldt, st), 'bn (stem partials)')
                        continue
                    L.check(lib.yolo_stem_conv_fwd(images.data_ptr(), L.ptr(self.pview[c.name + '.weight']), L.ptr(ones),
                                                   L.ptr(zeros), L.ptr(y.val), B, H, W, 3, c.cout, self.ldt, 1.0, st), 'stem')
                else:
                    L.check(lib.yolo_conv_fwd(C.byref(op['desc']), st), 'conv ' + c.name)
                npix = y.shape[0] * y.shape[1] * y.shape[2]
                p = self.net.params
                if self._bn3:
                    L.check(lib.yolo_bn_train_fwd(L.ptr(y.val), L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']),
                                                  L.ptr(op['res'].val) if op['res'] is not None else None, L.ptr(z.val),
                                                  L.ptr(op['mean']), L.ptr(op['invstd']), L.ptr(p[c.name + '.running_mean']),
                                                  L.ptr(p[c.name + '.running_var']), L.ptr(self.ws), npix, c.cout, BN_EPS,
                                                  BN_MOMENTUM, LEAKY_SLOPE, self.ldt, st), 'bn ' + c.name)
                    continue
                ws, wn = self._next_ws()
                if op['srows']:
                    L.check(lib.yolo_bn_train_fwd_partials(L.ptr(P.stats_f), op['srows'], lib.yolo_padded_channels(c.cout),
                                                           L.ptr(y.val), L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']),
                                                           L.ptr(op['res'].val) if op['res'] is not None else None, L.ptr(z.val),
                                                           L.ptr(op['mean']), L.ptr(op['invstd']), L.ptr(p[c.name + '.running_mean']),
                                                           L.ptr(p[c.name + '.running_var']), ws, wn, self.ws2[0].numel(), npix, c.cout,
                                                           BN_EPS, BN_MOMENTUM, LEAKY_SLOPE, self.ldt, st), 'bn (partials) ' + c.name)
                    continue
                L.check(lib.yolo_bn_train_fwd_pp(L.ptr(y.val), L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']),
                                                 L.ptr(op['res'].val) if op['res'] is not None else None, L.ptr(z.val),
                                                 L.ptr(op['mean']), L.ptr(op['invstd']), L.ptr(p[c.name + '.running_mean']),
                                                 L.ptr(p[c.name + '.running_var']), ws, wn, self.ws2[0].numel(), npix, c.cout, BN_EPS,
                                                 BN_MOMENTUM, LEAKY_SLOPE, self.ldt, st), 'bn ' + c.name)
            elif op['kind'] == 'out':
                L.check(lib.yolo_conv_fwd(C.byref(op['desc']), st), 'out conv')
            else:
                up, r, cat = op['up'], op['route'], op['cat']
                L.check(lib.yolo_upsample2x_concat(L.ptr(up.val), L.ptr(r.val), L.ptr(cat.val), r.shape[0], r.shape[1],
                                                   r.shape[2], up.shape[3], r.shape[3], self.ldt, st), 'upcat')

    # ---- backward -------------------------------------------------------------------------------------------
    def _accum(self, t, src):
        """grad[t] (+)= src (a tensor of the same shape).  First contribution aliases src."""
        t.ngot += 1
        if not t.ready:
            t.grad, t.ready = src, True
        else:
            L.check(self.lib.yolo_add(L.ptr(t.grad), L.ptr(src), L.ptr(t.grad), src.numel(), self.ldt, L.stream_ptr()), 'add')

    def _dgrad(self, c, dy, dy_shape, xin, cin_of_dy):
        """grad[xin] (+)= data gradient of conv c given dy (N,Ho,Wo,cin_of_dy) (dense)."""
        lib, st = self.lib, L.stream_ptr()
        wp, wd, ones, bias, zeros = self._prep[c.name]
        N, Hh, Ww, Cx = xin.shape
        s2 = self._prep_s2.get(c.name) if c.stride == 2 else None
        if s2 is not None and Hh == 2 * dy_shape[1] and Ww == 2 * dy_shape[2] and cin_of_dy == c.cout:
            # sub-pixel form: one 2x2-window conv over dy writes the four phases of dx (no dilated copy, 16/36 of the MFMAs)
            if not xin.ready:
                out, resid = torch.empty(xin.shape, dtype=self.tdt, device=self.dev), None
            else:
                out, resid = xin.grad, xin.grad
            sb = (None, None) if self._identity else (s2[1], s2[2])
            d = self._conv_desc(dy, dy_shape, s2[0], sb[0], sb[1], out, cin_of_dy, 4 * Cx, 2, 1, residual=resid)
            rc = None
            if getattr(self.net, 'tune', None) == 'measure':
                key = ('s2', dy_shape, cin_of_dy, Cx, resid is not None)
                if key not in self._dgrad_algo:
                    scratch = torch.zeros(xin.shape, dtype=self.tdt, device=self.dev)
                    dm = self._conv_desc(dy, dy_shape, s2[0], sb[0], sb[1], scratch, cin_of_dy, 4 * Cx, 2, 1,
                                         residual=scratch if resid is not None else None)
                    self._dgrad_algo[key] = self.net._measure_algo(dm, fn=lib.yolo_conv_dgrad_s2, algos=(2, 6, 10, 4))
                d.algo = self._dgrad_algo[key]
                if d.algo == 1:
                    rc = L.EUNSUPPORTED
            if rc is None:
                rc = lib.yolo_conv_dgrad_s2(C.byref(d), st)
            if rc == 0:
                xin.grad, xin.ready = out, True
                xin.ngot += 1
                return
            if rc != L.EUNSUPPORTED:
                L.check(rc, 'dgrad_s2 ' + c.name)
            del self._prep_s2[c.name]              # no variant fits this shape: the dilated form from now on
        if c.stride == 2:
            dil = torch.empty((N, Hh, Ww, cin_of_dy), dtype=self.tdt, device=self.dev)
            L.check(lib.yolo_dilate2x(L.ptr(dy), L.ptr(dil), N, Hh, Ww, dy_shape[1], dy_shape[2], cin_of_dy, self.ldt, st), 'dilate')
            src, sshape = dil, (N, Hh, Ww, cin_of_dy)
        else:
            src, sshape = dy, dy_shape
        if not xin.ready:
            out = torch.empty(xin.shape, dtype=self.tdt, device=self.dev)
            resid = None
        else:
            out, resid = xin.grad, xin.grad
        if self._identity:
            ones = zeros = None
        d = self._conv_desc(src, sshape, wd, ones, zeros, out, cin_of_dy, Cx, c.k, 1, residual=resid)
        # this call completes d(loss)/d(xin): take the BatchNorm-backward sums of the layer that produced xin in the epilogue
        P = self._P
        prod = P.prod.get(id(xin)) if (self._fuse_bwd and c.stride == 1 and xin.ngot + 1 == P.nuse.get(id(xin), 0)) else None
        if getattr(self.net, 'tune', None) == 'measure':
            key = (sshape, cin_of_dy, Cx, c.k, resid is not None)
            if key not in self._dgrad_algo:
                # time the variants on scratch outputs: the real `out` may already hold an accumulated gradient
                scratch = torch.zeros(xin.shape, dtype=self.tdt, device=self.dev)
                dm = self._conv_desc(src, sshape, wd, ones, zeros, scratch, cin_of_dy, Cx, c.k, 1,
                                     residual=scratch if resid is not None else None)
                self._dgrad_algo[key] = self.net._measure_algo(dm)
            d.algo = self._dgrad_algo[key]
        if prod is not None and self._pipe_kernel(d):
            pc, pp = prod['c'], self.net.params
            d.stats, d.stats_mode, d.stats_y = 1, 2, L.ptr(prod['yraw'].val)
            d.stats_mean, d.stats_invstd = L.ptr(prod['mean']), L.ptr(prod['invstd'])
            d.stats_gamma, d.stats_beta, d.stats_slope = L.ptr(pp[pc.name + '.gamma']), L.ptr(pp[pc.name + '.beta']), LEAKY_SLOPE
            rows = lib.yolo_conv_stats_rows(C.byref(d))
            if rows > 0:
                need = rows * 2 * lib.yolo_padded_channels(Cx)
                if self._stats_b is None or self._stats_b.numel() < need:
                    self._stats_b = torch.empty(need, dtype=torch.float32, device=self.dev)
                d.stats = L.ptr(self._stats_b)
                prod['brows'] = rows
            else:
                d.stats, d.stats_mode = None, 0
        L.check(lib.yolo_conv_fwd(C.byref(d), st), 'dgrad ' + c.name)
        xin.grad, xin.ready = out, True
        xin.ngot += 1

    def _wgrad_tag(self, c, cin):
        """Which weight-gradient kernel serves conv c ('walk' = csrc/wgrad_walk.hip); only YOLO_SIDE_FILTER (diagnostics:
        the classes that go to the side stream) looks at it."""
        if self.ldt == L.BF16 and c.k == 3 and c.stride == 1 and cin % 64 == 0 and c.cout % 64 == 0:
            return 'walk'
        return 'k%ds%d' % (c.k, c.stride)

    def _wgrad_algo_for(self, c, dy, xin):
        """yolo_conv_wgrad_algo id for conv c: 0 (the library's choice) unless the net was built with tune='measure' -- then the
        fastest of the kernels that take the shape, timed once per layer shape on a scratch gradient (the 8-wave row walk wins
        on two D53 shapes, the 16-column walker on the 13x13 ones, ...)."""
        if getattr(self.net, 'tune', None) != 'measure' or self.ldt != L.BF16 or L.lab_knob('YOLO_TRAIN_NO_WGRAD_TUNE'):
            return 0
        N, Hh, Ww, Cx = xin.shape
        key = (N, Hh, Ww, Cx, c.cout, c.k, c.stride)
        if key not in self._wgrad_algo and not getattr(self.net, 'measure_live', True):
            return 0                                      # (tune='plan', a shape the plan does not hold: the library's choice)
        if key not in self._wgrad_algo:
            cands = (0, 2, 3, 4) if (c.k == 3 and c.stride == 1) else (0, 1, 5) if c.k == 1 else (0,)
            best, best_t = 0, float('inf')
            if len(cands) > 1:
                lib, st = self.lib, L.stream_ptr()
                # the candidates run on THIS stream with the workspace the side stream's weight gradients share (their
                # finishing passes accumulate into it and zero it): nothing of the side stream may be in flight
                main = torch.cuda.current_stream()
                main.wait_stream(self._side)
                scratch = torch.zeros((c.cout, Cx, c.k, c.k), dtype=torch.float32, device=self.dev)
                call = lambda a: lib.yolo_conv_wgrad_algo(L.ptr(dy), L.ptr(xin.val), L.ptr(scratch), N, Hh, Ww, Cx, c.cout, c.k,
                                                          c.stride, 0, self.ldt, L.ptr(self.wg_ws), a, st)
                for a in cands:
                    if call(a) != 0:
                        continue
                    call(a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(6):
                        call(a)
                    e1.record(); e1.synchronize()
                    t = e0.elapsed_time(e1)
                    if t < best_t * 0.98:                       # (a later candidate must win by 2 %)
                        best, best_t = a, t
                self._side.wait_stream(main)
            self._wgrad_algo[key] = best
        return self._wgrad_algo[key]

    def _wgrad(self, dy, names, launch, tag=''):
        """Run launch(stream) -- a weight-gradient call reading dy, which the current stream has just produced -- on the
        side stream (all of them, in order: they share one workspace).  The gradient buckets hear about `names` one
        layer later, once the current stream has been made to wait for that layer's side-stream work."""
        main = torch.cuda.current_stream()
        flt = L.lab_knob('YOLO_SIDE_FILTER')                  # (diagnostics: exactly these classes go to the side stream)
        side = (tag in flt.split(',')) if flt is not None else True
        if not self._overlap or not side:
            if self._overlap:
                # (mixed main / side launches: earlier side-stream weight gradients use the same workspace, and a bucket
                #  these names complete may hold gradients the side stream is still writing)
                self._flush_wgrad()
                main.wait_stream(self._side)
            launch(main.cuda_stream)
            self.buckets.done(names)
            return
        ready = torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)
        if isinstance(dy, torch.Tensor):
            dy.record_stream(self._side)
        with torch.cuda.stream(self._side):
            launch(self._side.cuda_stream)
            done = torch.cuda.Event()
            done.record(self._side)
        self._flush_wgrad()
        self._pending_wgrad = (names, done)

    def _flush_wgrad(self):
        if self._pending_wgrad is not None:
            names, done = self._pending_wgrad
            self._pending_wgrad = None
            if self.buckets.active():
                # the main stream waits for the side stream only when these parameters COMPLETE a bucket (4 times per step,
                # not once per layer: the side stream runs its kernels in order, so its latest event covers the earlier ones)
                self.buckets.done(names, before_launch=lambda: torch.cuda.current_stream().wait_event(done))

    def _backward(self, P, exchange=True, capture=None):
        """capture: optional dict that receives, per conv name, the gradient w.r.t. the layer output as the layer saw it
        (`dz`, a copy: the buffer is re-used as the residual branch's gradient) and w.r.t. the raw convolution output
        (`dy`) -- parity tests re-derive every gradient of the step from them (tests/test_gpu_configs.py)."""
        lib, st = self.lib, L.stream_ptr()
        g = self.net.graph
        self.gflat.zero_()
        self._gb_slot.fill_(float(P.merged.shape[0]))               # this rank's shard size (see __init__)
        self.buckets.reset(enabled=exchange)
        self._pending_wgrad = None
        self._P = P
        for op in P.fwd:
            for k in ('x', 'z', 'up', 'route', 'cat', 'res'):
                t = op.get(k)
                if t is not None:
                    t.grad, t.ready, t.ngot = None, False, 0
            if op['kind'] == 'conv_bn':
                op['brows'] = 0
        B = P.merged.shape[0]
        for op in reversed(P.fwd):
            kind = op['kind']
            if kind == 'out':
                c, xin = op['c'], op['x']
                hw, cpad = op['hw'], op['cpad']
                src, src_bs, src_ps = op['src']            # this output's slice of d(loss)/d(logits), fp32
                L.check(lib.yolo_gather_rows(src, L.ptr(op['dyp']), B, hw, c.cout, cpad, src_bs, src_ps, self.ldt, st), 'gather')
                L.check(lib.yolo_bias_grad(L.ptr(op['dyp']), L.ptr(self.gview[c.name + '.bias']), B * hw, c.cout, cpad, self.ldt, st), 'db')
                N, Hh, Ww, Cx = xin.shape
                self._wgrad(op['dyp'], [c.name + '.weight', c.name + '.bias'], lambda s_, c=c, op=op, xin=xin, N=N, Hh=Hh, Ww=Ww, Cx=Cx, cpad=cpad: L.check(
                    lib.yolo_conv_wgrad(L.ptr(op['dyp']), L.ptr(xin.val), L.ptr(self.gview[c.name + '.weight']),
                                        N, Hh, Ww, Cx, c.cout, 1, 1, cpad, self.ldt, L.ptr(self.wg_ws), s_), 'wgrad out'), tag='out')
                self._dgrad(c, op['dyp'], (N, Hh, Ww, cpad), xin, cpad)
            elif kind == 'upcat':
                up, r, cat = op['up'], op['route'], op['cat']
                if not up.ready:
                    up.grad = torch.empty(up.shape, dtype=self.tdt, device=self.dev)
                if not r.ready:
                    r.grad = torch.empty(r.shape, dtype=self.tdt, device=self.dev)
                L.check(lib.yolo_upsample2x_concat_bwd(L.ptr(cat.grad), L.ptr(up.grad), L.ptr(r.grad), r.shape[0], r.shape[1],
                                                       r.shape[2], up.shape[3], r.shape[3], int(up.ready), int(r.ready), self.ldt, st),
                        'upcat bwd')
                up.ready = r.ready = True
                up.ngot += 1
                r.ngot += 1
            else:
                c, xin, y, z = op['c'], op['x'], op['yraw'], op['z']
                dz = z.grad
                npix = y.shape[0] * y.shape[1] * y.shape[2]
                p = self.net.params
                dy = torch.empty(y.shape, dtype=self.tdt, device=self.dev)
                if capture is not None and capture.get('_poison'):
                    dy.fill_(float('nan'))          # (diagnostics: an element the kernel does not write must show)
                if self._bn3:
                    L.check(lib.yolo_bn_train_bwd(L.ptr(dz), L.ptr(y.val), L.ptr(op['mean']), L.ptr(op['invstd']),
                                                  L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']), L.ptr(dy),
                                                  L.ptr(self.gview[c.name + '.gamma']), L.ptr(self.gview[c.name + '.beta']),
                                                  L.ptr(self.ws), npix, c.cout, LEAKY_SLOPE, self.ldt, st), 'bn bwd ' + c.name)
                elif op['brows']:
                    # the data gradient that completed dz took sum(da), sum(da * xhat) in its epilogue
                    ws, wn = self._next_ws()
                    L.check(lib.yolo_bn_train_bwd_partials(L.ptr(self._stats_b), op['brows'], lib.yolo_padded_channels(c.cout),
                                                           L.ptr(dz), L.ptr(y.val), L.ptr(op['mean']), L.ptr(op['invstd']),
                                                           L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']), L.ptr(dy),
                                                           L.ptr(self.gview[c.name + '.gamma']), L.ptr(self.gview[c.name + '.beta']),
                                                           ws, wn, self.ws2[0].numel(), npix, c.cout, LEAKY_SLOPE, self.ldt, st),
                            'bn bwd (partials) ' + c.name)
                else:
                    ws, wn = self._next_ws()
                    pr = self.probe
                    if pr is not None:                  # (measurement only, bench.py's training roofline: HIP events around the call)
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    L.check(lib.yolo_bn_train_bwd_pp(L.ptr(dz), L.ptr(y.val), L.ptr(op['mean']), L.ptr(op['invstd']),
                                                     L.ptr(p[c.name + '.gamma']), L.ptr(p[c.name + '.beta']), L.ptr(dy),
                                                     L.ptr(self.gview[c.name + '.gamma']), L.ptr(self.gview[c.name + '.beta']),
                                                     ws, wn, self.ws2[0].numel(), npix, c.cout, LEAKY_SLOPE, self.ldt, st), 'bn bwd ' + c.name)
                    if pr is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        pr.append(('bn_bwd', c.name, npix * c.cout, e0, e1))
                if capture is not None:
                    capture[c.name] = dict(dz=dz.clone(), dy=dy)
                if op['res'] is not None:
                    self._accum(op['res'], dz)          # the residual branch receives dz unchanged
                N, Hh, Ww, Cx = xin.shape
                names = [c.name + '.weight', c.name + '.gamma', c.name + '.beta']
                if c is g.stem:
                    def stem_wgrad(s_, c=c, dy=dy, xin=xin, N=N, Hh=Hh, Ww=Ww):
                        # (torch ops below run on the stream _wgrad has made current)
                        dw8 = torch.zeros((c.cout, 8, 3, 3), dtype=torch.float32, device=self.dev)
                        L.check(lib.yolo_conv_wgrad(L.ptr(dy), L.ptr(xin.val), L.ptr(dw8), N, Hh, Ww, 8, c.cout, 3, 1, 0, self.ldt,
                                                    L.ptr(self.wg_ws), s_), 'wgrad stem')
                        self.gview[c.name + '.weight'].copy_(dw8[:, :3])
                    self._wgrad(dy, names, stem_wgrad, tag='stem')
                else:
                    wa = self._wgrad_algo_for(c, dy, xin)
                    self._wgrad(dy, names, lambda s_, c=c, dy=dy, xin=xin, N=N, Hh=Hh, Ww=Ww, Cx=Cx, wa=wa: L.check(
                        lib.yolo_conv_wgrad_algo(L.ptr(dy), L.ptr(xin.val), L.ptr(self.gview[c.name + '.weight']), N, Hh, Ww,
                                                 Cx, c.cout, c.k, c.stride, 0, self.ldt, L.ptr(self.wg_ws), wa, s_), 'wgrad ' + c.name),
                                tag=self._wgrad_tag(c, Cx))
                    self._dgrad(c, dy, y.shape, xin, c.cout)
        self._flush_wgrad()
        if self._overlap:
            torch.cuda.current_stream().wait_stream(self._side)

    # ---- the reference's three calls: net(x) under autograd.record / loss.backward() / trainer.step(batch_size) -----
    def forward(self, images):
        """`self.net(bx)` in train mode (car/YOLO.py:381): (B,3,H,W) float32 NCHW -> list of 3 fp32 (B, HiWi, A, C) logits
        fine -> coarse with batch-statistics BatchNorm (running statistics updated); CarLPNet: (outs, [LP_output]).  The
        tensors are views of one buffer that the next forward of the same batch size overwrites."""
        L.require_current_device(self.dev, 'this Trainer')
        if images.dim() != 4 or images.shape[1] != 3 or images.dtype != torch.float32 or not images.is_cuda:
            raise ValueError('expected a (B,3,H,W) float32 CUDA tensor')
        images = images.contiguous()
        B, _, H, W = images.shape
        if (H, W) != self.size:
            raise ValueError('image size differs from the anchor grid the trainer was built for')
        P = self._plans.get(B)
        if P is None:
            P = self._plans[B] = self._build(B, H, W)
        if self.net._version != self._packed_version:
            # net.load_params / initialize since the last step (they write into the flat buffer's views in place)
            if any(self.net.params[n].data_ptr() != v.data_ptr() for n, v in self.pview.items()):
                raise L.YoloError('a parameter tensor of the net was replaced after the Trainer was built')
            self._repack()
        self._forward(P, images)
        self._fwd_B = B
        self._last = (P,)
        # the running statistics moved: an inference forward of the same net (the reference's _valid_iou every
        # valid_step, car/YOLO.py:501-534) must re-fold
        self.net._version += 1
        self._packed_version = self.net._version
        return self._outs(P)

    def _outs(self, P):
        g = self.net.graph
        hw = [op['hw'] for op in P.fwd if op['kind'] == 'out' and op['c'] is not g.lp_out][::-1]
        outs, o = [], 0
        for n in hw:
            outs.append(P.merged[:, o:o + n].view(P.merged.shape[0], n, P.A, g.per_anchor))
            o += n
        if g.lp_out is not None:
            return outs, [P.lp.view(P.lp.shape[0], P.lp_hw[0], P.lp_hw[1], g.lp_out.cout)]
        return outs

    def backward(self, grads, lp_grads=None, exchange=True, capture=None):
        """`sum(losses).backward()` (car/YOLO.py:394) for a loss the CALLER computed on forward()'s logits: grads = the list
        of 3 d(loss)/d(logits) tensors, shaped like forward()'s outputs (or ONE merged (B, sum HiWi, A, C) tensor);
        CarLPNet: lp_grads = d(loss)/d(LP_output).  Fills grads() and -- with a process group -- starts the bucketed
        SUM all-reduce of the gradient buffer (exchange=False: local gradients only)."""
        L.require_current_device(self.dev, 'this Trainer')
        if self._fwd_B is None:
            raise L.YoloError('backward() without a training-mode forward()')
        P = self._plans[self._fwd_B]
        B = P.merged.shape[0]
        if isinstance(grads, torch.Tensor):
            P.dmerged.copy_(grads.reshape(B, P.tot, P.AC))
        else:
            o = 0
            for gr in grads:
                n = gr.shape[1]
                P.dmerged[:, o:o + n].copy_(gr.reshape(B, n, P.AC))
                o += n
            if o != P.tot:
                raise ValueError('the gradients do not cover the %d cells of the three scales' % P.tot)
        if P.dlp is not None:
            if lp_grads is None:
                raise ValueError('a CarLPNet backward needs lp_grads')
            lg = lp_grads[0] if isinstance(lp_grads, (list, tuple)) else lp_grads
            P.dlp.copy_(lg.reshape(P.dlp.shape))
        self._backward(P, exchange=exchange, capture=capture)

    def step(self, batch_size=None):
        """`trainer.step(batch_size)` (car/YOLO.py:396): joins the gradient exchange, rescales by 1/batch_size and applies
        the MXNet Adam update on every rank.  batch_size=None: the SUM of the ranks' shard sizes of the last backward,
        taken from the slot the exchange itself reduced (no collective of its own, no host read)."""
        L.require_current_device(self.dev, 'this Trainer')
        lib, st = self.lib, L.stream_ptr()
        self.buckets.wait()                                    # KVStore sum-reduce of trainer.step (RCCL), bucketed
        self.t += 1
        n = self.nparam
        if batch_size is None:
            if self.buckets.active():
                L.check(lib.yolo_adam_step_dev(L.ptr(self.wflat), L.ptr(self.gflat), L.ptr(self.mflat), L.ptr(self.vflat), n,
                                               self.t, self.lr, self.b1, self.b2, self.eps, L.ptr(self._gb_slot), st), 'adam')
                batch_size = 0
            else:
                batch_size = self._fwd_B
        if batch_size:
            L.check(lib.yolo_adam_step(L.ptr(self.wflat), L.ptr(self.gflat), L.ptr(self.mflat), L.ptr(self.vflat), n,
                                       self.t, self.lr, self.b1, self.b2, self.eps, 1.0 / batch_size, st), 'adam')
        self._repack()
        self.net._version += 1
        self._packed_version = self.net._version

    # ---- one training step -----------------------------------------------------------------------------------
    def train_step(self, images, labels, global_batch=None, update=True, lp_labels=None, capture=None):
        """_train_batch (car/YOLO.py:350-399) with the reference's own targets and losses on the device.
        images (B,3,H,W) float32 CUDA; labels (B,nobj,6+ncls) float32 CUDA [cls,y,x,h,w,rot,dist...],
        cls < 0 = no object.  Returns losses (5,B) [score, box_yx, box_hw, rotate, class] (device)."""
        lib, st = self.lib, L.stream_ptr()
        labels = labels.to(self.dev, torch.float32).contiguous()
        C_ = self.net.graph.per_anchor
        if labels.dim() != 3 or labels.shape[0] != images.shape[0] or labels.shape[1] < 1 or labels.shape[2] != C_:
            # (the kernels take the row width from the net: a label tensor of another width would be read with the wrong stride)
            raise ValueError('labels must be (B=%d, nobj >= 1, %d) [cls, y, x, h, w, rot, class distribution...], got %s'
                             % (images.shape[0], C_, tuple(labels.shape)))
        if self.net.graph.lp_out is not None and lp_labels is not None:
            if lp_labels.dim() != 3 or lp_labels.shape[0] != images.shape[0] or lp_labels.shape[1] < 1:
                raise ValueError('lp_labels must be (B=%d, nobj >= 1, 10), got %s' % (images.shape[0], tuple(lp_labels.shape)))
        self.forward(images)
        P = self._plans[self._fwd_B]
        B, _, H, W = images.shape
        nobj, ncls = labels.shape[1], labels.shape[2] - 6
        rec = torch.empty((B, nobj, 7 + ncls), dtype=torch.float32, device=self.dev)
        L.check(lib.yolo_assign_targets(L.ptr(labels), L.ptr(self.anchors_ltrb), L.ptr(rec), B, nobj, ncls,
                                        C.byref(self.grid), st), 'assign')
        losses = torch.empty((5, B), dtype=torch.float32, device=self.dev)
        sc = self.scale
        s5 = (C.c_float * 5)(sc['score'], sc['box_yx'], sc['box_hw'], sc['rotate'] if self.car_rotate else 0.0, sc['class'])
        L.check(lib.yolo_loss_fwd_bwd(L.ptr(P.merged), L.ptr(rec), L.ptr(P.dmerged), L.ptr(losses), B, self.nbox, C_, nobj,
                                      s5, self.pos_w, self.neg_w, st), 'loss')
        self._last = (P, rec)
        if self.net.graph.lp_out is not None:
            # the five LP losses of CarLPNet's _train_batch (car_and_LP/YOLO.py:262-300, LP_detection.py:258-360)
            if lp_labels is None:
                raise ValueError('a CarLPNet step needs lp_labels (B, nobj, 10): rows of -1 = no plate')
            g = self.net.graph
            lp_labels = lp_labels.to(self.dev, torch.float32).contiguous()
            nlp, lw = lp_labels.shape[1], lp_labels.shape[2]
            LPC = g.lp_out.cout
            ncl = LPC - 7
            lrec = torch.empty((B, nlp, 8 + ncl), dtype=torch.float32, device=self.dev)
            step = g.steps()[0]
            L.check(lib.yolo_assign_targets_lp(L.ptr(lp_labels), L.ptr(lrec), B, nlp, lw, ncl, H, W, step,
                                               self.lp_r_max[0], self.lp_r_max[1], self.lp_r_max[2], st), 'assign lp')
            lp_losses = torch.empty((5, B), dtype=torch.float32, device=self.dev)
            ls = self.lp_scale
            l5 = (C.c_float * 5)(ls['LP_score'], ls['LP_xy'], ls['LP_z'], ls['LP_r'], ls['LP_class'])
            L.check(lib.yolo_loss_lp_fwd_bwd(L.ptr(P.lp), L.ptr(lrec), L.ptr(P.dlp), L.ptr(lp_losses), B, P.lp.shape[1],
                                             LPC, nlp, l5, self.lp_pos_w, self.lp_neg_w, st), 'lp loss')
            losses = torch.cat([losses, lp_losses], dim=0)
            self._last = (P, rec, lrec)
        self._backward(P, exchange=update, capture=capture)
        if update:
            self.step(global_batch)
        return losses

    def grads(self):
        return self.gview

    # ---- measured kernel choices as a value (N > 1: rank 0 measures, every rank runs rank 0's plan) -----------------
    def tuning_state(self):
        return {'algo': dict(self.net._algo_cache), 'dgrad': dict(self._dgrad_algo), 'wgrad': dict(self._wgrad_algo)}

    def load_tuning_state(self, state):
        self.net._algo_cache.update(state['algo'])
        self._dgrad_algo.update(state['dgrad'])
        self._wgrad_algo.update(state['wgrad'])
        return self

    def tune(self, images, labels, lp_labels=None):
        """One LOCAL step (no exchange, no update) whose only lasting effect is the measured kernel choices of every layer
        shape of this batch -- forward variants, data- and weight-gradient variants: what rank 0 runs before
        parallel.share_tuning() hands its choices to the other ranks.  The BatchNorm running statistics the forward
        moved are put back."""
        keep = {n: t.clone() for n, t in self.net.params.items() if n.endswith(('.running_mean', '.running_var'))}
        self.train_step(images, labels, update=False, lp_labels=lp_labels)
        for n, t in keep.items():
            self.net.params[n].copy_(t)
        self.net._version += 1
        torch.cuda.synchronize()
        return self.tuning_state()

    def merged_logits(self):
        P = self._last[0]
        return P.merged.view(P.merged.shape[0], P.tot, P.A, self.net.graph.per_anchor)
