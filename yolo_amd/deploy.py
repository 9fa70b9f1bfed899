"""Deployment seam: the frozen-graph runner the reference builds with
`yolo_gluon.init_executor` (yolo_gluon.py:204-242: load an exported checkpoint, bind it for inference, optionally in
half precision / through TensorRT) and the `/YOLO/box` row the video node publishes (car/video_node.py:235-255).

On MI355X the "engine" is a `CarNet` with BatchNorm folded into per-channel scale/bias and the weights packed for
the MFMA kernels (`CarNet.prepare`), running bf16 activations: there is no separate runtime to build.  Both halves of
`net.export` (yolo_gluon.py:245-272) are written and read: `export-NNNN.params` (the gluon parameter names prefixed with
`arg:` / `aux:`, matched by name, yolo_amd/mxparams.py) and `export-symbol.json` -- the reference's executor is built from
that file alone (`mxnet.model.load_checkpoint`, yolo_gluon.py:206-208: no spec.yaml), so `init_executor(folder, None,
size)` recovers the network structure from the symbol graph (`spec_from_symbol`) and, when a spec is given as well,
refuses a checkpoint whose graph disagrees with it.  The JSON layout is MXNet 1.x's `Symbol.tojson()` as recalled
(nodes / arg_nodes / node_row_ptr / heads; `attrs`, or `attr` / `param` in older files) -- parity unpinned like the rest.
"""
import json
import math
import os

import numpy as np

from . import mxparams


class Executor(object):
    """What `init_executor` returns, with the call signature the video node uses:
    `net_out = executor.forward(is_train=False, data=nd_img)` (car/video_node.py:230)."""

    def __init__(self, net):
        self.net = net
        self.outputs = None

    def forward(self, is_train=False, data=None):
        if is_train:
            raise ValueError('the deployment executor is inference-only (grad_req="null", yolo_gluon.py:236)')
        out = self.net(data)
        self.outputs = out if isinstance(out, list) else list(out[0]) + list(out[1])
        return self.outputs


def export_params(net, export_folder, epoch=0, prefix='carnet0_'):
    """The parameter half of yolo_gluon.export: <folder>/export-%04d.params, what HybridBlock.export writes --
    collect_params() with every name prefixed by `arg:` (or `aux:` for the running statistics)."""
    os.makedirs(export_folder, exist_ok=True)
    path = os.path.join(export_folder, 'export-%04d.params' % epoch)
    mxparams.write_params(path, mxparams.to_gluon(net.graph, net.collect_params(), prefix, export=True))
    return path


# ---- export-symbol.json -------------------------------------------------------------------------------------------
def _conv2d_nodes(nodes, c, names, src):
    """Append the nodes of one gluoncv _conv2d (Convolution [+ BatchNorm + LeakyReLU]) reading node `src`; returns the
    output node id.  Variable nodes carry the gluon parameter names."""
    def var(name):
        nodes.append({'op': 'null', 'name': name, 'inputs': []})
        return len(nodes) - 1
    w = var(names[c.name + '.weight'])
    ins = [[src, 0, 0], [w, 0, 0]]
    if not c.bn:
        ins.append([var(names[c.name + '.bias']), 0, 0])
    scope = names[c.name + '.weight'][:-len('weight')]
    nodes.append({'op': 'Convolution', 'name': scope + 'fwd',
                  'attrs': {'dilate': '(1, 1)', 'kernel': '(%d, %d)' % (c.k, c.k), 'layout': 'NCHW', 'no_bias': str(bool(c.bn)),
                            'num_filter': str(c.cout), 'num_group': '1', 'pad': '(%d, %d)' % (c.k // 2, c.k // 2),
                            'stride': '(%d, %d)' % (c.stride, c.stride)}, 'inputs': ins})
    out = len(nodes) - 1
    if c.bn:
        bscope = names[c.name + '.gamma'][:-len('gamma')]
        ids = [var(names['%s.%s' % (c.name, s)]) for s in ('gamma', 'beta', 'running_mean', 'running_var')]
        nodes.append({'op': 'BatchNorm', 'name': bscope + 'fwd',
                      'attrs': {'axis': '1', 'eps': '1e-05', 'fix_gamma': 'False', 'momentum': '0.9', 'use_global_stats': 'False'},
                      'inputs': [[out, 0, 0]] + [[i, 0, 0] for i in ids]})
        nodes.append({'op': 'LeakyReLU', 'name': bscope.replace('batchnorm', 'leakyrelu') + 'fwd',
                      'attrs': {'act_type': 'leaky', 'slope': '0.1'}, 'inputs': [[len(nodes) - 1, 0, 0]]})
        out = len(nodes) - 1
    return out


def symbol_json(graph, prefix='carnet0_'):
    """The symbol graph HybridBlock.export writes for CarNet.hybrid_forward (car/utils.py:68-95) -- or, for a graph with
    the licence-plate branch, CarLPNet.hybrid_forward (car_and_LP/YOLO.py:62-95; exported at car_and_LP/YOLO.py:386):
    five detection blocks chained through their tips on the input of the finest car block, a biased 1x1 and a transpose
    to NHWC, listed as one more head behind all_output[::-1] -- as a dict."""
    names = mxparams.gluon_param_names(graph, prefix)
    nodes = [{'op': 'null', 'name': 'data', 'inputs': []}]
    x = _conv2d_nodes(nodes, graph.stem, names, 0)
    routes, nadd = [], 0
    for down, res in graph.stages:
        x = _conv2d_nodes(nodes, down, names, x)
        for c1, c2 in res:
            y = _conv2d_nodes(nodes, c2, names, _conv2d_nodes(nodes, c1, names, x))
            nodes.append({'op': 'elemwise_add', 'name': '%s_plus%d' % (prefix.rstrip('_'), nadd), 'inputs': [[x, 0, 0], [y, 0, 0]]})
            nadd += 1
            x = len(nodes) - 1
        routes.append(x)
    routes = routes[-graph.num_pyramid:][::-1]
    heads, lp_head = [], None
    for i, (body, tip, outc, nA) in enumerate(graph.heads):
        if getattr(graph, 'lp_out', None) is not None and i >= len(graph.heads) - 1:
            t = x
            for lbody, ltip in graph.lp_blocks:                   # `_, LP_output = self.LP_branch[k](...)`: the tip output
                for c in list(lbody) + [ltip]:
                    t = _conv2d_nodes(nodes, c, names, t)
            t = _conv2d_nodes(nodes, graph.lp_out, names, t)
            nodes.append({'op': 'transpose', 'name': 'transpose%d' % len(graph.heads), 'attrs': {'axes': '(0, 2, 3, 1)'},
                          'inputs': [[t, 0, 0]]})
            lp_head = len(nodes) - 1
        for c in body:
            x = _conv2d_nodes(nodes, c, names, x)
        route = x
        o = _conv2d_nodes(nodes, outc, names, _conv2d_nodes(nodes, tip, names, route))
        nodes.append({'op': 'transpose', 'name': 'transpose%d' % i, 'attrs': {'axes': '(0, 2, 3, 1)'}, 'inputs': [[o, 0, 0]]})
        nodes.append({'op': 'Reshape', 'name': 'reshape%d' % i, 'attrs': {'shape': '(0, -1, %d, %d)' % (nA, graph.per_anchor)},
                      'inputs': [[len(nodes) - 1, 0, 0]]})
        heads.append(len(nodes) - 1)
        if i < len(graph.transitions):
            x = _conv2d_nodes(nodes, graph.transitions[i], names, route)
            for ax in (-1, -2):                                   # gluoncv _upsample: repeat along W, then along H
                nodes.append({'op': 'repeat', 'name': 'repeat%d' % (2 * i + (ax == -2)), 'attrs': {'axis': str(ax), 'repeats': '2'},
                              'inputs': [[x, 0, 0]]})
                x = len(nodes) - 1
            nodes.append({'op': 'Concat', 'name': 'concat%d' % i, 'attrs': {'dim': '1', 'num_args': '2'},
                          'inputs': [[x, 0, 0], [routes[i + 1], 0, 0]]})
            x = len(nodes) - 1
    return {'nodes': nodes, 'arg_nodes': [i for i, n in enumerate(nodes) if n['op'] == 'null'],
            'node_row_ptr': list(range(len(nodes) + 1)),
            'heads': [[h, 0, 0] for h in heads[::-1] + ([lp_head] if lp_head is not None else [])],     # all_output[::-1] (+ [LP_output])
            'attrs': {'mxnet_version': ['int', 10301]}}


def _attrs(node):
    return node.get('attrs') or node.get('attr') or node.get('param') or {}


def _tuple(text):
    return tuple(int(v) for v in text.strip('()[] ').replace(' ', '').split(',') if v)


def spec_from_symbol(sym):
    """Network structure from an exported symbol graph (a dict, or the path of export-symbol.json): the `layers` /
    `channels` of the spec, the number of pyramid scales, anchors per scale and values per anchor.  Anchor SIZES and the
    inner slice points are not part of the graph (the drivers read them from spec.yaml for the decode): `all_anchors`
    comes back as ones and `slice_point` as [per_anchor]; a head that is transposed but not reshaped is CarLPNet's
    LP_output (car_and_LP/YOLO.py:79) and gives `LP_slice_point` = [its channels].  The recovered graph is re-built and compared conv by conv
    with the file; anything that is not the CarNet topology raises ValueError."""
    if not isinstance(sym, dict):
        with open(sym) as f:
            sym = json.load(f)
    nodes = sym['nodes']
    chan, lvl, convs, adds = {}, {}, [], {}
    has_bn = set()
    for i, n in enumerate(nodes):
        op, ins = n['op'], [e[0] for e in n.get('inputs', [])]
        if op == 'null':
            if not ins and n['name'] == 'data':
                chan[i], lvl[i] = 3, 0
            continue
        a = _attrs(n)
        src = ins[0] if ins else None
        if src is not None and src not in chan:
            raise ValueError('node %s (%s) reads a node of unknown kind' % (n['name'], op))
        if op == 'Convolution':
            k, st = _tuple(a['kernel']), _tuple(a.get('stride', '(1, 1)'))
            if k[0] != k[1] or st[0] != st[1] or _tuple(a.get('pad', '(0, 0)'))[0] != k[0] // 2 or int(a.get('num_group', 1)) != 1:
                raise ValueError('convolution %s is not a square same-padded one' % n['name'])
            chan[i], lvl[i] = int(a['num_filter']), lvl[src] + (1 if st[0] == 2 else 0)
            convs.append(dict(id=i, cin=chan[src], cout=chan[i], k=k[0], stride=st[0],
                              bias=str(a.get('no_bias', 'False')).lower() not in ('true', '1')))
        elif op in ('BatchNorm', 'LeakyReLU', 'Activation', 'transpose', 'Reshape', 'reshape', 'identity', 'Cast', 'cast'):
            chan[i], lvl[i] = chan[src], lvl[src]
            if op == 'BatchNorm':
                has_bn.add(src)
        elif op in ('elemwise_add', '_Plus', '_plus', 'broadcast_add'):
            if chan[ins[0]] != chan[ins[1]] or lvl[ins[0]] != lvl[ins[1]]:
                raise ValueError('residual add %s joins tensors of different shapes' % n['name'])
            chan[i], lvl[i] = chan[src], lvl[src]
            adds[lvl[i]] = adds.get(lvl[i], 0) + 1
        elif op == 'repeat':
            if int(a.get('repeats', 1)) != 2:
                raise ValueError('repeat %s is not the 2x up-sampling' % n['name'])
            chan[i], lvl[i] = chan[src], lvl[src] - 0.5
        elif op == 'UpSampling':
            chan[i], lvl[i] = chan[src], lvl[src] - 1
        elif op in ('Concat', 'concat'):
            if len({lvl[j] for j in ins}) != 1:
                raise ValueError('concat %s joins maps of different resolution' % n['name'])
            chan[i], lvl[i] = sum(chan[j] for j in ins), lvl[src]
        else:
            raise ValueError('operator %s (%s) is not part of the CarNet graph' % (op, n['name']))
    if not convs:
        raise ValueError('no convolution in the symbol')
    for c in convs:
        c['bn'] = c['id'] in has_bn
    downs = [c for c in convs if c['stride'] == 2]
    channels = [convs[0]['cout']] + [c['cout'] for c in downs]
    layers = [adds.get(l + 1, 0) for l in range(len(downs))]
    per_scale, lp_c = [], None
    for h in sym['heads']:
        j = h[0]
        shape, transposed = None, False
        while nodes[j]['op'] != 'Convolution':
            if nodes[j]['op'] in ('Reshape', 'reshape'):
                shape = _tuple(_attrs(nodes[j])['shape'].replace('-1', '0'))
            transposed = transposed or nodes[j]['op'] == 'transpose'
            j = nodes[j]['inputs'][0][0]
        if shape is None and transposed and lp_c is None and h is sym['heads'][-1]:
            lp_c = chan[j]                                        # CarLPNet's [LP_output]: NHWC, no anchor axis
            continue
        if shape is None or len(shape) != 4:
            raise ValueError('output %s is not reshaped to (B, HW, A, C)' % nodes[h[0]]['name'])
        per_scale.append((shape[2], shape[3]))
    if len({c for _, c in per_scale}) != 1:
        raise ValueError('the scales disagree on the values per anchor')
    spec = {'layers': layers, 'channels': channels, 'slice_point': [per_scale[0][1]],
            'all_anchors': [[[1.0, 1.0]] * a for a, _ in per_scale]}
    if lp_c is not None:
        spec['LP_slice_point'] = [lp_c]
    from .spec import NetGraph
    g = NetGraph(spec)
    want = [(c.cin, c.cout, c.k, c.stride, bool(c.bn)) for c in mxparams.gluon_conv_order(g, 'forward')]
    got = [(c['cin'], c['cout'], c['k'], c['stride'], c['bn']) for c in convs]
    if sorted(want) != sorted(got) or want[:1 + sum(1 + 2 * n for n in layers)] != got[:1 + sum(1 + 2 * n for n in layers)]:
        raise ValueError('the symbol is not a CarNet: %d convolutions, the recovered spec %s builds %d' % (len(got), spec, len(want)))
    return spec


def export(net, export_folder, epoch=0, prefix='carnet0_'):
    """Both files of yolo_gluon.export / HybridBlock.export: <folder>/export-symbol.json and <folder>/export-%04d.params."""
    os.makedirs(export_folder, exist_ok=True)
    with open(os.path.join(export_folder, 'export-symbol.json'), 'w') as f:
        json.dump(symbol_json(net.graph, prefix), f, indent=2)
    return export_params(net, export_folder, epoch, prefix)


def init_executor(export_folder, spec, size, device='cuda:0', step=0, dtype='bf16', tune='auto'):
    """yolo_gluon.init_executor(export_folder, size, ctx, use_tensor_rt, step, fp16) for MI355X: load
    <folder>/export-symbol.json + export-%04d.params, fold BN, pack the weights, pre-build the launch plan for a
    (1,3,H,W) input.  spec = None: the structure comes from the symbol file alone, as in the reference (the anchors it
    carries are placeholders: decode with the driver's own spec.yaml); a spec AND a symbol file: they must agree."""
    import torch
    from .net import CarNet, CarLPNet
    sym_path = os.path.join(export_folder, 'export-symbol.json')
    if spec is None:
        spec = spec_from_symbol(sym_path)
    elif os.path.exists(sym_path):
        got = spec_from_symbol(sym_path)
        key = lambda sp: (list(sp['layers']), list(sp['channels']), sp['slice_point'][-1], [len(a) for a in sp['all_anchors']],
                          sp['LP_slice_point'][-1] if 'LP_slice_point' in sp else None)
        if key(got) != key(spec):
            raise ValueError('export-symbol.json describes another network than the spec: %s' % got)
    cls = CarLPNet if 'LP_slice_point' in spec else CarNet
    net = cls(spec, dtype=dtype, device=device, tune=tune)
    net.load_gluon_params(os.path.join(export_folder, 'export-%04d.params' % step))
    net.prepare()
    net(torch.zeros((1, 3, int(size[0]), int(size[1])), dtype=torch.float32, device=device))    # bind: build the plan
    return Executor(net)


_STEP = 360 // 24
_COS_OFFSET = np.array([math.cos(x * math.pi / 180) for x in range(0, 360, _STEP)])       # car/video_node.py:36-38
_SIN_OFFSET = np.array([math.sin(x * math.pi / 180) for x in range(0, 360, _STEP)])


def car_box_row(pred_car, depth_image=None):
    """Video.process (car/video_node.py:235-255): the row published on /YOLO/box from predict()'s (1, 6+24) output
    [score, y, x, h, w, rot, cls...]: element 5 becomes the azimuth = atan2 of the softmax-weighted mean direction of
    the 24 azimuth classes (15 degrees apart).  Returns a copy of the row."""
    row = np.array(pred_car[0], copy=True)
    x = row[-24:]
    prob = np.exp(x) / np.sum(np.exp(x), axis=0)
    c = sum(_COS_OFFSET * prob)
    s = sum(_SIN_OFFSET * prob)
    row[5] = math.atan2(s, c)
    return row
