import sys; import os; R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import numpy as np, torch
from yolo_amd import lib as L
from util import run_conv
lib = L.load(); dev = torch.device('cuda:0')
for val in (2.0 ** -20, 2.0 ** -16, 2.0 ** -13):
    x = np.full((1, 64, 8, 8), val, np.float32); w = np.ones((32, 64, 1, 1), np.float32)
    y = run_conv(lib, dev, x, w, np.ones(32, np.float32), np.zeros(32, np.float32), 1, 1.0, 'f16', out_f32=True)
    print('x = 2^%d (f16 %s): y = %.6g, expected %.6g' % (np.log2(val), 'subnormal' if val < 2.0 ** -14 else 'normal', float(y[0, 0, 0, 0]), 64 * val))
