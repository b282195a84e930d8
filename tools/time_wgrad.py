"""d/d weight of decnn.7 / decnn.4 alone (gpode_conv2d_bwd_weight[_bn], kernel + reduction of the partials) with HIP events.
GPODE_WGRAD_V1=1 selects the first engine.  usage: python tools/time_wgrad.py [images]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_gp_ode_amd import _lib
from vae_gp_ode_amd.ops import _ptr, _stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
#        name     Cin Cout Hi  Ht  K  S  P   MMAC per image
LAYERS = [('decnn.7', 32, 16, 13, 28, 5, 2, 1, 2.163e6), ('decnn.4', 64, 32, 6, 13, 5, 2, 1, 1.843e6)]
lib = _lib.load()
for name, Cin, Cout, Hi, Ht, K, S, P, macs in LAYERS:
    c = torch.randn(B, Cin, Hi, Hi, device='cuda')
    gy = torch.randn(B, Cout, Ht, Ht, device='cuda')
    table = torch.rand(Cin, 4, device='cuda') + 0.5
    gw = torch.empty(Cin, Cout, K, K, device='cuda')
    ws = torch.empty(max(int(lib.gpode_conv_wgrad_scratch(B, Cout, Cin, K)), 4), device='cuda')
    for bn in (False, True):
        def run():
            if bn:
                _lib.call('gpode_conv2d_bwd_weight_bn', _ptr(gy), _ptr(c), _ptr(table), _ptr(gw), _ptr(None), _ptr(ws), B, Cout, Ht, Ht, Cin, K, S, P, Hi, Hi, _stream())
            else:
                _lib.call('gpode_conv2d_bwd_weight', _ptr(gy), _ptr(c), _ptr(gw), _ptr(None), _ptr(ws), B, Cout, Ht, Ht, Cin, K, S, P, Hi, Hi, _stream())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%s d/d weight%s  %d images  %.1f us  %.1f TFLOP/s  frac %.2f   (%s engine)' % (
            name, ' + BN/ReLU input' if bn else '', B, ms * 1e3, 2 * macs * B / ms / 1e9, 2 * macs * B / ms / 1e9 / 157.3,
            'first' if os.environ.get('GPODE_WGRAD_V1') == '1' else 'second'))
