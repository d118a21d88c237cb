#!/usr/bin/env python3
"""python tools/trace_igemm.py : (diagnostics build, option igemm_trace) per-segment clock breakdown of k_igemm on a few shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import trace_build
trace_build.use()
import ffrnet_amd
eng = ffrnet_amd.Engine(0); eng.set_option('igemm_trace', 1); eng.reserve(256)
def conv(name, N, H, W, cin, cout, R, stride, tile, flags=0):
    pad = R // 2
    x = torch.randn(N, H, W, cin, device='cuda'); w = torch.randn(cout, R * R * cin, device='cuda') * 0.05
    Ho = (H + 2 * pad - R) // stride + 1
    bias = torch.zeros(cout, device='cuda'); out = torch.empty(N, Ho, Ho, cout, device='cuda')
    print(name, 'tile', tile, 'flags', flags, file=sys.stderr)
    for _ in range(2):
        eng.op_conv(x=x, N=N, H=H, W=W, in_pitch=cin, cin_pad=cin, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
                    out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=R, S=R, stride=stride, pad=pad, pad_mode=0,
                    border_bias=0, flags=flags, tile=tile, splitk=1)
conv('s1 64@56', 256, 56, 56, 64, 64, 3, 1, 2)
conv('s1 64@56', 256, 56, 56, 64, 64, 3, 1, 2, 0x100)
conv('s2 128@32 (no cuts)', 256, 32, 32, 128, 128, 3, 1, 1)
conv('s2 128@32 (no cuts)', 256, 32, 32, 128, 128, 3, 1, 1, 0x100)
conv('s3 256@14', 256, 14, 14, 256, 256, 3, 1, 1)
conv('s3 256@14', 256, 14, 14, 256, 256, 3, 1, 1, 0x100)
