#!/usr/bin/env python3
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).
env: SHAPE="H,W,cin,cout,R,stride,pad,mode" TILE=t N=256 REPS=3"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402

H, W, cin, cout, R, stride, pad, mode = [int(v) for v in os.environ.get('SHAPE', '14,14,256,256,3,1,1,0').split(',')]
N = int(os.environ.get('N', '256'))
tiles = [int(t) for t in os.environ.get('TILE', '1,2,3').split(',')]
reps = int(os.environ.get('REPS', '3'))
eng = ffrnet_amd.Engine(0)
eng.reserve(N)
x = torch.randn(N, H, W, cin, device='cuda')
w = torch.randn(cout, R * R * cin, device='cuda') * 0.05
bias = torch.randn(cout, device='cuda')
Ho = (H + 2 * pad - R) // stride + 1
Wo = (W + 2 * pad - R) // stride + 1
out = torch.empty(N, Ho, Wo, cout, device='cuda')
for t in tiles:
    for _ in range(reps):
        eng.op_conv(x=x, N=N, H=H, W=W, in_pitch=cin, cin_pad=cin, w=w, bias=bias, slope=None, resid=None,
                    res_pitch=0, out=out, out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=R, S=R,
                    stride=stride, pad=pad, pad_mode=mode, border_bias=0, flags=0, tile=t, splitk=1)
torch.cuda.synchronize()
print('done')
