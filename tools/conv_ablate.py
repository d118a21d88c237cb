#!/usr/bin/env python3
"""Timing ablations of k_igemm on GEMM-shaped 1x1 problems: occupancy (blocks per CU) and
which part of the loop costs what (flags 0x100 no DMA, 0x200 no fragment reads, 0x400 no barrier)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
eng = ffrnet_amd.Engine(0); eng.reserve(64)
def run(M, cout, K, tile, flags, reps=5):
    x = torch.randn(1, 1, M, K, device='cuda')        # N=1,H=1,W=M: 1x1 conv = plain GEMM
    w = torch.randn(cout, K, device='cuda') * 0.05
    bias = torch.zeros(cout, device='cuda'); out = torch.empty(M, cout, device='cuda')
    kw = dict(x=x, N=1, H=1, W=M, in_pitch=K, cin_pad=K, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
              out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=1, S=1, stride=1, pad=0, pad_mode=0,
              border_bias=0, flags=flags, tile=tile, splitk=1)
    for _ in range(2): eng.op_conv(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.op_conv(**kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * cout * K / ms / 1e9
K = 2304
for tile, bm, bn in ((1, 128, 128), (2, 128, 64), (3, 64, 64)):
    for per_cu in (1, 2, 3, 4, 8):
        M = 256 * per_cu * bm
        row = []
        for flags in (0, 0x100, 0x200, 0x400, 0x700):
            ms, tf = run(M, bn, K, tile, flags)
            row.append('f%03x %6.3fms %6.1fTF' % (flags, ms, tf))
        print('tile %dx%d blocks/CU=%d  ' % (bm, bn, per_cu) + '  '.join(row), flush=True)
