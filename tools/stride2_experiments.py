#!/usr/bin/env python3
"""The four stride-2 3x3 convolutions of the trunk (model_ir_se50.py:69 with stride 2) at batch 256, measured:
  (a) the direct implicit GEMM (k_igemm) with every tile shape that divides cout -- is 256x64 (one block per CU, a B tile
      reused by 256 rows) faster than 128x64 for the 64 -> 64 layer at 112 -> 56?
  (b) a LOWER BOUND for the polyphase Winograd form F(4x4, 2x2) with 25 xi and K = 4 cin: the 25 batched
      GEMMs [T x 4cin] * [4cin x cout] run here as ONE GEMM of 25 T rows through the same kernel (same FLOPs, same tile
      count; U is shared, which only helps), plus the bytes its V and M would move through HBM (V = 25 T 4cin floats
      written and read once, M = 25 T cout) priced at 5 TB/s, the rate the transform kernels of this library reach.
      The real form would add the transforms' own arithmetic on top.
usage: python tools/stride2_experiments.py [N]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = ffrnet_amd.Engine(0)
eng.reserve(N)


def timed(kw, reps=5):
    for _ in range(2):
        eng.op_conv(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.op_conv(**kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


for H, C in ((112, 64), (56, 128), (28, 256), (14, 512)):
    Ho = H // 2
    x = torch.randn(N, H, H, C, device='cuda')
    w = torch.randn(C, 9 * C, device='cuda') * 0.05
    bias = torch.randn(C, device='cuda')
    out = torch.empty(N, Ho, Ho, C, device='cuda')
    gflop = 2.0 * N * Ho * Ho * C * 9 * C / 1e9
    line = ['%3dx%-3d %3d->%-3d direct %.1f GFLOP:' % (H, H, C, C, gflop)]
    best = 1e30
    for tile in (0, 1, 2, 3, 4):
        bn = {0: 64, 1: 128, 2: 64, 3: 64, 4: 64}[tile]
        if C % bn:
            continue
        kw = dict(x=x, N=N, H=H, W=H, in_pitch=C, cin_pad=C, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
                  out_pitch=C, out_coff=0, cout_store=C, cout_pad=C, R=3, S=3, stride=2, pad=1, pad_mode=0, border_bias=0,
                  flags=0, tile=tile, splitk=1)
        us = timed(kw)
        best = min(best, us)
        line.append('tile %d %.0f us (%.1f TF)' % (tile, us, gflop / us * 1e3))
    print(' '.join(line), flush=True)
    del x, out
    # (b) polyphase F(4x4, 2x2): 25 xi, K = 4 cin, T = N * ceil(Ho / 4)^2 tiles
    T = N * math.ceil(Ho / 4) ** 2
    K = 4 * C
    rows = 25 * T
    xg = torch.randn(1, 1, rows, K, device='cuda')
    wg = torch.randn(C, K, device='cuda') * 0.05
    og = torch.empty(rows, C, device='cuda')
    bg = torch.zeros(C, device='cuda')
    gexec = 2.0 * rows * K * C / 1e9
    res = []
    for tile in ((1, 2) if C % 128 == 0 else (2, 4)):
        kw = dict(x=xg, N=1, H=1, W=rows, in_pitch=K, cin_pad=K, w=wg, bias=bg, slope=None, resid=None, res_pitch=0, out=og,
                  out_pitch=C, out_coff=0, cout_store=C, cout_pad=C, R=1, S=1, stride=1, pad=0, pad_mode=0, border_bias=0,
                  flags=0, tile=tile, splitk=1)
        res.append((timed(kw), tile))
    us, tile = min(res)
    v_bytes = 4.0 * rows * K
    m_bytes = 4.0 * rows * C
    io_us = (2 * v_bytes + 2 * m_bytes) / 5e12 * 1e6
    print('          polyphase F(4x4,2x2) lower bound: 25 GEMMs [%d x %d] * [%d x %d] = %.1f GFLOP executed (%.2f of direct): '
          '%.0f us (tile %d, %.1f TF) + V/M round trips %.2f GB at 5 TB/s = %.0f us  => >= %.0f us vs direct %.0f us'
          % (T, K, K, C, gexec, gexec / gflop, us, tile, gexec / us * 1e3, (2 * v_bytes + 2 * m_bytes) / 1e9, io_us,
             us + io_us, best), flush=True)
    del xg, og
