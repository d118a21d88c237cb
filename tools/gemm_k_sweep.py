#!/usr/bin/env python3
"""1x1-conv GEMMs [M x K] * [K x N] at fixed M, N and growing K: time = tiles * (K-tiles * t_k + t_fixed)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
eng = ffrnet_amd.Engine(0); eng.reserve(64)
def run(M, N, K, tile, reps=5):
    x = torch.randn(1, 1, M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05
    bias = torch.zeros(N, device='cuda'); out = torch.empty(M, N, device='cuda')
    kw = dict(x=x, N=1, H=1, W=M, in_pitch=K, cin_pad=K, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
              out_pitch=N, out_coff=0, cout_store=N, cout_pad=N, R=1, S=1, stride=1, pad=0, pad_mode=0,
              border_bias=0, flags=0, tile=tile, splitk=1)
    for _ in range(2): eng.op_conv(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.op_conv(**kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9
M, N = 36 * 4096, 256
for tile in (2, 1):
    for K in (64, 128, 256, 512, 1024, 2048):
        ms, tf = run(M, N, K, tile)
        print('tile %d  M=%d N=%d K=%-5d nkt=%-3d  %7.3f ms  %6.1f TF' % (tile, M, N, K, K // 32, ms, tf), flush=True)
