#!/usr/bin/env python3
"""What-if timings of k_igemm: the same M,N,K as plain GEMM (1x1) vs 3x3 conv, several M."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
eng = ffrnet_amd.Engine(0); eng.reserve(64)
def run(N, H, W, cin, cout, R, tile, reps=5, mode=0):
    pad = R // 2
    x = torch.randn(N, H, W, cin, device='cuda'); w = torch.randn(cout, R * R * cin, device='cuda') * 0.05
    bias = torch.zeros(cout, device='cuda'); out = torch.empty(N, H, W, cout, device='cuda')
    kw = dict(x=x, N=N, H=H, W=W, in_pitch=cin, cin_pad=cin, w=w, bias=bias, slope=None, resid=None, res_pitch=0, out=out,
              out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=R, S=R, stride=1, pad=pad, pad_mode=mode,
              border_bias=0, flags=0, tile=tile, splitk=1)
    for _ in range(2): eng.op_conv(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.op_conv(**kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * N * H * W * cout * R * R * cin / ms / 1e9
for tile in (1, 2, 3):
    print('tile', tile)
    for name, args in (('conv3x3 256@14 N=256', (256, 14, 14, 256, 256, 3)),
                       ('gemm    M=50176 N=256 K=2304', (256, 14, 14, 2304, 256, 1)),
                       ('gemm    M=50176 N=128 K=2304', (256, 14, 14, 2304, 128, 1)),
                       ('gemm    M=262144 N=128 K=2304', (256, 32, 32, 2304, 128, 1)),
                       ('gemm    M=65536 N=512 K=4608', (256, 16, 16, 4608, 512, 1)),
                       ('conv3x3 512@16 N=256', (256, 16, 16, 512, 512, 3)),
                       ('conv3x3 256@16 N=256 (M=65536)', (256, 16, 16, 256, 256, 3)),
                       ('conv3x3 reflect 256@16', (256, 16, 16, 256, 256, 3, 1)),
                       ):
        mode = args[6] if len(args) > 6 else 0
        ms, tf = run(*args[:6], tile, mode=mode)
        print('  %-34s %7.3f ms %6.1f TF' % (name, ms, tf), flush=True)
