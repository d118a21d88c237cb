#!/usr/bin/env python3
"""Per-shape timing of the implicit-GEMM conv kernel over every conv shape of the hot path
at batch 256 and every tile configuration (tuning harness; prints a table + JSON)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402

# name, count, H, W, cin, cout, R, stride, pad, mode
SHAPES = [
    ('s1.b0.c1 64@112', 1, 112, 112, 64, 64, 3, 1, 1, 0),
    ('s1.b0.c2 s2', 1, 112, 112, 64, 64, 3, 2, 1, 0),
    ('s1 64@56', 4, 56, 56, 64, 64, 3, 1, 1, 0),
    ('s2.b0.c1 64->128@56', 1, 56, 56, 64, 128, 3, 1, 1, 0),
    ('s2.b0.c2 s2', 1, 56, 56, 128, 128, 3, 2, 1, 0),
    ('s2 128@28', 6, 28, 28, 128, 128, 3, 1, 1, 0),
    ('s3.b0.c1 128->256@28', 1, 28, 28, 128, 256, 3, 1, 1, 0),
    ('s3.b0.c2 s2', 1, 28, 28, 256, 256, 3, 2, 1, 0),
    ('s3 256@14', 26, 14, 14, 256, 256, 3, 1, 1, 0),
    ('s4.b0.c1 256->512@14', 1, 14, 14, 256, 512, 3, 1, 1, 0),
    ('s4.b0.c2 s2', 1, 14, 14, 512, 512, 3, 2, 1, 0),
    ('s4 512@7', 4, 7, 7, 512, 512, 3, 1, 1, 0),
    ('sc 64->128', 1, 56, 56, 64, 128, 1, 2, 0, 0),
    ('sc 128->256', 1, 28, 28, 128, 256, 1, 2, 0, 0),
    ('sc 256->512', 1, 14, 14, 256, 512, 1, 2, 0, 0),
    ('rec 561->256', 1, 7, 7, 576, 256, 3, 1, 1, 1),
    ('rec 256->256', 2, 7, 7, 256, 256, 3, 1, 1, 1),
    ('rec 256->128', 1, 7, 7, 256, 128, 3, 1, 1, 1),
    ('rec 128->128', 2, 7, 7, 128, 128, 3, 1, 1, 1),
    ('rec 128->49', 1, 7, 7, 128, 64, 3, 1, 1, 1),
    ('rec 49->49', 2, 7, 7, 64, 64, 3, 1, 1, 1),
    ('rec 1024->512', 1, 7, 7, 1024, 512, 3, 1, 1, 1),
    ('rec 512->512', 4, 7, 7, 512, 512, 3, 1, 1, 1),
    ('rec 1536->512', 1, 7, 7, 1536, 512, 3, 1, 1, 1),
]


def main():
    N = int(os.environ.get('SWEEP_N', '256'))
    tiles = [int(t) for t in os.environ.get('SWEEP_TILES', '0,1,2,3,4').split(',')]
    eng = ffrnet_amd.Engine(0)
    eng.reserve(N)
    res = []
    tot = {t: 0.0 for t in tiles}
    totf = 0.0
    for name, cnt, H, W, cin, cout, R, stride, pad, mode in SHAPES:
        x = torch.randn(N, H, W, cin, device='cuda')
        w = torch.randn(cout, R * R * cin, device='cuda') * 0.05
        bias = torch.randn(cout, device='cuda')
        slope = torch.rand(cout, device='cuda')
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - R) // stride + 1
        out = torch.empty(N, Ho, Wo, cout, device='cuda')
        flops = 2.0 * N * Ho * Wo * cout * R * R * cin
        row = {'name': name, 'count': cnt, 'M': N * Ho * Wo, 'N': cout, 'K': R * R * cin, 'gflop': flops / 1e9}
        for t in tiles:
            if t and cout % (128 if t == 1 else 64):
                continue
            kw = dict(x=x, N=N, H=H, W=W, in_pitch=cin, cin_pad=cin, w=w, bias=bias, slope=slope, resid=None,
                      res_pitch=0, out=out, out_pitch=cout, out_coff=0, cout_store=cout, cout_pad=cout, R=R, S=R,
                      stride=stride, pad=pad, pad_mode=mode, border_bias=0, flags=0, tile=t, splitk=1 if t else 0)
            for _ in range(2):
                eng.op_conv(**kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                eng.op_conv(**kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row['t%d_ms' % t] = round(ms, 4)
            row['t%d_tf' % t] = round(flops / ms / 1e9, 1)
            tot[t] += ms * cnt
        totf += flops * cnt
        res.append(row)
        print('%-24s x%-2d M=%-8d N=%-4d K=%-6d ' % (name, cnt, row['M'], cout, row['K']) +
              '  '.join('t%d %7.3fms %6.1fTF' % (t, row['t%d_ms' % t], row['t%d_tf' % t])
                        for t in tiles if ('t%d_ms' % t) in row), flush=True)
    best = sum(min(r['t%d_ms' % t] for t in tiles if ('t%d_ms' % t) in r) * r['count'] for r in res)
    print('total GFLOP %.1f; heuristic(t0) %.3f ms = %.1f TF; best-of-tiles %.3f ms = %.1f TF'
          % (totf / 1e9, tot.get(0, 0), totf / max(tot.get(0, 1e-9), 1e-9) / 1e9, best, totf / best / 1e9))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'conv_sweep.json'), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
