#!/usr/bin/env python3
"""VERDICT r04 "Next round" #6: the exact 4+3 tiling of 7x7 maps, ONE 512 -> 512 convolution (model_ir_se50.py:67,69, stage 4) at 256,
512 and 1024 images: k_wino_fused on padded F(4x4) tiles (4 tiles per image, 36 xi each = 144 xi-tiles) vs k_wino_fused_mixed (one
tile of each type: 36 + 30 + 30 + 25 = 121; padded xi 124) -- and the same for one 256 -> 256 convolution on a 14x14 map.
Blocks per CU: 7x7 at 256 / 512 / 1024 images = 1 / 2 / 4 (the pairing of a 9-slot with a 7-slot block needs >= 2)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402

eng = ffrnet_amd.Engine(0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev)
    return t[len(t) // 2] * 1e3


for H, C, Ns in ((7, 512, (256, 512, 1024)), (14, 256, (128, 256, 512))):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    slope = torch.rand(C, generator=g) * 0.3 + 0.1
    for N in Ns:
        x = torch.randn(N, H, H, C, generator=g).cuda()
        eng.reserve(max(N, 8))
        # op_conv3x3 packs its weights per call (host): time the device side only, through the profiling classes
        res = {}
        for name, mode in (('padded F(4x4), 32x64 blocks', 1), ('exact 4+3 tiling', 4)):
            eng.profile_enable(True)
            for _ in range(4):
                out = eng.op_conv3x3(x, w, bias, slope, 0, mode, None)
            torch.cuda.synchronize()
            st = eng.profile_read()
            eng.profile_enable(False)
            res[name] = (st['wino_fused']['ms'] / st['wino_fused']['launches'] * 1e3, st['wino']['ms'] / max(1, st['wino']['launches']) * 1e3, out)
        (a, ai, oa), (b, bi, ob) = res['padded F(4x4), 32x64 blocks'], res['exact 4+3 tiling']
        err = ((oa - ob).abs().max() / oa.abs().max()).item()
        print('%2dx%-2d %d->%d, %4d images: padded F(4x4) %7.1f us (+ input transform %5.1f) | exact tiling %7.1f us (+ %5.1f) | fused kernel x%.3f, '
              'with transforms x%.3f | max rel diff %.1e' % (H, H, C, C, N, a, ai, b, bi, b / a, (b + bi) / (a + ai), err))
