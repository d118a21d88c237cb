#!/usr/bin/env python3
"""ms per forward at small batches, eager launches vs hipGraph replay."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd, bench
from ffrnet_amd import synth
spec_e, spec_r = bench.state_dict_specs()
eng = ffrnet_amd.Engine(0); eng.load_encoder(synth.synth_state_dict(spec_e)); eng.load_recnet(synth.synth_state_dict(spec_r))
for B in (1, 8, 32):
    x = synth.synth_images(B, seed=1).cuda()
    o = (torch.empty(B, 512, device='cuda'), torch.empty(B, 512, device='cuda'))
    def t(fn, it=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    eager = t(lambda: eng.embed(x, out=o))
    g = ffrnet_amd.GraphedEmbed(eng, B)
    graph = t(lambda: g(x))
    print('batch %3d: eager %.3f ms  hipGraph %.3f ms' % (B, eager, graph), flush=True)
