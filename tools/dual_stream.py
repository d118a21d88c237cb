#!/usr/bin/env python3
"""Experiment: two half batches on two HIP streams (two handles), so HBM-bound kernels of one half can
overlap the MFMA-bound GEMMs of the other."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd, bench
from ffrnet_amd import synth
spec_e, spec_r = bench.state_dict_specs()
sd_e, sd_r = synth.synth_state_dict(spec_e), synth.synth_state_dict(spec_r)
B = 256
x = synth.synth_images(B, seed=1).cuda()
def make(n):
    e = ffrnet_amd.Engine(0); e.load_encoder(sd_e); e.load_recnet(sd_r); e.reserve(n); return e
def run(nsplit, steps=10):
    engs = [make(B // nsplit) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = [x[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for i in range(nsplit)]
    outs = [(torch.empty(B // nsplit, 512, device='cuda'), torch.empty(B // nsplit, 512, device='cuda')) for _ in range(nsplit)]
    def step():
        for e, s, xi, o in zip(engs, streams, xs, outs):
            with torch.cuda.stream(s):
                e.embed(xi, out=o)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('streams %d: %.3f ms/step, %.0f emb/s' % (nsplit, dt / steps * 1e3, B * steps / dt), flush=True)
for n in (1, 2, 4):
    run(n)
