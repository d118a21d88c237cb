#!/usr/bin/env python3
"""ms per forward of Engine.embed at batch 256 (no result checks: for timing ablations)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd, bench
from ffrnet_amd import synth
spec_e, spec_r = bench.state_dict_specs()
eng = ffrnet_amd.Engine(0); eng.load_encoder(synth.synth_state_dict(spec_e)); eng.load_recnet(synth.synth_state_dict(spec_r))
B = int(os.environ.get('B', '256'))
x = synth.synth_images(B, seed=1).cuda(); o = (torch.empty(B, 512, device='cuda'), torch.empty(B, 512, device='cuda'))
for _ in range(3): eng.embed(x, out=o)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.embed(x, out=o)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
eng.profile_enable(True); eng.embed(x, out=o); torch.cuda.synchronize(); st = eng.profile_read()
print('%.3f ms/forward  %.0f emb/s' % (dt * 1e3, B / dt), {k: round(v['ms'], 3) for k, v in st.items() if v['launches']})
