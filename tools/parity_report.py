#!/usr/bin/env python3
"""Prints the parity margins (max-abs-err / max-abs-ref) of the HIP path against the committed
goldens of the reference (config 1, 8 images) and against the oracle at other batch sizes."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ffrnet_amd, ffr_oracle as O
from ffrnet_amd import synth
specs = json.load(open(os.path.join(ROOT, 'tests/golden/g0_state_dict_keys.json')))
sd_e = synth.synth_state_dict(specs['encoder']); sd_r = synth.synth_state_dict(specs['recnet'])
eng = ffrnet_amd.Engine(0); eng.load_encoder(sd_e); eng.load_recnet(sd_r)
g = np.load(os.path.join(ROOT, 'tests/golden/g1_config1.npz'))
def rel(a, b):
    a = a.double().cpu(); b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max()).item()
x = synth.synth_images(8, seed=123).cuda()
fm, f = eng.encoder_forward(x); f_new, feat_new = eng.recnet_forward(fm)
print('vs reference goldens (8 images): f %.2e  f_new %.2e  featmap0 %.2e  feat_new0 %.2e'
      % (rel(f, g['f']), rel(f_new, g['f_new']), rel(fm[0], g['featmap0']), rel(feat_new[0], g['feat_new0'])))
rowl2 = lambda a, b: ((a.cpu() - torch.as_tensor(b)).norm(dim=1) / torch.as_tensor(b).norm(dim=1)).max().item()
print('per-row rel-L2: f %.2e  f_new %.2e' % (rowl2(f, g['f']), rowl2(f_new, g['f_new'])))
g2 = np.load(os.path.join(ROOT, 'tests/golden/g2_stage_taps.npz'))
for nb, name in [(0, 'input_layer')] + [(i + 1, 'body.%d' % i) for i in (0, 2, 3, 6, 7, 20, 21, 23)]:
    got = eng.encoder_trunk_nhwc(x[:1], nb).permute(0, 3, 1, 2).cpu()[0].reshape(-1)
    step = max(1, got.numel() // 256)
    err = np.abs(got[::step][:256].numpy() - g2[name + '.samples']).max() / float(g2[name + '.absmax'])
    print('  tap %-12s %.2e' % (name, err))
