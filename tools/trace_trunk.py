import os, sys, json, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import ffrnet_amd
from ffrnet_amd import synth
specs = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/g0_state_dict_keys.json')))
eng = ffrnet_amd.Engine(0); eng.load_encoder(synth.synth_state_dict(specs['encoder']))
x = synth.synth_images(256, seed=1).cuda()
eng.encoder_trunk_nhwc(x, 24)
