"""Embed one synthetic batch and save (f_new, f) -- run with different experiment knobs (FFR_OPT_<NAME>=<int> in the
environment of THIS script -> Engine.set_option before the weights are packed) by
tests/test_gpu_parity.py::test_experiment_knobs_keep_parity."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402
from ffrnet_amd import synth  # noqa: E402

out, B = sys.argv[1], int(sys.argv[2])
specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
eng = ffrnet_amd.Engine(0)
print('options', eng.set_options_from_env())
eng.load_encoder(synth.synth_state_dict(specs['encoder']))
eng.load_recnet(synth.synth_state_dict(specs['recnet']))
f_new, f = eng.embed(synth.synth_images(B, seed=77).cuda())
torch.cuda.synchronize()
torch.save({'f_new': f_new.cpu(), 'f': f.cpu()}, out)
print('OK')
