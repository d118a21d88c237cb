#!/usr/bin/env python3
"""FFR_WF_TRACE=1 python tools/wf_trace.py : one forward at batch 256 with per-launch phase stamps of k_wino_fused on stderr."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd  # noqa: E402
from ffrnet_amd import synth  # noqa: E402
import bench  # noqa: E402

spec_e, spec_r = bench.state_dict_specs()
eng = ffrnet_amd.Engine(0)
eng.load_encoder(synth.synth_state_dict(spec_e))
eng.load_recnet(synth.synth_state_dict(spec_r))
B = int(os.environ.get('B', '256'))
x = synth.synth_images(B, seed=1).cuda()
eng.embed(x)
torch.cuda.synchronize()
