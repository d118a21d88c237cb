#!/usr/bin/env python3
"""python tools/wf_trace.py : (diagnostics build, option wf_trace) one forward at batch 256 with per-launch phase stamps of k_wino_fused on stderr."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import trace_build  # noqa: E402
trace_build.use()
import ffrnet_amd  # noqa: E402
from ffrnet_amd import synth  # noqa: E402
import bench  # noqa: E402

spec_e, spec_r = bench.state_dict_specs()
eng = ffrnet_amd.Engine(0)
eng.set_options_from_env()
eng.set_option(os.environ.get('TRACE', 'wf_trace'), 1)
eng.load_encoder(synth.synth_state_dict(spec_e))
eng.load_recnet(synth.synth_state_dict(spec_r))
B = int(os.environ.get('B', '256'))
x = synth.synth_images(B, seed=1).cuda()
eng.embed(x)
torch.cuda.synchronize()
