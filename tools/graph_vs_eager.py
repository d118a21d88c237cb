"""hipGraph replay (GraphedEmbed) vs eager launches of one forward at batch 256: 16.93 vs 16.93 ms -- the forward is GPU bound,
the graph only matters for small batches (launch bound)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffrnet_amd
from ffrnet_amd import synth
from ffrnet_amd.native import GraphedEmbed
specs = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g0_state_dict_keys.json')))
eng = ffrnet_amd.Engine(0)
eng.load_encoder(synth.synth_state_dict(specs['encoder']))
eng.load_recnet(synth.synth_state_dict(specs['recnet']))
B = 256
x = synth.synth_images(B, seed=1).cuda()
f_new = torch.empty((B, 512), device='cuda'); f = torch.empty((B, 512), device='cuda')
def timeit(fn, n=40, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('eager ms', timeit(lambda: eng.embed(x, out=(f_new, f))))
g = GraphedEmbed(eng, B)
print('graph ms', timeit(lambda: g(x)))
print('eager ms', timeit(lambda: eng.embed(x, out=(f_new, f))))
