"""Is the training backward deterministic and independent of stale device memory?  Runs the G8-scenario backward
(Winograd mode) three times: fresh engine, after poisoning freed device memory with NaN, and again; compares bitwise."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ffrnet_amd
import test_gpu_train as T

specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
tc = T.build_train_case(specs)
og = tc['out_grads']
stacked = []
for i in range(7):
    a, b = og[i], og[7 + i]
    if a is None and b is None:
        stacked.append(None); continue
    shp = tc['out_non'][i].shape
    stacked.append(torch.cat([a if a is not None else torch.zeros(shp), b if b is not None else torch.zeros(shp)]).cuda())


def poison(gb=6):
    xs = [torch.full((256, 1024, 1024), float('nan'), device='cuda') for _ in range(gb)]
    torch.cuda.synchronize()
    del xs
    torch.cuda.empty_cache()


def run(mode):
    eng = ffrnet_amd.Engine(0)
    eng.train_init(tc['sd_r'])
    eng.train_option('winograd', mode)
    outs = eng.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2)
    eng.train_zero_grad()
    eng.train_backward(stacked)
    torch.cuda.synchronize()
    g = {k: eng.train_get(k, 'grad') for k in tc['keys']}
    o = [x.cpu() for x in outs]
    eng.close()
    return g, o


for mode in (0, 1):
    g1, o1 = run(mode)
    poison()
    g2, o2 = run(mode)
    g3, o3 = run(mode)
    for name, (ga, gb_) in (('fresh vs poisoned', (g1, g2)), ('poisoned vs again', (g2, g3))):
        bad = [k for k in g1 if not torch.equal(ga[k], gb_[k])]
        nan = [k for k in g1 if not torch.isfinite(gb_[k]).all()]
        print('mode %d %s: %d of %d gradient tensors differ, %d non-finite; first: %s' % (mode, name, len(bad), len(g1), len(nan), bad[:3]))
    print('mode %d outputs identical:' % mode, all(torch.equal(a, b) for a, b in zip(o1, o2)))
