"""Per-parameter gradient error of the native RecNet backward vs the oracle's autograd (G8 scenario)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ffrnet_amd
from ffrnet_amd import synth
import test_gpu_train as T

specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
tc = T.build_train_case(specs)
eng = ffrnet_amd.Engine(0)
eng.train_init(tc['sd_r'])
eng.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2, want=())
og = tc['out_grads']
stacked = []
for i in range(7):
    a, b = og[i], og[7 + i]
    if a is None and b is None:
        stacked.append(None); continue
    shp = tc['out_non'][i].shape
    a = a if a is not None else torch.zeros(shp)
    b = b if b is not None else torch.zeros(shp)
    stacked.append(torch.cat([a, b]).cuda())
print('cotangents given:', [T.NAMES[i] for i in range(7) if stacked[i] is not None])
eng.train_zero_grad()
eng.train_backward(stacked)
torch.cuda.synchronize()
for k in tc['keys']:
    got = eng.train_get(k, 'grad'); ref = tc['param_grads'][k]
    print('%-44s err %.3e   |ref|max %.3e  |got|max %.3e' % (k, T.rel(got, ref), ref.abs().max().item(), got.abs().max().item()))
# ---- intermediates of Conv4Channel (first group = clean images) -------------------------------------
import ffr_oracle as O
import torch.nn.functional as F
sd = tc['sd_r']
fm = tc['fm'][:4]
with torch.no_grad():
    _, ssc = O.self_similarity(fm)
    flat = fm.reshape(4, 512, -1)
    cat = torch.cat((flat, ssc), 2)
    h1pre = F.linear(cat, sd['Conv4Channel.0.weight'], sd['Conv4Channel.0.bias'])
g_cat = eng.train_debug('cat', (8 * 512, 576))[:4 * 512]
g_h1 = eng.train_debug('h1pre', (8 * 512, 64))[:4 * 512, :32]
print('ss_channel err', T.rel(g_cat[:, :512], ssc.reshape(-1, 512)), ' Xt err', T.rel(g_cat[:, 512:561], flat.reshape(-1, 49)))
d = (g_h1 - h1pre.reshape(-1, 32)).abs()
print('h1pre max abs err', d.max().item(), 'sign flips', ((g_h1 > 0) != (h1pre.reshape(-1, 32) > 0)).sum().item())
d32a = eng.train_debug('d32a', (8 * 512, 64))
ref_b = tc['param_grads']['Conv4Channel.0.bias']
got_b = eng.train_get('Conv4Channel.0.bias', 'grad')
print('colsum(d32a) vs ref db0', T.rel(d32a[:, :32].double().sum(0), ref_b), ' native db0 vs colsum(d32a)', T.rel(got_b, d32a[:, :32].double().sum(0)))
print('d32a pad cols absmax', d32a[:, 32:].abs().max().item())
catg = eng.train_debug('cat', (8 * 512, 576))
dW = d32a[:, :32].double().t() @ catg.double()          # [32][576] native column order
ref_W = tc['param_grads']['Conv4Channel.0.weight']
ref_nat = torch.cat([ref_W[:, 49:], ref_W[:, :49]], 1)
print('d32a^T cat vs ref dW0', T.rel(dW[:, :561], ref_nat))
fm1 = tc['fm'][4:]
with torch.no_grad():
    _, ssc1 = O.self_similarity(fm1)
    cat1 = torch.cat((fm1.reshape(4, 512, -1), ssc1), 2)
    h1pre1 = F.linear(cat1, sd['Conv4Channel.0.weight'], sd['Conv4Channel.0.bias']).reshape(-1, 32)
g1 = eng.train_debug('h1pre', (8 * 512, 64))[4 * 512:, :32]
print('group 1: h1pre max abs err', (g1 - h1pre1).abs().max().item(), 'sign flips', ((g1 > 0) != (h1pre1 > 0)).sum().item(),
      'min |h1pre|', h1pre1.abs().min().item())
