"""Where does NativeTrainer.step deviate from the oracle?  cotangents (GPU torch vs CPU oracle) and gradients."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ffrnet_amd
from ffrnet_amd import synth
import torch_losses as train
import test_gpu_train as T

specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
tc = T.build_train_case(specs)
sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
non, ocl, label = synth.synth_train_batch(4, seed=301)
eng = ffrnet_amd.Engine(0)
eng.load_encoder(sd_e)
eng.train_init(tc['sd_r'])
n = 4
with torch.no_grad():
    fm, f_enc = eng.encoder_forward(torch.cat((non, ocl)).cuda())
print('encoder featmap err', T.rel(fm, tc['fm']))
lab = torch.cat((label, label)).cuda()
outs = eng.train_forward(fm, lab, groups=2, want=('f_new', 'pred_loss', 'pred_label', 'feat_space', 'feat_channel'))
f_new, pred_loss, pred_label, _, _, feat_space, feat_channel = outs
leaves = [t.detach().requires_grad_(True) for t in (f_new, pred_loss, feat_space, feat_channel)]
lf, lp, ls, lc = leaves
items = train.trainer_losses(lf[:n], lf[n:], lp[:n], lp[n:], ls[:n], ls[n:], lc[:n], lc[n:], fm[:n], f_enc[:n], f_enc[n:],
                             label.cuda().long())
torch.autograd.backward(sum(items))
print('losses gpu', [float(l) for l in items], 'oracle', tc['losses'])
og = tc['out_grads']
for name, gpu_g, idx in (('f_new', lf.grad, 0), ('pred_loss', lp.grad, 1), ('feat_space', ls.grad, 5), ('feat_channel', lc.grad, 6)):
    ref = torch.cat([og[idx], og[7 + idx]])
    print('cotangent %-12s err %.3e  |ref|max %.3e' % (name, T.rel(gpu_g, ref), ref.abs().max().item()))
eng.train_zero_grad()
eng.train_backward([lf.grad, lp.grad, None, None, None, ls.grad, lc.grad])
torch.cuda.synchronize()
rows = []
for k in tc['keys']:
    got = eng.train_get(k, 'grad'); ref = tc['param_grads'][k]
    rows.append((T.rel(got, ref), k))
rows.sort(reverse=True)
for e, k in rows[:12]:
    print('%-44s err %.3e' % (k, e))
