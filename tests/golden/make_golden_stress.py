#!/usr/bin/env python3
"""Generate the STRESS-FAMILY goldens G11 by running THE REFERENCE ITSELF on CPU.

G1-G10 use one weight seed of one benign distribution.  G11 repeats the config-1 run, the stage taps and a
600-pair verification run for three more weight families (ffr-net_amd/synth.py):

  benign_s1 : the G1 distribution, seed 1 (encoder and RecNet).
  kaiming   : a 'trained' encoder + RecNet exactly as the reference initialises it before training
              (init_weights(self.recnet, 'kaiming'), models/recnet.py:13-42 via models/trainer.py:65-66).
  trained   : encoder AND RecNet trained-like: per-channel weight scales over three decades, BatchNorm gamma of
              both signs, PReLU slopes in [-0.5, 1.5], a third of the SE gates saturated, and BatchNorm running
              statistics that match the activations, as a trained checkpoint's do.

The running statistics of the 'trained' sets cannot be drawn independently of the weights (24 residual blocks
would over/underflow), so they are CALIBRATED here: one pass of 16 synthetic images through the reference's own
modules in float64 with every BatchNorm in training mode and momentum 1 (running stats := batch stats), the
SE fc2 of each block rescaled so that its pre-sigmoid output has sigma 6.2 (P(|z| > 6) = 1/3), then every
statistic perturbed (var x logU[0.7, 1.4], mean + 0.1 sigma N(0,1)) because a checkpoint's EMA never equals the
test batch's statistics.  The calibrated vectors are DATA (tests/golden/g11_calib_<family>.npz, fp32); with them
synth.synth_state_dict(..., family='trained', calib=...) regenerates the weights bit for bit on the GPU box.

Per family the fixture holds, from the reference in fp32 (its own arithmetic) AND in float64 (the exact answer,
so the tests can state the reference's own rounding error beside the product's):
  f, f_new (8 x 512), featmap0, feat_new0, the G2 stage taps, and a 600-pair run of
  lfw/lfw_eval.py:226-252 calculate_distance + :110-118 KFold(600, 10) + :255-259 get_fold_accuracy.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_stress.py   (~10 min)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from ffrnet_amd import synth  # noqa: E402

TAPS = ['input_layer'] + ['body.%d' % i for i in (0, 2, 3, 6, 7, 20, 21, 23)]
N_PAIRS, PAIR_BATCH = 600, 100
PAIR_BLOCK = 60           # KFold(600, 10) test folds are [60 i, 60 (i + 1)): 30 'same' + 30 'different' pairs each, as LFW's 300 + 300
GATE_SIGMA = 6.2          # P(|N(0, 6.2)| > 6) = 0.333: a third of the SE gates beyond sigmoid(+-6) = 0.9975 / 0.0025

FAMILIES = synth.STRESS_FAMILIES     # name -> (encoder family, encoder seed, RecNet family, RecNet seed, image seed, pair seed)


def bn_modules(net):
    return [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]


def calibrate(enc, rec, x, rec_trained):
    """Fill the running statistics of enc (and of rec when rec_trained) from one pass of x, rescale every SE fc2.
    enc / rec: the reference's modules in float64 holding the pre-calibration weights.  Returns {key: array}."""
    scales = {}
    nets = [enc] + ([rec] if rec_trained else [])
    for net in nets:
        net.eval()
        for m in bn_modules(net):
            m.momentum = 1.0
            m.train()
    with torch.no_grad():
        h = enc.input_layer(x)
        for bi, blk in enumerate(enc.body):
            se = blk.res_layer[5]
            z = {}
            hook = se.fc2.register_forward_hook(lambda m, i, o: z.__setitem__('z', o.detach()))
            blk(h)
            s = GATE_SIGMA / float(z['z'].std())
            se.fc2.weight.mul_(s)
            scales['body.%d.res_layer.5.fc2.weight:scale' % bi] = np.float64(s)
            h = blk(h)
            hook.remove()
            sat = float((z['z'].abs() > 6).double().mean())
            if bi in (0, 7, 23):
                print('    block %2d: fc2 scale %8.3f, saturated gates %.3f, trunk std %.3g' % (bi, s, sat, float(h.std())))
        featmap = enc.bn(h)
        enc.output_layer(h)
        if rec_trained:
            rec(featmap)
    out = dict(scales)
    for prefix, net in zip(('enc:', 'rec:'), nets):
        for k, v in net.state_dict().items():
            if k.endswith('running_mean') or k.endswith('running_var'):
                out[prefix + k] = v.clone().double().numpy()
    for net in nets:
        net.eval()
    return out


def perturb(calib):
    """A checkpoint's EMA statistics are never the test batch's: var x logU[0.7, 1.4], mean + 0.1 sigma N(0, 1)."""
    out = {}
    for k, v in calib.items():
        if k.endswith('running_var'):
            g = synth._gen('perturb:' + k, 0x511)
            out[k] = v * np.exp(g.uniform(np.log(0.7), np.log(1.4), v.shape))
    for k, v in calib.items():
        if k.endswith('running_mean'):
            g = synth._gen('perturb:' + k, 0x511)
            out[k] = v + 0.1 * np.sqrt(out[k[:-len('running_mean')] + 'running_var']) * g.standard_normal(v.shape)
        elif not k.endswith('running_var'):
            out[k] = v
    return {k: (np.asarray(v, np.float32) if np.ndim(v) else np.float64(v)) for k, v in out.items()}


def split_calib(calib):
    ce = {k[4:]: v for k, v in calib.items() if k.startswith('enc:')}
    ce.update({k: v for k, v in calib.items() if k.endswith(':scale')})
    cr = {k[4:]: v for k, v in calib.items() if k.startswith('rec:')}
    return ce, cr


def run_config1(enc, rec, x):
    taps = {}
    hooks = [enc.input_layer.register_forward_hook(lambda m, i, o: taps.__setitem__('input_layer', o.detach()))]
    for bi in range(24):
        hooks.append(enc.body[bi].register_forward_hook(
            lambda m, i, o, bi=bi: taps.__setitem__('body.%d' % bi, o.detach())))
    with torch.no_grad():
        featmap, f = enc(x)
        f_new, feat_new = rec(featmap)
    for h in hooks:
        h.remove()
    return featmap, f, f_new, feat_new, taps


def gate_saturation(enc, x):
    sat = []
    hooks = [blk.res_layer[5].fc2.register_forward_hook(
        lambda m, i, o: sat.append(float((o.abs() > 6).double().mean()))) for blk in enc.body]
    with torch.no_grad():
        enc(x)
    for h in hooks:
        h.remove()
    return np.array(sat)


def main(only=None):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m_enc, m_rec, m_lfw = mg.import_reference()
    spec = json.load(open(os.path.join(HERE, 'g0_state_dict_keys.json')))
    for fam, (fe, se, fr, sr, img_seed, pair_seed) in FAMILIES.items():
        if only and fam not in only:
            continue
        t_start = time.time()
        print('family', fam)
        enc = m_enc.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
        rec = m_rec.RecNet(channel=512, shape=7, norm_type='bn', relu_type='prelu')
        ce = cr = None
        if fe == 'trained':
            enc.load_state_dict(synth.synth_state_dict(spec['encoder'], se, fe))
            rec.load_state_dict(synth.synth_state_dict(spec['recnet'], sr, fr))
            enc.double()
            rec.double()
            xc = synth.synth_images(16, 112, 112, seed=img_seed + 1000).double()
            calib = perturb(calibrate(enc, rec, xc, fr == 'trained'))
            np.savez_compressed(os.path.join(HERE, 'g11_calib_%s.npz' % fam), **calib)
            ce, cr = split_calib(calib)
            enc = m_enc.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
            rec = m_rec.RecNet(channel=512, shape=7, norm_type='bn', relu_type='prelu')
        sd_e = synth.synth_state_dict(spec['encoder'], se, fe, ce)
        sd_r = synth.synth_state_dict(spec['recnet'], sr, fr, cr if fr == 'trained' else None)
        enc.load_state_dict(sd_e)
        rec.load_state_dict(sd_r)
        enc.eval()
        rec.eval()

        x = synth.synth_images(8, 112, 112, seed=img_seed)
        out = dict(input_checksum=np.float64(x.double().sum().item()))
        featmap, f, f_new, feat_new, taps = run_config1(enc, rec, x)
        out.update(f=f.numpy(), f_new=f_new.numpy(), featmap0=featmap[0].numpy(), feat_new0=feat_new[0].numpy(),
                   featmap_absmax=np.float32(featmap.abs().max().item()),
                   feat_new_absmax=np.float32(feat_new.abs().max().item()))
        for name in TAPS:
            t = taps[name][0]
            out['tap.' + name + '.absmax'] = np.float64(t.abs().max().item())
            out['tap.' + name + '.std'] = np.float64(t.double().std().item())
            out['tap.' + name + '.samples'] = mg.strided(t).numpy()
        out['tap.body.23.full'] = taps['body.23'][0].numpy()
        bnv = np.concatenate([v.numpy().ravel() for k, v in sd_e.items() if k.endswith('running_var')])
        slopes = np.concatenate([v.numpy().ravel() for k, v in sd_e.items()
                                 if synth._role(k, spec['encoder']) == 'prelu'])
        out['enc_running_var_range'] = np.array([bnv.min(), np.median(bnv), bnv.max()])
        out['enc_prelu_range'] = np.array([slopes.min(), slopes.max()])
        out['enc_gate_saturation'] = gate_saturation(enc, x)

        # pairs in the reference's fp32 arithmetic
        i1, i2, lab = synth.synth_pairs(N_PAIRS, seed=pair_seed, block=PAIR_BLOCK)
        loader = [dict(img1=i1[s:s + PAIR_BATCH], img2=i2[s:s + PAIR_BATCH], label=lab[s:s + PAIR_BATCH],
                       idx=torch.arange(s, s + PAIR_BATCH)) for s in range(0, N_PAIRS, PAIR_BATCH)]
        pred_new, pred = m_lfw.calculate_distance(loader, enc, rec, use_gpu=False)
        folds = m_lfw.KFold(n=N_PAIRS, n_folds=10, shuffle=False)
        res_new = [m_lfw.get_fold_accuracy(fd, pred_new, 1) for fd in folds]
        res = [m_lfw.get_fold_accuracy(fd, pred, 0) for fd in folds]
        out.update(pair_seed=np.int64(pair_seed), n_pairs=np.int64(N_PAIRS), pair_batch=np.int64(PAIR_BATCH),
                   pair_block=np.int64(PAIR_BLOCK),
                   pair_checksum=np.float64(i1.double().sum().item() + i2.double().sum().item()),
                   scores_new=pred_new[:, 0], scores=pred[:, 0], labels=pred[:, 1],
                   best_thr_new=np.array([r[0] for r in res_new]), test_acc_new=np.array([r[1] for r in res_new]),
                   best_thr=np.array([r[0] for r in res]), test_acc=np.array([r[1] for r in res]),
                   acc_new=np.float64(sum(a for _, a in res_new) / 10), acc=np.float64(sum(a for _, a in res) / 10))

        # the same modules in float64: the exact answer the two fp32 implementations are both approximating
        enc.double()
        rec.double()
        featmap, f, f_new, feat_new, taps = run_config1(enc, rec, x.double())
        out.update(f_f64=f.numpy(), f_new_f64=f_new.numpy(), featmap0_f64=featmap[0].numpy(),
                   feat_new0_f64=feat_new[0].numpy())
        for name in TAPS:
            out['tap.' + name + '.samples_f64'] = mg.strided(taps[name][0]).numpy()
        loader = [dict(img1=b['img1'].double(), img2=b['img2'].double(), label=b['label'], idx=b['idx']) for b in loader]
        pred_new64, pred64 = m_lfw.calculate_distance(loader, enc, rec, use_gpu=False)
        out.update(scores_new_f64=pred_new64[:, 0], scores_f64=pred64[:, 0])
        np.savez_compressed(os.path.join(HERE, 'g11_%s.npz' % fam), **out)

        def rel(a, b):
            return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
        print('  reference fp32 vs its own float64: f %.2e  f_new %.2e  featmap0 %.2e  feat_new0 %.2e  scores %.2e / %.2e'
              % (rel(out['f'], out['f_f64']), rel(out['f_new'], out['f_new_f64']),
                 rel(out['featmap0'], out['featmap0_f64']), rel(out['feat_new0'], out['feat_new0_f64']),
                 np.abs(out['scores_new'] - out['scores_new_f64']).max(), np.abs(out['scores'] - out['scores_f64']).max()))
        print('  featmap absmax %.3g  feat_new absmax %.3g  acc_new %.4f  acc %.4f  thr_new %s  thr %s'
              % (out['featmap_absmax'], out['feat_new_absmax'], out['acc_new'], out['acc'],
                 sorted(set(np.round(out['best_thr_new'], 3))), sorted(set(np.round(out['best_thr'], 3)))))
        print('  encoder running_var min / median / max %s  PReLU slopes %s  saturated gates per block %.2f..%.2f'
              % (out['enc_running_var_range'], out['enc_prelu_range'],
                 out['enc_gate_saturation'].min(), out['enc_gate_saturation'].max()))
        print('  %.0f s' % (time.time() - t_start))


if __name__ == '__main__':
    main(sys.argv[1:])
