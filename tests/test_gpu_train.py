"""GPU parity of the RecNet training step through the C ABI (include/ffrnet_train.h):
single operators against torch CPU autograd, the whole step against the oracle (oracle/ffr_oracle_train.py)
and the golden G8 captured from the reference's own Trainer code."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ffrnet_amd
from ffrnet_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_losses as TL  # noqa: E402

GRAD_TOL = 2e-4     # gradients: max-abs-err / max-abs-ref per tensor (fp32 re-association over <= 12544-row sums)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def engine():
    return ffrnet_amd.Engine(0)


CONVLAYER_CASES = [   # (G, N, cin, cout)
    (1, 4, 64, 64), (2, 3, 128, 128), (1, 5, 49, 49), (2, 2, 561, 256), (1, 3, 128, 49), (1, 2, 1024, 512),
]


@pytest.mark.parametrize('case', CONVLAYER_CASES)
def test_convlayer_train_forward_backward(engine, case):
    """reflect-pad -> conv3x3 -> BatchNorm2d(train) -> PReLU, models/recnet.py:78-85: outputs, batch / running
    statistics, data gradient and all four parameter gradients vs torch autograd on the CPU."""
    G, N, cin, cout = case
    g = torch.Generator().manual_seed(77 + cin + cout)
    x = torch.randn(G * N, 7, 7, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    gamma = torch.rand(cout, generator=g) * 0.5 + 0.75
    beta = torch.randn(cout, generator=g) * 0.1
    slope = torch.rand(cout, generator=g) * 0.3 + 0.1
    da = torch.randn(G * N, 7, 7, cout, generator=g)
    res = engine.op_convlayer_train(x.cuda(), G, w, gamma, beta, slope, da.cuda())
    torch.cuda.synchronize()
    # torch reference, one BatchNorm batch per group
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, gr, br, sr = (t.clone().requires_grad_(True) for t in (w, gamma, beta, slope))
    rm, rv = torch.zeros(cout), torch.zeros(cout)
    outs = []
    for gi in range(G):
        y = F.conv2d(F.pad(xr[gi * N:(gi + 1) * N], (1,) * 4, mode='reflect'), wr)
        outs.append(F.prelu(F.batch_norm(y, rm, rv, gr, br, True, 0.1, 1e-5), sr))
    out = torch.cat(outs)
    out.backward(da.permute(0, 3, 1, 2))
    assert rel(res['out'].permute(0, 3, 1, 2), out) < 2e-5
    assert rel(res['running_mean'], rm) < 2e-5 and rel(res['running_var'], rv) < 2e-5
    assert rel(res['dx'].permute(0, 3, 1, 2), xr.grad) < GRAD_TOL
    assert rel(res['dw'], wr.grad) < GRAD_TOL
    assert rel(res['dgamma'], gr.grad) < GRAD_TOL
    assert rel(res['dbeta'], br.grad) < GRAD_TOL
    assert rel(res['dslope'], sr.grad) < GRAD_TOL


# ---- PReLU kinks ------------------------------------------------------------------------------------------
# RecNet has 2.1 M PReLU inputs in an 8-image step; a forward that differs from the reference's by 1e-5 (F(4x4,3x3) in fp32,
# another encoder) puts a few dozen of them on the other side of zero, and each such element changes its own gradient by
# (1 - slope) -- percents of a parameter gradient's norm, although NOTHING is wrong (the oracle does the same to itself:
# a 1e-5 relative perturbation of its float64 inputs moves its own gradients by 7e-3..1e-2 in relative L2, a 1e-6 one by
# 3e-6; tools/train_grad_report.py).  The reference-pinned check of a gradient therefore takes the SIGN PATTERN the GPU
# forward produced (ffr_train_debug_copy: y * scale + shift per ConvLayer, the three row pre-activations of
# Conv4Channel), evaluates the oracle's gradient formula on THAT side of every kink (ffr_oracle_train._prelu, ctl['mask']),
# and (a) holds the GPU gradients to it tightly, per tensor, in the default (Winograd) arithmetic and with the real slopes;
# (b) requires every element whose side differs from the oracle's own to lie within the forward tolerance of zero -- the
# only place where two correct fp32 forwards may disagree.
def gpu_kink_masks(eng, groups, n, slot=0):
    """{oracle PReLU name: [groups * n, ...] bool, True = the identity side} of the forward held in `slot`."""
    import ffr_oracle_train as OT
    imgs = groups * n
    masks = {}
    for name, dbg in OT.PRELU_LAYERS:
        if dbg.startswith('h'):
            masks[name] = eng.train_debug(dbg, (imgs * 512, 64), slot)[:, :32].reshape(imgs, 512, 32) > 0
            continue
        c = {'sp': (256, 256, 256, 128, 128, 128, 49, 49, 49), 'fm': (512,) * 3, 'mg': (512,) * 3}[dbg[:2]][int(dbg[2])]
        cp = (c + 63) // 64 * 64
        y = eng.train_debug('y.' + dbg, (groups, n * 49, cp), slot).double()
        sc = eng.train_debug('scale.' + dbg, (groups, 1, cp), slot).double()
        sh = eng.train_debug('shift.' + dbg, (groups, 1, cp), slot).double()
        z = (y * sc + sh).reshape(imgs, 7, 7, cp)[..., :c].permute(0, 3, 1, 2)      # the kernel's fma, exactly (float64 holds the product)
        masks[name] = z > 0
    return masks


def oracle_grads_on_masks(sd_r, fm, f_enc, label, masks, n):
    """The oracle's parameter gradients of Trainer.backward's loss with every PReLU element on the side `masks` says
    (None: its own side).  fm / f_enc: [2 n, ...] clean then occluded.  -> ({key: grad}, natural pre-activations, items)"""
    import ffr_oracle_train as OT
    keys = OT.trainable_keys(sd_r)
    params = {k: sd_r[k].clone().requires_grad_(True) for k in keys}
    running = {k: v.clone() for k, v in sd_r.items() if k not in params}
    ctl = [{'mask': {k: m[:n] for k, m in masks.items()}} if masks else {}, {'mask': {k: m[n:] for k, m in masks.items()}} if masks else {}]
    out_non = OT.recnet_train_forward(params, fm[:n], label, running, ctl[0])
    out_ocl = OT.recnet_train_forward(params, fm[n:], label, running, ctl[1])
    items = OT.trainer_losses(out_non, out_ocl, fm[:n], f_enc[:n], f_enc[n:], label)
    grads = torch.autograd.grad(sum(items), [params[k] for k in keys], allow_unused=True)
    pre = {k: torch.cat([ctl[0]['pre'][k], ctl[1]['pre'][k]]) for k in ctl[0]['pre']}
    return {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(keys, grads)}, pre, [float(i.detach()) for i in items]


def golden_side_masks(g8, pre, n):
    """The side of every PReLU kink the REFERENCE's run (golden G8) was on: the oracle's own signs on this host, with
    the elements the golden recorded as near zero (|x| < 1e-4 of the layer's largest) put on the golden's side -- any
    host's fp32 rounding moves pre-activations by ~1e-6, so only recorded elements can differ."""
    masks = {}
    for k, x in pre.items():
        m = (x > 0)
        for gi, tag in enumerate(('non', 'ocl')):
            part = m[gi * n:(gi + 1) * n].reshape(-1).clone()
            idx = torch.from_numpy(g8['kink.%s.%s.idx' % (k, tag)])
            part[idx] = torch.from_numpy(g8['kink.%s.%s.side' % (k, tag)])
            m[gi * n:(gi + 1) * n] = part.reshape(m[gi * n:(gi + 1) * n].shape)
        masks[k] = m
    return masks


def check_kinks(masks, pre, tol):
    """Every element whose side differs from the oracle's own lies within `tol` (relative to its layer's largest
    pre-activation) of zero.  -> number of such elements."""
    flipped = 0
    for k, m in masks.items():
        nat = pre[k]
        diff = m != (nat > 0)
        if diff.any():
            margin = (nat[diff].abs().max() / nat.abs().max()).item()
            assert margin < tol, (k, int(diff.sum()), margin)
            flipped += int(diff.sum())
    return flipped


def grad_err(got, ref):
    # a BatchNorm bias in front of [identity -> conv -> BatchNorm] has a gradient that is zero up to rounding
    # (the next BatchNorm removes the shift): errors are taken relative to at least 1e-3
    return ((got.double().cpu() - ref.double()).abs().max() / max(ref.abs().max().item(), 1e-3)).item()


# ---- the whole RecNet training step ---------------------------------------------------------------
@pytest.fixture(scope='module')
def train_case(specs):
    return build_train_case(specs)


def build_train_case(specs, smooth=False, family=None):
    """The G8 scenario (4 clean + 4 occluded images, synthetic weights; `family`: the weights of golden G12 instead) through
    the ORACLE with autograd: outputs, gradients wrt the seven outputs of both RecNet calls, parameter gradients (unclipped)."""
    import ffr_oracle as O
    import ffr_oracle_train as OT
    if family:
        sd_e, sd_r = synth.stress_state_dicts(family, specs['encoder'], specs['recnet'], os.path.join(ROOT, 'tests', 'golden'))
    else:
        sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
        sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    if smooth:      # PReLU slopes of 1: the network has no kinks, gradients are continuous in every rounding
        for k in sd_r:
            if k.endswith('func.weight'):
                sd_r[k] = torch.ones_like(sd_r[k])
    non, ocl, label = synth.synth_train_batch(4, seed=301)
    with torch.no_grad():
        fm_non, fe_non = O.encoder_forward(sd_e, non)
        fm_ocl, fe_ocl = O.encoder_forward(sd_e, ocl)
    keys = OT.trainable_keys(sd_r)
    params = {k: sd_r[k].clone().requires_grad_(True) for k in keys}
    running = {k: v.clone() for k, v in sd_r.items() if k not in params}
    out_non = OT.recnet_train_forward(params, fm_non, label, running)
    out_ocl = OT.recnet_train_forward(params, fm_ocl, label, running)
    items = OT.trainer_losses(out_non, out_ocl, fm_non, fe_non, fe_ocl, label)
    grads = torch.autograd.grad(sum(items), [params[k] for k in keys], allow_unused=True)
    pg = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(keys, grads)}
    # cotangents of the 7-tuples: the PARTIAL derivatives of the loss, outputs taken as independent leaves
    # (pred_loss is a function of pred_label, f_new of feat_space ... inside the graph)
    leaf_non = [o.detach().clone().requires_grad_(True) for o in out_non]
    leaf_ocl = [o.detach().clone().requires_grad_(True) for o in out_ocl]
    items_l = OT.trainer_losses(leaf_non, leaf_ocl, fm_non, fe_non, fe_ocl, label)
    og = torch.autograd.grad(sum(items_l), leaf_non + leaf_ocl, allow_unused=True)
    return dict(sd_r=sd_r, f_enc=torch.cat([fe_non, fe_ocl]), fm=torch.cat([fm_non, fm_ocl]), label=torch.cat([label, label]), out_non=out_non,
                out_ocl=out_ocl, out_grads=og, param_grads=pg, running=running, keys=keys,
                losses=[float(l.detach()) for l in items])


NAMES = ['f_new', 'pred_loss', 'pred_label', 'M_space', 'M_channel', 'feat_space', 'feat_channel']


def test_train_forward_matches_oracle_and_golden(engine, train_case, golden_dir):
    """RecNet.forward(input, label) in train() mode (models/recnet.py:398-429), both groups in one call."""
    tc = train_case
    engine.train_init(tc['sd_r'])
    outs = engine.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2)
    torch.cuda.synchronize()
    g8 = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    for gi, (tag, ref) in enumerate((('non', tc['out_non']), ('ocl', tc['out_ocl']))):
        for nm, o, r in zip(NAMES, outs, ref):
            got = o[gi * 4:(gi + 1) * 4].cpu()
            r = r.detach().reshape(got.shape)
            assert rel(got, r) < 1e-4, (tag, nm, rel(got, r))
        assert rel(outs[0][gi * 4:(gi + 1) * 4], torch.from_numpy(g8['full.%s.f' % tag])) < 1e-4
        assert rel(outs[3][gi * 4], torch.from_numpy(g8['full.%s.M_space0' % tag]).reshape(49, 49)) < 1e-4
        assert rel(outs[6][gi * 4], torch.from_numpy(g8['full.%s.feat_channel0' % tag])) < 1e-4
    sd_after = engine.train_state_dict()
    for k, v in tc['running'].items():
        if k.endswith(('running_mean', 'running_var')):
            assert rel(sd_after[k], v) < 1e-4, k
            assert rel(sd_after[k], torch.from_numpy(g8['after.' + k]).float()) < 1e-4, k
        elif k.endswith('num_batches_tracked'):
            assert int(sd_after[k]) == 2


@pytest.mark.parametrize('family', [None, 'kaiming'])
def test_train_backward_matches_oracle_and_golden(engine, train_case, golden_dir, specs, family):
    """loss.backward() through RecNet (models/trainer.py:179-180): the 76 parameter gradients for the reference's four
    losses (cotangents of the 7-tuple from the oracle's autograd), in BOTH arithmetics -- direct convolutions and the
    DEFAULT (Winograd F(4x4,3x3) forward and data-gradient convolutions) -- with the reference's real PReLU slopes,
    per tensor, against (1) the oracle on the GPU's side of every kink (see "PReLU kinks" above) and (2) golden G8, the
    clipped gradients of the reference's own Trainer, plus exactly the difference the transplanted kinks make."""
    # family 'kaiming' (golden G12): the weights a training run STARTS from -- RecNet as init_weights(self.recnet, 'kaiming') leaves it
    # (models/trainer.py:65-66) behind the trained-like encoder of golden G11; feature maps up to 74, unclipped gradients up to 1e2
    tc = build_train_case(specs, family=family) if family else train_case
    og = tc['out_grads']
    stacked = []
    for i in range(7):
        a, b = og[i], og[7 + i]
        if a is None and b is None:
            stacked.append(None)
            continue
        ref_shape = tc['out_non'][i].shape
        a = a if a is not None else torch.zeros(ref_shape)
        b = b if b is not None else torch.zeros(ref_shape)
        stacked.append(torch.cat([a, b]).cuda())
    g8 = np.load(os.path.join(golden_dir, 'g12_train_step_%s.npz' % family if family else 'g8_train_step.npz'))
    own, pre, _ = oracle_grads_on_masks(tc['sd_r'], tc['fm'], tc['f_enc'], tc['label'][:4], None, 4)
    for k in tc['keys']:
        assert grad_err(own[k], tc['param_grads'][k]) < 1e-6     # the helper IS build_train_case's computation
    # the oracle on the side of every kink the reference's run was on reproduces the golden's clipped gradients on THIS host
    gmasks = golden_side_masks(g8, pre, 4)
    print('%d near-zero PReLU inputs of this host\'s oracle run lie on the other side than in the golden\'s run'
          % check_kinks(gmasks, pre, 1e-5))
    natural, _, _ = oracle_grads_on_masks(tc['sd_r'], tc['fm'], tc['f_enc'], tc['label'][:4], gmasks, 4)
    for mode, tol, fwd_tol in ((0, 1e-4, 1e-5), (1, 1e-4, 1e-4)):      # measured: 7e-6 / 1.4e-5 (1 and 4 kinks transplanted)
        engine.train_init(tc['sd_r'])
        engine.train_option('winograd', mode)
        engine.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2, want=())
        masks = gpu_kink_masks(engine, 2, 4)
        flipped = check_kinks(masks, pre, fwd_tol)
        engine.train_zero_grad()
        engine.train_backward(stacked)
        torch.cuda.synchronize()
        ref, _, _ = oracle_grads_on_masks(tc['sd_r'], tc['fm'], tc['f_enc'], tc['label'][:4], masks, 4)
        worst = ('', 0.0)
        moved = 0.0
        for k in tc['keys']:
            got = engine.train_get(k, 'grad')
            e = grad_err(got, ref[k])
            worst = max(worst, (k, e), key=lambda t: t[1])
            assert e < tol, (mode, k, e)
            moved = max(moved, grad_err(ref[k], natural[k]))
            # the reference's own (clipped) gradients: digest = [sum, abs-sum, 64 strided samples] + what the kinks moved
            d = g8['grad.' + k]

            def samples(t):
                f = t.clamp(-1.0, 1.0).reshape(-1)
                return f[::max(1, f.numel() // 64)][:64].double().cpu()
            expect = torch.from_numpy(d[2:]) + samples(ref[k]) - samples(natural[k])
            scale = max(ref[k].abs().max().item(), 1e-3)          # the tensor's largest gradient, as for `e`
            assert (samples(got) - expect).abs().max().item() / scale < tol, (mode, k)
            asum = d[1] + ref[k].clamp(-1, 1).double().abs().sum().item() - natural[k].clamp(-1, 1).double().abs().sum().item()
            assert abs(got.clamp(-1.0, 1.0).double().abs().sum().item() - asum) <= tol * max(d[1], 1e-3 * got.numel()), (mode, k)
        print('%s winograd=%d: %d of 2.1 M PReLU inputs on the other side of zero (all within %.0e of it); they move the '
              'oracle\'s own gradients by up to %.2e; worst GPU gradient error on the same side %.2e (%s)'
              % (family or 'benign', mode, flipped, fwd_tol, moved, worst[1], worst[0]))
    engine.train_option('winograd', 1)


def test_adam_and_clip_match_torch(engine, specs):
    """clip_grad_value_(1.0) + torch.optim.Adam (models/trainer.py:120,182-187): three steps with given gradients."""
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    engine.train_init(sd_r)
    keys = ['Conv4Space.4.conv2d.weight', 'Conv4Channel.0.weight', 'Conv4Space.5.conv2.norm.norm.bias',
            'Conv4Channel.7.func.weight']
    params = [sd_r[k].clone().requires_grad_(True) for k in keys]
    opt = torch.optim.Adam(params, 0.1, betas=(0.9, 0.999), weight_decay=0)
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        engine.train_zero_grad()
        for k, p in zip(keys, params):
            grad = torch.randn(p.shape, generator=g) * (3.0 if step == 1 else 0.3)     # some beyond the clip value
            engine.train_set(k, grad, 'grad')
            p.grad = grad.clone()
        torch.nn.utils.clip_grad_value_(params, 1.0)
        opt.step()
        engine.train_adam_step(0.1, (0.9, 0.999), 1e-8, 0.0, 1.0)
    torch.cuda.synchronize()
    for k, p in zip(keys, params):
        assert rel(engine.train_get(k, 'param'), p) < 1e-5, k
        assert rel(engine.train_get(k, 'exp_avg'), opt.state[p]['exp_avg']) < 1e-5, k
        assert rel(engine.train_get(k, 'exp_avg_sq'), opt.state[p]['exp_avg_sq']) < 1e-5, k
    # untouched parameters (zero gradient) do not move
    assert torch.equal(engine.train_get('Conv4Merge.0.conv2d.weight', 'param'), sd_r['Conv4Merge.0.conv2d.weight'])
    assert engine.train_info()['adam_step'] == 3


def test_native_trainer_step_matches_reference(specs, golden_dir):
    """One whole iteration (train.py:46-54) through NativeTrainer vs the golden captured from the reference's
    Trainer: the four loss items, the accuracy, clipped gradients, running statistics, updated parameters."""
    import ffr_oracle_train as OT
    g8 = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    non, ocl, label = synth.synth_train_batch(4, seed=301)
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=float(g8['lr']))
    eng.train_option('winograd', 0)          # direct convolutions for the tight comparison; Winograd mode below
    items = tr.step(non.cuda(), ocl.cuda(), label.cuda())
    torch.cuda.synchronize()
    got = np.array([float(l) for l in items])
    assert np.allclose(got, g8['losses'], rtol=1e-4), (got, g8['losses'])
    assert float(tr.accuracy) == float(g8['accuracy'])
    sd_after = tr.state_dict()
    for k in sd_after:
        if k.endswith(('running_mean', 'running_var')):
            assert rel(sd_after[k], torch.from_numpy(g8['after.' + k]).float()) < 1e-4, k
    # The feature maps here are the GPU encoder's (1e-5 from the CPU's), so the gradients are held to the oracle evaluated
    # ON those feature maps and on the GPU's side of every PReLU kink ("PReLU kinks" above): per tensor, tight, real slopes.
    import ffr_oracle as O  # noqa: F401
    fm, f_enc = eng.encoder_forward(torch.cat((non, ocl)).cuda())
    fm, f_enc = fm.cpu(), f_enc.cpu()
    keys = OT.trainable_keys(sd_r)
    _, pre, _ = oracle_grads_on_masks(sd_r, fm, f_enc, label, None, 4)

    def check_iteration(tol, fwd_tol, tag):
        masks = gpu_kink_masks(eng, 2, 4)
        flipped = check_kinks(masks, pre, fwd_tol)
        ref, _, ref_items = oracle_grads_on_masks(sd_r, fm, f_enc, label, masks, 4)
        worst = 0.0
        for k in keys:
            got_g = eng.train_get(k, 'grad')
            e = grad_err(got_g, ref[k])
            worst = max(worst, e)
            assert e < tol, (tag, k, e)
        print('%s: %d kinks transplanted, worst gradient error %.2e' % (tag, flipped, worst))
        return ref

    check_iteration(1e-4, 1e-5, 'whole iteration, direct')
    for k in keys:
        # the optimiser inside the step: Adam's first update of the clipped native gradient, torch's formula
        gc = eng.train_get(k, 'grad').clamp(-1.0, 1.0)
        m, v = 0.1 * gc, 0.001 * gc * gc
        expect = sd_r[k] - (0.1 / 0.1) * m / ((v.sqrt() / (1.0 - 0.999) ** 0.5) + 1e-8)
        assert (sd_after[k] - expect).abs().max().item() < 2e-5, k
    # DEFAULT mode (Winograd forward / data-gradient convolutions): same losses, gradients per tensor as tight as above
    tr2 = ffrnet_amd.NativeTrainer(eng, sd_r, lr=float(g8['lr']))
    items2 = tr2.step(non.cuda(), ocl.cuda(), label.cuda())
    assert np.allclose(np.array([float(l) for l in items2]), g8['losses'], rtol=1e-4)
    check_iteration(1e-4, 1e-4, 'whole iteration, default (Winograd)')
    # the same iteration with the loss items evaluated by torch ops instead of the native loss kernels
    tr3 = ffrnet_amd.NativeTrainer(eng, sd_r, lr=float(g8['lr']))
    items3 = TL.step_torch_losses(tr3, non.cuda(), ocl.cuda(), label.cuda())
    assert np.allclose(np.array([float(l) for l in items3]), np.array([float(l) for l in items2]), rtol=1e-4)
    assert float(tr3.accuracy) == float(tr2.accuracy)
    check_iteration(1e-4, 1e-4, 'whole iteration, default, torch loss items')


def test_train_backward_kink_free_network_both_modes(engine, specs):
    """With all PReLU slopes set to 1 the network is smooth, so rounding cannot flip anything: the direct and the
    Winograd mode must both give the oracle's gradients tightly (this is what separates 'kink noise' from a bug)."""
    tc = build_train_case(specs, smooth=True)
    og = tc['out_grads']
    stacked = []
    for i in range(7):
        a, b = og[i], og[7 + i]
        if a is None and b is None:
            stacked.append(None)
            continue
        shp = tc['out_non'][i].shape
        stacked.append(torch.cat([a if a is not None else torch.zeros(shp), b if b is not None else torch.zeros(shp)]).cuda())
    for mode, fused in ((0, 1), (1, 0), (1, 2)):      # direct / Winograd as transform kernels + batched GEMM / Winograd in k_wino_fused
        engine.train_init(tc['sd_r'])
        engine.train_option('winograd', mode)
        engine.train_option('fused', fused)
        engine.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2, want=())
        engine.train_zero_grad()
        engine.train_backward(stacked)
        torch.cuda.synchronize()
        for k in tc['keys']:
            ref = tc['param_grads'][k]
            got = engine.train_get(k, 'grad')
            # a BatchNorm bias in front of [identity -> conv -> BatchNorm] has a gradient that is zero up to rounding
            # (the next BatchNorm removes the shift): errors are taken relative to at least 1e-3
            e = ((got.double() - ref.double()).abs().max() / max(ref.abs().max().item(), 1e-3)).item()
            assert e < (1e-4 if mode == 0 else 3e-4), (mode, fused, k, e)
    engine.train_option('fused', 1)


def test_recnet_shell_train_branch_two_iterations(specs):
    """ffrnet_amd.RecNet in train() mode as the reference's Trainer drives it (models/trainer.py:139-187):
    recnet(featmap, label) twice, the four losses, loss.backward(), clip_grad_value_, torch.optim.Adam.step();
    then a second iteration, which must see the moved parameters.  Held to the oracle's train_step."""
    import ffr_oracle as O
    import ffr_oracle_train as OT
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    for k in sd_r:                      # smooth network (PReLU slope 1): no kink noise, tight tolerances
        if k.endswith('func.weight'):
            sd_r[k] = torch.ones_like(sd_r[k])
    non, ocl, label = synth.synth_train_batch(4, seed=301)
    with torch.no_grad():
        fm_non, fe_non = O.encoder_forward(sd_e, non)
        fm_ocl, fe_ocl = O.encoder_forward(sd_e, ocl)
    rec = ffrnet_amd.RecNet(norm_type='bn', relu_type='prelu')
    rec.load_state_dict(sd_r)
    rec.cuda().train()
    # plain SGD between the two iterations: Adam's g / (|g| + eps) turns the rounding noise of mathematically zero
    # gradients into full-size steps, which would make iteration 2 incomparable (Adam itself: test_adam_and_clip_*)
    opt = torch.optim.SGD([p for p in rec.parameters() if p.requires_grad], 0.05)
    sd_o = {k: v.clone() for k, v in sd_r.items()}
    ost = OT.new_adam_state(sd_o)
    dev = torch.device('cuda', 0)
    for it in range(2):
        # oracle iteration on the same feature maps (encoder_forward is deterministic: recomputed inside)
        ref = OT.train_step(sd_e, sd_o, ost, non, ocl, label, apply_update=False)
        for k, g in ref['grads'].items():
            sd_o[k] -= 0.05 * g
        out_non = rec(fm_non.to(dev), label.to(dev))
        out_ocl = rec(fm_ocl.to(dev), label.to(dev))
        assert len(out_non) == 7 and out_non[1].shape == (4, 10575)
        items = TL.trainer_losses(out_non[0], out_ocl[0], out_non[1], out_ocl[1], out_non[5], out_ocl[5], out_non[6],
                                 out_ocl[6], fm_non.to(dev), fe_non.to(dev), fe_ocl.to(dev), label.to(dev))
        assert np.allclose([float(l) for l in items], ref['losses'], rtol=2e-4), (it, [float(l) for l in items], ref['losses'])
        opt.zero_grad()
        sum(items).backward()
        torch.nn.utils.clip_grad_value_(rec.parameters(), 1.0)
        for k, p in rec.named_parameters():
            r = ref['grads'][k]
            e = ((p.grad.cpu().double() - r.double()).abs().max() / max(r.abs().max().item(), 1e-3)).item()
            assert e < 5e-4, (it, k, e)
        opt.step()
    sd_m = rec.state_dict()
    for k, v in sd_o.items():
        if k.endswith('num_batches_tracked'):
            assert int(sd_m[k]) == 4
        elif k.endswith(('running_mean', 'running_var')):
            assert rel(sd_m[k], v) < 1e-4, k


def test_native_loss_items_and_their_gradients(engine, train_case):
    """ffr_train_losses: the four loss items of Trainer.backward (models/trainer.py:154-178) and their partial
    derivatives wrt the RecNet outputs, against the oracle's values and autograd cotangents."""
    tc = train_case
    engine.train_init(tc['sd_r'])
    engine.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2, want=())
    out5 = engine.train_losses(tc['f_enc'].cuda()).cpu()
    assert np.allclose(out5[:4].numpy(), tc['losses'], rtol=1e-4), (out5, tc['losses'])
    acc = (tc['out_ocl'][2].detach().argmax(1) == tc['label'][:4]).float().mean().item()
    assert float(out5[4]) == acc
    og = tc['out_grads']
    df = engine.train_debug('df_ext', (8, 512))
    assert rel(df, torch.cat([og[0], og[7]])) < 1e-4
    dcos = engine.train_debug('dcos', (8, 10624))
    assert rel(dcos[:, :10575], 30.0 * torch.cat([og[1], og[8]])) < 1e-4
    assert dcos[:, 10575:].abs().max() == 0
    ext = engine.train_debug('extM', (8 * 49, 1024)).reshape(8, 49, 1024)
    d_fs = torch.cat([og[5], og[12]]).reshape(8, 512, 49).permute(0, 2, 1)
    d_fc = torch.cat([og[6], og[13]]).reshape(8, 512, 49).permute(0, 2, 1)
    assert rel(ext[:, :, :512], d_fs) < 2e-4
    assert rel(ext[:, :, 512:], d_fc) < 2e-4
    # and the backward fed by them gives the gradients of the direct-mode reference run
    engine.train_option('winograd', 0)
    engine.train_init(tc['sd_r'])
    engine.train_option('winograd', 0)
    engine.train_forward(tc['fm'].cuda(), tc['label'].cuda(), groups=2, want=())
    engine.train_losses(tc['f_enc'].cuda())
    engine.train_zero_grad()
    engine.train_backward_losses()
    torch.cuda.synchronize()
    masks = gpu_kink_masks(engine, 2, 4)
    ref, _, _ = oracle_grads_on_masks(tc['sd_r'], tc['fm'], tc['f_enc'], tc['label'][:4], masks, 4)
    for k in tc['keys']:
        assert grad_err(engine.train_get(k, 'grad'), ref[k]) < 2e-4, k
    engine.train_option('winograd', 1)


def test_training_full_size_properties(specs):
    """BASELINE configs[4] per-GPU shape (128 pairs per iteration), where the oracle is too slow: properties.
    Bitwise reproducibility of a whole iteration, direct vs Winograd mode agree on the loss items, a few Adam steps
    on a fixed batch lower the total loss, everything stays finite."""
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    non, ocl, label = synth.synth_train_batch(128, seed=77)
    non, ocl, label = non.cuda(), ocl.cuda(), label.cuda()
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    first = []
    for mode, fused in ((1, 1), (1, 1), (0, 1), (1, 0)):
        tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
        eng.train_option('winograd', mode)
        eng.train_option('fused', fused)
        items = torch.stack(tr.step(non, ocl, label)).cpu()
        first.append((items, tr.flat_grads.clone()))
    assert torch.equal(first[0][0], first[1][0]) and torch.equal(first[0][1], first[1][1])       # same mode: bitwise
    assert torch.allclose(first[0][0], first[2][0], rtol=2e-4)                                   # Winograd vs direct
    l2 = ((first[0][1] - first[2][1]).norm() / first[2][1].norm()).item()
    assert l2 < 2e-2, l2          # whole flat gradient, kink noise included (256-image batch)
    # the fused kernel and the transform-kernel form of the same Winograd arithmetic (summation order differs only)
    assert torch.allclose(first[0][0], first[3][0], rtol=2e-5)
    l2f = ((first[0][1] - first[3][1]).norm() / first[3][1].norm()).item()
    assert l2f < 2e-2, l2f
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
    totals = []
    for _ in range(6):
        totals.append(float(torch.stack(tr.step(non, ocl, label)).sum()))
    assert all(np.isfinite(totals)) and totals[-1] < totals[0], totals
    assert torch.isfinite(tr.flat_params).all()
    assert 0.0 <= float(tr.accuracy) <= 1.0


def test_native_trainer_resume_is_exact(specs, tmp_path):
    """Checkpoint / resume of the native trainer (models/trainer.py:201-224 layout): parameters, BatchNorm running
    statistics, num_batches_tracked (continues from the checkpoint value), Adam moments and step count.  Two steps +
    save + load into a fresh trainer + one step == three uninterrupted steps, bit for bit; labels are validated."""
    from ffrnet_amd import checkpoint
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    non, ocl, label = (t.cuda() for t in synth.synth_train_batch(6, seed=88))
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
    for _ in range(3):
        tr.step(non, ocl, label)
    want = tr.flat_params.clone()
    want_sd = tr.state_dict()
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
    for _ in range(2):
        tr.step(non, ocl, label)
    path = str(tmp_path / '0000002.pth.gzip')
    checkpoint.save_recnet_checkpoint(tr, path, extra_info={'epoch': 0, 'iter': 2})
    ck = checkpoint.load(path)
    nbt = [int(v) for k, v in ck['RecNet'].items() if k.endswith('num_batches_tracked')]
    assert set(nbt) == {4} and ck['optimizer']['step'] == 2          # two BatchNorm batches (clean, occluded) per step
    eng2 = ffrnet_amd.Engine(0)
    eng2.load_encoder(sd_e)
    tr2 = ffrnet_amd.NativeTrainer(eng2, ck['RecNet'], lr=1e-3)
    tr2.load_optimizer_state_dict(ck['optimizer'])
    tr2.step(non, ocl, label)
    assert torch.equal(tr2.flat_params, want)
    got_sd = tr2.state_dict()
    for k in want_sd:
        assert torch.equal(got_sd[k].cpu(), want_sd[k].cpu()), k
    assert {int(v) for k, v in got_sd.items() if k.endswith('num_batches_tracked')} == {6}
    with pytest.raises(RuntimeError):
        tr2.step(non, ocl, label.clone().fill_(10575))               # class id out of range: loud, as the reference
    with pytest.raises(RuntimeError):
        tr2.step(non, ocl, label[:3])                                # one label per pair


def test_train_forward_full_size_values(specs):
    """BASELINE configs[4] per-GPU shape: 128 pairs = 256 images through the train-mode forward, VALUES checked.  The
    oracle's restatement (torch ops) is evaluated on the device for this size -- an independent implementation (stock
    torch / MIOpen kernels) of RecNet.forward(input, label), models/recnet.py:398-429, BatchNorm batch statistics per
    group included -- on the very feature maps the native encoder produced."""
    import ffr_oracle_train as OT
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    non, ocl, label = synth.synth_train_batch(128, seed=78)
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.train_init(sd_r)
    with torch.no_grad():
        fm, _ = eng.encoder_forward(torch.cat((non, ocl)).cuda())
    lab = label.cuda()
    outs = eng.train_forward(fm, torch.cat((lab, lab)), groups=2)
    sd_dev = {k: v.cuda() for k, v in sd_r.items()}
    names = ('f_new', 'pred_loss', 'pred_label', 'M_space', 'M_channel', 'feat_space', 'feat_channel')
    with torch.no_grad():
        for g in range(2):
            ref = OT.recnet_train_forward(sd_dev, fm[128 * g:128 * (g + 1)], lab.long(), None)
            for name, got, want in zip(names, outs, ref):
                got = got[128 * g:128 * (g + 1)]
                assert rel(got.reshape(want.shape), want) < 2e-4, (g, name, rel(got.reshape(want.shape), want))


@pytest.mark.parametrize('n', [1, 3])
def test_train_forward_small_and_odd_batches(engine, specs, n):
    """Edge batches: one image per group (BatchNorm statistics over 49 positions only) and an odd count."""
    import ffr_oracle_train as OT
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    g = torch.Generator().manual_seed(40 + n)
    fm = torch.randn(2 * n, 512, 7, 7, generator=g) * 0.6
    label = torch.randint(0, 10575, (n,), generator=g)
    keys = OT.trainable_keys(sd_r)
    params = {k: sd_r[k] for k in keys}
    running = {k: v.clone() for k, v in sd_r.items() if k not in params}
    with torch.no_grad():
        ref = [OT.recnet_train_forward(params, fm[i * n:(i + 1) * n], label, running) for i in range(2)]
    engine.train_init(sd_r)
    outs = engine.train_forward(fm.cuda(), torch.cat([label, label]).cuda(), groups=2)
    torch.cuda.synchronize()
    for gi in range(2):
        for nm, o, r in zip(NAMES, outs, ref[gi]):
            got = o[gi * n:(gi + 1) * n].cpu()
            assert rel(got, r.reshape(got.shape)) < 2e-4, (n, gi, nm)
    sd_after = engine.train_state_dict()
    for k, v in running.items():
        if k.endswith(('running_mean', 'running_var')):
            assert rel(sd_after[k], v) < 2e-4, k


def test_trained_weights_feed_the_eval_path(specs):
    """After training iterations the exported state_dict (torch layouts, running statistics) must drive the
    verification path like any checkpoint: oracle eval forward == native eval forward on it, and the RecNet
    embedding differs from the one before training."""
    import ffr_oracle as O
    sd_e = synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = synth.synth_state_dict(specs['recnet'], seed=0)
    non, ocl, label = synth.synth_train_batch(8, seed=91)
    eng = ffrnet_amd.Engine(0)
    eng.load_encoder(sd_e)
    eng.load_recnet(sd_r)
    x = synth.synth_images(4, seed=92)
    f_before, _ = eng.embed(x.cuda())
    tr = ffrnet_amd.NativeTrainer(eng, sd_r, lr=1e-3)
    for _ in range(3):
        tr.step(non.cuda(), ocl.cuda(), label.cuda())
    sd_t = tr.state_dict()
    assert set(sd_t) == set(sd_r) and all(sd_t[k].shape == sd_r[k].shape for k in sd_r)
    assert int(sd_t['Conv4Space.0.norm.norm.num_batches_tracked']) == 6
    eng.load_recnet(sd_t)
    f_after, _ = eng.embed(x.cuda())
    with torch.no_grad():
        ref, _ = O.embed(sd_e, sd_t, x)
    assert rel(f_after, ref) < 1e-3
    assert rel(f_after, f_before.cpu()) > 1e-3
