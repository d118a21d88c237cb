"""Deterministic synthetic weights, images and verification pairs.

The reference's real checkpoints (se50.pth / FFRNet.pth, README.md:19-20) and the
LFW / CASIA images (README.md:26-30) are download links and are not available, so
every parity test, the smoke test and the benchmark run on seeded synthetic data.

The generator is counter based (numpy Philox keyed by crc32 of the state_dict key)
so the same tensors are regenerated bit for bit on the GPU box without depending on
torch's RNG stream.  BatchNorm running statistics are deliberately NON trivial: with
the default (mean 0, var 1, bias 0) a wrong pre-conv BN fold at zero padded borders
(reference pretrain/model_ir_se50.py:66-67) is invisible.
"""
import zlib

import numpy as np
import torch

# Block tables: (in_channel, depth, stride) of every bottleneck.
# Restates get_blocks(num_layers) / get_block of pretrain/model_ir_se50.py:78-105.
IR_UNITS = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}
IRSE50_STAGES = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))


def ir_blocks(num_layers=50):
    blocks = []
    for (cin, depth, _), units in zip(IRSE50_STAGES, IR_UNITS[num_layers]):
        blocks.append((cin, depth, 2))
        blocks.extend((depth, depth, 1) for _ in range(units - 1))
    return blocks


def irse50_blocks():
    return ir_blocks(50)


def _gen(key, seed):
    k = (int(seed) << 32) | (zlib.crc32(key.encode()) & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=k))


FAMILIES = ('benign', 'kaiming', 'trained')


def _role(key, spec):
    stem, leaf = key.rsplit('.', 1)
    is_bn = (stem + '.running_mean') in spec
    if leaf in ('num_batches_tracked', 'running_mean', 'running_var'):
        return leaf
    if is_bn:
        return 'bn_' + leaf
    if leaf == 'weight' and len(spec[key]) == 1:
        return 'prelu'
    return leaf                                     # 'weight' | 'bias' of a Conv2d / Linear


def _conv_feeds_bn(key, spec):
    """True when the Conv2d / Linear that owns `key` is followed directly by a BatchNorm (so its
    per-channel output scale is what that BatchNorm's running_var records in a trained network)."""
    stem = key.rsplit('.', 1)[0]
    if stem.endswith('.conv2d'):                    # ConvLayer of models/recnet.py:52-85: conv2d -> norm -> relu
        return (stem[:-len('conv2d')] + 'norm.norm.running_var') in spec
    head, _, idx = stem.rpartition('.')
    if idx.isdigit():                               # Sequential(..., Conv2d, BatchNorm2d) of model_ir_se50.py
        return ('%s.%d.running_var' % (head, int(idx) + 1)) in spec
    return False


def synth_tensor(key, shape, spec, seed=0, family='benign', calib=None):
    """One tensor of a state_dict, chosen by the role its key plays.

    family 'benign'  : gain-1 normal weights, BN gamma / running_var in U(0.75, 1.25), PReLU slopes in U(0.1, 0.4)
                       (goldens G1-G10).
    family 'kaiming' : RecNet exactly as the reference initialises it before training
                       (init_weights(self.recnet, 'kaiming'), models/recnet.py:13-42 via models/trainer.py:65-66):
                       Conv2d / Linear weights N(0, 2 / fan_in), their biases 0, BatchNorm2d gamma N(1, 0.02), beta 0,
                       running statistics (0, 1), PReLU 0.25; AddMarginProduct keeps its own init.
    family 'trained' : what a trained checkpoint looks like (se50.pth / FFRNet.pth, README.md:19-20): per-channel
                       weight scales over three decades, gamma of both signs, PReLU slopes in [-0.5, 1.5], and
                       running statistics that MATCH the activations: `calib` (tests/golden/g11_calib_*.npz, written
                       by tests/golden/make_golden_stress.py from a calibration pass through the reference) supplies
                       running_mean / running_var of every BatchNorm and the scale of every SE fc2 (a third of the
                       gates saturated).  Without `calib` the running statistics are (0, 1): the pre-calibration net.
    """
    if family not in FAMILIES:
        raise ValueError('synth_tensor: unknown family %r' % (family,))
    g = _gen(key, seed)
    shape = tuple(int(s) for s in shape)
    role = _role(key, spec)
    if role == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.int64)
    if family == 'benign':
        if role == 'running_mean':
            a = g.standard_normal(shape) * 0.1
        elif role == 'running_var':
            a = g.uniform(0.75, 1.25, shape)
        elif role == 'bn_weight':
            a = g.uniform(0.75, 1.25, shape)
        elif role == 'bn_bias':
            a = g.standard_normal(shape) * 0.1
        elif role == 'prelu':
            a = g.uniform(0.1, 0.4, shape)
        elif role == 'weight':
            fan_in = int(np.prod(shape[1:]))
            # variance-preserving normal (gain 1): with the kaiming gain sqrt(2) the 24
            # residual adds grow the trunk output to ~1e4 and every sigmoid on the path
            # saturates, which would hide errors instead of exposing them
            a = g.standard_normal(shape) * np.sqrt(1.0 / fan_in)
        elif role == 'bias':
            a = g.standard_normal(shape) * 0.05
        else:
            raise KeyError('synth_tensor: unknown parameter role for %r' % key)
    elif family == 'kaiming':
        if role == 'running_mean':
            a = np.zeros(shape)
        elif role == 'running_var':
            a = np.ones(shape)
        elif role == 'bn_weight':
            a = 1.0 + 0.02 * g.standard_normal(shape)
        elif role in ('bn_bias', 'bias'):
            a = np.zeros(shape)
        elif role == 'prelu':
            a = np.full(shape, 0.25)
        elif role == 'weight' and key == 'classifier.weight':
            a = g.standard_normal(shape) * np.sqrt(1.0 / shape[1])
        elif role == 'weight':
            a = g.standard_normal(shape) * np.sqrt(2.0 / int(np.prod(shape[1:])))
        else:
            raise KeyError('synth_tensor: unknown parameter role for %r' % key)
    else:
        if role in ('running_mean', 'running_var'):
            if calib is not None:
                a = np.asarray(calib[key], dtype=np.float64).reshape(shape)
            else:
                a = np.zeros(shape) if role == 'running_mean' else np.ones(shape)
        elif role == 'bn_weight':
            a = np.exp(g.uniform(np.log(0.05), np.log(2.0), shape)) * np.where(g.uniform(0, 1, shape) < 0.15, -1.0, 1.0)
        elif role == 'bn_bias':
            a = g.standard_normal(shape) * 0.3
        elif role == 'prelu':
            a = g.uniform(-0.5, 1.5, shape)
        elif role == 'weight':
            fan_in = int(np.prod(shape[1:]))
            a = g.standard_normal(shape) * np.sqrt(1.0 / fan_in)
            lo, hi = (1e-2, 1e1) if _conv_feeds_bn(key, spec) else (0.1, 4.0)
            ch = np.exp(g.uniform(np.log(lo), np.log(hi), shape[0]))        # per-output-channel variance
            a = a * np.sqrt(ch).reshape((-1,) + (1,) * (len(shape) - 1))
            if key.endswith('.fc2.weight') and calib is not None:           # SEModule gate scale (model_ir_se50.py:24-25)
                a = a * float(calib[key + ':scale'])
        elif role == 'bias':
            a = g.standard_normal(shape) * 0.1
        else:
            raise KeyError('synth_tensor: unknown parameter role for %r' % key)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).astype(np.float32))


def synth_state_dict(spec, seed=0, family='benign', calib=None):
    """spec: {key: shape}.  Returns {key: tensor} in the same key order."""
    return {k: synth_tensor(k, s, spec, seed, family, calib) for k, s in spec.items()}


# The weight families of the stress goldens G11 (tests/golden/make_golden_stress.py):
# name -> (encoder family, encoder seed, RecNet family, RecNet seed, image seed, pair seed)
STRESS_FAMILIES = {
    'benign_s1': ('benign', 1, 'benign', 1, 223, 17),
    'kaiming': ('trained', 12, 'kaiming', 11, 224, 18),     # the 'trained' family's encoder (its fp32 conditioning is known)
    'trained': ('trained', 12, 'trained', 12, 225, 19),
}


def stress_state_dicts(name, spec_enc, spec_rec, golden_dir):
    """(encoder state_dict, RecNet state_dict) of one G11 family; the calibrated BatchNorm statistics / SE scales of
    the 'trained' sets are read from golden_dir/g11_calib_<name>.npz (data, written by make_golden_stress.py)."""
    import os
    fe, se, fr, sr = STRESS_FAMILIES[name][:4]
    ce = cr = None
    if fe == 'trained' or fr == 'trained':
        z = np.load(os.path.join(golden_dir, 'g11_calib_%s.npz' % name))
        ce = {k[4:]: z[k] for k in z.files if k.startswith('enc:')}
        ce.update({k: z[k] for k in z.files if k.endswith(':scale')})
        cr = {k[4:]: z[k] for k in z.files if k.startswith('rec:')}
    return (synth_state_dict(spec_enc, se, fe, ce if fe == 'trained' else None),
            synth_state_dict(spec_rec, sr, fr, cr if fr == 'trained' else None))


def synth_images(n, h=112, w=112, seed=123):
    """[n,3,h,w] fp32 in [-1,1): the input contract of data/dataloader.py:24-28
    (ToTensor + Normalize(0.5,0.5)), BGR order irrelevant for random data."""
    g = np.random.Generator(np.random.Philox(key=(int(seed) << 32) | 0x1A6E5))
    return torch.from_numpy(g.uniform(-1.0, 1.0, (n, 3, h, w)).astype(np.float32))


def synth_pairs(n_pairs, h=112, w=112, seed=7, block=600):
    """Synthetic verification pairs laid out like LFW pairs.txt as parsed by
    data/dataset.py:42-53: blocks of `block` pairs, first half 'same', second half
    'different'.  A 'same' pair is (img, img with the lower half replaced + noise);
    a 'different' pair is two independent images.  Returns img1, img2, labels."""
    g = np.random.Generator(np.random.Philox(key=(int(seed) << 32) | 0xBA125))
    img1 = g.uniform(-1.0, 1.0, (n_pairs, 3, h, w)).astype(np.float32)
    img2 = g.uniform(-1.0, 1.0, (n_pairs, 3, h, w)).astype(np.float32)
    labels = np.zeros(n_pairs, dtype=np.int64)
    half = block // 2
    for i in range(n_pairs):
        if (i % block) < half:
            labels[i] = 1
            # strength of the identity signal varies so scores straddle thresholds
            a = g.uniform(0.2, 1.0)
            noise = g.standard_normal((3, h, w)).astype(np.float32) * 0.05
            img2[i] = a * img1[i] + (1.0 - a) * img2[i] + noise
            img2[i, :, h // 2:, :] = g.uniform(-1.0, 1.0, (3, h - h // 2, w))
    np.clip(img2, -1.0, 1.0, out=img2)
    return torch.from_numpy(img1), torch.from_numpy(img2), torch.from_numpy(labels)


def synth_train_batch(n, seed=301, n_classes=10575):
    """A synthetic training batch shaped like data/dataset.py's CASIA items (train.py:46-48):
    clean images, the same images with a lower-right block replaced ('occluded'), identity labels."""
    non = synth_images(n, 112, 112, seed=seed)
    ocl = non.clone()
    ocl[:, :, 56:, 40:] = synth_images(n, 112, 112, seed=seed + 1)[:, :, 56:, 40:]
    g = np.random.Generator(np.random.Philox(key=(int(seed) << 32) | 0x1ABE1))
    label = torch.from_numpy(g.integers(0, n_classes, n)).long()
    return non, ocl, label
