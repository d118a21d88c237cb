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


def synth_tensor(key, shape, spec, seed=0):
    """One tensor of a state_dict, chosen by the role its key plays."""
    g = _gen(key, seed)
    shape = tuple(int(s) for s in shape)
    stem = key.rsplit('.', 1)[0]
    leaf = key.rsplit('.', 1)[1]
    is_bn = (stem + '.running_mean') in spec
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == 'running_mean':
        a = g.standard_normal(shape) * 0.1
    elif leaf == 'running_var':
        a = g.uniform(0.75, 1.25, shape)
    elif leaf == 'weight' and is_bn:
        a = g.uniform(0.75, 1.25, shape)
    elif leaf == 'bias' and is_bn:
        a = g.standard_normal(shape) * 0.1
    elif leaf == 'weight' and len(shape) == 1:
        a = g.uniform(0.1, 0.4, shape)              # PReLU slopes
    elif leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        # variance-preserving normal (gain 1): with the kaiming gain sqrt(2) the 24
        # residual adds grow the trunk output to ~1e4 and every sigmoid on the path
        # saturates, which would hide errors instead of exposing them
        a = g.standard_normal(shape) * np.sqrt(1.0 / fan_in)
    elif leaf == 'bias':
        a = g.standard_normal(shape) * 0.05
    else:
        raise KeyError('synth_tensor: unknown parameter role for %r' % key)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).astype(np.float32))


def synth_state_dict(spec, seed=0):
    """spec: {key: shape}.  Returns {key: tensor} in the same key order."""
    return {k: synth_tensor(k, s, spec, seed) for k, s in spec.items()}


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
