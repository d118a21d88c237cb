#!/usr/bin/env python3
"""A few RecNet training iterations on a synthetic batch (needs an MI355X): the counterpart of the loop body of
train.py:41-54 (Trainer.set_input / forward / optimizer_parameters), with the periodic hand-over of the trained weights
to the verification path (train.py:74-93).

    python examples/train_synthetic.py [--pairs 64] [--iters 20]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
from ffrnet_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=64, help='image pairs per GPU and iteration (run.py: 64 in total)')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--lr', type=float, default=1e-3)
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
    eng = ffrnet_amd.Engine(local)
    eng.load_encoder(synth.synth_state_dict(specs['encoder']))                 # frozen, eval (models/trainer.py:62-63,79)
    trainer = ffrnet_amd.NativeTrainer(eng, synth.synth_state_dict(specs['recnet']), lr=a.lr, betas=(0.9, 0.999),
                                       weight_decay=0.0, loss_weight=(1, 1, 1, 1))
    trainer.broadcast_params(0)
    dev = torch.device('cuda', local)
    non, ocl, label = (t.to(dev) for t in synth.synth_train_batch(a.pairs, seed=100 + rank))
    for it in range(1, a.iters + 1):
        items = trainer.step(non, ocl, label)                                   # device tensors, no sync
        if it in (5000, 10000, 15000):
            trainer.lr *= 0.5                                                   # MultiStepLR per iteration, trainer.py:82-84
        if rank == 0 and (it % 5 == 0 or it == 1):
            print('iter %3d  ss %.4f  triplet %.4f  identity %.4f  cls %.4f  acc %.3f' %
                  (it, *[float(x) for x in items], float(trainer.accuracy)))
    eng.load_recnet(trainer.state_dict())                                       # hand the weights to the eval path
    f_new, f = eng.embed(non[:4])
    if rank == 0:
        print('embeddings with the trained RecNet:', tuple(f_new.shape), 'finite:', bool(torch.isfinite(f_new).all()))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
