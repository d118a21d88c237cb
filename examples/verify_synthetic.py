#!/usr/bin/env python3
"""Verification end to end on synthetic LFW-shaped pairs (needs an MI355X): the counterpart of
`python train.py --phase test` -> eval_lfw -> lfw_eval.get_avg_accuracy (train.py:101-113, lfw/lfw_eval.py:272-287).

    python examples/verify_synthetic.py [--pairs 600] [--batch 100]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/verify_synthetic.py

With real weights: replace the synthetic state_dicts by torch.load('pretrain/se50.pth') and
ffrnet_amd.checkpoint.load_recnet_checkpoint('check_points/FFR-Net/latest.pth.gzip').
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ffrnet_amd
from ffrnet_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=600)
    ap.add_argument('--batch', type=int, default=100)
    a = ap.parse_args()
    world, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    specs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g0_state_dict_keys.json')))
    eng = ffrnet_amd.Engine(local)
    eng.load_encoder(synth.synth_state_dict(specs['encoder']))
    eng.load_recnet(synth.synth_state_dict(specs['recnet']))
    i1, i2, lab = synth.synth_pairs(a.pairs, seed=7, block=max(2, a.pairs // 10))     # 10 blocks: half same, half different
    dev = torch.device('cuda', local)
    loader = [dict(img1=i1[s:s + a.batch].to(dev), img2=i2[s:s + a.batch].to(dev), label=lab[s:s + a.batch],
                   idx=torch.arange(s, min(s + a.batch, a.pairs))) for s in range(0, a.pairs, a.batch)]
    # pairs sharded over the ranks, ONE all-gather of the embeddings per batch, scores and the 10-fold threshold protocol on
    # the device (ffr_cosine_scores, ffr_lfw_fold_accuracy); with the two nn.Module shells the call is the reference's:
    #     acc_new, acc = ffrnet_amd.lfw.get_avg_accuracy(encoder, recnet, data_loader)
    acc_new, acc = ffrnet_amd.lfw.get_avg_accuracy(eng.embed, loader)
    if int(os.environ.get('RANK', '0')) == 0:
        print('f_new (RecNet)   %d pairs, 10-fold accuracy %.4f' % (a.pairs, acc_new))
        print('f (encoder)      %d pairs, 10-fold accuracy %.4f' % (a.pairs, acc))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
