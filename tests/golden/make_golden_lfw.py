#!/usr/bin/env python3
"""Generate tests/golden/g9_lfw_protocol.npz by running THE REFERENCE ITSELF on CPU at the
full size of BASELINE.json configs[3]: 6000 synthetic pairs, pair batches of 512.

Run in the build container only (needs /root/reference, which never travels):
    python tests/golden/make_golden_lfw.py            # ~10 min on 8 cores

What runs is the reference's own code, imported as in make_golden.py:
  lfw/lfw_eval.py:226-252  calculate_distance(loader, encoder, recnet, use_gpu=False)
  lfw/lfw_eval.py:110-118  KFold(n=6000, n_folds=10, shuffle=False)
  lfw/lfw_eval.py:255-259  get_fold_accuracy(fold, predicts, new)   (called directly: the
                           mp.Pool of :276-283 only distributes these ten calls)
  lfw/lfw_eval.py:261-270  the mean over the folds with its literal /10
Stored: the two score vectors, labels, ten (best_thr, test_acc) per embedding, both means.
The pairs are regenerated from ffr-net_amd/synth.py (seed 7, block 600), never stored.
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

N_PAIRS, BATCH, SEED, BLOCK = 6000, 512, 7, 600


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m_enc, m_rec, m_lfw = mg.import_reference()
    enc = m_enc.Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    rec = m_rec.RecNet(channel=512, shape=7, norm_type='bn', relu_type='prelu')
    spec = json.load(open(os.path.join(HERE, 'g0_state_dict_keys.json')))
    enc.load_state_dict(synth.synth_state_dict(spec['encoder'], seed=0))
    rec.load_state_dict(synth.synth_state_dict(spec['recnet'], seed=0))
    enc.eval()
    rec.eval()

    i1, i2, lab = synth.synth_pairs(N_PAIRS, seed=SEED, block=BLOCK)
    loader = [dict(img1=i1[s:s + BATCH], img2=i2[s:s + BATCH], label=lab[s:s + BATCH],
                   idx=torch.arange(s, min(s + BATCH, N_PAIRS))) for s in range(0, N_PAIRS, BATCH)]
    t0 = time.time()
    pred_new, pred = m_lfw.calculate_distance(loader, enc, rec, use_gpu=False)
    t_embed = time.time() - t0
    folds = m_lfw.KFold(n=N_PAIRS, n_folds=10, shuffle=False)
    t0 = time.time()
    res_new = [m_lfw.get_fold_accuracy(fd, pred_new, 1) for fd in folds]
    res = [m_lfw.get_fold_accuracy(fd, pred, 0) for fd in folds]
    t_fold = time.time() - t0
    acc_new = sum(a for _, a in res_new) / 10        # lfw_eval.py:268
    acc = sum(a for _, a in res) / 10
    np.savez_compressed(
        os.path.join(HERE, 'g9_lfw_protocol.npz'),
        n_pairs=np.int64(N_PAIRS), batch=np.int64(BATCH), seed=np.int64(SEED), block=np.int64(BLOCK),
        input_checksum=np.float64(i1.double().sum().item() + i2.double().sum().item()),
        scores_new=pred_new[:, 0], scores=pred[:, 0], labels=pred[:, 1], idx=pred[:, 2],
        best_thr_new=np.array([r[0] for r in res_new]), test_acc_new=np.array([r[1] for r in res_new]),
        best_thr=np.array([r[0] for r in res]), test_acc=np.array([r[1] for r in res]),
        acc_new=np.float64(acc_new), acc=np.float64(acc),
        ref_cpu_seconds=np.array([t_embed, t_fold]))
    print('G9: acc_new %.6f acc %.6f  embed %.1f s (%.1f img/s)  folds %.1f s'
          % (acc_new, acc, t_embed, 2 * N_PAIRS / t_embed, t_fold))
    print('  thr_new', [round(float(r[0]), 3) for r in res_new])
    print('  thr    ', [round(float(r[0]), 3) for r in res])


if __name__ == '__main__':
    main()
