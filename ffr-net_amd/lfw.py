"""Verification harness around the hot path: the counterpart of lfw/lfw_eval.py.

  calculate_distance  lfw/lfw_eval.py:226-252  (same call order, cosine formula, rows)
  KFold / eval_acc / find_best_threshold / get_fold_accuracy / get_avg_accuracy
                      lfw/lfw_eval.py:110-118,137-162,255-287

Differences from the reference, by design: pair batches can be sharded over the ranks of
a torch.distributed group (one process per GPU, RCCL all-gather of the 512-d embeddings
over xGMI; the reference is single GPU here), scores are computed on device once per
shard without a `.tolist()` sync per batch, and the threshold sweep is vectorised and
runs in-process (the reference forks 10 workers, which is unsafe after HIP init).
Tie rules are kept exactly: same iff score > thr (:142), best thr = LAST argmax (:159).
"""
import numpy as np
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def KFold(n=6000, n_folds=10, shuffle=False):
    if shuffle:
        raise NotImplementedError('the reference only ever calls KFold(shuffle=False)')
    idx = np.arange(n)
    folds = []
    for i in range(n_folds):
        lo, hi = i * n // n_folds, (i + 1) * n // n_folds
        folds.append([np.concatenate((idx[:lo], idx[hi:])), idx[lo:hi]])
    return folds


THRESHOLDS = np.arange(-1.0, 1.0, 0.005)


def eval_acc(threshold, diff):
    diff = np.asarray(diff)
    pred = diff[:, 0].astype(np.float64) > threshold
    true = diff[:, 1].astype(np.int64) == 1
    return 1.0 * np.count_nonzero(pred == true) / len(true)


def _acc_table(diff, thresholds):
    """accuracy of every threshold on the rows of diff: [len(thresholds)]"""
    s = diff[:, 0].astype(np.float64)
    y = diff[:, 1].astype(np.int64) == 1
    pred = s[None, :] > thresholds[:, None]
    return np.count_nonzero(pred == y[None, :], axis=1) * 1.0 / len(y)


def find_best_threshold(thresholds, predicts):
    thresholds = np.asarray(thresholds)
    acc = _acc_table(np.asarray(predicts), thresholds)
    # the reference starts from best_acc = 0 and updates on >=: last index of the maximum
    best = len(acc) - 1 - int(np.argmax(acc[::-1]))
    return thresholds[best]


def get_fold_accuracy(fold, predicts):
    best = find_best_threshold(THRESHOLDS, predicts[fold[0]])
    return best, eval_acc(best, predicts[fold[1]])


def get_accuracy_from_predicts(predicts, n_folds=10):
    """-> (mean accuracy, [(best_thr, test_acc)] per fold).  The reference fixes
    n=6000, 10 folds and divides by a literal 10 (lfw_eval.py:268,274)."""
    predicts = np.asarray(predicts)
    res = [get_fold_accuracy(f, predicts) for f in KFold(len(predicts), n_folds)]
    return sum(a for _, a in res) / n_folds, res


# ---------------------------------------------------------------------------------
def cosine_scores_torch(a, b):
    """lfw_eval.py:246,248 on whatever device a, b live (used on gathered embeddings)."""
    return torch.sum(a * b, dim=1) / (a.norm(dim=1) * b.norm(dim=1) + 1e-8)


def shard_bounds(n, rank, world):
    """contiguous split along dim 0, like data_parallel's scatter (models/trainer.py:70)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_rows(t, n_total, group=None):
    """All-gather row shards of equal padded size and trim: [n_total, ...]."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return t
    world = dist.get_world_size(group)
    if world == 1:
        return t
    per = (n_total + world - 1) // world
    if t.size(0) < per:
        pad = torch.zeros((per - t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat((t, pad), 0)
    out = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out[:n_total]


def calculate_distance(data_loader, embed_fn, group=None, score_fn=None):
    """Counterpart of lfw_eval.calculate_distance(data_loader, encoder, recnet).

    data_loader yields dicts with img1, img2, label, idx (data/dataset.py:84-88).
    embed_fn(img[N,3,112,112]) -> (f_new[N,512], f[N,512]) is the hot path
    (Engine.embed, or encoder+recnet shells).  With an initialised process group each
    rank embeds its contiguous shard of every pair batch and the embeddings are
    all-gathered (RCCL) before scoring, so every rank returns the full arrays.
    Returns two [n_pairs,3] float64 arrays (score, label, idx): (f_new based, f based).
    """
    score_fn = score_fn or cosine_scores_torch
    rank = dist.get_rank(group) if (dist and dist.is_initialized()) else 0
    world = dist.get_world_size(group) if (dist and dist.is_initialized()) else 1
    s_new, s_old, labels, idxs = [], [], [], []
    for data in data_loader:
        img1, img2 = data['img1'], data['img2']
        n = img1.size(0)
        lo, hi = shard_bounds(n, rank, world)
        if hi > lo:
            both = torch.cat((img1[lo:hi], img2[lo:hi]), 0)
            f_new, f = embed_fn(both)
            m = hi - lo
            e = torch.cat((f_new[:m], f_new[m:], f[:m], f[m:]), 1)        # [m, 4*512]
        else:
            e = torch.zeros((0, 2048), dtype=torch.float32, device=img1.device)
        e = all_gather_rows(e, n, group)
        d = e.size(1) // 4
        s_new.append(score_fn(e[:, :d], e[:, d:2 * d]))
        s_old.append(score_fn(e[:, 2 * d:3 * d], e[:, 3 * d:]))
        labels.append(torch.as_tensor(data['label']).reshape(-1).double().cpu())
        idxs.append(torch.as_tensor(data['idx']).reshape(-1).double().cpu())
    s_new = torch.cat(s_new).double().cpu().numpy()      # one device->host sync at the end
    s_old = torch.cat(s_old).double().cpu().numpy()
    lab, idx = torch.cat(labels).numpy(), torch.cat(idxs).numpy()
    return np.array([s_new, lab, idx]).T, np.array([s_old, lab, idx]).T


def get_avg_accuracy(embed_fn, data_loader, group=None, n_folds=10):
    """-> (avg_acc_new, avg_acc), as lfw_eval.get_avg_accuracy(encoder, recnet, loader)."""
    pred_new, pred = calculate_distance(data_loader, embed_fn, group)
    return (get_accuracy_from_predicts(pred_new, n_folds)[0],
            get_accuracy_from_predicts(pred, n_folds)[0])
