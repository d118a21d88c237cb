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
    """lfw_eval.py:246,248 in stock torch ops: the EXPLICIT fallback for callers that bring their own
    embed function (e.g. CPU tensors in the gloo tests).  With an Engine the scores come from the library
    (ffr_cosine_scores)."""
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


def engine_of(embed_fn):
    """The native Engine behind an embed function (Engine.embed, GraphedEmbed, pair_embed), or None."""
    from .native import Engine, GraphedEmbed
    if isinstance(embed_fn, Engine):
        return embed_fn
    if isinstance(embed_fn, GraphedEmbed):
        return embed_fn.engine
    owner = getattr(embed_fn, '__self__', None)
    if isinstance(owner, Engine):
        return owner
    if isinstance(owner, GraphedEmbed):
        return owner.engine
    return getattr(embed_fn, 'engine', None)


class pair_embed(object):
    """embed function of an (encoder, recnet) pair as the reference calls them back to back
    (lfw_eval.py:240-244):  img -> (f_new, f).

    With the two drop-in shells (ffrnet_amd.Backbone / RecNet) both weight sets live in ONE native handle and a
    batch runs as one ffr_embed call (the NCHW featmap between the modules is never formed); any other pair of
    callables is simply called in the reference's order."""

    def __init__(self, encoder, recnet):
        from .modules import Backbone, RecNet
        self.encoder, self.recnet = encoder, recnet
        self.native = isinstance(encoder, Backbone) and isinstance(recnet, RecNet)
        self.engine = None

    def _bind(self, device):
        eng = self.encoder._engine(device)              # (re)loads the encoder weights when they changed
        rsig, keep = self.recnet._signature(device)
        # what the HANDLE holds decides, not what this object loaded last: another pair_embed over the same Backbone
        # shell, or a direct eng.load_recnet, may have replaced the RecNet weights in between (load_recnet clears the
        # signature)
        if eng._recnet_sig != rsig:
            eng.load_recnet(self.recnet.state_dict())
            eng._recnet_sig, eng._recnet_keep = rsig, keep    # `keep`: the tensors stay alive, so their ids stay unique
        self.engine = eng
        return eng

    def bind(self, device):
        if self.native:
            self._bind(device)
        return self

    def __call__(self, img):
        if not self.native:
            featmap, f = self.encoder(img)
            f_new, _ = self.recnet(featmap)
            return f_new, f
        if self.encoder.training or self.recnet.training:
            raise NotImplementedError('ffrnet_amd.lfw: verification runs the eval() forward (train.py:103-104); '
                                      'call encoder.eval() / recnet.eval() first')
        self.encoder._require_native(img, 'lfw.calculate_distance')
        return self._bind(img.device).embed(img)


def _embed_of(encoder, recnet=None):
    if recnet is None:
        return encoder                                   # already an embed function
    return pair_embed(encoder, recnet)


def target_device(embed_fn):
    """The ROCm device an embed function computes on: the Engine's own device (never merely the CURRENT device: a
    rank that forgot torch.cuda.set_device would otherwise feed cuda:0 to an Engine on cuda:r), for the two shells
    the device their parameters live on, else the current device like the reference's `.cuda()` (lfw_eval.py:236-238).
    None for foreign embed functions (they get the tensors as the loader made them)."""
    eng = engine_of(embed_fn)
    if eng is not None:
        return eng.device
    if getattr(embed_fn, 'native', False):
        for p in embed_fn.encoder.parameters():
            if p.is_cuda:
                return p.device
            break
        return torch.device('cuda', torch.cuda.current_device())
    return None


class ShardFeeder(object):
    """The input side of calculate_distance (replaces lfw_eval.py:236-238's three `.cuda()` calls).

    For every pair batch of the loader: take THIS rank's contiguous shard `[lo:hi]` of img1 and img2 ON THE HOST, put
    the two slices behind each other in a pinned staging buffer and copy that one block to `device` on a side stream
    (non-blocking), one batch ahead of the consumer -- the copy of batch k+1 runs under the kernels of batch k, and a
    rank moves only its own 2*(hi-lo) images instead of the whole pair batch (8 x less host traffic at 8 GPUs).
    Tensors that already live on the device are sliced there (and must be on `device`).  `device=None`: no copies at
    all (foreign embed functions, CPU tests).  Yields (data, both[2m,3,H,W], m, n).

    `stats`: batches, shard_bytes (bytes of image data handed to the embed function), h2d_bytes (bytes this rank
    copied host -> device), full_batch_bytes (what copying whole pair batches would have moved)."""

    def __init__(self, data_loader, rank=0, world=1, device=None, prefetch=True):
        self.loader, self.rank, self.world, self.device = data_loader, rank, world, device
        self.prefetch = prefetch and device is not None
        self.stats = dict(batches=0, shard_bytes=0, h2d_bytes=0, full_batch_bytes=0)
        self._stage = [None, None]       # pinned host buffers
        self._free = [None, None]        # event after which a staging buffer may be overwritten
        self._side = None

    def _produce(self, k, data):
        """Slice, stage and start the copy of one batch; -> (data, both, m, n, ready_event | None)."""
        img1, img2 = data['img1'], data['img2']
        n = img1.size(0)
        lo, hi = shard_bounds(n, self.rank, self.world)
        m = hi - lo
        st = self.stats
        st['batches'] += 1
        st['full_batch_bytes'] += (img1.numel() + img2.numel()) * img1.element_size()
        a, b = img1[lo:hi], img2[lo:hi]                   # the shard is cut BEFORE anything is copied anywhere
        st['shard_bytes'] += (a.numel() + b.numel()) * a.element_size()
        if m == 0:
            return data, None, 0, n, None
        dev = self.device
        if dev is None:
            return data, torch.cat((a, b), 0), m, n, None
        if a.is_cuda or b.is_cuda:
            for t, nm in ((a, 'img1'), (b, 'img2')):
                if not t.is_cuda or t.device.index != dev.index:
                    raise RuntimeError('ffrnet_amd.lfw: %s is on %s but the embed function computes on %s'
                                       % (nm, t.device, dev))
            return data, torch.cat((a, b), 0), m, n, None
        # host tensors: one pinned block [2m,...], one asynchronous copy on the side stream
        slot = k & 1
        shape = (2 * m,) + tuple(a.shape[1:])
        need = 2 * m * a[0].numel()
        buf = self._stage[slot]
        if buf is None or buf.numel() < need or buf.dtype != a.dtype:
            buf = torch.empty(need, dtype=a.dtype, pin_memory=True)
            self._stage[slot] = buf
        elif self._free[slot] is not None:
            self._free[slot].synchronize()                # the copy that last read this buffer (two batches ago) is done
        host = buf[:need].view(shape)
        host[:m].copy_(a)
        host[m:].copy_(b)
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._side):
            both = host.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._free[slot] = ev
        st['h2d_bytes'] += need * a.element_size()
        return data, both, m, n, ev

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        try:
            cur = self._produce(k, next(it))
        except StopIteration:
            return
        while cur is not None:
            nxt = None
            if self.prefetch:                             # start batch k+1's copy before batch k is consumed
                try:
                    nxt = self._produce(k + 1, next(it))
                except StopIteration:
                    nxt = False
            data, both, m, n, ev = cur
            if ev is not None:
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                both.record_stream(main)                  # allocated on the side stream, consumed on this one
            yield data, both, m, n
            if nxt is None:
                try:
                    nxt = self._produce(k + 1, next(it))
                except StopIteration:
                    nxt = False
            cur = nxt or None
            k += 1


last_feed_stats = None      # ShardFeeder.stats of the most recent calculate_distance call in this process


def calculate_distance(data_loader, encoder, recnet=None, flag=0, use_flip=False, use_gpu=True, group=None,
                       score_fn=None, device_scores=False):
    """lfw_eval.calculate_distance(data_loader, encoder, recnet, flag, use_flip, use_gpu), lfw/lfw_eval.py:226-252.

    `encoder, recnet`: the two modules (the drop-in shells run natively, fused), or ONE embed function
    `embed_fn(img[N,3,112,112]) -> (f_new[N,512], f[N,512])` in `encoder`'s place (Engine.embed, GraphedEmbed)
    with recnet=None.  data_loader yields dicts with img1, img2, label, idx (data/dataset.py:84-88); with `use_gpu`
    host images go to the device the embed function computes on (the reference's `.cuda()`, :236-238) -- through
    ShardFeeder: shard first, pinned staging, the next batch's copy under this batch's kernels.

    With an initialised process group each rank embeds its contiguous shard of every pair batch and ONE
    all-gather (RCCL) of the packed [f1_new | f2_new | f1 | f2] rows precedes scoring, so every rank returns the
    full arrays.  Scores: `ffr_cosine_scores` whenever a native Engine is behind the embed function; the torch
    formula only for foreign embed functions (or an explicit `score_fn`).
    Returns two [n_pairs,3] float64 arrays (score, label, idx): (f_new based, f based); with device_scores=True
    additionally the two fp32 score vectors still on the device.
    """
    global last_feed_stats
    embed_fn = _embed_of(encoder, recnet)
    rank = dist.get_rank(group) if (dist and dist.is_initialized()) else 0
    world = dist.get_world_size(group) if (dist and dist.is_initialized()) else 1
    s_new, s_old, labels, idxs = [], [], [], []
    dev = target_device(embed_fn) if use_gpu else None
    if dev is not None and hasattr(embed_fn, 'bind'):
        embed_fn.bind(dev)            # both weight sets in the handle BEFORE the loop: a rank whose first shard is
    eng = engine_of(embed_fn)         # empty scores with the same function as every other rank
    feeder = ShardFeeder(data_loader, rank, world, dev)
    last_feed_stats = feeder.stats
    for data, both, m, n in feeder:
        if m:
            f_new, f = embed_fn(both)
            e = torch.cat((f_new[:m], f_new[m:], f[:m], f[m:]), 1)        # [m, 4*512]
        else:
            e = torch.zeros((0, 2048), dtype=torch.float32, device=dev if dev is not None else data['img1'].device)
        e = all_gather_rows(e, n, group)
        d = e.size(1) // 4
        fn = score_fn
        if fn is None:
            fn = eng.cosine_scores if (eng is not None and e.is_cuda) else cosine_scores_torch
        s_new.append(fn(e[:, :d], e[:, d:2 * d]))
        s_old.append(fn(e[:, 2 * d:3 * d], e[:, 3 * d:]))
        labels.append(torch.as_tensor(data['label']).reshape(-1).double().cpu())
        idxs.append(torch.as_tensor(data['idx']).reshape(-1).double().cpu())
    if not s_new:                                # empty loader: the reference returns empty arrays as well
        z = np.zeros((0, 3))
        return (z, z.copy(), torch.zeros(0), torch.zeros(0)) if device_scores else (z, z.copy())
    dev_new, dev_old = torch.cat(s_new), torch.cat(s_old)
    h_new = dev_new.double().cpu().numpy()      # one device->host sync at the end
    h_old = dev_old.double().cpu().numpy()
    lab, idx = torch.cat(labels).numpy(), torch.cat(idxs).numpy()
    out = (np.array([h_new, lab, idx]).T, np.array([h_old, lab, idx]).T)
    if device_scores:
        return out + (dev_new.float(), dev_old.float())
    return out


def get_avg_accuracy(encoder, recnet=None, data_loader=None, flag=0, verbose=False, group=None, n_folds=10,
                     details=False):
    """lfw_eval.get_avg_accuracy(encoder, recnet, data_loader, flag, verbose), lfw/lfw_eval.py:272-287
    -> (avg_acc_new, avg_acc).  Also callable as get_avg_accuracy(embed_fn, data_loader).

    The ten `get_fold_accuracy` calls the reference spreads over a process pool (:276-283; forking after HIP
    initialisation is unsafe) run as ONE device launch per score vector (`ffr_lfw_fold_accuracy`: same thresholds,
    same `>` / last-argmax rules, bit-equal numbers) whenever a native Engine is behind the embed function; the
    numpy restatement above serves foreign embed functions."""
    if data_loader is None:                              # (embed_fn, data_loader)
        encoder, recnet, data_loader = encoder, None, recnet
    embed_fn = _embed_of(encoder, recnet)
    pred_new, pred, dev_new, dev_old = calculate_distance(data_loader, embed_fn, None, flag, group=group,
                                                          device_scores=True)
    eng = engine_of(embed_fn)
    if eng is not None and dev_new.is_cuda:
        lab = torch.from_numpy(pred_new[:, 1].astype(np.int32))
        acc_new, res_new = eng.lfw_fold_accuracy(dev_new, lab, n_folds)
        acc, res = eng.lfw_fold_accuracy(dev_old, lab, n_folds)
    else:
        acc_new, res_new = get_accuracy_from_predicts(pred_new, n_folds)
        acc, res = get_accuracy_from_predicts(pred, n_folds)
    if verbose:
        for name, r, a in (('f_new', res_new, acc_new), ('f', res, acc)):
            for thr, ta in r:
                print('Best threshold: {:.4f}; Test accuracy: {:.4f}'.format(thr, ta))
            print('Average accuracy ({}): {}'.format(name, a))
    if details:
        return acc_new, acc, dict(pred_new=pred_new, pred=pred, folds_new=res_new, folds=res)
    return acc_new, acc
