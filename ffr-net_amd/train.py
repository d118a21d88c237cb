"""Training harness around the native RecNet training step: the counterpart of models/trainer.py.

  Trainer.forward              models/trainer.py:139-152   encoder (frozen, eval) on the clean and the occluded
                                                           image, RecNet (train-mode BatchNorm) on both
  Trainer.backward             models/trainer.py:154-180   the four loss items
  Trainer.optimizer_parameters models/trainer.py:182-187   zero_grad, backward, clip_grad_value_(1.0), Adam
  nn.parallel.data_parallel    models/trainer.py:70-72     replaced by one process per GPU and ONE all-reduce of the
                                                           flat fp32 gradient buffer (RCCL over xGMI); BatchNorm
                                                           statistics stay per replica, as in the reference

What runs where: everything of an iteration is native (ffr-net_amd/csrc/train*.{cpp,hip}, wgrad.hip): encoder
forward, RecNet forward, the four loss items and their gradients, the whole RecNet backward, gradient clipping
and Adam; torch only carries the device buffers and the all-reduce.  `trainer_losses` below restates the loss
items with torch ops on the device: `NativeTrainer.step_torch_losses` uses it as the cross-check of the native
loss kernels.  There is no CPU path.
"""
import torch
import torch.nn.functional as F

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None

TRIPLET_MARGIN = 0.1     # models/trainer.py:39


def cosine_sim(x1, x2):
    """models/recnet.py:220-224."""
    return torch.bmm(F.normalize(x1, dim=2), F.normalize(x2, dim=2).permute(0, 2, 1))


def self_similarity(x):
    """selfSimilarity, models/recnet.py:226-236 -> (ss_space [N,HW,H,W], ss_channel [N,C,C])."""
    n, c, h, w = x.shape
    v = x.reshape(n, c, h * w)
    vt = v.permute(0, 2, 1)
    return cosine_sim(vt, vt).reshape(n, h * w, h, w), cosine_sim(v, v)


def triplet_loss(x, y, z):
    """TripletLoss.forward, models/trainer.py:38-43."""
    pos = 1 - torch.sum(F.normalize(x) * F.normalize(y), 1)
    neg = 1 - torch.sum(F.normalize(x) * F.normalize(z), 1)
    return F.relu((pos - neg) + TRIPLET_MARGIN).mean(), pos.mean(), neg.mean()


def trainer_losses(f_non, f_ocl, pred_loss_non, pred_loss_ocl, space_non, space_ocl, channel_non, channel_ocl,
                   feat_map_non, f_enc_non, f_enc_ocl, label, loss_weight=(1, 1, 1, 1)):
    """The four weighted loss items of Trainer.backward, models/trainer.py:154-178."""
    ss_space, ss_channel = self_similarity(feat_map_non)
    ss_space_non, _ = self_similarity(space_non)
    ss_space_ocl, _ = self_similarity(space_ocl)
    _, ss_channel_non = self_similarity(channel_non)
    _, ss_channel_ocl = self_similarity(channel_ocl)
    mse = F.mse_loss
    l_space = (mse(ss_space, ss_space_non) + mse(ss_space, ss_space_ocl)) / 2
    l_channel = (mse(ss_channel, ss_channel_non) + mse(ss_channel, ss_channel_ocl)) / 2
    items = [(l_space + l_channel) / 2,
             triplet_loss(f_ocl, f_enc_non, f_enc_ocl)[0],
             (mse(f_non, f_enc_non) + mse(f_ocl, f_enc_non)) / 2,
             F.cross_entropy(pred_loss_non, label) / (1e-8 + loss_weight[3]) + F.cross_entropy(pred_loss_ocl, label)]
    return [l * w for l, w in zip(items, loss_weight)]


def average_gradients(flat, group=None):
    """Data-parallel gradient exchange: ONE all-reduce of the flat fp32 gradient buffer (29.9 M floats,
    119.7 MB for RecNet), then the mean over the ranks -- what gathering the outputs on one device and
    calling backward does in nn.parallel.data_parallel (models/trainer.py:70-72) for mean-reduced losses."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, group=group)
        flat.div_(world)
    return flat


class FlatBuffer(object):
    """Zero-copy torch view of a native device buffer (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


class NativeTrainer(object):
    """One training iteration of train.py:46-54 on the native path.

    engine: ffrnet_amd.Engine with the encoder loaded; recnet_state_dict: the 121-entry RecNet state_dict.
    Hyper-parameters as run.py / utils/options.py: Adam lr, (beta1, beta2), weight_decay, loss_weight;
    clip_value 1.0 (models/trainer.py:183).  With an initialised process group the flat gradient buffer is
    averaged over the ranks before the optimiser step.
    """

    def __init__(self, engine, recnet_state_dict, lr=0.1, betas=(0.9, 0.999), weight_decay=0.0,
                 loss_weight=(1, 1, 1, 1), clip_value=1.0, group=None):
        if not engine.has_encoder:
            raise RuntimeError('ffrnet_amd: load the encoder before building a NativeTrainer')
        self.engine = engine
        self.lr, self.betas, self.weight_decay = lr, betas, weight_decay
        self.loss_weight, self.clip_value, self.group = tuple(loss_weight), clip_value, group
        engine.train_init(recnet_state_dict)
        info = engine.train_info()
        self._grads = torch.as_tensor(FlatBuffer(info['grads'], info['n_flat']), device=engine.device)
        self._params = torch.as_tensor(FlatBuffer(info['params'], info['n_flat']), device=engine.device)
        self.loss_items = None
        self.accuracy = None

    @property
    def flat_grads(self):
        return self._grads

    @property
    def flat_params(self):
        return self._params

    def world_size(self):
        if dist is None or not dist.is_available() or not dist.is_initialized():
            return 1
        return dist.get_world_size(self.group)

    def broadcast_params(self, src=0):
        """Same initial weights on every rank (the reference replicates the module on every call)."""
        if self.world_size() > 1:
            dist.broadcast(self._params, src, group=self.group)

    def step(self, img_non, img_ocl, label):
        """Trainer.set_input + forward + optimizer_parameters (models/trainer.py:131-187), everything native:
        one launch-only call up to the gradients (ffr_train_iteration), the gradient exchange, clip + Adam.
        Returns the four weighted loss items as a device tensor view (no host sync)."""
        out5 = self.engine.train_iteration(img_non, img_ocl, label, self.loss_weight)
        average_gradients(self._grads, self.group)
        self.engine.train_adam_step(self.lr, self.betas, 1e-8, self.weight_decay, self.clip_value)
        self.loss_items = [out5[i] for i in range(4)]
        self.accuracy = out5[4]
        return self.loss_items

    def step_torch_losses(self, img_non, img_ocl, label):
        """The same iteration with the four loss items evaluated by torch ops (`trainer_losses`) and their autograd
        cotangents handed to ffr_train_backward: the cross-check of the native loss kernels."""
        eng = self.engine
        n = img_non.size(0)
        with torch.no_grad():
            fm, f_enc = eng.encoder_forward(torch.cat((img_non, img_ocl), 0))
        label = label.to(fm.device)
        outs = eng.train_forward(fm, torch.cat((label, label)), groups=2,
                                 want=('f_new', 'pred_loss', 'pred_label', 'feat_space', 'feat_channel'))
        f_new, pred_loss, pred_label, _, _, feat_space, feat_channel = outs
        leaves = [t.detach().requires_grad_(True) for t in (f_new, pred_loss, feat_space, feat_channel)]
        lf, lp, ls, lc = leaves
        items = trainer_losses(lf[:n], lf[n:], lp[:n], lp[n:], ls[:n], ls[n:], lc[:n], lc[n:], fm[:n], f_enc[:n],
                               f_enc[n:], label.long(), self.loss_weight)
        torch.autograd.backward(sum(items))
        eng.train_zero_grad()
        eng.train_backward([lf.grad, lp.grad, None, None, None, ls.grad, lc.grad])
        average_gradients(self._grads, self.group)
        eng.train_adam_step(self.lr, self.betas, 1e-8, self.weight_decay, self.clip_value)
        self.accuracy = (pred_label[n:].argmax(1) == label).float().mean()
        self.loss_items = [l.detach() for l in items]
        return self.loss_items

    def state_dict(self):
        return self.engine.train_state_dict()
