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
and Adam; torch only carries the device buffers and the all-reduce.  There is no CPU path and no torch-op path
(the torch restatement of the loss items that cross-checks the native loss kernels lives in tests/torch_losses.py).
"""
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None

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


def average_gradients_overlapped(engine, flat, comm_stream, group=None):
    """The same exchange, bucket by bucket in the order the backward finishes them (classifier 21.8 MB, Conv4Merge
    47 MB, ChannelFlipMerge 38 MB, Conv4Channel 0.2 MB, Conv4Space 12 MB), each all-reduce enqueued on `comm_stream`
    behind a device-side wait for its bucket: RCCL moves the first 100 MB over xGMI while the backward is still
    computing the rest.  Called right after the iteration has been enqueued; the compute stream then waits for
    `comm_stream`.  The reduction order inside a bucket is RCCL's (deterministic for a fixed world size)."""
    world = dist.get_world_size(group)
    cur = torch.cuda.current_stream(flat.device)
    with torch.cuda.stream(comm_stream):
        for bid, off, cnt in engine.train_buckets():
            engine.train_bucket_wait(bid, comm_stream)
            piece = flat[off:off + cnt]
            dist.all_reduce(piece, group=group)
            piece.div_(world)
    cur.wait_stream(comm_stream)
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
                 loss_weight=(1, 1, 1, 1), clip_value=1.0, group=None, overlap=True):
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
        # gradient exchange overlapped with the backward (second stream, per-bucket events) when there are ranks
        self.overlap = overlap
        self._comm = None

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
        if self.overlap and self.world_size() > 1:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.engine.device)
            average_gradients_overlapped(self.engine, self._grads, self._comm, self.group)
        else:
            average_gradients(self._grads, self.group)
        self.engine.train_adam_step(self.lr, self.betas, 1e-8, self.weight_decay, self.clip_value)
        self.loss_items = [out5[i] for i in range(4)]
        self.accuracy = out5[4]
        return self.loss_items

    def state_dict(self):
        return self.engine.train_state_dict()

    def optimizer_state_dict(self):
        """Adam's state for a checkpoint (the 'optimizer' entry of Trainer.save_model, models/trainer.py:216-224):
        step count and both moment buffers per parameter, in torch layout on the host."""
        eng = self.engine
        keys = [k for k in eng._train_spec if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
        return {'kind': 'ffrnet_amd.NativeTrainer/adam', 'step': eng.train_info()['adam_step'],
                'lr': self.lr, 'betas': tuple(self.betas), 'weight_decay': self.weight_decay,
                'exp_avg': {k: eng.train_get(k, 'exp_avg') for k in keys},
                'exp_avg_sq': {k: eng.train_get(k, 'exp_avg_sq') for k in keys}}

    def load_optimizer_state_dict(self, state):
        if state.get('kind') != 'ffrnet_amd.NativeTrainer/adam':
            raise ValueError('not a NativeTrainer optimizer state')
        eng = self.engine
        for which in ('exp_avg', 'exp_avg_sq'):
            for k, v in state[which].items():
                eng.train_set(k, v, which)
        eng.train_option('adam_step', int(state['step']))
