"""Drop-in nn.Module shells for the reference's two hot-path modules.

    Backbone(num_layers, drop_ratio, mode)      pretrain/model_ir_se50.py:108-141
    ir_se_50_512(weights_path)                  pretrain/model_ir_se50.py:143-154
    RecNet(channel, shape, norm_type, relu_type) models/recnet.py:347-429

Same constructor signatures, same forward signatures / return tuples and the SAME
state_dict key set (402 / 121 entries), so se50.pth / FFRNet.pth style checkpoints and
`net.load_state_dict(other.state_dict())` (models/trainer.py:98-113) work unchanged.
The parameter tree below only HOLDS weights; all arithmetic of forward() runs in
libffrnet_hip.so (hand-written gfx950 kernels) through ffrnet_amd.native.Engine.
The packed / BN-folded device copy is a cache owned by the native handle and is
rebuilt whenever a parameter or buffer changes: every forward compares the identity and
version counter of each tensor in the tree with what was packed (load_state_dict, .to(),
in-place edits, a Parameter replaced on a sub-module).

Backbone covers every configuration the reference defines (num_layers 50 / 100 / 152, mode 'ir' / 'ir_se'); it is eval-only (the reference never trains it, models/trainer.py:62-63); RecNet runs
natively in eval() (label=None -> 2-tuple) and in train() mode (label given -> the 7-tuple,
differentiable through the native backward).  There is no CPU or stock-torch fallback anywhere
in this package.
"""
import torch
import torch.nn as nn

from .native import Engine
from .synth import ir_blocks


def l2_norm(input, axis=1):
    """pretrain/model_ir_se50.py:13-16 (host helper, no eps)."""
    return input / torch.norm(input, 2, axis, True)


# ---- parameter holders (names fixed by the state_dict key contract) ---------------
class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('ffrnet_amd: sub-modules only hold weights; call the top-level '
                           'Backbone / RecNet, whose forward runs in the HIP library')


class Flatten(_Holder):
    pass


class SEModule(_Holder):
    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, 1, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, 1, bias=False)


class bottleneck_IR_SE(_Holder):
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(
                nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(
            nn.BatchNorm2d(in_channel),
            nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
            nn.PReLU(depth),
            nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
            nn.BatchNorm2d(depth),
            SEModule(depth, 16))


class bottleneck_IR(_Holder):
    """pretrain/model_ir_se50.py:38-54: bottleneck_IR_SE without the SEModule (Backbone mode 'ir')."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(
                nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(
            nn.BatchNorm2d(in_channel),
            nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
            nn.PReLU(depth),
            nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
            nn.BatchNorm2d(depth))


class _NativeModule(nn.Module):
    """Shared cache logic: one Engine per device, reloaded when any tensor changed."""
    _kind = None

    def _tensors(self):
        """Every parameter and buffer tensor of the tree (the 402 / 121 state_dict entries), read straight from the
        sub-modules' `_parameters` / `_buffers` dicts: 0.1 ms, where building a state_dict per forward cost 0.9 ms
        (measured) -- more than a small-batch forward.  Because the walk is repeated on every forward, replacing a
        Parameter or buffer on ANY sub-module (`sub.weight = nn.Parameter(...)`, `load_state_dict(assign=True)`,
        pruning, parametrisation) changes the identities below and is seen."""
        out = []
        for m in self.modules():
            for t in m._parameters.values():
                if t is not None:
                    out.append(t)
            for t in m._buffers.values():
                if t is not None:
                    out.append(t)
        return out

    def invalidate_native_cache(self):
        """Force a re-pack on the next forward (never needed for edits made through the module tree)."""
        self.__dict__.pop('_native_cache', None)

    def _signature(self, device):
        """(device, identity and version of every tensor).  The cache entry keeps the tensor list itself alive, so an
        id() cannot be reused by a new tensor while it is part of a stored signature."""
        lst = self._tensors()
        return (device.index, tuple(id(t) for t in lst), tuple(t._version for t in lst)), lst

    def _engine(self, device):
        sig, lst = self._signature(device)
        cache = self.__dict__.setdefault('_native_cache', {})
        ent = cache.get(device.index)
        if ent is None or ent[0] != sig:
            eng = ent[1] if ent is not None else Engine(device.index)
            getattr(eng, 'load_' + self._kind)(self.state_dict())
            cache[device.index] = (sig, eng, lst)
            ent = cache[device.index]
        return ent[1]

    def _reject_replica(self, what):
        if getattr(self, '_is_replica', False):
            # nn.parallel.data_parallel / DataParallel (models/trainer.py:70-72) replicate the shell per device IN THREADS:
            # a replica's tensors are new objects on every call, so the packed native copy would be rebuilt (43.8 M
            # weights through the host) per forward and replica.  The MI355X design is one process per GPU.
            raise RuntimeError(
                'ffrnet_amd.%s was called as a torch data_parallel replica.  This package runs ONE PROCESS PER GPU '
                '(torch.distributed over RCCL; lfw.calculate_distance shards the pair batch itself): call the module '
                'directly on its own device and pass gpu_ids of length 1 to the reference Trainer (INTEGRATION.md 1)'
                % what)

    def _require_native(self, x, what):
        self._reject_replica(what)
        if self.training:
            raise NotImplementedError(
                'ffrnet_amd.%s: only the eval() forward is implemented natively (the reference '
                'keeps the encoder in eval and verifies with recnet.eval(), models/trainer.py:75-79); '
                'call .eval() first' % what)
        if not (torch.is_tensor(x) and x.is_cuda):
            raise RuntimeError('ffrnet_amd.%s: input must be a ROCm device tensor; this package has '
                               'no CPU path' % what)
        if x.dtype != torch.float32:
            raise RuntimeError('ffrnet_amd.%s: input must be float32' % what)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == '_native_cache':
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class Backbone(_NativeModule):
    _kind = 'encoder'

    def __init__(self, num_layers, drop_ratio, mode='ir'):
        super().__init__()
        assert num_layers in [50, 100, 152], 'num_layers should be 50,100, or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        self.num_layers, self.mode = num_layers, mode
        unit = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE        # model_ir_se50.py:113-116
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, (3, 3), 1, 1, bias=False),
                                         nn.BatchNorm2d(64), nn.PReLU(64))
        self.output_layer = nn.Sequential(nn.BatchNorm2d(512), nn.Dropout(drop_ratio), Flatten(),
                                          nn.Linear(512 * 7 * 7, 512), nn.BatchNorm1d(512))
        self.bn = nn.BatchNorm2d(512)
        self.body = nn.Sequential(*[unit(c, d, s) for c, d, s in ir_blocks(num_layers)])

    def forward(self, x):
        """-> (featmap[N,512,7,7], l2_norm(feat)[N,512]); model_ir_se50.py:136-141."""
        self._require_native(x, 'Backbone')
        return self._engine(x.device).encoder_forward(x)


def ir_se_50_512(weights_path='./pretrain/se50.pth', **kwargs):
    """model_ir_se50.py:143-154."""
    model = Backbone(num_layers=50, drop_ratio=0.6, mode='ir_se')
    if weights_path:
        model.load_state_dict(torch.load(weights_path, map_location='cpu'))
    return model


class ReluLayer(_Holder):
    def __init__(self, channels, relu_type='relu'):
        super().__init__()
        if relu_type.lower() != 'prelu':
            raise NotImplementedError('ffrnet_amd: only relu_type="prelu" is native')
        self.func = nn.PReLU(channels)


class NormLayer(_Holder):
    def __init__(self, channels, norm_type='bn'):
        super().__init__()
        if norm_type.lower() != 'bn':
            raise NotImplementedError('ffrnet_amd: only norm_type="bn" is native')
        self.norm = nn.BatchNorm2d(channels)


class ConvLayer(_Holder):
    """models/recnet.py:52-85 with norm 'bn' (=> conv bias=False, :57), relu 'prelu'."""

    def __init__(self, in_channels, out_channels, kernel_size=3, norm_type='bn', relu_type='prelu'):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, 1, bias=False)
        self.relu = ReluLayer(out_channels, relu_type)
        self.norm = NormLayer(out_channels, norm_type)


class ResidualBlock(_Holder):
    def __init__(self, inplanes, planes, kernel_size=3, norm_type='bn', relu_type='prelu'):
        super().__init__()
        self.conv1 = ConvLayer(inplanes, planes, kernel_size, norm_type, relu_type)
        self.conv2 = ConvLayer(planes, planes, kernel_size, norm_type, relu_type)


class AddMarginProduct(_Holder):
    """Weight holder of the training-only CosFace head (models/recnet.py:238-270)."""

    def __init__(self, in_features, out_features=10575, s=30.0, m=0.40):
        super().__init__()
        self.in_features, self.out_features, self.s, self.m = in_features, out_features, s, m
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)


class RecNet(_NativeModule):
    _kind = 'recnet'

    def __init__(self, channel=512, shape=7, norm_type='bn', relu_type='prelu'):
        super().__init__()
        if channel != 512 or shape != 7:
            raise NotImplementedError('ffrnet_amd: RecNet is native for channel=512, shape=7 only')
        self.channel, self.shape = channel, shape
        a = dict(norm_type=norm_type, relu_type=relu_type)
        s2 = shape ** 2
        self.Conv4Space = nn.Sequential(
            ConvLayer(channel + s2, 256, **a), ResidualBlock(256, 256, **a),
            ConvLayer(256, 128, **a), ResidualBlock(128, 128, **a),
            ConvLayer(128, s2, **a), ResidualBlock(s2, s2, **a), nn.Sigmoid())
        self.Conv4Channel = nn.Sequential(
            nn.Linear(channel + s2, 32), ReluLayer(512, 'prelu'), nn.Linear(32, channel),
            nn.Linear(channel, 32), ReluLayer(512, 'prelu'), nn.Linear(32, channel),
            nn.Linear(channel, 32), ReluLayer(512, 'prelu'), nn.Linear(32, channel),
            nn.Sigmoid())
        self.ChannelFlipMerge = nn.Sequential(ConvLayer(channel * 2, channel, **a),
                                              ResidualBlock(channel, channel, **a))
        self.Conv4Merge = nn.Sequential(ConvLayer(channel * 3, channel, **a),
                                        ResidualBlock(channel, channel, **a))
        self.pool5_7x7 = nn.AvgPool2d(kernel_size=[7, 7], stride=[1, 1], padding=0)
        self.classifier = AddMarginProduct(channel)

    def forward(self, input, label=None):
        """label=None (eval) -> (f_new[N,512], feat_new[N,512,7,7]), models/recnet.py:398-426.
        label given (train) -> the 7-tuple of models/recnet.py:427-429, differentiable with respect to the
        module's parameters through the native backward (include/ffrnet_train.h)."""
        if label is not None:
            return self._forward_train(input, label)
        self._require_native(input, 'RecNet')
        return self._engine(input.device).recnet_forward(input)

    # -- training branch: RecNet.forward(input, label) in train() mode ------------------------------------
    def _train_engine(self, device):
        """Engine whose training state mirrors this module's parameters: built from state_dict() once per device,
        afterwards only tensors whose version moved (an optimiser step, load_state_dict) are re-imported, on
        the device."""
        cache = self.__dict__.setdefault('_native_cache', {})
        ent = cache.get(('train', device.index))
        sd = self.state_dict(keep_vars=True)
        if ent is None:
            eng = Engine(device.index)
            eng.train_init({k: v.detach() for k, v in sd.items()})
            ent = [eng, {k: (id(v), v._version) for k, v in sd.items()}, 0, [None, None]]
            cache[('train', device.index)] = ent
            return ent
        eng, seen = ent[0], ent[1]
        for k, v in sd.items():
            sig = (id(v), v._version)
            if seen.get(k) == sig or k.endswith('num_batches_tracked'):
                continue
            if k.endswith(('running_mean', 'running_var')):
                eng.train_import(k, v.detach().to(device), 'running')
            else:
                eng.train_import(k, v.detach().to(device), 'param')
            seen[k] = sig
        return ent

    def _forward_train(self, input, label):
        self._reject_replica('RecNet')
        if not self.training:
            raise NotImplementedError('ffrnet_amd.RecNet: forward(input, label) is the train() branch '
                                      '(models/recnet.py:427-429); call .train() first')
        if not (torch.is_tensor(input) and input.is_cuda and input.dtype == torch.float32):
            raise RuntimeError('ffrnet_amd.RecNet: input must be a float32 ROCm device tensor; this package has no CPU path')
        ent = self._train_engine(input.device)
        eng = ent[0]
        # two activation contexts: the reference keeps exactly two forwards alive (clean, occluded) until backward
        ticket = ent[2]
        slot = ticket % 2
        ent[2] += 1
        ent[3][slot] = ticket
        names = [k for k, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        outs = _RecNetTrainFn.apply(eng, (slot, ticket, ent[3]), names, input, label, *params)
        # BatchNorm buffers follow the native running statistics (models/recnet.py:141-143 in train mode)
        with torch.no_grad():
            seen = ent[1]
            for k, b in self.named_buffers():
                if k.endswith(('running_mean', 'running_var')):
                    eng.train_export(k, b.data, 'running')
                elif k.endswith('num_batches_tracked'):
                    b += 1
                seen[k] = (id(b), b._version)
        return outs


class _RecNetTrainFn(torch.autograd.Function):
    """RecNet.forward(input, label) in train() mode with the native backward behind torch.autograd: parameter
    gradients come back in torch layout; the input gets none (the encoder is frozen, models/trainer.py:62-63)."""

    @staticmethod
    def forward(ctx, eng, slot, names, input, label, *params):
        ctx.eng, ctx.names = eng, names
        # unused outputs (M_channel [n,512,512], pred_label [n,10575], ...) must reach backward as None, not as dense
        # zero tensors the native backward would transpose and consume (> 270 MB of memset per iteration at n = 128)
        ctx.set_materialize_grads(False)
        ctx.slot, ctx.ticket, ctx.live = slot
        ctx.shapes = [tuple(p.shape) for p in params]
        outs = eng.train_forward(input.detach(), label, groups=1, slot=ctx.slot)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        if ctx.live[ctx.slot] != ctx.ticket:
            raise RuntimeError('ffrnet_amd.RecNet: the activations of this forward were overwritten -- at most two '
                               'train-mode forwards may be outstanding before backward (models/trainer.py:144-145)')
        eng.train_zero_grad()
        eng.train_backward(list(gouts), slot=ctx.slot)
        grads = []
        dev = eng.device
        for k, shp in zip(ctx.names, ctx.shapes):
            grads.append(eng.train_export(k, torch.empty(shp, device=dev, dtype=torch.float32), 'grad'))
        return (None, None, None, None, None) + tuple(grads)
