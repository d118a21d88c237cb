"""ctypes binding of libffrnet_hip.so (include/ffrnet.h).

PyTorch is used for device memory and streams only: tensors are handed over as raw
device pointers (`tensor.data_ptr()`) and launches go to torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KCLASS_NAMES = ['conv_igemm', 'stem', 'se', 'combine', 'head', 'selfsim', 'channel',
                'space', 'layout', 'score', 'wino', 'wino_fused', 'wgrad',
                'train_bn', 'train_loss', 'train_optim', 'train_xform', 'train_elem']


class NativeLibraryMissing(RuntimeError):
    pass


class TensorDesc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.c_void_p), ('ndim', C.c_int32),
                ('shape', C.c_int64 * 4)]


class MemStats(C.Structure):
    _fields_ = [('encoder_weight_bytes', C.c_size_t), ('recnet_weight_bytes', C.c_size_t), ('mixed_tile_weight_bytes', C.c_size_t),
                ('workspace_bytes', C.c_size_t), ('encoder_load_seconds', C.c_double), ('recnet_load_seconds', C.c_double),
                ('mixed_tile_pack_seconds', C.c_double)]


class KClassStat(C.Structure):
    _fields_ = [('launches', C.c_int64), ('ms', C.c_double), ('flops', C.c_double),
                ('bytes', C.c_double), ('flops_executed', C.c_double), ('flops_useful', C.c_double)]


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('N', C.c_int), ('H', C.c_int), ('W', C.c_int),
                ('in_pitch', C.c_int), ('cin_pad', C.c_int),
                ('w', C.c_void_p), ('bias', C.c_void_p), ('slope', C.c_void_p),
                ('resid', C.c_void_p), ('res_pitch', C.c_int),
                ('out', C.c_void_p), ('out_pitch', C.c_int), ('out_coff', C.c_int),
                ('cout_store', C.c_int), ('cout_pad', C.c_int),
                ('R', C.c_int), ('S', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('pad_mode', C.c_int), ('border_bias', C.c_int), ('flags', C.c_int),
                ('tile', C.c_int), ('splitk', C.c_int)]


_LIB_PATH = os.path.join(_HERE, 'libffrnet_hip.so')


def lib_path():
    return _LIB_PATH


def set_library(path):
    """Use another build of the library (tools/trace_build.py: the -DFFR_TRACE diagnostics build).  Must be called
    before the first Engine is created; the product never calls it."""
    global _LIB_PATH
    if _LIB is not None:
        raise RuntimeError('ffrnet_amd: the native library is already loaded')
    _LIB_PATH = os.path.abspath(path)


# every symbol include/ffrnet.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ('ffr_create', C.c_int, [C.POINTER(_P), C.c_int]),
    ('ffr_destroy', None, [_P]),
    ('ffr_last_error', C.c_char_p, [_P]),
    ('ffr_version', C.c_char_p, []),
    ('ffr_load_encoder', C.c_int, [_P, C.POINTER(TensorDesc), C.c_int]),
    ('ffr_load_recnet', C.c_int, [_P, C.POINTER(TensorDesc), C.c_int]),
    ('ffr_encoder_forward', C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ('ffr_recnet_forward', C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    ('ffr_embed', C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    ('ffr_embed_u8', C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    ('ffr_cosine_scores', C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    ('ffr_lfw_fold_accuracy', C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    ('ffr_workspace_bytes', C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    ('ffr_reserve', C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    ('ffr_generation', C.c_ulonglong, [_P]),
    ('ffr_memory_stats', C.c_int, [_P, _P]),
    ('ffr_set_option', C.c_int, [_P, C.c_char_p, C.c_longlong]),
    ('ffr_get_option', C.c_int, [_P, C.c_char_p, C.POINTER(C.c_longlong)]),
    ('ffr_probe_mfma_peak', C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    ('ffr_profile_enable', C.c_int, [_P, C.c_int]),
    ('ffr_profile_read', C.c_int, [_P, C.POINTER(KClassStat)]),
    ('ffr_op_conv', C.c_int, [_P, C.POINTER(ConvDesc), _P]),
    ('ffr_op_conv3x3', C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ('ffr_encoder_trunk_nhwc', C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    ('ffr_recnet_debug', C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    # include/ffrnet_train.h
    ('ffr_op_convlayer_train', C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ('ffr_train_init', C.c_int, [_P, C.POINTER(TensorDesc), C.c_int]),
    ('ffr_train_info', C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int)]),
    ('ffr_train_get', C.c_int, [_P, C.c_int, C.c_char_p, _P, C.c_size_t]),
    ('ffr_train_set', C.c_int, [_P, C.c_int, C.c_char_p, _P, C.c_size_t]),
    ('ffr_train_zero_grad', C.c_int, [_P, _P]),
    ('ffr_train_forward', C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    ('ffr_train_backward', C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    ('ffr_train_adam_step', C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    ('ffr_train_debug_copy', C.c_int, [_P, C.c_int, C.c_char_p, _P, C.c_size_t]),
    ('ffr_train_option', C.c_int, [_P, C.c_char_p, C.c_int]),
    ('ffr_train_buckets', C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    ('ffr_train_bucket_wait', C.c_int, [_P, C.c_int, _P]),
    ('ffr_train_export', C.c_int, [_P, C.c_int, C.c_char_p, _P, _P]),
    ('ffr_train_import', C.c_int, [_P, C.c_int, C.c_char_p, _P, _P]),
    ('ffr_train_losses', C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_double), _P, _P]),
    ('ffr_train_backward_losses', C.c_int, [_P, C.c_int, _P]),
    ('ffr_train_iteration', C.c_int, [_P, _P, _P, _P, C.c_int, C.POINTER(C.c_double), _P, _P]),
]


def load_library():
    """dlopen the in-tree library once; no fallback of any kind."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            'ffrnet_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library lacks a symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _check_dev(t, name, shape_tail=None, device=None):
    """Boundary check of a tensor handed to the library as a raw pointer.  `device`: the handle's device -- a tensor
    on ANOTHER GPU is rejected (its pointer would be dereferenced by kernels running on the handle's GPU: peer reads
    over xGMI at best, a fault at worst; one process per GPU must `torch.cuda.set_device(local_rank)` or pass explicit
    devices)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('ffrnet_amd: %s is on %s; the native HIP path needs a ROCm device '
                           'tensor (there is no CPU fallback)' % (name, t.device))
    if device is not None and t.device.index != device.index:
        raise RuntimeError('ffrnet_amd: %s is on %s but this Engine lives on %s; move the tensor there (or create the '
                           'Engine on the tensor\'s device: one process per GPU calls torch.cuda.set_device(local_rank) '
                           'before anything else)' % (name, t.device, device))
    if t.dtype != torch.float32:
        raise RuntimeError('ffrnet_amd: %s must be float32, got %s' % (name, t.dtype))
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise RuntimeError('ffrnet_amd: %s expected shape [N,%s], got %s'
                           % (name, ','.join(map(str, shape_tail)), list(t.shape)))


class Engine(object):
    """One native handle on one ROCm device: packed weights + workspace arena."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.device = torch.device('cuda', device if isinstance(device, int) else
                                   (device.index or 0))
        self._h = C.c_void_p(0)
        self._ck(self.lib.ffr_create(C.byref(self._h), self.device.index), create=True)
        self.has_encoder = False
        self.has_recnet = False
        self._recnet_sig = None      # lfw.pair_embed: signature of the RecNet shell whose weights this handle holds

    # -- plumbing -------------------------------------------------------------
    def _ck(self, rc, create=False):
        if rc != 0:
            msg = self.lib.ffr_last_error(None if create else self._h)
            raise RuntimeError('ffrnet native error %d: %s' % (rc, (msg or b'?').decode()))

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self.lib.ffr_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _descs(sd):
        keep, items = [], []
        for k, v in sd.items():
            if not torch.is_tensor(v) or not v.is_floating_point():
                continue
            t = v.detach().to('cpu', torch.float32).contiguous()
            if t.dim() < 1 or t.dim() > 4:
                continue
            keep.append(t)
            d = TensorDesc()
            d.name = k.encode()
            d.data = t.data_ptr()
            d.ndim = t.dim()
            for i, s in enumerate(t.shape):
                d.shape[i] = s
            items.append(d)
        arr = (TensorDesc * len(items))(*items)
        return arr, len(items), keep

    # -- weights --------------------------------------------------------------
    def load_encoder(self, state_dict):
        arr, n, keep = self._descs(state_dict)
        self._ck(self.lib.ffr_load_encoder(self._h, arr, n))
        self.has_encoder = True
        nblk = 0
        while 'body.%d.res_layer.1.weight' % nblk in state_dict:
            nblk += 1
        self.num_layers = {24: 50, 49: 100, 50: 152}[nblk]

    def load_recnet(self, state_dict):
        arr, n, keep = self._descs(state_dict)
        self._ck(self.lib.ffr_load_recnet(self._h, arr, n))
        self.has_recnet = True
        self._recnet_sig = None      # whoever loaded through a shell records its signature after this call

    # -- forward --------------------------------------------------------------
    def encoder_forward(self, x, want_f=True, want_featmap=True):
        _check_dev(x, 'x', device=self.device)
        if x.dim() != 4 or x.size(1) != 3:
            raise RuntimeError('ffrnet_amd: encoder input must be [N,3,H,W], got %s' % list(x.shape))
        x = x.contiguous()
        n, _, h, w = x.shape
        if h % 16 or w % 16:
            raise RuntimeError('ffrnet_amd: H and W must be multiples of 16, got %dx%d' % (h, w))
        if want_f and (h, w) != (112, 112):
            # same failure as the reference: Linear(512*7*7, 512), model_ir_se50.py:124
            raise RuntimeError('mat1 and mat2 shapes cannot be multiplied (%dx%d and 25088x512)'
                               % (n, 512 * (h // 16) * (w // 16)))
        fm = torch.empty((n, 512, h // 16, w // 16), device=x.device, dtype=torch.float32) \
            if want_featmap else None
        f = torch.empty((n, 512), device=x.device, dtype=torch.float32) if want_f else None
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_encoder_forward(self._h, _ptr(x), n, h, w, _ptr(fm), _ptr(f),
                                                  self._stream()))
        return fm, f

    def recnet_forward(self, featmap, want_feat_new=True):
        _check_dev(featmap, 'input', device=self.device)
        if featmap.dim() != 4 or tuple(featmap.shape[1:]) != (512, 7, 7):
            c = featmap.size(1) + featmap.size(2) * featmap.size(3) if featmap.dim() == 4 else -1
            # the reference fails in Conv4Space's first conv (models/recnet.py:363)
            raise RuntimeError('RecNet expects input [N,512,7,7] (561 channels after the '
                               'self-similarity concat), got %s (%d channels)'
                               % (list(featmap.shape), c))
        featmap = featmap.contiguous()
        n = featmap.size(0)
        f_new = torch.empty((n, 512), device=featmap.device, dtype=torch.float32)
        feat_new = torch.empty((n, 512, 7, 7), device=featmap.device, dtype=torch.float32) \
            if want_feat_new else None
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_recnet_forward(self._h, _ptr(featmap), n, _ptr(f_new),
                                                 _ptr(feat_new), self._stream()))
        return f_new, feat_new

    def embed(self, x, want_f=True, out=None):
        """x[N,3,112,112] -> (f_new[N,512], f[N,512]); `out` = preallocated (f_new, f)."""
        _check_dev(x, 'x', (3, 112, 112), device=self.device)
        x = x.contiguous()
        n = x.size(0)
        if out is not None:
            f_new, f = out
            for o, nm in ((f_new, 'out[0]'), (f, 'out[1]')):
                if o is None:
                    continue
                _check_dev(o, nm, device=self.device)
                if tuple(o.shape) != (n, 512) or not o.is_contiguous():
                    raise RuntimeError('ffrnet_amd: %s must be a contiguous [%d,512] tensor, got %s' % (nm, n, list(o.shape)))
        else:
            f_new = torch.empty((n, 512), device=x.device, dtype=torch.float32)
            f = torch.empty((n, 512), device=x.device, dtype=torch.float32) if want_f else None
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_embed(self._h, _ptr(x), n, _ptr(f_new), _ptr(f), self._stream()))
        return f_new, f

    def embed_u8(self, img, flip=None, want_f=True):
        """img[N,112,112,3] uint8 RGB (device) -> (f_new, f); flip: optional uint8[N] h-flip flags."""
        if not (torch.is_tensor(img) and img.is_cuda and img.dtype == torch.uint8):
            raise RuntimeError('ffrnet_amd: embed_u8 needs a uint8 ROCm device tensor [N,112,112,3]')
        if img.device.index != self.device.index:
            raise RuntimeError('ffrnet_amd: img is on %s but this Engine lives on %s' % (img.device, self.device))
        if img.dim() != 4 or tuple(img.shape[1:]) != (112, 112, 3):
            raise RuntimeError('ffrnet_amd: embed_u8 expects [N,112,112,3] (HWC, RGB), got %s' % list(img.shape))
        img = img.contiguous()
        n = img.size(0)
        fl = flip.to(device=img.device, dtype=torch.uint8).contiguous() if flip is not None else None
        f_new = torch.empty((n, 512), device=img.device, dtype=torch.float32)
        f = torch.empty((n, 512), device=img.device, dtype=torch.float32) if want_f else None
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_embed_u8(self._h, _ptr(img), _ptr(fl), n, _ptr(f_new), _ptr(f), self._stream()))
        return f_new, f

    def cosine_scores(self, a, b):
        _check_dev(a, 'a', device=self.device)
        _check_dev(b, 'b', device=self.device)
        if a.shape != b.shape or a.dim() != 2:
            raise RuntimeError('cosine_scores: a and b must both be [n,dim]')
        a, b = a.contiguous(), b.contiguous()
        s = torch.empty((a.size(0),), device=a.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_cosine_scores(self._h, _ptr(a), _ptr(b), a.size(0), a.size(1),
                                                _ptr(s), self._stream()))
        return s

    def lfw_fold_accuracy(self, scores, labels, n_folds=10):
        """Device fold protocol: scores[n] fp32, labels[n] -> (mean accuracy, [(best_thr, acc)] per fold).
        The mean divides by n_folds (the reference hard-codes 10 = its n_folds)."""
        _check_dev(scores, 'scores', device=self.device)
        scores = scores.contiguous()
        lab = labels.to(device=scores.device, dtype=torch.int32).contiguous()
        thr = torch.empty(n_folds, device=scores.device, dtype=torch.float64)
        acc = torch.empty(n_folds, device=scores.device, dtype=torch.float64)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_lfw_fold_accuracy(self._h, _ptr(scores), _ptr(lab), scores.numel(), n_folds,
                                                    _ptr(thr), _ptr(acc), self._stream()))
        thr, acc = thr.cpu().tolist(), acc.cpu().tolist()
        return sum(acc) / n_folds, list(zip(thr, acc))

    # -- arena / measurement --------------------------------------------------
    def workspace_bytes(self, n, h=112, w=112):
        return int(self.lib.ffr_workspace_bytes(self._h, n, h, w))

    def reserve(self, n, h=112, w=112):
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_reserve(self._h, n, h, w))

    def set_option(self, name, value):
        """Experiment knob of this handle (include/ffrnet.h: ffr_set_option); the library reads no environment."""
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_longlong(0)
        self._ck(self.lib.ffr_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_options_from_env(self, prefix='FFR_OPT_'):
        """tools/ only: FFR_OPT_<NAME>=<int> in the environment of a measurement script -> set_option(name, int).
        Read here, in the script's own host code; libffrnet_hip.so itself never looks at the environment."""
        done = {}
        for k, v in sorted(os.environ.items()):
            if k.startswith(prefix):
                self.set_option(k[len(prefix):].lower(), int(v))
                done[k[len(prefix):].lower()] = int(v)
        return done

    def generation(self):
        """Changes whenever the handle released device memory a captured hipGraph may point into."""
        return int(self.lib.ffr_generation(self._h))

    def memory_stats(self):
        """Device bytes and packing seconds of the handle (include/ffrnet.h: ffr_mem_stats)."""
        st = MemStats()
        self._ck(self.lib.ffr_memory_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in MemStats._fields_}

    def probe_mfma_peak(self, iters=20000):
        """(TFLOP/s, shader clock in GHz) of a register-resident fp32-MFMA loop on this device."""
        tf, ghz = C.c_double(0), C.c_double(0)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_probe_mfma_peak(self._h, iters, C.byref(tf), C.byref(ghz), self._stream()))
        return tf.value, ghz.value

    def profile_enable(self, on=True):
        self._ck(self.lib.ffr_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        arr = (KClassStat * len(KCLASS_NAMES))()
        self._ck(self.lib.ffr_profile_read(self._h, arr))
        return {KCLASS_NAMES[i]: dict(launches=int(arr[i].launches), ms=arr[i].ms,
                                      flops=arr[i].flops, bytes=arr[i].bytes, flops_executed=arr[i].flops_executed,
                                      flops_useful=arr[i].flops_useful)
                for i in range(len(KCLASS_NAMES))}

    # -- test hooks -----------------------------------------------------------
    def op_conv(self, **kw):
        d = ConvDesc()
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                v = v.data_ptr()
            elif v is None:
                v = 0 if k not in ('x', 'w', 'bias', 'slope', 'resid', 'out') else None
            setattr(d, k, v)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_op_conv(self._h, C.byref(d), self._stream()))

    def op_conv3x3(self, x_nhwc, w, bias, slope=None, pad_mode=0, use_wino=True, resid=None):
        """x[N,H,W,cin] device NHWC, w[cout,cin,3,3] / bias / slope host tensors -> out[N,H,W,cout]."""
        _check_dev(x_nhwc, 'x', device=self.device)
        x_nhwc = x_nhwc.contiguous()
        n, hh, ww, cin = x_nhwc.shape
        wh = w.detach().float().cpu().contiguous()
        bh = bias.detach().float().cpu().contiguous()
        sh = slope.detach().float().cpu().contiguous() if slope is not None else None
        out = torch.empty((n, hh, ww, wh.size(0)), device=x_nhwc.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_op_conv3x3(self._h, _ptr(x_nhwc), n, hh, ww, cin, C.c_void_p(wh.data_ptr()),
                                             C.c_void_p(bh.data_ptr()),
                                             C.c_void_p(sh.data_ptr()) if sh is not None else C.c_void_p(0),
                                             wh.size(0), pad_mode, int(use_wino),
                                             _ptr(resid.contiguous()) if resid is not None else C.c_void_p(0),
                                             _ptr(out), self._stream()))
        return out

    def op_convlayer_train(self, x_nhwc, G, w, gamma, beta, slope, da_nhwc, need_dx=True):
        """One ConvLayer in train() mode (models/recnet.py:78-85), forward + backward (test hook).
        x_nhwc [G*N,7,7,cin] and da_nhwc [G*N,7,7,cout] device tensors (logical channel counts);
        w [cout,cin,3,3], gamma/beta/slope [cout] host tensors.
        Returns dict(out, dx, dw[cout,cin,3,3], dgamma, dbeta, dslope, running_mean, running_var, mean, invstd)."""
        _check_dev(x_nhwc, 'x', device=self.device)
        gn, hh, ww, cin = x_nhwc.shape
        assert hh == 7 and ww == 7 and gn % G == 0
        cout = w.size(0)
        cin_pad, cout_pad = (cin + 31) // 32 * 32, (cout + 63) // 64 * 64
        rows = gn * 49
        dev = x_nhwc.device
        xp = torch.zeros((rows, cin_pad), device=dev)
        xp[:, :cin] = x_nhwc.reshape(rows, cin)
        dap = torch.zeros((rows, cout_pad), device=dev)
        dap[:, :cout] = da_nhwc.reshape(rows, cout)
        host = [t.detach().float().cpu().contiguous() for t in (w, gamma, beta, slope)]
        out = torch.empty((rows, cout_pad), device=dev)
        dx = torch.zeros((rows, cin_pad), device=dev) if need_dx else None
        dw = torch.empty((cout_pad, 9, cin_pad), device=dev)
        dvec = torch.empty((5, cout_pad), device=dev)
        stats = torch.empty((2, G, cout_pad), device=dev)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_op_convlayer_train(self._h, _ptr(xp), G, gn // G, cin, cout,
                                                     *[C.c_void_p(t.data_ptr()) for t in host], _ptr(dap), _ptr(out),
                                                     _ptr(dx), _ptr(dw), _ptr(dvec), _ptr(stats), self._stream()))
        dwt = dw[:cout, :, :cin].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
        return dict(out=out[:, :cout].reshape(gn, 7, 7, cout), dx=dx[:, :cin].reshape(gn, 7, 7, cin) if need_dx else None,
                    dw=dwt, dgamma=dvec[0, :cout], dbeta=dvec[1, :cout], dslope=dvec[2, :cout],
                    running_mean=dvec[3, :cout], running_var=dvec[4, :cout], mean=stats[0, :, :cout],
                    invstd=stats[1, :, :cout])

    # -- RecNet training step (include/ffrnet_train.h) --------------------------------------------
    N_CLASSES = 10575
    validate_labels = True     # range check of class ids (the reference fails loudly in scatter_ / CrossEntropyLoss)

    def _labels(self, label, n, device):
        """int32 device copy of `label` after the checks the native kernels do not make: one id per image, every id in
        [0, N_CLASSES).  A CPU tensor is checked before the copy (free); a device tensor costs one small sync, which
        `validate_labels = False` removes."""
        if not torch.is_tensor(label) or label.numel() != n:
            raise RuntimeError('ffrnet_amd: expected %d class labels, got %s' %
                               (n, list(label.shape) if torch.is_tensor(label) else type(label)))
        if self.validate_labels and n:
            lo, hi = int(label.min()), int(label.max())
            if lo < 0 or hi >= self.N_CLASSES:
                raise RuntimeError('ffrnet_amd: class label out of range [0, %d): min %d, max %d' % (self.N_CLASSES, lo, hi))
        return label.reshape(-1).to(device, torch.int32).contiguous()

    def train_init(self, recnet_state_dict):
        """Device-resident training state (flat parameter / gradient / Adam buffers) from a RecNet state_dict."""
        arr, n, keep = self._descs(recnet_state_dict)
        self._ck(self.lib.ffr_train_init(self._h, arr, n))
        self._train_spec = {k: tuple(v.shape) for k, v in recnet_state_dict.items()}
        # num_batches_tracked continues from the loaded values (the native counter counts updates since this call)
        self._nbt0 = {k: int(v) for k, v in recnet_state_dict.items() if k.endswith('num_batches_tracked')}

    def train_info(self):
        p, g, n, nbt, step = _P(0), _P(0), C.c_size_t(0), C.c_longlong(0), C.c_int(0)
        self._ck(self.lib.ffr_train_info(self._h, C.byref(p), C.byref(g), C.byref(n), C.byref(nbt), C.byref(step)))
        return dict(params=p.value, grads=g.value, n_flat=n.value, num_batches_tracked=nbt.value, adam_step=step.value)

    def train_get(self, key, which='param'):
        """One tensor in torch layout on the host; which: param | grad | exp_avg | exp_avg_sq | running."""
        code = {'param': 0, 'grad': 1, 'exp_avg': 2, 'exp_avg_sq': 3, 'running': 4}[which]
        out = torch.empty(self._train_spec[key], dtype=torch.float32)
        self._ck(self.lib.ffr_train_get(self._h, code, key.encode(), C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def train_set(self, key, value, which='param'):
        code = {'param': 0, 'grad': 1, 'exp_avg': 2, 'exp_avg_sq': 3}[which]
        v = value.detach().to('cpu', torch.float32).contiguous()
        self._ck(self.lib.ffr_train_set(self._h, code, key.encode(), C.c_void_p(v.data_ptr()), v.numel()))

    def train_state_dict(self):
        """The RecNet state_dict as the reference would save it (models/trainer.py:216-224)."""
        info = self.train_info()
        sd = {}
        for k, shape in self._train_spec.items():
            if k.endswith('num_batches_tracked'):
                sd[k] = torch.tensor(info['num_batches_tracked'] + self._nbt0.get(k, 0), dtype=torch.long)
            elif k.endswith(('running_mean', 'running_var')):
                sd[k] = self.train_get(k, 'running')
            else:
                sd[k] = self.train_get(k, 'param')
        return sd

    def train_zero_grad(self):
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_zero_grad(self._h, self._stream()))

    def train_forward(self, featmap, label, groups=1, slot=0, want=('f_new', 'pred_loss', 'pred_label', 'M_space',
                                                                      'M_channel', 'feat_space', 'feat_channel')):
        """RecNet.forward(input, label) in train() mode on `groups` BatchNorm batches stacked along dim 0.
        Returns the reference's 7-tuple (entries not in `want` are None)."""
        _check_dev(featmap, 'featmap', device=self.device)
        featmap = featmap.contiguous()
        n = featmap.size(0)
        if featmap.dim() != 4 or tuple(featmap.shape[1:]) != (512, 7, 7) or n % groups:
            raise RuntimeError('ffrnet_amd: RecNet input must be [G*N,512,7,7], got %s' % list(featmap.shape))
        lab = self._labels(label, n, featmap.device)
        dev = featmap.device
        shapes = dict(f_new=(n, 512), pred_loss=(n, self.N_CLASSES), pred_label=(n, self.N_CLASSES), M_space=(n, 49, 49),
                      M_channel=(n, 512, 512), feat_space=(n, 512, 7, 7), feat_channel=(n, 512, 7, 7))
        names = ('f_new', 'pred_loss', 'pred_label', 'M_space', 'M_channel', 'feat_space', 'feat_channel')
        outs = [torch.empty(shapes[k], device=dev, dtype=torch.float32) if k in want else None for k in names]
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_forward(self._h, slot, _ptr(featmap), _ptr(lab), groups, n // groups,
                                                *[_ptr(o) for o in outs], self._stream()))
        return tuple(outs)

    def train_backward(self, grads, slot=0):
        """grads: 7 tensors or None (gradients wrt the 7-tuple of train_forward); adds into the flat gradient buffer."""
        gs = [g.contiguous().float() if g is not None else None for g in grads]
        for g in gs:
            if g is not None:
                _check_dev(g, 'grad', device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_backward(self._h, slot, *[_ptr(g) for g in gs], self._stream()))

    _WHICH = {'param': 0, 'grad': 1, 'exp_avg': 2, 'exp_avg_sq': 3, 'running': 4}

    def train_export(self, key, out, which='grad'):
        """One entry in torch layout into the device tensor `out` (no host round trip)."""
        _check_dev(out, 'out', device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_export(self._h, self._WHICH[which], key.encode(), _ptr(out), self._stream()))
        return out

    def train_import(self, key, value, which='param'):
        _check_dev(value, 'value', device=self.device)
        v = value.detach().contiguous().float()
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_import(self._h, self._WHICH[which], key.encode(), _ptr(v), self._stream()))

    def train_losses(self, f_enc, loss_weight=(1, 1, 1, 1), slot=0):
        """The four weighted loss items + accuracy (device tensor [5]) of the forward in `slot` (G = 2)."""
        _check_dev(f_enc, 'f_enc', device=self.device)
        out = torch.empty(5, device=f_enc.device, dtype=torch.float32)
        lw = (C.c_double * 4)(*[float(x) for x in loss_weight])
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_losses(self._h, slot, _ptr(f_enc.contiguous()), lw, _ptr(out), self._stream()))
        return out

    def train_backward_losses(self, slot=0):
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_backward_losses(self._h, slot, self._stream()))

    def train_iteration(self, img_non, img_ocl, label, loss_weight=(1, 1, 1, 1)):
        """Encoder + RecNet train forward + losses + zero_grad + backward in one native call -> device tensor [5]."""
        _check_dev(img_non, 'img_non', device=self.device)
        _check_dev(img_ocl, 'img_ocl', device=self.device)
        n = img_non.size(0)
        if tuple(img_non.shape[1:]) != (3, 112, 112) or img_ocl.shape != img_non.shape:
            raise RuntimeError('ffrnet_amd: training images must be [N,3,112,112] pairs')
        lab = self._labels(label, n, img_non.device)
        out = torch.empty(5, device=img_non.device, dtype=torch.float32)
        lw = (C.c_double * 4)(*[float(x) for x in loss_weight])
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_iteration(self._h, _ptr(img_non.contiguous()), _ptr(img_ocl.contiguous()), _ptr(lab), n,
                                                  lw, _ptr(out), self._stream()))
        return out

    def train_buckets(self):
        """[(offset, count)] of the gradient buckets in the order the backward finishes them, with their ids."""
        n = C.c_int(0)
        off = (C.c_size_t * 6)()
        order = (C.c_int * 5)()
        self._ck(self.lib.ffr_train_buckets(self._h, C.byref(n), off, order))
        return [(int(order[k]), int(off[order[k]]), int(off[order[k] + 1] - off[order[k]])) for k in range(n.value)]

    def train_bucket_wait(self, i, stream):
        """`stream` (torch.cuda.Stream) waits on the device until the enqueued backward has finished bucket i."""
        self._ck(self.lib.ffr_train_bucket_wait(self._h, int(i), C.c_void_p(stream.cuda_stream)))

    def train_option(self, name, value):
        self._ck(self.lib.ffr_train_option(self._h, name.encode(), int(value)))

    def train_debug(self, name, shape, slot=0):
        out = torch.empty(shape, dtype=torch.float32)
        self._ck(self.lib.ffr_train_debug_copy(self._h, slot, name.encode(), C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def train_adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=1.0):
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_train_adam_step(self._h, lr, betas[0], betas[1], eps, weight_decay, clip_value,
                                                  self._stream()))

    def encoder_trunk_nhwc(self, x, n_blocks):
        _check_dev(x, 'x', device=self.device)
        x = x.contiguous()
        n, _, h, w = x.shape
        chans, div = 64, 1
        from .synth import ir_blocks
        for cin, depth, stride in ir_blocks(getattr(self, 'num_layers', 50))[:n_blocks]:
            chans, div = depth, div * stride
        out = torch.empty((n, h // div, w // div, chans), device=x.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_encoder_trunk_nhwc(self._h, _ptr(x), n, h, w, n_blocks,
                                                     _ptr(out), self._stream()))
        return out

    def recnet_debug(self, featmap):
        _check_dev(featmap, 'input', (512, 7, 7), device=self.device)
        featmap = featmap.contiguous()
        n, dev = featmap.size(0), featmap.device
        o = dict(ss_space=torch.empty((n, 49, 49), device=dev),
                 M_space=torch.empty((n, 49, 49), device=dev),
                 feat_space=torch.empty((n, 512, 7, 7), device=dev),
                 feat_channel_raw=torch.empty((n, 512, 7, 7), device=dev),
                 feat_channel=torch.empty((n, 512, 7, 7), device=dev),
                 ss_channel0=torch.empty((512, 512), device=dev), M_channel0=torch.empty((512, 512), device=dev))
        with torch.cuda.device(self.device):
            self._ck(self.lib.ffr_recnet_debug(
                self._h, _ptr(featmap), n, _ptr(o['ss_space']), _ptr(o['M_space']),
                _ptr(o['feat_space']), _ptr(o['feat_channel_raw']), _ptr(o['feat_channel']),
                _ptr(o['ss_channel0']), _ptr(o['M_channel0']), self._stream()))
        return o


class GraphedEmbed(object):
    """hipGraph replay of Engine.embed for a fixed batch size (small batches are launch bound: one
    forward is ~190 launches).  The whole forward is enqueued by ONE ffr_embed call that neither
    allocates nor synchronises once the workspace is reserved, so it captures as is.
        g = GraphedEmbed(engine, n);  f_new, f = g(x)       # x[n,3,112,112] on the device
    The returned tensors are the graph's static outputs (overwritten by the next call)."""

    def __init__(self, engine, n):
        self.engine, self.n = engine, n
        dev = engine.device
        self.x = torch.zeros((n, 3, 112, 112), device=dev, dtype=torch.float32)
        self.f_new = torch.empty((n, 512), device=dev, dtype=torch.float32)
        self.f = torch.empty((n, 512), device=dev, dtype=torch.float32)
        self.captures = 0
        self._capture()

    def _capture(self):
        """The graph holds raw pointers into the handle's workspace and packed weights: it is only valid for the
        allocation generation it was captured in (Engine.generation())."""
        engine, dev = self.engine, self.engine.device
        engine.reserve(self.n)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up outside the capture
            engine.embed(self.x, out=(self.f_new, self.f))
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            engine.embed(self.x, out=(self.f_new, self.f))
        self.generation = engine.generation()
        self.captures += 1

    def __call__(self, x):
        _check_dev(x, 'x', (3, 112, 112), device=self.engine.device)
        if x.size(0) != self.n:
            raise RuntimeError('GraphedEmbed was captured for batch %d, got %d' % (self.n, x.size(0)))
        if self.engine.generation() != self.generation:
            # a larger batch, a weight reload or ffr_train_init re-allocated what the graph points into
            self._capture()
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.f_new, self.f
