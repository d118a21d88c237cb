"""Checkpoint interop with the reference's files.

  se50.pth            plain torch.save of the Backbone state_dict    pretrain/model_ir_se50.py:151-153
  <iter>.pth.gzip     gzip(torch.save({'RecNet': state_dict, 'optimizer': ..., 'epoch': e, 'iter': i}))
                      models/trainer.py:201-224, utils/utils.py:110-123

These helpers read / write exactly those containers and feed them either to the nn.Module shells
or straight to a native Engine (packed, BN-folded device copy) without building modules.
"""
import gzip
import os

import torch


def load(read_path, map_location='cpu'):
    """utils.load: gzip-aware torch.load (always onto the host)."""
    if read_path.endswith('.gzip'):
        with gzip.open(read_path, 'rb') as f:
            return torch.load(f, map_location=map_location)
    return torch.load(read_path, map_location=map_location)


def save(obj, save_path):
    """utils.save: torch.save into a gzip container."""
    with gzip.GzipFile(save_path, 'wb') as f:
        torch.save(obj, f)


def latest_checkpoint(ckpt_dir):
    """Trainer.load_model('latest'): last *.pth.gzip in lexicographic order."""
    weights = sorted(x for x in os.listdir(ckpt_dir) if x.endswith('pth.gzip'))
    if not weights:
        raise FileNotFoundError('no *.pth.gzip in %s' % ckpt_dir)
    return os.path.join(ckpt_dir, weights[-1])


def load_recnet_checkpoint(recnet, file_path):
    """Trainer.load_model: recnet.load_state_dict(weights['RecNet'], strict=False); returns the
    resume point {'epoch', 'iter'} the reference keeps."""
    weights = load(file_path)
    recnet.load_state_dict(weights['RecNet'], strict=False)
    return {'epoch': weights.get('epoch'), 'iter': weights.get('iter')}


def save_recnet_checkpoint(recnet, file_path, optimizer_state=None, extra_info=None):
    """Trainer.save_model layout.  `recnet`: an nn.Module (pass the torch optimizer's state_dict()) or a NativeTrainer
    (its Adam step count and moment buffers are saved unless optimizer_state is given)."""
    if optimizer_state is None and hasattr(recnet, 'optimizer_state_dict'):
        optimizer_state = recnet.optimizer_state_dict()
    d = {'RecNet': recnet.state_dict(), 'optimizer': optimizer_state if optimizer_state is not None else {}}
    if extra_info is not None:
        d.update(extra_info)
    save(d, file_path)


def load_engine(engine, encoder_path=None, recnet_path=None):
    """Load reference checkpoint files straight into a native Engine."""
    if encoder_path:
        engine.load_encoder(load(encoder_path))
    if recnet_path:
        w = load(recnet_path)
        engine.load_recnet(w['RecNet'] if isinstance(w, dict) and 'RecNet' in w else w)
    return engine
