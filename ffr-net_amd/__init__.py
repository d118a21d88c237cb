"""ffrnet_amd -- MI355X-native FFR-Net embedding path (IR-SE50 + RecNet forward).

Host-side mirror of the reference's nn.Module interface for the hot path
(pretrain/model_ir_se50.py Backbone / ir_se_50_512, models/recnet.py RecNet) on top of
the C-ABI library libffrnet_hip.so (include/ffrnet.h).  There is no CPU fallback: a
forward without the HIP library or on a non-ROCm tensor raises.
"""
from . import synth  # noqa: F401
from .native import Engine, GraphedEmbed, NativeLibraryMissing, lib_path  # noqa: F401
from .modules import Backbone, RecNet, ir_se_50_512, l2_norm  # noqa: F401
from . import lfw  # noqa: F401
from . import checkpoint  # noqa: F401
from . import train  # noqa: F401
from .train import NativeTrainer  # noqa: F401

__all__ = ['Backbone', 'RecNet', 'ir_se_50_512', 'l2_norm', 'Engine', 'GraphedEmbed',
           'NativeLibraryMissing', 'lib_path', 'synth', 'lfw', 'checkpoint', 'train', 'NativeTrainer']
