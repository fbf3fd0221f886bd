"""gan-leaks_amd: MI355X-native implementation of the GAN-Leaks full-black-box (fbb) attack path.

Layout mirrors the part of the reference this replaces:
    attack_models/fbb.py, utils.py, eval_roc.py      drop-in entry points (same names/arguments)
    gan_models/dcgan/model_torch.py, wgangp/model.py generator classes (load_state_dict compatible)
    csrc/ + libganleaks_hip.so                       HIP kernels behind the C ABI (include/ganleaks.h)
    shard.py                                         bank sharding + RCCL min-reduce (one process per GPU)

Import as `import ganleaks_amd` (shim at the repository root; the directory name has a hyphen).
Importing the package does not load the HIP library; the first computation does and raises if the
library or the GPU is missing -- there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import synth  # noqa: F401
from ._lib import Context, DeviceArray, GanLeaksError, build, device_count  # noqa: F401
from .attack import Bank, GeneratedBank, attack, prepare_images  # noqa: F401
