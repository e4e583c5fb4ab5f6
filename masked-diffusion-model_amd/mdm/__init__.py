"""mdm -- MI355X-native masked-diffusion train step and reverse sampler.

Host side (Python on PyTorch-ROCm for device memory / streams / torch.distributed)
over libmdm_hip.so (hand-written HIP kernels for gfx950, C ABI in include/mdm_hip.h).
Keeps the reference's call surface: Scheduler(args), Sampler(dataset, args, Scheduler,
dataset_hist).sample(model, timesteps), Trainer(...).train(...), model(x, t).sample.
"""
from . import _lib  # noqa: F401
from . import evaluate  # noqa: F401
from ._lib import BF16, F32  # noqa: F401
from .dist import GradComm  # noqa: F401
from .optim import EMA, Accelerator, AdamW, get_lr_scheduler  # noqa: F401
from .sampler import Sampler  # noqa: F401
from .scheduler import Scheduler  # noqa: F401
from .trainer import BaseTrainer, Trainer  # noqa: F401
from .unet import UNet, unet6_config  # noqa: F401
from .unet2d import UNet2D, my_model_config  # noqa: F401
