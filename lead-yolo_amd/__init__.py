"""lead-yolo_amd: MI355X-native (gfx950 HIP) implementation of the LEAD-YOLO detector hot path.

Import name: `lead_yolo_amd` (the directory name carries a hyphen; see ../lead_yolo_amd/__init__.py)."""
from . import capi, ops, pack  # noqa: F401
from .modules import *  # noqa: F401,F403
from .modules import Lazy  # noqa: F401
from .model import DEFAULT_CFG, DetectionModel, Model, load_cfg, make_divisible, parse_model  # noqa: F401
from .loss import ComputeLoss  # noqa: F401
from .ddp import GradReducer  # noqa: F401
from .train import GraphedTrainStep, ModelEMA, forward_backward, optimizer_step, smart_optimizer, train_step  # noqa: F401
from .optim import FusedSGD  # noqa: F401
from .graph import GraphedForward  # noqa: F401
from .nms import nms_padded, non_max_suppression  # noqa: F401,E402
