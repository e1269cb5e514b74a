"""lead-yolo_amd: MI355X-native (gfx950 HIP) implementation of the LEAD-YOLO detector hot path.

Import name: `lead_yolo_amd` (the directory name carries a hyphen; see ../lead_yolo_amd/__init__.py)."""
from . import capi, pack  # noqa: F401
from .modules import *  # noqa: F401,F403
