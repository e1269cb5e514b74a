"""Import harness for the upstream reference checkout (TEST INFRASTRUCTURE, build container only).

The reference (a pure-Python YOLOv5 fork) lives read-only at /root/reference and exists ONLY in the
build container: it never travels to the GPU box.  This module imports it unmodified so that
`oracle/gen_golden.py` can emit golden vectors and `tests/` can pin `oracle/functional.py` against the
real thing.  Third-party packages the reference imports at module scope but that are off the
arithmetic path (cv2, IPython, torchvision, timm, thop, seaborn, ...) are served as MagicMock
packages through a `sys.meta_path` finder (recipe: SURVEY.md §8c).

Nothing under lead-yolo_amd/ (the product) may import this file.
"""
import importlib
import importlib.abc
import importlib.machinery
import logging
import os
import sys
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("LEADYOLO_REFERENCE", "/root/reference")

_STUB_ROOTS = ("cv2", "IPython", "timm", "torchvision", "thop", "seaborn", "git", "ultralytics",
               "tensorboard", "clearml", "comet_ml", "wandb")


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__path__ = []
        m.__all__ = []          # makes `from timm.models.layers import *` a no-op
        m.__loader__ = self
        return m

    def exec_module(self, module):
        return None


class _StubFinder(importlib.abc.MetaPathFinder):
    def __init__(self, roots):
        self.roots = set(roots)
        self.loader = _StubLoader()

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self.loader, is_package=True)
        return None


_loaded = None


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def load():
    """Returns a namespace with the reference's `models.yolo`, `models.common`, `models.rfa`,
    `utils.loss`, `utils.torch_utils`, `utils.general` modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference checkout not present at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot  # noqa: F401  must precede the IPython stub (utils/plots.py:31)
    missing = []
    for root in _STUB_ROOTS:
        try:
            importlib.import_module(root)
        except Exception:
            missing.append(root)
    sys.meta_path.append(_StubFinder(missing))
    logging.disable(logging.CRITICAL)
    try:
        yolo = importlib.import_module("models.yolo")
        common = importlib.import_module("models.common")
        rfa = importlib.import_module("models.rfa")
        loss = importlib.import_module("utils.loss")
        torch_utils = importlib.import_module("utils.torch_utils")
        general = importlib.import_module("utils.general")
        metrics = importlib.import_module("utils.metrics")
    finally:
        logging.disable(logging.NOTSET)
    logging.getLogger("yolov5").setLevel(logging.ERROR)

    class NS:
        pass
    ns = NS()
    ns.yolo, ns.common, ns.rfa, ns.loss = yolo, common, rfa, loss
    ns.torch_utils, ns.general, ns.metrics = torch_utils, general, metrics
    ns.stubbed = missing
    _loaded = ns
    return ns
