"""Injection shim for a user-side checkout of the reference (qingqing-zijin/LEAD-YOLO).

The reference resolves yaml module names with eval() in `models/yolo.py`'s module globals at
parse_model call time (models/yolo.py:427, whitelist set :434-444, fuse() exact-type checks
:224-231), so replacing those globals before `Model(...)` is built swaps in the HIP modules while
train.py / detect.py / val.py stay byte-unchanged:

    import lead_yolo_amd.inject as inj
    inj.patch()                       # after `import models.yolo`, before Model(cfg)

or `python -m lead_yolo_amd.run detect.py --weights ... --source ...` (run.py).
This module never imports the reference itself; the caller's checkout must be importable.
"""
import importlib

NAMES = ("PatchEmbed_FasterNet", "PatchMerging_FasterNet", "BasicStage", "RFCBAMConv", "C3_CA", "CA_Bottleneck", "CoordAtt")


def patch(yolo_module=None, names=NAMES):
    """Rebind the hot-path class names in `models.yolo` (and `models.common` / `models.rfa`, which the
    reference star-imports from) to the HIP-backed classes.  Returns the dict of replaced originals."""
    from . import modules as M
    mods = [yolo_module or importlib.import_module("models.yolo")]
    for extra in ("models.common", "models.rfa"):
        try:
            mods.append(importlib.import_module(extra))
        except ImportError:
            pass
    replaced = {}
    ymod = mods[0]
    # the reference's constructor probes strides with a CPU forward (models/yolo.py:289): run it in
    # shape-probe mode (modules.SHAPE_PROBE) so that only shapes, never values, come from the CPU
    cls = getattr(ymod, "DetectionModel", None)
    if cls is not None and not getattr(cls.__init__, "_ly_wrapped", False):
        orig_init = cls.__init__

        def init(self, *a, **kw):
            M.SHAPE_PROBE = True
            try:
                orig_init(self, *a, **kw)
            finally:
                M.SHAPE_PROBE = False
        init._ly_wrapped = True
        replaced[(ymod.__name__, "DetectionModel.__init__")] = orig_init
        cls.__init__ = init
    for name in names:
        new = getattr(M, name)
        for mod in mods:
            if hasattr(mod, name):
                replaced.setdefault((mod.__name__, name), getattr(mod, name))
                setattr(mod, name, new)
    return replaced


def unpatch(replaced):
    for (mod_name, name), old in replaced.items():
        mod = importlib.import_module(mod_name)
        if name == "DetectionModel.__init__":
            mod.DetectionModel.__init__ = old
        else:
            setattr(mod, name, old)
