"""Import shim: the package sources live in ../lead-yolo_amd/ (a hyphen is not importable), this
module gives them the stable import path `lead_yolo_amd` (pickled checkpoints reference classes as
`lead_yolo_amd.modules.<Name>`)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "lead-yolo_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
