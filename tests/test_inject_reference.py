"""Injection shim against the real reference checkout (build container only; skipped elsewhere)."""
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present")


def test_patch_builds_reference_model_with_hip_modules():
    ref = ref_import.load()
    import lead_yolo_amd as L
    import lead_yolo_amd.inject as inj
    torch.manual_seed(0)
    plain = ref.yolo.Model(ref_import.REFERENCE_ROOT + "/models/LEAD-YOLO.yaml")
    replaced = inj.patch(ref.yolo)
    try:
        torch.manual_seed(0)
        m = ref.yolo.Model(ref_import.REFERENCE_ROOT + "/models/LEAD-YOLO.yaml")   # reference parse_model, our classes
        kinds = {type(mod if not isinstance(mod, torch.nn.Sequential) else mod[0]).__module__ for mod in m.model
                 if mod.type.split(".")[-1] in inj.NAMES}
        assert kinds == {"lead_yolo_amd.modules"}
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v.shape)) for k, v in plain.state_dict().items()]
        assert torch.equal(m.stride, plain.stride)           # strides from the shape-probe forward
        assert torch.equal(m.model[-1].anchors, plain.model[-1].anchors)
        assert L.modules.SHAPE_PROBE is False
        with pytest.raises(RuntimeError):                    # outside the constructor a CPU tensor still fails loudly
            m.eval()(torch.zeros(1, 3, 64, 64))
        fm = m.fuse()                                         # reference fuse() on our PatchEmbed/PatchMerging
        assert not hasattr(fm.model[0], "norm")
    finally:
        inj.unpatch(replaced)
    assert ref.yolo.DetectionModel.__init__ is replaced[("models.yolo", "DetectionModel.__init__")]
