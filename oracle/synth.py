"""Deterministic synthetic weights / inputs shared by the golden generator and the tests
(TEST INFRASTRUCTURE).  A fixture stores only (seed, key->shape table, expected outputs); both the
generator and the tests rebuild the identical state dict from the recipe below, which keeps the
committed fixtures small.  torch's CPU generator is deterministic for a given torch build; every
fixture also carries a checksum of the regenerated state so a drifted RNG fails loudly, not subtly.
"""
import zlib

import numpy as np
import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def synth_state(shapes, seed):
    """shapes: ordered list of (key, shape, dtype_str).  Conv/linear weights ~ N(0, 1/fan_in);
    BN weight ~ U(.5,1.5); BN/conv bias and running_mean ~ N(0,.1); running_var ~ U(.5,1.5)."""
    g = _gen(seed)
    keys = {k for k, _, _ in shapes}
    state = {}
    for key, shape, dt in shapes:
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            state[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith("running_var"):
            state[key] = torch.rand(shape, generator=g) + 0.5
        elif key.endswith("running_mean"):
            state[key] = torch.randn(shape, generator=g) * 0.1
        elif key.endswith("anchors"):
            state[key] = None  # filled by caller
        elif key.endswith(".bias"):
            state[key] = torch.randn(shape, generator=g) * 0.1
        elif key.endswith(".weight") and key[:-6] + "running_mean" in keys:
            state[key] = torch.rand(shape, generator=g) + 0.5
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            state[key] = torch.randn(shape, generator=g) / max(fan_in, 1) ** 0.5
    return state


def synth_input(shape, seed, scale=1.0):
    return torch.randn(tuple(shape), generator=_gen(seed)) * scale


def synth_images(b, s, seed, ch=3):
    """COCO-shaped synthetic batch: uint8 uniform 0..255 -> float / 255."""
    u8 = torch.randint(0, 256, (b, ch, s, s), generator=_gen(seed), dtype=torch.uint8)
    return u8


def synth_targets(b, seed, per_image=7):
    """targets f32[n,6] = (img, cls=0, x, y, w, h); xy ~ U(.1,.9), wh ~ U(.02,.22); img sorted."""
    g = _gen(seed)
    n = b * per_image
    img = torch.arange(b).repeat_interleave(per_image).float()
    xy = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(n, 2, generator=g) * 0.2 + 0.02
    return torch.cat((img[:, None], torch.zeros(n, 1), xy, wh), 1)


def shapes_of(state_dict):
    return [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()]


def checksum(state):
    c = 0
    for k in sorted(state):
        v = state[k]
        if v is None:
            continue
        c = zlib.crc32(np.ascontiguousarray(v.detach().numpy()).tobytes(), c)
    return c
