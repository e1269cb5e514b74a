"""tests/golden/ssdd16.npz: 16 images of the SSDD test split bundled with the reference (data/SSDD/images/test, data/SSDD/labels/test; the
dataset LEAD-YOLO.yaml's nc = 1 recipe trains on, data/SSDD.yaml), letterboxed to 320 x 320 the way val.py's loader does (resize the long
side to the target, pad with 114; utils/augmentations.py `letterbox`) with their labels mapped into the letterboxed frame.  Data only:
the images are single-channel SAR chips stored as (JPEG-noisy) RGB triples; their luma plane is kept, ONE uint8 plane each, and the tests
feed it to all three input channels.
tests/golden/ssdd_train48.npz (round 6): 48 images of the TRAIN split (data/SSDD/images/train), same format — what the accuracy test trains
on, so that the 16 test-split images it scores are held out (data/SSDD.yaml: train and test are disjoint image sets).
    python oracle/gen_ssdd_fixture.py          (build container: /root/reference present)"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/data/SSDD"
S, N = 320, 16


def main():
    one("test", 16, "ssdd16.npz")
    one("train", 48, "ssdd_train48.npz")


def one(split, N, name):
    files = sorted(glob.glob(os.path.join(REF, "images", split, "*.jpg")))[:N]
    assert len(files) == N, f"SSDD {split} images not found (this script runs in the build container only)"
    imgs = np.full((N, S, S), 114, dtype=np.uint8)
    targets, names = [], []
    for i, f in enumerate(files):
        im = Image.open(f).convert("RGB")
        w, h = im.size
        r = S / max(w, h)
        nw, nh = int(round(w * r)), int(round(h * r))
        g = np.asarray(im.convert("L").resize((nw, nh), Image.BILINEAR))
        top, left = (S - nh) // 2, (S - nw) // 2
        imgs[i, top:top + nh, left:left + nw] = g
        lab = os.path.join(REF, "labels", split, os.path.basename(f)[:-4] + ".txt")
        if os.path.exists(lab):
            for line in open(lab):
                p = line.split()
                if len(p) == 5:
                    c, x, y, bw, bh = int(p[0]), *map(float, p[1:])
                    targets.append([i, c, (x * nw + left) / S, (y * nh + top) / S, bw * nw / S, bh * nh / S])
        names.append(os.path.basename(f))
    targets = np.asarray(targets, dtype=np.float32).reshape(-1, 6)
    out = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(out, imgs=imgs, targets=targets, names=np.asarray(names))
    print(f"{out}: {os.path.getsize(out) / 1024:.0f} KiB, {len(targets)} boxes in {N} images")


if __name__ == "__main__":
    sys.exit(main())
