"""Generates tests/golden/frames_cases.npz: the reference's frame preparation (ho3d.py:35-40, 136-149) on small seeded
frames.  crop_and_pad_image is the REAL reference function (datasets/utils.py imports cleanly); torchvision is absent, so
ToTensor / Resize(antialias=True) / Normalize are spelled with the torch calls torchvision's tensor path makes.

    python tests/golden/make_frames_fixture.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from datasets.utils import crop_and_pad_image  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def img_transform(crop_hwc_u8, size):
    t = torch.from_numpy(np.ascontiguousarray(crop_hwc_u8)).permute(2, 0, 1).float().div(255)          # ToTensor
    t = F.interpolate(t[None], size=(size, size), mode="bilinear", antialias=True, align_corners=False)[0]
    return ((t - MEAN) / STD).numpy()                                                                 # Normalize


def smooth_frame(rng, h, w):
    """Band-limited + noisy content so that both the antialias filter and the edges matter."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 90 * np.sin(xx / (7 + 3 * c) + c) * np.cos(yy / (5 + 2 * c)) for c in range(3)], axis=-1)
    img += rng.standard_normal(img.shape) * 25
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {   # name: (frame h, w, output size, boxes)
    "down_2x":     (120, 160, 32, [[10, 20, 90, 100], [40, 0, 160, 120]]),          # 80 -> 32, 120 -> 32 (scale 2.5 / 3.75)
    "up_and_1to1": (96, 128, 64, [[30, 10, 62, 42], [20, 20, 84, 84]]),              # 32 -> 64 (upsample), 64 -> 64 (identity)
    "outside":     (96, 128, 48, [[-20, -30, 50, 40], [100, 60, 170, 130], [-40, 10, 200, 250]]),   # leaves the frame on every side
    "non_square":  (96, 128, 40, [[5, 7, 105, 57], [17, 3, 44, 92]]),                # 100x50 and 27x89 windows
    "odd_scale":   (200, 200, 56, [[3, 5, 190, 192], [11, 13, 84, 86]]),              # 187 -> 56 (3.34), 73 -> 56 (1.30)
    "empty_box":   (64, 64, 32, [[10, 10, 10, 30], [5, 5, 37, 37]]),                 # first box empty: black image (ho3d.py:138-140)
}


def main():
    rng = np.random.default_rng(4242)
    out = {}
    for name, (h, w, size, boxes) in CASES.items():
        frames = np.stack([smooth_frame(rng, h, w) for _ in boxes])
        res = []
        for f, b in zip(frames, boxes):
            x1, y1, x2, y2 = b
            crop = np.zeros([10, 10, 3], np.uint8) if (x2 <= x1 or y2 <= y1) else crop_and_pad_image(f, b)
            res.append(img_transform(crop, size))
        out[f"{name}.frames"], out[f"{name}.boxes"] = frames, np.array(boxes, np.int32)
        out[f"{name}.size"] = np.int32(size)
        out[f"{name}.out"] = np.stack(res).astype(np.float32)
    path = os.path.join(HERE, "frames_cases.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
