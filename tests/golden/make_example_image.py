#!/usr/bin/env python3
"""Fixture generator (build container only): the one real training image the reference ships, as DATA.

    python tests/golden/make_example_image.py        ->  tests/golden/example_image.npz

Reads /root/reference/example_data/imgs/r_0.png (800 x 800 RGBA, the frame of example_data/transforms_train.json) and stores what the
reference's Blender loader turns it into with `factor: 2` (configs/example.yaml:4; rnerf/datasets.py:340-347): the image / 255 resized to
400 x 400 with cv2.INTER_AREA — for an exact halving that is the mean of each 2 x 2 block.  Stored exactly, as the uint16 SUM of the four
8-bit values per channel (`rgba_sum4`, [400, 400, 4]); the loader's pixel is sum / (4 * 255).  White background is off in the example
config (white_bkgd: false -> images[..., :3], datasets.py:356-357).  The file holds pixels only — no text of any reference source."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/example_data/imgs/r_0.png"


def main():
    im = np.array(Image.open(SRC))
    assert im.shape == (800, 800, 4) and im.dtype == np.uint8, (im.shape, im.dtype)
    s = im.astype(np.uint16).reshape(400, 2, 400, 2, 4).sum(axis=(1, 3)).astype(np.uint16)
    out = os.path.join(HERE, "example_image.npz")
    np.savez_compressed(out, rgba_sum4=s)
    print(out, os.path.getsize(out), "bytes; mean rgb", (s[..., :3] / 1020.0).mean(axis=(0, 1)), "alpha coverage", float((s[..., 3] > 0).mean()))


if __name__ == "__main__":
    main()
