#!/usr/bin/env python3
"""Makes tests/golden/reference_picture.npz from data files the reference checkout holds (no code of it runs):

  raytracer/sppm15.jpg   a picture the reference's authors rendered with their program (1024 x 1024) -> 128 x 128, RGB
  raytracer/timg.jpg     the floor's texture (Scene.h:131, :141)                                      -> longest side 128, BGR
  raytracer/planet.jpg   the small sphere's texture (Scene.h:138-139)                                  -> longest side 128, BGR

    python tests/golden/make_reference_picture.py [/root/reference]

The picture is the only output of the reference itself that the checkout holds.  It is a lossy JPEG of an unknown
iteration count from a generator its own threads race on, and it shows another state of the program than the checkout's
sources: plain walls where Scene.h:133-135 now puts wall.jpg on three of them, a teapot texture (blue.jpg) the checkout
lacks, and about half the radiance.  What it can pin is structure: camera, geometry, the teapot's placement, mirror and glass
paths, the floor's texture mapping, the fall-off of the light.  tests/test_reference_picture.py compares by correlation."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", "raytracer")


def small_bgr(name, side=128):
    img = Image.open(os.path.join(REF, name)).convert("RGB")
    k = side / max(img.size)
    img = img.resize((max(1, round(img.size[0] * k)), max(1, round(img.size[1] * k))), Image.LANCZOS)
    return np.ascontiguousarray(np.asarray(img)[:, :, ::-1])      # rows x cols x BGR, what cv::imread hands Texture (Element.h:47-59)


picture = np.asarray(Image.open(os.path.join(REF, "sppm15.jpg")).convert("RGB").resize((128, 128), Image.BOX))
np.savez_compressed(os.path.join(HERE, "reference_picture.npz"), picture_rgb=picture, timg_bgr=small_bgr("timg.jpg"),
                    planet_bgr=small_bgr("planet.jpg"), source="raytracer/sppm15.jpg, timg.jpg, planet.jpg of wondergo2017/raytrace3, down-sampled")
print("picture", picture.shape, "->", os.path.join(HERE, "reference_picture.npz"))
