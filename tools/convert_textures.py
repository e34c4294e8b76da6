#!/usr/bin/env python3
"""Converts the texture images of a raytrace3 checkout (timg.jpg, wall.jpg, planet.jpg, ... next to its main.cpp)
into the binary PPM files the drop-in Texture class reads (include/raytrace3/Element.h): <name>.ppm in the output
directory, optionally down-sampled.  Point SPPM_DATA_DIR at that directory (teapot.bpt lives there as well).

    python tools/convert_textures.py /path/to/raytrace3/raytracer raytrace3_amd/data --max-side 1024

Needs Pillow; nothing here is on the accelerated path."""
import argparse
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src", help="directory with the .jpg / .png textures")
    ap.add_argument("dst", help="directory that receives the .ppm files")
    ap.add_argument("--max-side", type=int, default=0, help="down-sample so that no side is longer than this (0: keep)")
    args = ap.parse_args()
    try:
        from PIL import Image
    except ImportError:
        sys.exit("Pillow is not installed")
    os.makedirs(args.dst, exist_ok=True)
    done = 0
    for path in sorted(glob.glob(os.path.join(args.src, "*.jpg")) + glob.glob(os.path.join(args.src, "*.png"))):
        img = Image.open(path).convert("RGB")
        if args.max_side and max(img.size) > args.max_side:
            k = args.max_side / max(img.size)
            img = img.resize((max(1, round(img.size[0] * k)), max(1, round(img.size[1] * k))), Image.LANCZOS)
        out = os.path.join(args.dst, os.path.splitext(os.path.basename(path))[0] + ".ppm")
        img.save(out, format="PPM")
        print(f"{path} -> {out} ({img.size[0]}x{img.size[1]})")
        done += 1
    if not done:
        sys.exit(f"no .jpg / .png files in {args.src}")


if __name__ == "__main__":
    main()
