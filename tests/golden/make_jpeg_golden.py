"""Generates tests/golden/jpeg_libjpeg_turbo.npz: small RGB images and the JPEG files Pillow's libjpeg-turbo writes
for them (quality / restart interval in the key) -- the library behind the reference's cv2.imwrite
(reference main.py:100-101).  The scalar oracle (oracle/c/jpeg_oracle.c) must reproduce these bytes exactly, with or
without Pillow installed.      python tests/golden/make_jpeg_golden.py"""
import io
import os

import numpy as np
from PIL import Image, features


def main():
    rng = np.random.default_rng(2024)
    out = {"libjpeg_turbo_version": np.array(features.version_feature("libjpeg_turbo") or "unknown")}
    yy, xx = np.mgrid[0:32, 0:48]
    smooth = np.stack([(yy * 7 + xx * 3) % 256, (128 + 100 * np.sin(xx / 5.0)).astype(int), (yy * xx) % 256], -1).astype(np.uint8)
    cases = {"smooth": smooth, "noise": rng.integers(0, 256, (32, 48, 3), dtype=np.uint8),
             "white": np.full((16, 16, 3), 255, np.uint8)}
    for name, img in cases.items():
        for quality, restart in ((95, 0), (95, 2), (50, 1)):
            b = io.BytesIO()
            kw = {"restart_marker_blocks": restart} if restart else {}
            Image.fromarray(img, "RGB").save(b, "JPEG", quality=quality, **kw)
            out["%s_q%d_ri%d_rgb" % (name, quality, restart)] = img
            out["%s_q%d_ri%d_jpg" % (name, quality, restart)] = np.frombuffer(b.getvalue(), np.uint8)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_libjpeg_turbo.npz"), **out)


if __name__ == "__main__":
    main()
