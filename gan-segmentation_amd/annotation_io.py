"""Annotator sample files: what the reference's GUI saves next to a
hand-drawn mask and what its few-shot dataset reads back.

    data/img_%06d.jpg      the generated image, RGB (PIL default JPEG settings)        seg_annotator.py:333
    data/feat_%06d.pickle  pickled Python list of the 9 CHW fp32 block outputs          seg_annotator.py:336-337
    data/mask_%06d.png     the drawn mask: <64 ignore, 64..192 background, >192 class 1 seg_datasets.py:85-106

``export_sample`` lets the MI355X generator produce the first two for an annotation session (the
mask stays a human's job; the Tk GUI itself is out of scope), ``load_sample`` is the reader.
"""
import os
import pickle

import numpy as np


def export_sample(dst_dir, image_id, img, feats):
    """img (R,R,3) u8 RGB, feats: list of (C,R_r,R_r) fp32 arrays (numpy or torch, any device)."""
    from PIL import Image
    os.makedirs(dst_dir, exist_ok=True)
    feats_np = []
    for f in feats:
        if hasattr(f, "detach"):
            f = f.detach().cpu().numpy()
        f = np.ascontiguousarray(f, dtype=np.float32)
        if f.ndim != 3:
            raise ValueError("each feature must be (C,H,W); got shape %s" % (f.shape,))
        feats_np.append(f)
    if hasattr(img, "detach"):
        img = img.detach().cpu().numpy()
    Image.fromarray(np.ascontiguousarray(img, dtype=np.uint8), "RGB").save(os.path.join(dst_dir, "img_%06d.jpg" % image_id))
    with open(os.path.join(dst_dir, "feat_%06d.pickle" % image_id), "wb") as fp:
        pickle.dump(feats_np, fp)


def preprocess_mask(mask_u8):
    """The thresholds of reference seg_datasets.py:85-106: int32 labels -1 (ignore), 0, 1."""
    m = np.asarray(mask_u8)
    out = np.zeros(m.shape, np.int32)
    out[m > 192] = 1
    out[m < 64] = -1
    return out


def load_sample(db_dir, image_id):
    """-> (mask int32 or None, img RGB u8, [features])  (reference seg_datasets.py:60-83)."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(db_dir, "img_%06d.jpg" % image_id)).convert("RGB"))
    with open(os.path.join(db_dir, "feat_%06d.pickle" % image_id), "rb") as fp:
        feats = pickle.load(fp)
    mpath = os.path.join(db_dir, "mask_%06d.png" % image_id)
    mask = preprocess_mask(np.asarray(Image.open(mpath).convert("L"))) if os.path.exists(mpath) else None
    return mask, img, feats
