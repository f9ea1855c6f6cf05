"""Image loading / preprocessing -- public names of the reference's `starster.image`
(starster/image.py:43-139; docs/api.rst "Image").  Host side, once per image, out of the hot path
(SURVEY.md section 2 row 5); torchvision is not a dependency here: the bicubic resize uses torch.
"""
__all__ = ("load_image", "load_images", "process_image", "prepare_images_for_mast3r")

import numpy as np
import torch


def process_image(img: torch.Tensor, size: int = 224) -> torch.Tensor:
    """(3,H,W) float image in [0,1] -> resized so the long side is `size`, centre-cropped to multiples of 16,
    normalised to [-1,1] (reference image.py:43-76)."""
    C, H, W = img.shape
    nh, nw = (int(x * size / max(H, W)) for x in (H, W))   # truncation, as the reference (image.py:62)
    out = torch.nn.functional.interpolate(img[None].float(), size=(nh, nw), mode="bicubic", align_corners=False,
                                          antialias=True)[0]
    ch, cw = (nh // 16) * 16, (nw // 16) * 16
    t, l = (nh - ch) // 2, (nw - cw) // 2
    out = out[:, t:t + ch, l:l + cw]
    return out * 2 - 1


def make_pair_indices(n: int, symmetric: bool = True):
    """All index pairs of the fully connected graph (reference image.py:24-40): (i, j) for j < i, then the
    mirrored pairs when symmetric."""
    pairs = [(i, j) for i in range(n) for j in range(i)]
    if symmetric:
        pairs += [(j, i) for (i, j) in pairs]
    return pairs


def load_image(path: str, size: int = 224) -> torch.Tensor:
    from PIL import Image
    from PIL.ImageOps import exif_transpose
    arr = np.asarray(exif_transpose(Image.open(path)).convert("RGB"), dtype=np.float32) / 255.0
    return process_image(torch.from_numpy(arr).permute(2, 0, 1), size)


def load_images(paths, size: int = 224):
    return [load_image(p, size) for p in paths]


def prepare_images_for_mast3r(imgs):
    """list of (3,H,W) tensors in [-1,1] -> the list of dicts Mast3r's pair maker consumes (image.py:112-139)."""
    out = []
    for i, img in enumerate(imgs):
        out.append(dict(img=img[None], true_shape=np.int32([img.shape[1:]]), idx=i, instance=str(i)))
    return out
