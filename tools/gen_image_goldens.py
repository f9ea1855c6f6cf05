#!/usr/bin/env python
"""Generates tests/golden/image_preprocess.npz by RUNNING the reference's own /root/reference/starster/image.py
(`process_image` :43-76, `load_image(s)` :79-109, `prepare_images_for_mast3r` :112-139, `make_pair_indices` :24-40) in this
container.  Its one absent import, torchvision, is a stub: `ToTensor` / `Normalize` as documented, and
`functional.resize(img, size, BICUBIC)` = `torch.nn.functional.interpolate(mode="bicubic", antialias=True,
align_corners=False)` -- what torchvision does for float tensors [U]; the resize itself is therefore NOT pinned, the size
rule, the centre crop to multiples of 16, the normalisation, the file loading and the Mast3r dict layout are.
Run:  python tools/gen_image_goldens.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/starster/image.py"
CASES = [(60, 91, 48), (48, 64, 64), (97, 33, 40), (32, 32, 32), (41, 131, 48), (25, 17, 80)]   # small: noise does not compress


def load_reference_image():
    tv = types.ModuleType("torchvision"); T = types.ModuleType("torchvision.transforms")
    F = types.ModuleType("torchvision.transforms.functional")

    class ToTensor:
        def __call__(self, a):   # HWC uint8 array -> CHW float in [0, 1]
            return torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float().div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.m = torch.tensor(mean).view(-1, 1, 1); self.s = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.m) / self.s

    class InterpolationMode:
        BICUBIC = "bicubic"

    def resize(img, size, interpolation=None):
        assert interpolation == "bicubic"
        return torch.nn.functional.interpolate(img[None].float(), size=tuple(size), mode="bicubic", align_corners=False,
                                               antialias=True)[0]
    F.resize = resize
    T.ToTensor, T.Normalize, T.InterpolationMode, T.functional = ToTensor, Normalize, InterpolationMode, F
    tv.transforms = T
    sys.modules.update({"torchvision": tv, "torchvision.transforms": T, "torchvision.transforms.functional": F})
    spec = importlib.util.spec_from_file_location("reference_starster_image", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_image()
    out = {"cases": np.array(CASES)}
    for k, (H, W, size) in enumerate(CASES):
        x = torch.rand(3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
        out[f"in{k}"] = x.numpy()
        out[f"out{k}"] = ref.process_image(x, size).numpy()
    # a file through load_image: 8-bit RGB PNG
    from PIL import Image
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 256, (30, 53, 3), dtype=np.uint8)
    d = tempfile.mkdtemp()
    p = os.path.join(d, "a.png")
    Image.fromarray(arr).save(p)
    out["png"] = arr
    out["png_loaded_224"] = ref.load_image(p, 64).numpy()          # (key names from the first version: sizes 64 and 40 now)
    out["png_loaded_96"] = ref.load_images([p, p], size=40)[1].numpy()
    dicts = ref.prepare_images_for_mast3r([torch.zeros(3, 32, 48), torch.ones(3, 16, 16)])
    out["dict_true_shape"] = np.stack([dd["true_shape"] for dd in dicts])
    out["dict_true_shape_dtype_is_int32"] = np.array(int(all(dd["true_shape"].dtype == np.int32 for dd in dicts)))
    out["dict_img_shapes"] = np.array([dd["img"].shape for dd in dicts[:1]] + [dicts[1]["img"].shape])
    out["dict_idx"] = np.array([dd["idx"] for dd in dicts]); out["dict_instance_is_str_idx"] = np.array(
        int(all(dd["instance"] == str(dd["idx"]) for dd in dicts) and sorted(dicts[0]) == ["idx", "img", "instance", "true_shape"]))
    out["pairs3_sym"] = np.array(ref.make_pair_indices(3)); out["pairs4_asym"] = np.array(ref.make_pair_indices(4, symmetric=False))
    path = os.path.join(ROOT, "tests", "golden", "image_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, [out[f"out{k}"].shape for k in range(len(CASES))], out["png_loaded_224"].shape)


if __name__ == "__main__":
    main()
