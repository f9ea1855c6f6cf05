"""Host-side image preprocessing keeps the reference's size rules (starster/image.py:43-76, 24-40)."""
import numpy as np
import pytest
import torch

from starst3r_amd import image


@pytest.mark.parametrize("H,W,size", [(300, 451, 224), (480, 640, 512), (1080, 1920, 512), (97, 33, 224), (224, 224, 224)])
def test_process_image_size_rules(H, W, size):
    x = torch.rand(3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
    y = image.process_image(x, size)
    nh, nw = int(H * size / max(H, W)), int(W * size / max(H, W))       # image.py:62 (truncation)
    cy, cx = nh // 2, nw // 2                                             # image.py:65-66
    assert y.shape == (3, 2 * ((cy // 8) * 8), 2 * ((cx // 8) * 8))       # image.py:69-73: multiples of 16
    assert y.dtype == torch.float32 and float(y.min()) >= -2.0 and float(y.max()) <= 2.0   # Normalize(0.5, 0.5); bicubic overshoots on noise


def test_process_image_is_centre_crop_of_the_resize():
    x = torch.rand(3, 130, 210, generator=torch.Generator().manual_seed(1))
    y = image.process_image(x, 200)
    full = torch.nn.functional.interpolate(x[None], size=(int(130 * 200 / 210), 200), mode="bicubic",
                                           align_corners=False, antialias=True)[0] * 2 - 1
    nh, nw = full.shape[1:]
    cy, cx = nh // 2, nw // 2
    hh, wh = (cy // 8) * 8, (cx // 8) * 8
    assert torch.equal(y, full[:, cy - hh:cy + hh, cx - wh:cx + wh])


def test_pair_indices_and_mast3r_dicts():
    assert image.make_pair_indices(3, symmetric=False) == [(1, 0), (2, 0), (2, 1)]
    assert image.make_pair_indices(3) == [(1, 0), (2, 0), (2, 1), (0, 1), (0, 2), (1, 2)]
    d = image.prepare_images_for_mast3r([torch.zeros(3, 32, 48), torch.zeros(3, 16, 16)])
    assert d[1]["img"].shape == (1, 3, 16, 16) and d[0]["true_shape"].tolist() == [[32, 48]] and d[1]["instance"] == "1"


def test_against_the_reference_run(tmp_path):
    """tests/golden/image_preprocess.npz: /root/reference/starster/image.py executed in place (tools/gen_image_goldens.py;
    torchvision stubbed -- the bicubic resize itself is the same torch call on both sides and therefore not pinned)."""
    import os
    from PIL import Image
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_preprocess.npz"))
    for k, (H, W, size) in enumerate(z["cases"].tolist()):
        y = image.process_image(torch.from_numpy(z[f"in{k}"]), size)
        assert y.shape == z[f"out{k}"].shape
        np.testing.assert_allclose(y.numpy(), z[f"out{k}"], rtol=0, atol=1e-6)     # x * 2 - 1 vs (x - 0.5) / 0.5
    p = str(tmp_path / "a.png")
    Image.fromarray(z["png"]).save(p)
    np.testing.assert_allclose(image.load_image(p, 64).numpy(), z["png_loaded_224"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(image.load_images([p, p], size=40)[1].numpy(), z["png_loaded_96"], rtol=0, atol=1e-6)
    d = image.prepare_images_for_mast3r([torch.zeros(3, 32, 48), torch.ones(3, 16, 16)])
    assert np.array_equal(np.stack([dd["true_shape"] for dd in d]), z["dict_true_shape"])
    assert all(dd["true_shape"].dtype == np.int32 for dd in d) and int(z["dict_true_shape_dtype_is_int32"]) == 1
    assert [list(dd["img"].shape) for dd in d] == z["dict_img_shapes"].tolist()
    assert [dd["idx"] for dd in d] == z["dict_idx"].tolist() and all(dd["instance"] == str(dd["idx"]) for dd in d)
    assert sorted(d[0]) == ["idx", "img", "instance", "true_shape"] and int(z["dict_instance_is_str_idx"]) == 1
    assert image.make_pair_indices(3) == [tuple(r) for r in z["pairs3_sym"].tolist()]
    assert image.make_pair_indices(4, symmetric=False) == [tuple(r) for r in z["pairs4_asym"].tolist()]
