"""Host logic of starst3r_amd.forward / matching.merge_corres (no GPU): unique + sort + first-occurrence index like
numpy.unique on the packed (idx2, idx1) int64 view, and the upstream cache file layout."""
import hashlib

import numpy as np
import torch

from starst3r_amd import forward, matching


def test_merge_corres_ret_index_equals_numpy_unique():
    rng = np.random.default_rng(0)
    i1 = rng.integers(0, 50, 400); i2 = rng.integers(0, 60, 400)
    a, b, idx = matching.merge_corres(torch.tensor(i1), torch.tensor(i2), ret_xy=False, ret_index=True)
    # upstream: np.unique(np.c_[idx2, idx1].view(np.int64), return_index=True) -- the little-endian int64 view sorts
    # on idx1 first, idx2 second
    packed, first = np.unique(np.c_[i2, i1].astype(np.int32).view(np.int64), return_index=True)
    uniq = packed[:, None].view(np.int32)                     # columns (idx2, idx1)
    assert np.array_equal(b.numpy(), uniq[:, 0]) and np.array_equal(a.numpy(), uniq[:, 1])
    assert np.array_equal(idx.numpy(), first)
    a2, b2 = matching.merge_corres(torch.tensor(i1), torch.tensor(i2), ret_xy=False)
    assert torch.equal(a, a2) and torch.equal(b, b2)
    xy1, xy2, _ = matching.merge_corres(torch.tensor(i1), torch.tensor(i2), (10, 8), (12, 8), ret_xy=True, ret_index=True)
    assert np.array_equal(xy1.numpy(), np.stack([uniq[:, 1] % 8, uniq[:, 1] // 8], 1))
    assert np.array_equal(xy2.numpy(), np.stack([uniq[:, 0] % 8, uniq[:, 0] // 8], 1))


def test_cache_reuse_and_mirrored_correspondence_file(tmp_path):
    """A pair whose three files exist is not inferred (model=None must not be touched); a correspondence file stored
    under the opposite order is mirrored."""
    h = lambda s: hashlib.md5(s.encode()).hexdigest()
    assert forward.hash_md5("0.png") == h("0.png")
    a, b = dict(instance="0.png", idx=0), dict(instance="1.png", idx=1)
    X = torch.zeros(4, 4, 3); Cf = torch.ones(4, 4)
    p1 = tmp_path / "forward" / h("0.png") / (h("1.png") + ".pth"); p1.parent.mkdir(parents=True)
    p2 = tmp_path / "forward" / h("1.png") / (h("0.png") + ".pth"); p2.parent.mkdir(parents=True)
    torch.save((X, Cf, X, Cf), p1); torch.save((X, Cf, X, Cf), p2)
    cdir = tmp_path / "corres_conf=desc_conf_subsample=8"; cdir.mkdir()
    xy1 = torch.tensor([[1., 2.]]); xy2 = torch.tensor([[3., 0.]])
    torch.save(((1.0, 2.0, 1), (xy2, xy1, torch.tensor([2.0]))), cdir / f"{h('1.png')}-{h('0.png')}.pth")
    res, _ = forward.forward_mast3r([(a, b), (b, a)], None, str(tmp_path))
    assert list(res) == [("0.png", "1.png")]
    (q1, q2), qc = res["0.png", "1.png"]
    assert q1 == str(p1) and q2 == str(p2)
    score, (m1, m2, cf) = torch.load(qc)
    assert torch.equal(m1, xy1) and torch.equal(m2, xy2) and score[2] == 1
