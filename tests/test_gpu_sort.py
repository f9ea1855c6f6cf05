"""GPU tests of the hand-written onesweep radix sort (csrc/radix_sort.hip) through the C ABI: bit-exact against
numpy's stable argsort, on every key shape of the pipeline and on the edge sizes of the tiling (0, 1, one short of /
exactly / one past a tile and a wave), heavy duplication (stability), partial last digits and begin_bit > 0."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from starst3r_amd import ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return ops.get_context("cuda:0")


def _keys(rng, n, bits, mode, dtype):
    hi = (1 << bits) - 1
    if mode == "random":
        k = rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
    elif mode == "few":          # 7 distinct keys: stability is all that orders the values
        k = rng.integers(0, 6, n, dtype=np.uint64, endpoint=True) * np.uint64(max(hi // 7, 1))
    elif mode == "runs":         # tile-key like: long runs of neighbouring values
        k = (np.arange(n, dtype=np.uint64) // np.uint64(37)) % np.uint64(min(hi, 65279) + 1)
    elif mode == "sorted_desc":
        k = np.sort(rng.integers(0, hi, n, dtype=np.uint64, endpoint=True))[::-1].copy()
    else:
        raise ValueError(mode)
    return k.astype(dtype)


SIZES = [0, 1, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385, 32769, 100_003, 1_000_003,
         1_700_003, 4_200_003]   # the tile shape changes at 100 large tiles (16384 / 8192 items): both shapes are covered


@pytest.mark.parametrize("kb,bits", [(4, 32), (4, 17), (4, 16), (4, 3), (8, 49), (8, 36), (8, 64)])
@pytest.mark.parametrize("mode", ["random", "few", "runs", "sorted_desc"])
def test_sort_pairs_equals_stable_argsort(ctx, kb, bits, mode):
    from starst3r_amd import ops
    rng = np.random.default_rng(kb * 100 + bits)
    dt = np.uint32 if kb == 4 else np.uint64
    for n in SIZES:
        k = _keys(rng, n, min(bits, 63), mode, dt)
        if bits == 64:
            k = k | (rng.integers(0, 1, n, dtype=np.uint64, endpoint=True) << np.uint64(63))   # exercise the top bit
        v = rng.permutation(n).astype(np.int32)
        tk = torch.tensor(k.view(np.int32 if kb == 4 else np.int64), device="cuda:0")
        tv = torch.tensor(v, device="cuda:0")
        ko, vo = ops.radix_sort_pairs(ctx, tk, tv, 0, bits)
        torch.cuda.synchronize()
        mask = dt((1 << bits) - 1) if bits < 8 * kb else dt(~dt(0))
        order = np.argsort(k & mask, kind="stable")
        assert np.array_equal(ko.cpu().numpy().view(dt), k[order]), (n, "keys")
        assert np.array_equal(vo.cpu().numpy(), v[order]), (n, "values")
        # inputs untouched
        assert np.array_equal(tk.cpu().numpy().view(dt), k)


def test_sort_bit_window_and_keys_only(ctx):
    """Only bits [begin, end) take part; equal windows keep their input order.  Keys-only calls (vals NULL) work."""
    from starst3r_amd import ops
    rng = np.random.default_rng(5)
    n = 300_001
    k = rng.integers(0, 2**32 - 1, n, dtype=np.uint64).astype(np.uint32)
    v = np.arange(n, dtype=np.int32)
    tk = torch.tensor(k.view(np.int32), device="cuda:0"); tv = torch.tensor(v, device="cuda:0")
    for b0, b1 in [(8, 24), (5, 9), (31, 32), (0, 1)]:
        ko, vo = ops.radix_sort_pairs(ctx, tk, tv, b0, b1)
        window = (k >> np.uint32(b0)) & np.uint32((1 << (b1 - b0)) - 1)
        order = np.argsort(window, kind="stable")
        assert np.array_equal(vo.cpu().numpy(), v[order]), (b0, b1)
        assert np.array_equal(ko.cpu().numpy().view(np.uint32), k[order]), (b0, b1)
    ko, vo = ops.radix_sort_pairs(ctx, tk, None, 0, 32)
    assert vo is None and np.array_equal(ko.cpu().numpy().view(np.uint32), np.sort(k))


def test_sort_full_size_shapes(ctx):
    """The two sort shapes of the SYNTH-1M train step (8 M x 32-bit level-1 keys, 26 M x 16-bit tile keys): sortedness,
    stability (values ascending inside runs of equal keys) and a permutation checksum."""
    from starst3r_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for n, bits in [(8_000_000, 32), (26_000_000, 16)]:
        k = torch.randint(0, 2**bits if bits < 32 else 2**31, (n,), device="cuda:0", generator=g, dtype=torch.int64)
        k = (k * (2 if bits == 32 else 1)).to(torch.int64)           # reach bit 31 as well
        k32 = (k & 0xFFFFFFFF).to(torch.int64)
        ki = torch.where(k32 >= 2**31, k32 - 2**32, k32).to(torch.int32)
        v = torch.arange(n, device="cuda:0", dtype=torch.int32)
        ko, vo = ops.radix_sort_pairs(ctx, ki, v, 0, bits)
        ku = ko.to(torch.int64) & 0xFFFFFFFF
        assert bool((ku[1:] >= ku[:-1]).all())
        same = ku[1:] == ku[:-1]
        assert bool((vo[1:][same] > vo[:-1][same]).all())            # stable: original order inside equal keys
        assert int(vo.to(torch.int64).sum()) == n * (n - 1) // 2
        assert bool(((ki[vo.long()]) == ko).all())                   # each value still travels with its key
