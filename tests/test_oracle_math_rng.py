"""Pins for the oracle's deterministic math and RNG primitives (oracle/o_math.h, oracle/o_rng.h)."""
import numpy as np
import pytest


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-300)


def test_expf_accuracy(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-87, 88, 40001), np.linspace(-1, 1, 4001)]).astype(np.float32)
    got = np.array([L.eazo_expf(float(x)) for x in xs], np.float32)
    assert _ulp_err(got, np.exp(xs.astype(np.float64))).max() <= 2.0
    assert L.eazo_expf(-3.4e38) == 0.0 and L.eazo_expf(0.0) == 1.0 and np.isinf(L.eazo_expf(89.0))
    # subnormal results are reached with one rounding
    assert abs(L.eazo_expf(-100.0) - np.exp(-100.0)) <= 2 * float(np.spacing(np.float32(np.exp(-100.0))))


def test_logf_accuracy(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.geomspace(1e-38, 1e38, 20001), np.linspace(0.5, 2.0, 4001)]).astype(np.float32)
    got = np.array([L.eazo_logf(float(x)) for x in xs], np.float32)
    ref = np.log(xs.astype(np.float64))
    ok = np.abs(ref) > 1e-3
    assert _ulp_err(got[ok], ref[ok]).max() <= 2.0
    assert np.abs(got[~ok] - ref[~ok]).max() < 1e-7
    assert L.eazo_logf(1.0) == 0.0 and np.isinf(L.eazo_logf(0.0))


def test_tanhf_accuracy(oracle):
    L = oracle.lib()
    xs = np.linspace(-20, 20, 40001).astype(np.float32)
    got = np.array([L.eazo_tanhf(float(x)) for x in xs], np.float32)
    assert np.abs(got - np.tanh(xs.astype(np.float64))).max() <= 1.3e-7
    assert L.eazo_tanhf(0.0) == 0.0 and L.eazo_tanhf(50.0) == 1.0 and L.eazo_tanhf(-50.0) == -1.0


def test_tree_sum_is_pairwise(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 7, 16, 200, 256):
        v = rng.standard_normal(n).astype(np.float32)
        P = 1
        while P < n:
            P *= 2
        buf = np.zeros(P, np.float32)
        buf[:n] = v
        while buf.size > 1:
            buf = (buf[0::2] + buf[1::2]).astype(np.float32)
        assert L.eazo_tree_sum(v.ctypes.data, n) == buf[0]


# Random123 known-answer vectors for Threefry-2x32 (20 rounds): (counter, key) -> output
THREEFRY_KAT = [
    ((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
    ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
    ((0x243F6A88, 0x85A308D3), (0x13198A2E, 0x03707344), (0xC4923A9C, 0x483DF7A0)),
]


@pytest.mark.parametrize("ctr,key,expected", THREEFRY_KAT)
def test_threefry_known_answers(oracle, ctr, key, expected):
    out = np.zeros(2, np.uint32)
    oracle.lib().eazo_threefry2x32(key[0], key[1], ctr[0], ctr[1], out.ctypes.data)
    assert (int(out[0]), int(out[1])) == expected


def test_gumbel_moments(oracle):
    g = oracle.gumbel((5, 6), 200000).astype(np.float64)
    assert abs(g.mean() - 0.5772156649) < 0.01
    assert abs(g.var() - np.pi ** 2 / 6) < 0.03
    assert np.isfinite(g).all()


def test_key_fold_matches_product_host_code(oracle, eaz):
    """The C-ABI's host-side key derivation (no GPU needed) agrees with the oracle's."""
    import ctypes as C

    L = eaz.load_library()
    for key in (0, 7, 0x1234567890ABCDEF):
        for tag in (1, 2, 0x100, 0x11F):
            out = C.c_uint64()
            assert L.eaz_key_fold(key, tag, C.byref(out)) == 0
            k0, k1 = oracle.fold((key & 0xFFFFFFFF, key >> 32), tag)
            assert out.value == (k0 | (k1 << 32))
