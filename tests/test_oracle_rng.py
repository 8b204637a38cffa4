"""Known-answer tests that pin the oracle's generators.

ChaCha12 / seed_from_u64 vectors: SURVEY.md Appendix A -- captured from the survey's RNG whose renders
reproduced the reference's committed docs/semesterbild.png pixel-for-pixel along row prefixes
(rand 0.9.1 StdRng; the crate source is not in the container).
Philox4x32-10 vectors: the Random123 distribution's kat_vectors (Salmon et al., SC'11).
pcg4d (the counter-mode generator since round 5; Jarzynski & Olano, JCGT 9(3) 2020): the paper publishes the function, no vectors -- it is
pinned by three restatements written independently from the listing (C++ in the oracle, vectorised numpy in tools/rng_battery.py, Python
integers in tests/kat_f32.py), by its algebra (every step is invertible: the inverse below undoes it, so it is a bijection of 128 bits and the
blocks of one path are distinct) and by the vectors frozen here from their agreement.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

APPENDIX_A = {  # seed: (key words, u32[0..3], u32[16..17], u32[62..65], u32[128])
    0: ("f973f2ec 45cdb581 7346f087 ad6cad06 e3a3d0d0 67e71733 72ea9bf2 fe7d8ad7", "cd2c6f7f bb2a3fb2 8e27697b c6017c94",
        "9e0d7fac bfd4a4ae", "6fdc7e07 fa202be2 4c0bcc72 eadd98ee", "ddf70276"),
    1: ("721dd8ea 4e10265d f83b9c89 2e78ce42 da03d3ba c2d29799 ac560212 1bfb6673", "d3301861 f9681a64 cc0d694a b0f4d125",
        "1cb3b3a6 85353f1c", "0c3f0b5d 3c25aa00 06cc05a3 f4c4c9f5", "a4dbf589"),
    299: ("eb787412 a1638c04 529643fc 74604ba3 745d1654 6612cb9a deb30e32 6ba274c4", "65b495d1 ee8afbbf e8a99d05 bbbd3a26",
          "85f9f693 d52ae773", "6b3e0917 d4b27f44 1977fc1c 9e9b78c2", "c3391bb3"),
    599: ("c8938b8a 61441041 9e0f226f 5285f794 c0d29729 6f86fea0 db5aedbc 633445a1", "def4f7e3 60c45fca 7d58f486 a8b56c58",
          "4047391a 348b2f0e", "cdceb4bb 42bafa41 f70db215 1da04a59", "58850501"),
}


def _hex(s):
    return [int(w, 16) for w in s.split()]


def test_chacha12_seed_from_u64_and_stream(oracle_mod):
    for seed, (key, w0, w16, w62, w128) in APPENDIX_A.items():
        assert oracle_mod.chacha_key(seed).tolist() == _hex(key)
        w = oracle_mod.chacha_words(seed, 129).tolist()
        assert w[0:4] == _hex(w0)
        assert w[16:18] == _hex(w16)          # second block of the first 4-block refill
        assert w[62:66] == _hex(w62)          # across the refill boundary (block counter 4)
        assert [w[128]] == _hex(w128)         # third refill


def test_float_conversions(oracle_mod):
    L = oracle_mod.lib()
    w = oracle_mod.chacha_words(0, 7)
    f01 = [np.float32(L.oracle_u32_to_f01(int(v))).view(np.uint32).item() for v in w[:4]]
    assert f01 == [0x3f4d2c6f, 0x3f3b2a3f, 0x3f0e2769, 0x3f46017c]
    assert w[4:7].tolist() == [0xcf310a16, 0x069dc102, 0xabe5f6d0]
    r11 = [np.float32(L.oracle_u32_to_range11(int(v))).view(np.uint32).item() for v in w[4:7]]
    assert r11 == [0x3f1e6214, 0xbf72c480, 0x3eaf97d8]
    # range: [0,1) and [-1,1), 24 / 23 bits
    assert L.oracle_u32_to_f01(0xFFFFFFFF) < 1.0 and L.oracle_u32_to_f01(0) == 0.0
    assert -1.0 <= L.oracle_u32_to_range11(0) and L.oracle_u32_to_range11(0xFFFFFFFF) < 1.0


def test_range11_short_form_equals_rands_formula_for_every_mantissa():
    """The device computes random_range(-1.0..1.0) as (the float in [2, 4) with mantissa k) - 3 (csrc/device/rt_rng.h u32_to_range11);
    the oracle and the rand crate compute (value1_2 - 1.0) * 2.0 + -1.0.  All 2^23 values of k = w >> 9, in IEEE f32: same bits."""
    k = np.arange(1 << 23, dtype=np.uint32)
    v12 = (k | np.uint32(0x3F800000)).view(np.float32)
    rand_form = ((v12 - np.float32(1.0)) * np.float32(2.0) + np.float32(-1.0)).astype(np.float32)
    short_form = ((k | np.uint32(0x40000000)).view(np.float32) - np.float32(3.0)).astype(np.float32)
    assert np.array_equal(rand_form.view(np.uint32), short_form.view(np.uint32))


def test_philox4x32_10_kat(oracle_mod):
    assert oracle_mod.philox(0, 0, 0, 0, 0, 0).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    assert oracle_mod.philox(f, f, f, f, f, f).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle_mod.philox(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


PCG4D_VECTORS = {      # frozen from the agreement of the three restatements (see the module docstring)
    (0, 0, 0, 0): [251852841, 760645481, 850445371, 3542436074],
    (1, 2, 3, 4): [908250390, 4044648920, 3775961919, 45698095],
}


def pcg4d_inverse(x, y, z, w):
    """Undoes pcg4d step by step on Python integers: the multiply-adds in reverse order, the xorshift by 16 (its own inverse), the LCG step by
    the modular inverse of its multiplier."""
    M = 1 << 32
    w = (w - y * z) % M; z = (z - x * y) % M; y = (y - z * x) % M; x = (x - y * w) % M
    x, y, z, w = [v ^ (v >> 16) for v in (x, y, z, w)]
    w = (w - y * z) % M; z = (z - x * y) % M; y = (y - z * x) % M; x = (x - y * w) % M
    inv = pow(1664525, -1, M)
    return [((v - 1013904223) * inv) % M for v in (x, y, z, w)]


def test_pcg4d_three_restatements_vectors_and_bijection(oracle_mod):
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kat_f32 as K
    import rng_battery as B
    for args, want in PCG4D_VECTORS.items():
        assert oracle_mod.pcg4d(*args).tolist() == want == K.pcg4d(*args)
    rng = np.random.default_rng(20251005)
    v = rng.integers(0, 1 << 32, size=(4, 4096), dtype=np.uint64).astype(np.uint32)
    v[:, :8] = np.array([[0, 1, 0xFFFFFFFF, 0x80000000, 0, 0xFFFFFFFF, 7, 0x7FFFFFFF]] * 4, np.uint32)          # edge words
    nb = B.pcg4d(v[0], v[1], v[2], v[3])
    for i in range(0, 4096, 37):
        a = [int(v[k, i]) for k in range(4)]
        out = oracle_mod.pcg4d(*a).tolist()
        assert out == K.pcg4d(*a) == [int(nb[k][i]) for k in range(4)]
        assert pcg4d_inverse(*out) == a                                    # a bijection of 128 bits
    # the addressing of the product: base = pcg4d(x, sample, key lo, key hi); block j after ray r = pcg4d(base + (0, 0, r, j))
    for (k0, k1, x, s, r, j) in [(5, 0, 1, 2, 0, 0), (5, 0, 1, 2, 3, 1), (0xFFFFFFFF, 0xFFFFFFFF, 799, 255, 29, 7)]:
        b = K.pcg4d(x, s, k0, k1)
        want = K.pcg4d(b[0], b[1], (b[2] + r) & 0xFFFFFFFF, (b[3] + j) & 0xFFFFFFFF)
        assert oracle_mod.ctr_block(k0, k1, x, s, r, j, gen=2).tolist() == want == K.ctr_block(k0, k1, x, s, r, j)
        assert [int(w[0]) for w in B.block("pcg4d", *[np.array([t], np.uint32) for t in (k0, k1, x, s, r, j)])] == want
    assert oracle_mod.ctr_gen() == 2                                       # what oracle.render's counter mode uses unless a test selects another
    # generators 0 / 1 keep Philox's addressing (A/B builds of the device library: -DMI355RT_CTR_GEN=0 / 1)
    assert oracle_mod.ctr_block(1, 2, 3, 4, 5, 6, gen=0).tolist() == oracle_mod.philox(1, 2, 3, 4, 5, 6).tolist()
    assert oracle_mod.ctr_block(1, 2, 3, 4, 5, 6, gen=1).tolist() == oracle_mod.philox_rounds(1, 2, 3, 4, 5, 6, 7).tolist() != oracle_mod.philox(1, 2, 3, 4, 5, 6).tolist()


def test_ctr_stream_first_sample_of_a_row_is_keyed_by_row(oracle_mod, abi):
    """Two rows never share a key and a seed offset of k equals shifting the row by k."""
    a = oracle_mod.ctr_block(5, 0, 1, 2, 0, 0).tolist()
    b = oracle_mod.ctr_block(6, 0, 1, 2, 0, 0).tolist()
    assert a != b
