"""The Wang hash (dpu/shared/kernels/partition.c:20-28) is a bijection on 32-bit keys: every step is invertible modulo
2^32.  The packed-tuple path (dpu_olap_amd/csrc/pack6.h) rebuilds foreign keys from hash bits with that inverse; this is
the inverse restated in numpy, independently of the device code, against the oracle's forward hash (itself pinned by the
reference's known-answer values, tests/test_oracle_golden.py).  CPU only."""
import numpy as np

import oracle


def wang_inv_np(h):
    M = np.uint64(0xFFFFFFFF)
    k = h.astype(np.uint64)
    k ^= k >> np.uint64(16)                                                                   # key ^= key >> 16
    k = ((k + np.uint64(1)) * np.uint64(pow((1 - (1 << 11)) % (1 << 32), -1, 1 << 32))) & M   # key += ~(key << 11)
    k ^= (k >> np.uint64(6)) ^ (k >> np.uint64(12)) ^ (k >> np.uint64(18)) ^ (k >> np.uint64(24)) ^ (k >> np.uint64(30))   # key ^= key >> 6
    k = (k * np.uint64(pow(9, -1, 1 << 32))) & M                                              # key += key << 3
    k ^= (k >> np.uint64(10)) ^ (k >> np.uint64(20)) ^ (k >> np.uint64(30))                  # key ^= key >> 10
    k = ((k + np.uint64(1)) * np.uint64(pow((1 - (1 << 15)) % (1 << 32), -1, 1 << 32))) & M   # key += ~(key << 15)
    return k.astype(np.uint32)


def test_wang_hash_inverse_round_trip():
    rng = np.random.default_rng(1)
    k = np.concatenate([np.array([0, 1, 2, 3, 8, 9, 12, 0xFFFFFFFF, 1 << 30, 1 << 31], dtype=np.uint32),
                        rng.integers(0, 1 << 32, size=1 << 22, dtype=np.uint32)])
    np.testing.assert_array_equal(wang_inv_np(oracle.wang_hash_np(k)), k)
    np.testing.assert_array_equal(oracle.wang_hash_np(wang_inv_np(k)), k)


def test_wang_hash_known_preimages(kats):
    """Appendix A of SURVEY.md / the reference's partition test: wang(0) = 0x4636b9c9, wang(2) = 0xff4d1170, ..."""
    h = np.array([0x4636b9c9, 0x62baf5a0, 0xff4d1170, 0x2bf062cf, 0x5a4d3aaf, 0xff60a88f, 0xecd809ff, 0xdc8b039a, 0xd18d6e72], dtype=np.uint32)
    np.testing.assert_array_equal(wang_inv_np(h), np.array([0, 1, 2, 3, 8, 9, 12, 0xFFFFFFFF, 1 << 30], dtype=np.uint32))
