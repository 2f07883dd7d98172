"""TEST INFRASTRUCTURE (oracle side): Philox4x32-10 in numpy, the checker for the in-kernel reparameterisation noise of
xh_poe_multi (include/xlstm_hved.h; RA_HVED.py:741-747 draws eps ~ N(0,1) with normal_()).

The algorithm is third-party and published: J. Salmon, M. Moraes, R. Dror, D. Shaw, "Parallel random numbers: as easy as 1, 2, 3",
SC'11 (Random123; also the generator behind torch's device RNG).  Pinned by the known-answer vectors of the Random123 distribution
(kat_vectors: philox4x32 10 rounds), tests/test_oracle_golden.py::test_philox_reference_known_answers.

Only tests/ may import this file; the product draws its noise in csrc/eltwise.hip."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# (counter[4], key[2]) -> output[4], Random123 kat_vectors, "philox4x32 10"
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c, k):
    """c: (..., 4) uint32 counters, k: (2,) key words -> (..., 4) uint32."""
    c0, c1, c2, c3 = [np.asarray(c[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(k[0]), int(k[1])
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def poe_noise_words(seed, counter, stream, n):
    """The counter blocks xh_poe_multi uses for elements 0..n-1 of level `stream` in draw `counter` (csrc/eltwise.hip philox_block):
    counter words (i lo, i hi | stream << 24, draw lo, draw hi), key (seed lo, seed hi)."""
    i = np.arange(n, dtype=np.uint64)
    c = np.stack([i & MASK, (i >> np.uint64(32)) | np.uint64(stream << 24), np.full(n, counter & 0xFFFFFFFF, np.uint64),
                  np.full(n, (counter >> 32) & 0xFFFFFFFF, np.uint64)], -1)
    return philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def poe_noise(seed, counter, stream, n):
    """The normals: Box-Muller on the first two words' upper 24 bits (fp32 arithmetic on the device; float64 here)."""
    w = poe_noise_words(seed, counter, stream, n)
    u1 = ((w[:, 0] >> 8).astype(np.float64) + 0.5) / 16777216.0
    u2 = ((w[:, 1] >> 8).astype(np.float64) + 0.5) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
