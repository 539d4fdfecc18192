""" TEST INFRASTRUCTURE -- numpy restatement of the on-device collocation sampler (include/pinn.h `pinn_sample_points`,
kernel `pinn_sample_kernel` in pydens_amd/csrc/pinn_aux_kernels.h).

The reference draws its points on the host (pydens/model_torch.py:430-434: `torch.rand((N, 1))` per column, or
`sampler.sample(N)` of a batchflow NumpySampler product); there is no reference arithmetic to match, only the
distribution.  The generator itself is third-party published arithmetic: Philox4x32-10 of Salmon, Moraes, Dror, Shaw,
"Parallel random numbers: as easy as 1, 2, 3" (SC'11), constants and known-answer vectors of the Random123 library
(kat_vectors, `philox4x32 10` rows) -- `KNOWN_ANSWERS` below, checked by tests/test_sampler.py.  Only tests/ import
this module. """
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)

# (counter[4], key[2]) -> output[4]
KNOWN_ANSWERS = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """ vectorised over uint32 arrays (broadcast); returns four uint32 arrays. """
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


UNIFORM, NORMAL, CONST = 0, 1, 2


def sample_points(n, columns, seed, call_index):
    """ [n, d] float32 exactly as pinn_sample_kernel fills it; `columns` = [(kind, a, b), ...]. Uniform and constant
    columns are bit-exact; normal columns go through log / cos / sqrt (compare with a tolerance). """
    d = len(columns)
    i = np.arange(n, dtype=np.uint64)
    i_lo, i_hi = (i & MASK).astype(np.uint32), (i >> np.uint64(32)).astype(np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c_lo, c_hi = call_index & 0xFFFFFFFF, (call_index >> 32) & 0x0FFFFFFF
    scale = np.float32(2.0 ** -24)
    out = np.empty((n, d), dtype=np.float32)
    words = None
    for c, (kind, a, b) in enumerate(columns):
        a, b = np.float32(a), np.float32(b)
        if c % 4 == 0:
            words = philox4x32_10(i_lo, i_hi, np.uint32(c_lo), np.uint32(c_hi | ((c // 4) << 28)), k0, k1)
        if kind == UNIFORM:
            u = (words[c % 4] >> np.uint32(8)).astype(np.float32) * scale
            out[:, c] = a + (b - a) * u
        elif kind == NORMAL:
            q = philox4x32_10(i_lo, i_hi, np.uint32(c_lo), np.uint32(c_hi | ((8 + c) << 28)), k0, k1)
            u1 = ((q[0] >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * scale
            u2 = (q[1] >> np.uint32(8)).astype(np.float32) * scale
            z = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
            out[:, c] = a + b * z.astype(np.float32)
        else:
            out[:, c] = a
    return out
