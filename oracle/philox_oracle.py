"""TEST INFRASTRUCTURE (oracle/): numpy restatement of the on-device noise stream of the HIP kernels.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path never does.

The reference draws its Brownian increments from torch's CPU generator (solver.py:367, 381: randn(K, d, N + 1)); that stream
is the `noise='reference'` mode and is pinned by the golden fixtures.  The throughput mode `noise='philox'` has no counterpart in
the reference: its definition is this file and include/psp.h (psp_philox_normal_fill):

  * generator: Philox4x32-7 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the smallest
    Crush-resistant member of the family; rounds 1 - 3 of this repository used the 10-round default), checked below against the
    known-answer vectors of the Random123 distribution (kat_vectors, philox4x32 with 7 and with 10 rounds);
  * counter (c0, c1, c2, c3) = (global trajectory index, time step n, call index 4 b + q, iteration l); key = (seed & 2^32-1,
    seed >> 32);
  * the four 32-bit outputs -> four N(0, 1) values by two Box-Muller pairs on 24-bit uniforms u = ((r >> 8) + 1/2) 2^-24:
        z0 = sqrt(-2 ln u0) cos(2 pi u1), z1 = sqrt(-2 ln u0) sin(2 pi u1), z2, z3 likewise from (u2, u3);
  * call (b, q) of trajectory k at step n supplies the increments xi_{n+1}[k, 16 b + 4 r + q], r = 0..3 (csrc/hjb_kernels.h
    philox_block: the T layout's four features of lane (k mod 16, q) in 16-feature block b).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


ROUNDS = 7          # csrc/hjb_kernels.h: PSP_PHILOX_ROUNDS


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=ROUNDS):
    """Vectorised over numpy uint32 arrays (broadcast); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, rounds=10)


# known-answer vectors of Random123 (kat_vectors: "philox4x32 <rounds> <ctr x4> <key x2> <expected x4>")
KAT7 = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def normal4(r0, r1, r2, r3):
    """Two Box-Muller pairs in float64 (the device evaluates them in fp32 with hardware log / sin / cos)."""
    u = [((r >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24 for r in (r0, r1, r2, r3)]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    return ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]), rb * np.cos(2 * np.pi * u[3]), rb * np.sin(2 * np.pi * u[3])


def normal_stream(N, K, d, k_offset=0, seed=42, iteration=0):
    """(N + 1, K, d) float64, slice 0 zero: what psp_philox_normal_fill materialises (include/psp.h)."""
    out = np.zeros((N + 1, K, d))
    nb = (d + 15) // 16
    n = np.arange(N, dtype=np.uint32)[:, None, None]
    k = (np.arange(K, dtype=np.uint64) + np.uint64(k_offset)).astype(np.uint32)[None, :, None]
    idx = np.arange(4 * nb, dtype=np.uint32)[None, None, :]
    r = philox4x32(k, n, idx, np.uint32(iteration), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z = normal4(*r)                                          # four arrays (N, K, 4 nb)
    for c in range(4 * nb):
        b, q = c >> 2, c & 3
        for rr in range(4):
            f = 16 * b + 4 * rr + q
            if f < d:
                out[1:, :, f] = z[rr][:, :, c]
    return out
