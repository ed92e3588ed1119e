"""Oracle: restatement of the build's OWN device RNG (TEST INFRASTRUCTURE).

The reference draws everything from NumPy's global MT19937 stream; the device-RNG throughput mode
of the HIP kernels uses counter-based Philox4x32-10 instead (avddpg_amd/csrc/common.h).  This file
restates that generator in NumPy so the integer outputs (replay indices) can be checked bit for
bit and the float outputs to a few ulp.  It pins the kernels to their specification, not to the
reference (the two streams are only distributionally equal -- see DESIGN.md, RNG).
"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
STREAM_RESET_A, STREAM_RESET_B, STREAM_OU, STREAM_NORMAL, STREAM_REPLAY = 1, 2, 3, 4, 5


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c0, c1, c2, c3


def philox_at(seed, counter, index, stream):
    index = np.asarray(index, dtype=np.uint64)
    z = np.zeros_like(index)
    return philox4x32_10(index, z + np.uint64(stream), z + np.uint64(counter & 0xFFFFFFFF),
                         z + np.uint64((counter >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def replay_indices(n_agents, B, rng_range, seed, counter):
    total = n_agents * B
    calls = (total + 3) // 4
    w = philox_at(seed, counter, np.arange(calls), STREAM_REPLAY)
    words = np.stack(w, axis=1).reshape(-1)[:total]
    return ((words * np.uint64(rng_range)) >> np.uint64(32)).astype(np.int32).reshape(n_agents, B)


def box_muller(a, b):
    u1 = ((a >> np.uint64(8)) + np.uint64(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    u2 = (b >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = np.float32(6.283185307179586) * u2
    return (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)


def normals(n, seed, counter, stream):
    w = philox_at(seed, counter, np.arange(n), stream)
    return box_muller(w[0], w[1])[0]


def uniform_pm1(a):
    """common.h uniform_pm1: U[-1, 1) from the top 24 bits of one word."""
    return (a >> np.uint64(8)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)


def uniforms(n, seed, counter, stream, half_width):
    """avd_uniform_f32: U(-half_width, half_width) per index (word x of the block)."""
    w = philox_at(seed, counter, np.arange(n), stream)
    return (uniform_pm1(w[0]) * np.float32(half_width)).astype(np.float32)
