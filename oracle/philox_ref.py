"""NumPy restatement of the production noise generator.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

The reference draws from numpy PCG64 generators (StochasticProcessModel.py:27); the HIP kernel draws from
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 - the algorithm
of the Random123 library, which is not vendored anywhere here).  This module restates the published block
function and the pair-stream layout of mbt_gym_amd/csrc/philox.hpp so tests can pin the device generator:
  * `philox4x32_10` is checked against the Random123 known-answer vectors (tests/test_philox_oracle.py);
  * `pair_stream_noise` must reproduce the device's uniforms bit for bit and its Box-Muller normals to ~1e-6.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key, rounds=10):
    """ctr: 4 uint32 arrays (broadcastable), key: 2 uint32 scalars/arrays -> 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in ctr)
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            c1 = (p1 & MASK).astype(np.uint32)
            c3 = (p0 & MASK).astype(np.uint32)
            c0, c2 = n0, n2
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform24(w):
    """[0,1) on the 2^-24 grid, exact in float32 (philox.hpp: uniform24)."""
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24)).astype(np.float32)


TILE = 512  # lanes per tile; lanes g and g + 256 of a tile share one Philox pair (csrc/philox.hpp)


def _low_bytes(blk):
    """w0 | w1 << 8 | w2 << 16 | w3 << 24 of the low bytes of a block's four words (philox.hpp: low_bytes)."""
    b = [w & np.uint32(0xFF) for w in blk]
    return b[0] | (b[1] << np.uint32(8)) | (b[2] << np.uint32(16)) | (b[3] << np.uint32(24))


def pair_stream_noise(seed, trajectory_offset, step, n):
    """(u_arr (n,2), u_fill (n,2), z (n)) for lanes [offset, offset+n) at philox step `step` - the layout
    documented at the top of mbt_gym_amd/csrc/philox.hpp.  `trajectory_offset` must be a multiple of 512."""
    assert trajectory_offset % TILE == 0
    tiles = (n + TILE - 1) // TILE
    n_pad = tiles * TILE
    half = TILE // 2
    pairs = np.arange(tiles * half, dtype=np.uint64) + np.uint64(trajectory_offset // 2)
    plo = (pairs & MASK).astype(np.uint32)
    phi = (pairs >> np.uint64(32)).astype(np.uint32)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    st = np.full_like(plo, np.uint32(step))
    blocks = [philox4x32_10((plo, phi, st, np.full_like(plo, np.uint32(b))), key) for b in range(2)]
    local = np.arange(tiles * half)
    lower = (local // half) * TILE + local % half  # local lane of each pair's first member; the second is + 256
    u_arr = np.empty((n_pad, 2), np.float32)
    u_fill = np.empty((n_pad, 2), np.float32)
    for member, blk in enumerate(blocks):
        lanes = lower + member * half
        u_arr[lanes, 0], u_arr[lanes, 1] = uniform24(blk[0]), uniform24(blk[1])
        u_fill[lanes, 0], u_fill[lanes, 1] = uniform24(blk[2]), uniform24(blk[3])
    wr, wt = (_low_bytes(blk) for blk in blocks)  # the bytes the uniforms discard: radius from block 0, angle from block 1
    u1 = ((wr >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0**-24
    r = np.sqrt(-2.0 * np.log(u1))
    theta = 2.0 * np.pi * (wt >> np.uint32(8)).astype(np.float64) * 2.0**-24
    z = np.empty((n_pad,), np.float64)
    z[lower], z[lower + half] = r * np.cos(theta), r * np.sin(theta)
    return u_arr[:n], u_fill[:n], z[:n]


class PhiloxNoise:
    """Noise source for OracleEnv that follows the device stream (uniforms exact, normals in float64)."""

    def __init__(self, seed, trajectory_offset=0, first_step=0):
        self.seed, self.offset, self.step = seed, trajectory_offset, first_step

    def draw(self, n):
        u_arr, u_fill, z = pair_stream_noise(self.seed, self.offset, self.step, n)
        self.step += 1
        return u_arr.astype(np.float64), u_fill.astype(np.float64), z.reshape(n, 1)


def policy_exploration_noise(seed, trajectory_offset, step, n, n_actions=2):
    """eps (n, n_actions) of a learned policy's exploration at philox step `step` (csrc/policy_mlp.hpp: explore_and_clip):
    block (pair, step, 8) -> Box-Muller(w0, w1) -> the lower lane's (e0, e1), Box-Muller(w2, w3) -> the upper lane's; block
    word 3 = 9 the same for action components 2 and 3.  float64 here; the device uses the hardware transcendentals."""
    assert trajectory_offset % TILE == 0
    tiles = (n + TILE - 1) // TILE
    half = TILE // 2
    pairs = np.arange(tiles * half, dtype=np.uint64) + np.uint64(trajectory_offset // 2)
    plo, phi = (pairs & MASK).astype(np.uint32), (pairs >> np.uint64(32)).astype(np.uint32)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    st = np.full_like(plo, np.uint32(step))
    local = np.arange(tiles * half)
    lower = (local // half) * TILE + local % half
    eps = np.zeros((tiles * TILE, 4))

    def box_muller(wr, wt):
        u1 = ((wr >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0**-24
        r = np.sqrt(-2.0 * np.log(u1))
        theta = 2.0 * np.pi * (wt >> np.uint32(8)).astype(np.float64) * 2.0**-24
        return r * np.cos(theta), r * np.sin(theta)

    for block, first in ((8, 0), (9, 2)):
        if first >= n_actions:
            break
        w = philox4x32_10((plo, phi, st, np.full_like(plo, np.uint32(block))), key)
        eps[lower, first], eps[lower, first + 1] = box_muller(w[0], w[1])
        eps[lower + half, first], eps[lower + half, first + 1] = box_muller(w[2], w[3])
    return eps[:n, :n_actions]
