"""Known-answer tests of the Philox4x32-10 restatement (oracle/philox_ref.py) against the vectors published
with Random123 (kat_vectors: `philox4x32 10` rows).  CPU only; the device generator is compared with this
restatement in tests/test_gpu_rng.py."""
import numpy as np

from oracle.philox_ref import pair_stream_noise, philox4x32_10

KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
     (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_random123_known_answers():
    for ctr, key, want in KAT:
        got = philox4x32_10([np.uint32(c) for c in ctr], key)
        assert tuple(int(x) for x in got) == want


def test_pair_stream_is_a_pure_function_of_global_lane_and_step():
    full = pair_stream_noise(50, 0, 7, 1536)
    tail = pair_stream_noise(50, 1024, 7, 512)
    for a, b in zip(full, tail):
        np.testing.assert_array_equal(a[1024:], b)
    head = pair_stream_noise(50, 0, 7, 700)  # a ragged request is a prefix of the padded one
    for a, b in zip(full, head):
        np.testing.assert_array_equal(a[:700], b)
    other_step = pair_stream_noise(50, 0, 8, 1536)
    assert not np.array_equal(full[0], other_step[0])


def test_uniforms_are_float32_exact_and_normals_standard():
    u_arr, u_fill, z = pair_stream_noise(3, 0, 0, 1 << 16)
    for u in (u_arr, u_fill):
        assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
        assert np.array_equal(u, np.round(u.astype(np.float64) * 2**24) / 2**24)
        assert abs(float(u.mean()) - 0.5) < 0.005
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
