"""CPU: the synthetic frame generator is deterministic (CRC of seed 1000 committed)."""
import zlib

import numpy as np


def test_seed_1000_crc(synth):
    img = synth.make_frame(1000)
    assert img.shape == (480, 752) and img.dtype == np.uint8
    assert zlib.crc32(img.tobytes()) == 0x978C1887
    assert np.array_equal(img, synth.make_frame(1000))
    assert not np.array_equal(img, synth.make_frame(1001))


def test_splitmix_known_answers(synth):
    # SplitMix64 reference outputs for seed 0 (Steele, Lea, Flood 2014; same values as java.util.SplittableRandom)
    out = synth.splitmix64(0, np.arange(3))
    assert [int(v) for v in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_stream_is_shifted_crops(synth):
    frames, offs = synth.make_stream(7, 6, H=120, W=160, margin=16)
    assert frames.shape == (6, 120, 160)
    for t in range(1, 6):
        dx, dy = offs[t] - offs[t - 1]
        assert abs(dx) <= 8 and abs(dy) <= 8
        # overlap region identical after the integer shift
        a = frames[t - 1][max(dy, 0):120 + min(dy, 0), max(dx, 0):160 + min(dx, 0)]
        b = frames[t][max(-dy, 0):120 + min(-dy, 0), max(-dx, 0):160 + min(-dx, 0)]
        assert np.array_equal(a, b)


def test_descriptor_sets(synth):
    q, c = synth.make_descriptor_sets(2000, n=500)
    d = np.unpackbits(q ^ c, axis=1).sum(axis=1)
    near = d < 64
    assert 0.7 < near.mean() < 0.9
    assert 12 < d[near].mean() < 28
