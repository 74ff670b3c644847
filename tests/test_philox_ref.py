"""The numpy Philox used as a checker reproduces Random123's published
known-answer vectors for philox4x32-10 (kat_vectors of the Random123 package)."""
import numpy as np

import philox_ref as P


def test_known_answers():
    z = P.philox4x32_10([[0, 0, 0, 0]], (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = 0xFFFFFFFF
    o = P.philox4x32_10([[f, f, f, f]], (f, f))[0]
    assert [hex(int(v)) for v in o] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = P.philox4x32_10([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]],
                        (0xa4093822, 0x299f31d0))[0]
    assert [hex(int(v)) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_isotropic_statistics():
    d = P.isotropic(200000, 0x5EED2026, 3)
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-15
    assert np.abs(d.mean(axis=0)).max() < 5e-3
    assert np.abs((d * d).mean(axis=0) - 1 / 3).max() < 5e-3
