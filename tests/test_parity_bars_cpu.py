"""CPU: the image-parity bars of the full-size GPU tests (tests/util.py assert_image_parity) hold the line where it is
(round-5 verdict next #5: the old bars were FRACTIONS of the image — 1e-4 of 640 000 pixels = 64 — while the measured
differences are 0-2 pixels per image; a regression flipping 50 pixels per image passed).  Here the checker itself is
checked: the measured handful of threshold flips passes, a deliberately injected 50-pixel flip fails on every count."""
import numpy as np
import pytest

import util as U


def _images(seed=0, H=800, W=800):
    g = np.random.default_rng(seed)
    o = dict(color=g.random((3, H, W), dtype=np.float32), depth=g.random((1, H, W), dtype=np.float32) * 2,
             alpha=g.random((1, H, W), dtype=np.float32), final_T=g.random((H, W), dtype=np.float32),
             n_contrib=g.integers(0, 4000, (H, W)).astype(np.uint32))
    h = {k: v.copy() for k, v in o.items()}
    h["color"] *= np.float32(1 + 2e-7)      # the rounding-level difference of two fp32 programs
    return h, o


def _flip(h, n, seed=1):
    """n pixels gain one contributor of weight ~1/255 (what an alpha on the other side of the threshold does)."""
    g = np.random.default_rng(seed)
    H, W = h["n_contrib"].shape
    ys, xs = g.integers(0, H, n), g.integers(0, W, n)
    h["n_contrib"][ys, xs] += 1
    h["color"][:, ys, xs] += np.float32(1 / 255) * 0.4
    h["alpha"][0, ys, xs] += np.float32(1 / 255)
    h["depth"][0, ys, xs] += np.float32(1.9 / 255)
    h["final_T"][ys, xs] *= np.float32(1 - 1 / 255)


def test_the_measured_handful_of_threshold_flips_passes():
    h, o = _images()
    _flip(h, 2)
    nc, ft, px, p = U.assert_image_parity(h, o, "two flips")
    assert (nc, ft, px) == (2, 2, 2) and p > 60


def test_a_fifty_pixel_flip_fails_every_count():
    h, o = _images()
    _flip(h, 50)
    nc, ft, px, p = U.image_parity_counts(h, o)
    assert nc == ft == px == 50
    assert (h["n_contrib"] != o["n_contrib"]).mean() < 1e-4       # ... which the old fractional bar waved through
    with pytest.raises(AssertionError):
        U.assert_image_parity(h, o, "fifty flips")
    for key in ("n_contrib", "final_T", "color"):                  # each count alone is enough to fail
        h2, _ = _images()
        h2[key] = h[key]
        with pytest.raises(AssertionError):
            U.assert_image_parity(h2, o, key)


def test_the_mask_restricts_the_comparison():
    h, o = _images(H=64, W=64)
    h["n_contrib"][:32] += 1
    mask = np.zeros((64, 64), bool)
    mask[32:] = True
    assert U.image_parity_counts(h, o, mask)[0] == 0 and U.image_parity_counts(h, o)[0] == 32 * 64
