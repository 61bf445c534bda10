"""The reference-built STag oracle (oracle/_ref): loads, and its pieces behave as the reference code reads."""
import numpy as np
import pytest

from oracle import stag_ref

pytestmark = pytest.mark.skipif(not stag_ref.available(), reason="oracle/_ref/libstag_ref.so not built")


def test_constants_are_the_reference_ones():
    assert stag_ref.constants() == dict(EDGE_VERTICAL=1, EDGE_HORIZONTAL=2, ANCHOR_PIXEL=254)


def test_smooth5_is_the_fixed_binomial_kernel():
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    s = stag_ref.smooth5(img)
    k = np.array([1, 4, 6, 4, 1])
    assert np.array_equal(s[2:7, 2:7], (np.outer(k, k) * 255 + 128) >> 8)
    assert stag_ref.smooth5(np.full((5, 7), 93, np.uint8)).tolist() == [[93] * 7] * 5  # reflect-101 keeps a constant


def test_vertical_step_edge_gives_vertical_anchors_sorted_by_gradient():
    img = np.zeros((12, 16), np.uint8)
    img[:, 8:] = 200
    grad, dirs = stag_ref.gradient(img, 16)
    assert (grad[0, :] == 15).all() and (grad[:, 0] == 15).all()  # border = GRADIENT_THRESH - 1
    assert grad[5, 7] == 600 and grad[5, 8] == 600 and dirs[5, 7] == 1
    edge, order = stag_ref.anchors(grad, dirs, 16, 0, 1)
    ys, xs = np.nonzero(edge)
    assert set(xs.tolist()) == {7, 8} and ys.min() == 2 and ys.max() == 9
    g = grad.reshape(-1)[order]
    assert (np.diff(g) >= 0).all()
    # equal gradients: descending offsets (the --C[grad] placement of SortAnchorsByGradValue)
    assert (np.diff(order) < 0).all()
