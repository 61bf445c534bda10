"""BASELINE cfg 3 at full size (256 x 1920x1080 resident in HBM) through size-independent properties:
the batch result of every frame equals the single-frame result (frames are independent units), a
sample of frames equals the oracle, and detection is idempotent."""
import numpy as np
import pytest

import oracle
from fiducials_amd.detector import ArucoDetector
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.synth import K_DEFAULT, make_frame

pytestmark = pytest.mark.gpu


def test_cfg3_batch_256_properties():
    torch = pytest.importorskip("torch")
    d = get_predefined_dictionary(6)
    uniq = np.stack([make_frame(d, 1000 + i).image for i in range(8)])
    B = 256
    host = np.concatenate([uniq] * (B // 8))
    dev = torch.from_numpy(host).cuda()
    det = ArucoDetector(d, max_width=1920, max_height=1080, max_batch=B, max_markers=64)
    res = det.detect_markers_device(dev.data_ptr(), B, 1920, 1080)
    poses = det.pose_last(0.14, K_DEFAULT, np.zeros(5))
    # frames 8k + j are copies of frame j: identical results (no cross-frame state)
    for f in range(B):
        j = f % 8
        assert res[f][1].tolist() == res[j][1].tolist()
        assert np.array_equal(res[f][0], res[j][0])
        assert np.array_equal(poses[f].tvecs, poses[j].tvecs)
    # sample vs oracle
    for j in range(8):
        oids, ocorners = oracle.detect(uniq[j], d)
        assert res[j][1].tolist() == oids.tolist() and len(oids) == 20
        assert np.array_equal(res[j][0], ocorners)
    # idempotence: second run of the same batch
    res2 = det.detect_markers_device(dev.data_ptr(), B, 1920, 1080)
    assert all(np.array_equal(a[0], b[0]) and a[1].tolist() == b[1].tolist() for a, b in zip(res, res2))
    det.close()
