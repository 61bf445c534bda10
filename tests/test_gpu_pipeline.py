"""fid_submit_device / fid_collect (batches in flight on several contexts, fiducials_amd/pipeline.py) against fid_detect_device:
the same markers and poses batch by batch, in submission order; and the one-batch-per-context rule."""
import numpy as np
import pytest

import oracle
from fiducials_amd.detector import ArucoDetector, FidError
from fiducials_amd.dictionary import get_predefined_dictionary
from fiducials_amd.pipeline import BatchPipeline
from fiducials_amd.synth import K_DEFAULT, make_frame

pytestmark = pytest.mark.gpu


def _batches(torch, n_batches, B, w=640, h=480):
    d = get_predefined_dictionary(6)
    frames = np.stack([make_frame(d, 4000 + i, width=w, height=h, n_markers=4, side_range=(60, 110)).image for i in range(n_batches * B)])
    return [torch.from_numpy(frames[k * B:(k + 1) * B].copy()).cuda() for k in range(n_batches)], frames


@pytest.mark.parametrize("depth,ordered", [(1, True), (2, True), (3, True), (2, False)])
def test_batches_in_flight_equal_one_call_after_the_other(depth, ordered):
    torch = pytest.importorskip("torch")
    B, NB, w, h = 32, 5, 640, 480  # (32 frames: fid_detect_device cuts the batch in two, a chained batch is one piece)
    dev, frames = _batches(torch, NB, B, w, h)
    ref = ArucoDetector(6, max_width=w, max_height=h, max_batch=B, max_markers=32)
    want = []
    for k in range(NB):
        m = ref.detect_markers_device(dev[k].data_ptr(), B, w, h)
        want.append((m, ref.pose_last(0.14, K_DEFAULT, np.zeros(5))))
    ref.close()
    # (the reference run itself equals the oracle on the first batch)
    d = get_predefined_dictionary(6)
    for f in range(6):
        oids, ocorners = oracle.detect(frames[f], d)
        assert want[0][0][f][1].tolist() == oids.tolist() and np.array_equal(want[0][0][f][0], ocorners)
    got = []
    # ordered: fid_order_after -- a batch starts behind the previous one's find_starts and is laid out as one piece
    with BatchPipeline(6, depth=depth, fiducial_len=0.14, K=K_DEFAULT, D=np.zeros(5), ordered=ordered, max_width=w, max_height=h,
                       max_batch=B, max_markers=32) as pipe:
        for k in range(NB):
            done = pipe.push(dev[k].data_ptr(), B, w, h)
            assert (done is None) == (k < depth)
            if done is not None:
                got.append(done)
        got += pipe.flush()
        assert pipe.flush() == []
        assert all(d.last_launches() == (1 if ordered and depth > 1 else 2) for d in pipe.detectors)
    assert len(got) == NB
    for k in range(NB):
        (gm, gp), (wm, wp) = got[k], want[k]
        for f in range(B):
            assert gm[f][1].tolist() == wm[f][1].tolist() and len(gm[f][1]) > 0, (k, f)
            assert np.array_equal(gm[f][0], wm[f][0]), (k, f)
            assert np.array_equal(gp[f].rvecs, wp[f].rvecs) and np.array_equal(gp[f].tvecs, wp[f].tvecs), (k, f)


def test_one_batch_per_context_at_a_time():
    torch = pytest.importorskip("torch")
    B, w, h = 4, 640, 480
    dev, _ = _batches(torch, 1, B, w, h)
    det = ArucoDetector(6, max_width=w, max_height=h, max_batch=B, max_markers=32)
    try:
        with pytest.raises(FidError):
            det.collect()  # nothing submitted
        det.submit_device(dev[0].data_ptr(), B, w, h)
        for call in (lambda: det.submit_device(dev[0].data_ptr(), B, w, h),
                     lambda: det.detect_markers_device(dev[0].data_ptr(), B, w, h),
                     lambda: det.detect_markers_batch(np.zeros((B, h, w), np.uint8)),
                     lambda: det.pose_last(0.14, K_DEFAULT, np.zeros(5))):
            with pytest.raises(FidError):
                call()
        a = det.collect()  # the batch in flight is untouched by the refused calls
        b = det.detect_markers_device(dev[0].data_ptr(), B, w, h)
        assert all(x[1].tolist() == y[1].tolist() and np.array_equal(x[0], y[0]) for x, y in zip(a, b))
        assert sum(len(x[1]) for x in a) > 0
    finally:
        det.close()


def test_host_fed_batches_in_flight_equal_fid_detect_batch():
    """fid_submit_batch: batches in host memory (pageable and pinned) through two contexts in turn against fid_detect_batch."""
    torch = pytest.importorskip("torch")
    B, NB, w, h = 64, 4, 640, 480  # (64 frames: fid_detect_batch sends a batch up in four pieces, a chained batch in one)
    _, frames = _batches(torch, NB, B, w, h)
    host = [frames[k * B:(k + 1) * B].copy() for k in range(NB)]
    ref = ArucoDetector(6, max_width=w, max_height=h, max_batch=B, max_markers=32)
    want = []
    for k in range(NB):
        m = ref.detect_markers_batch(host[k])
        want.append((m, ref.pose_last(0.14, K_DEFAULT, np.zeros(5))))
    assert ref.last_launches() == 4
    ref.close()
    for arrs in (host, [torch.from_numpy(a).pin_memory().numpy() for a in host]):
        with BatchPipeline(6, depth=2, fiducial_len=0.14, K=K_DEFAULT, D=np.zeros(5), max_width=w, max_height=h, max_batch=B,
                           max_markers=32) as pipe:
            got = [d for d in (pipe.push_host(a) for a in arrs) if d is not None] + pipe.flush()
            assert all(d.last_launches() == 1 for d in pipe.detectors)
            with pytest.raises(ValueError):
                pipe.push_host(arrs[0][:, :, ::2])  # not contiguous: refused, never copied behind the caller's back
        assert len(got) == NB
        for k in range(NB):
            (gm, gp), (wm, wp) = got[k], want[k]
            for f in range(B):
                assert gm[f][1].tolist() == wm[f][1].tolist() and len(gm[f][1]) > 0, (k, f)
                assert np.array_equal(gm[f][0], wm[f][0]), (k, f)
                assert np.array_equal(gp[f].tvecs, wp[f].tvecs), (k, f)


def test_a_refused_host_batch_queues_no_copy_and_leaves_the_context_usable():
    """fid_detect_batch / fid_submit_batch validate geometry, encoding and limits BEFORE any host -> device copy is queued (a
    refused call must not leave DMA from the caller's buffer in flight: the caller may free it on the error), the order set by
    fid_order_after does not outlive the refused call, and the context works on afterwards."""
    d = get_predefined_dictionary(6)
    a = ArucoDetector(6, max_width=640, max_height=480, max_batch=2, max_markers=32)
    b = ArucoDetector(6, max_width=640, max_height=480, max_batch=2, max_markers=32)
    try:
        frames = np.stack([make_frame(d, 4100 + i, width=640, height=480, n_markers=4, side_range=(60, 110)).image for i in range(2)])
        want = [oracle.detect(f, d) for f in frames]
        too_tall = np.zeros((2, 481, 640), np.uint8)
        for bad in (too_tall, np.zeros((3, 480, 640), np.uint8)):  # taller than the context; more frames than max_batch
            with pytest.raises(FidError) as e:
                a.detect_markers_batch(bad)
            assert e.value.status == 1
            with pytest.raises(FidError):
                a.submit_batch(bad, after=b)
            del bad  # (nothing reads it any more)
        # a batch in flight on b, a ordered behind it, a's submit refused: the next good submit on a is not held by a stale event
        b.submit_batch(frames)
        with pytest.raises(FidError):
            a.submit_batch(too_tall, after=b)
        resb = b.collect()
        b.close()  # (b's events are gone now)
        b = None
        resa = a.detect_markers_batch(frames)
        for res in (resa, resb):
            for (corners, ids), (oids, ocorners) in zip(res, want):
                assert ids.tolist() == oids.tolist() and np.array_equal(corners, ocorners)
    finally:
        a.close()
        if b is not None:
            b.close()


def test_every_refusal_drops_the_order_and_keeps_the_layout_of_the_last_call_that_ran():
    """Round-4 advisor: the refusals in front of check_call (null pointer, too many frames, a frame stride smaller than a frame,
    a batch in flight) returned with the order of fid_order_after still set -- the next good call then waited on an event of a
    context that may be gone -- and a refused call that had already laid the result block out for ITS frame count left
    fid_pose_last / fid_tap_read addressing the wrong layout."""
    import ctypes as C

    torch = pytest.importorskip("torch")
    d = get_predefined_dictionary(6)
    a = ArucoDetector(6, max_width=640, max_height=480, max_batch=2, max_markers=32)
    b = ArucoDetector(6, max_width=640, max_height=480, max_batch=2, max_markers=32)
    K = K_DEFAULT.copy()
    K[0, 0] = K[1, 1] = 1400.0 * 640 / 1920
    K[0, 2], K[1, 2] = 320.0, 240.0
    try:
        frames = np.stack([make_frame(d, 4200 + i, width=640, height=480, n_markers=4, side_range=(60, 110)).image for i in range(2)])
        want = [oracle.detect(f, d) for f in frames]
        res0 = a.detect_markers_batch(frames)
        poses0 = a.pose_last(0.14, K, np.zeros(5))
        counts0 = a.tap_counts().copy()
        L, ptr = a._L, frames.ctypes.data
        dev = torch.from_numpy(frames).cuda()
        b.submit_batch(frames)
        refusals = [
            lambda: L.fid_submit_batch(a._ctx, None, 2, 640, 480, 640, 640 * 480, 0),                 # no image
            lambda: L.fid_submit_batch(a._ctx, ptr, 3, 640, 480, 640, 640 * 480, 0),                  # more than max_batch
            lambda: L.fid_submit_batch(a._ctx, ptr, 2, 640, 480, 640, 640 * 479, 0),                  # frame stride < a frame
            lambda: L.fid_submit_batch(a._ctx, ptr, 1, 641, 480, 641, 641 * 480, 0),                  # wider than the context
            lambda: L.fid_submit_device(a._ctx, None, 2, 640, 480, 640, 640 * 480, 0),
            lambda: L.fid_submit_device(a._ctx, C.c_void_p(dev.data_ptr()), 1, 640, 481, 640, 640 * 481, 0),
            lambda: L.fid_detect_device(a._ctx, None, 2, 640, 480, 640, 640 * 480, 0, a._out, a.max_markers, a._n),
            lambda: L.fid_detect_device(a._ctx, C.c_void_p(dev.data_ptr()), 1, 640, 480, 600, 640 * 480, 0, a._out, a.max_markers, a._n),
            lambda: L.fid_detect_batch(a._ctx, ptr, 1, 640, 480, 640, 100, 0, a._out, a.max_markers, a._n),
        ]
        for k, call in enumerate(refusals):
            a._check(L.fid_order_after(a._ctx, b._ctx))
            assert call() != 0, k
            # the layout is still the one of the two-frame call: the same counters, the same poses
            assert np.array_equal(a.tap_counts(), counts0), k
            p = a.pose_last(0.14, K, np.zeros(5))
            assert all(np.array_equal(p[f].tvecs, poses0[f].tvecs) for f in range(2)), k
        resb = b.collect()
        b.close()  # (b's events are gone now: a stale order would wait on a destroyed event)
        b = None
        res1 = a.detect_markers_device(dev.data_ptr(), 2, 640, 480)
        for res in (res0, res1, resb):
            for (corners, ids), (oids, ocorners) in zip(res, want):
                assert ids.tolist() == oids.tolist() and np.array_equal(corners, ocorners)
    finally:
        a.close()
        if b is not None:
            b.close()


def test_a_blocking_host_call_behind_fid_order_after_and_alone_give_the_same():
    """fid_detect_batch of one piece copies on the context's main stream behind the clear of its result block (no copy
    stream, no event); when fid_order_after has told the context to wait for another one's batch, it takes the copy-stream
    road instead.  Both roads, and the same context used alternately for blocking and submitted calls, give the oracle's
    markers (a stale event handle or a result block cleared at the wrong moment would show here)."""
    d = get_predefined_dictionary(6)
    frames = np.stack([make_frame(d, 300 + i, width=640, height=480, n_markers=5).image for i in range(4)])
    want = [oracle.detect(f, d) for f in frames]
    a = ArucoDetector(6, max_width=640, max_height=480, max_batch=4)
    b = ArucoDetector(6, max_width=640, max_height=480, max_batch=4)

    def check(res, idx):
        for k, i in enumerate(idx):
            assert res[k][1].tolist() == want[i][0].tolist()
            assert np.array_equal(res[k][0], want[i][1])

    try:
        check(b.detect_markers_batch(frames[:1]), [0])            # one piece, blocking: main-stream copy
        a.submit_batch(frames)                                     # a batch in flight on the other context ...
        b._check(b._L.fid_order_after(b._ctx, a._ctx))             # ... that b is told to run behind
        check(b.detect_markers_batch(frames[1:3]), [1, 2])         # blocking call with an order: copy-stream road
        check(a.collect(), [0, 1, 2, 3])
        check(b.detect_markers_batch(frames[3:4]), [3])            # and the plain road again (the order held for one call)
        b.submit_batch(frames[:2])
        check(b.collect(), [0, 1])
        check(b.detect_markers_batch(frames[2:3]), [2])
    finally:
        a.close()
        b.close()
