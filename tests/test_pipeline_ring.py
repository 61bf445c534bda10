"""The bookkeeping of fiducials_amd.pipeline.BatchPipeline without a GPU: which context takes which batch, which one it is
ordered after, when a batch is collected, in which order results come back.  The contexts are recorders with ArucoDetector's
submit / collect surface (the arithmetic behind it is tested on the GPU: tests/test_gpu_pipeline.py)."""
import numpy as np
import pytest

from fiducials_amd.pipeline import BatchPipeline


class Recorder:
    log = []

    def __init__(self, dictionary, **kw):
        self.name = len([e for e in Recorder.log if e[0] == "new"])
        self.kw = kw
        self.in_flight = None
        self.closed = False
        Recorder.log.append(("new", self.name))

    def submit_device(self, ptr, n, w, h, after=None, **kw):
        assert self.in_flight is None, "one batch per context at a time"
        self.in_flight = ptr
        Recorder.log.append(("submit", self.name, ptr, None if after is None else after.name))

    def submit_batch(self, images, after=None, **kw):
        assert self.in_flight is None
        self.in_flight = int(images[0, 0, 0])
        Recorder.log.append(("submit_host", self.name, self.in_flight, None if after is None else after.name))

    def collect(self, unpack=True):
        assert self.in_flight is not None
        b, self.in_flight = self.in_flight, None
        Recorder.log.append(("collect", self.name, b))
        return [("markers of", b)]

    def pose_last(self, fiducial_len, K, D, unpack=True):
        assert self.in_flight is None  # (fid_pose_last is refused while a batch is in flight)
        Recorder.log.append(("pose", self.name))
        return [("poses", fiducial_len)]

    def close(self):
        self.closed = True


@pytest.fixture(autouse=True)
def _fresh_log():
    Recorder.log = []
    yield


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_batches_go_round_the_ring_and_come_back_in_order(depth):
    pipe = BatchPipeline("dict", depth=depth, fiducial_len=0.14, K=np.eye(3), detector_factory=Recorder, max_batch=4)
    assert [d.kw for d in pipe.detectors] == [{"max_batch": 4}] * depth
    out = []
    for b in range(7):
        done = pipe.push(100 + b, 4, 64, 48)
        assert (done is None) == (b < depth)  # the ring is full after `depth` batches: from then on one comes back per push
        if done is not None:
            out.append(done)
    out += pipe.flush()
    assert pipe.flush() == []
    assert [m[0][1] for m, _ in out] == [100 + b for b in range(7)]  # submission order
    assert all(p == [("poses", 0.14)] for _, p in out)
    subs = [e for e in Recorder.log if e[0] == "submit"]
    assert [e[1] for e in subs] == [b % depth for b in range(7)]  # context b % depth takes batch b ...
    assert [e[3] for e in subs] == [None if depth == 1 else (b - 1) % depth for b in range(7)]  # ... ordered after its ring predecessor
    # a context is collected (and its poses fetched) before it is handed the next batch
    for name in range(depth):
        mine = [e[0] for e in Recorder.log if e[0] in ("submit", "collect", "pose") and e[1] == name]
        assert mine == ["submit", "collect", "pose"] * (len(mine) // 3)
    pipe.close()
    assert pipe.detectors == []


def test_unordered_and_host_fed_pushes():
    pipe = BatchPipeline("dict", depth=2, ordered=False, detector_factory=Recorder)
    frames = [np.full((2, 4, 4), b, np.uint8) for b in range(3)]
    got = [pipe.push_host(f) for f in frames]
    assert got[0] is None and got[1] is None and got[2][0] == [("markers of", 0)] and got[2][1] is None  # (no camera: no poses)
    assert [e[3] for e in Recorder.log if e[0] == "submit_host"] == [None, None, None]
    rest = pipe.flush()
    assert [m[0][1] for m, _ in rest] == [1, 2]
    with pytest.raises(ValueError):
        BatchPipeline("dict", depth=0, detector_factory=Recorder)
    with BatchPipeline("dict", depth=2, detector_factory=Recorder) as p2:
        ds = list(p2.detectors)
    assert all(d.closed for d in ds)
