"""A stream of batches through several contexts in turn.

One context works on one batch at a time, and the last third of a batch's kernels (sort, too-close filter, identification,
corner refinement, pose: a few thousand candidates per frame) is latency-bound: the chip is far from full while it runs.  With
`depth` contexts, batch k + 1 is submitted (fid_submit_device) before batch k is collected (fid_collect), so that end runs under
the front of the next batch; fid_order_after holds batch k + 1 back until batch k is past its chip-filling kernels, so two fronts
never fight for the chip.  Results come back in submission order, one batch late per extra context.

The node's shape: imageCallback (aruco_detect.cpp:332-350) is called once per frame for as long as the camera runs -- the
frames keep coming, and a detection is published when it is ready.
"""
from __future__ import annotations

from collections import deque

import numpy as np

from .detector import ArucoDetector


class BatchPipeline:
    def __init__(self, dictionary, depth: int = 2, fiducial_len: float | None = None, K=None, D=None, ordered: bool = True,
                 detector_factory=ArucoDetector, **detector_kwargs):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.ordered = ordered  # a batch starts when the one before it is past its chip-filling kernels (fid_order_after)
        # (detector_factory: the class the contexts are made of -- the CPU tests of this ring's bookkeeping pass a recorder)
        self.detectors = [detector_factory(dictionary, **detector_kwargs) for _ in range(depth)]
        self._pose = None
        if fiducial_len is not None:
            self._pose = (float(fiducial_len), np.asarray(K, dtype=np.float64), np.zeros(5) if D is None else np.asarray(D, dtype=np.float64))
        self._next = 0
        self.last_collected = 0
        self._busy: deque[int] = deque()  # contexts with a batch in flight, oldest first

    @property
    def depth(self) -> int:
        return len(self.detectors)

    def _collect_oldest(self, unpack: bool):
        i = self._busy.popleft()
        self.last_collected = i
        det = self.detectors[i]
        markers = det.collect(unpack=unpack)
        poses = det.pose_last(*self._pose, unpack=unpack) if self._pose else None
        return markers, poses

    def push(self, data_ptr: int, nframes: int, width: int, height: int, unpack: bool = True, **kw):
        """Submit one batch (frames resident in HBM; they must stay there until the batch's results have been returned).
        Returns the (markers, poses) of the oldest batch in flight when all contexts were busy, else None."""
        done = None
        if len(self._busy) == self.depth:
            done = self._collect_oldest(unpack)
        i = self._next
        self._next = (i + 1) % self.depth
        # (the context before this one in the ring even when it is idle: a batch of a chain is laid out as one piece)
        prev = self.detectors[(i - 1) % self.depth] if (self.ordered and self.depth > 1) else None
        self.detectors[i].submit_device(data_ptr, nframes, width, height, after=prev, **kw)
        self._busy.append(i)
        return done

    def push_host(self, images: np.ndarray, unpack: bool = True, **kw):
        """push() for a batch in HOST memory (fid_submit_batch): its copy runs under the kernels of the batch before it.  The array
        must stay unchanged until the batch's results have been returned."""
        done = None
        if len(self._busy) == self.depth:
            done = self._collect_oldest(unpack)
        i = self._next
        self._next = (i + 1) % self.depth
        prev = self.detectors[(i - 1) % self.depth] if (self.ordered and self.depth > 1) else None
        self.detectors[i].submit_batch(images, after=prev, **kw)
        self._busy.append(i)
        return done

    def flush(self, unpack: bool = True):
        """Results of every batch still in flight, oldest first."""
        out = []
        while self._busy:
            out.append(self._collect_oldest(unpack))
        return out

    def close(self):
        for d in self.detectors:
            d.close()
        self.detectors = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
