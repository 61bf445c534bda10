"""Synthetic camera frames for the BASELINE.json workloads.

mono8 frames: background 128 + low-frequency gradient (+-30) + N(0, sigma=2) sensor noise; markers are
`drawMarker`-semantics rasters (fiducials_amd.dictionary.draw_marker: (n+2)^2 cells, 1-cell black
border, bit 1 = white) with a one-cell white quiet zone, placed on a jittered grid, posed through a
pin-hole camera K = [1400 0 960; 0 1400 540; 0 0 1], D = 0 with in-plane rotation U(-pi, pi) and an
out-of-plane tilt <= 35 deg, rendered by inverse homography with 3x3 supersampling, then blurred with a
Gaussian sigma = 0.8.  RNG: numpy default_rng(seed); the bench uses seed = 1000 + frame_index
(cfg 3) and 10000*s + i for stream s (cfg 4).

The generator returns the ground-truth corner positions (TL, TR, BR, BL of the canonical marker) for an
accuracy report; parity is judged against the oracle, not against these.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .dictionary import Dictionary, draw_marker

K_DEFAULT = np.array([[1400.0, 0.0, 960.0], [0.0, 1400.0, 540.0], [0.0, 0.0, 1.0]])
MARKER_LEN = 0.14  # metres, node default ~fiducial_len (aruco_detect.cpp:612)


@dataclass
class SynthFrame:
    image: np.ndarray  # (H, W) uint8
    ids: np.ndarray  # (M,) int32
    corners: np.ndarray  # (M, 4, 2) float64 ground truth
    rvecs: np.ndarray  # (M, 3)
    tvecs: np.ndarray  # (M, 3)


def _rodrigues(r):
    a = np.linalg.norm(r)
    if a < 1e-12:
        return np.eye(3)
    k = r / a
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * (Kx @ Kx)


def _rot_to_rvec(R):
    c = (np.trace(R) - 1) / 2
    c = min(1.0, max(-1.0, c))
    ang = np.arccos(c)
    if ang < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(ang))
    return ax * ang


def _gauss_kernel(sigma):
    r = max(1, int(np.ceil(3 * sigma)))
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-x * x / (2 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _blur(img, sigma):
    k = _gauss_kernel(sigma)
    r = len(k) // 2
    p = np.pad(img, ((0, 0), (r, r)), mode="edge")
    out = np.zeros_like(img)
    for i, w in enumerate(k):
        out += w * p[:, i:i + img.shape[1]]
    p = np.pad(out, ((r, r), (0, 0)), mode="edge")
    out2 = np.zeros_like(img)
    for i, w in enumerate(k):
        out2 += w * p[i:i + img.shape[0], :]
    return out2


def make_frame(d: Dictionary, seed: int, width: int = 1920, height: int = 1080, n_markers: int = 20,
               ids: np.ndarray | None = None, K: np.ndarray | None = None, noise_sigma: float = 2.0,
               side_range=(96.0, 160.0), max_tilt_deg: float = 35.0, pinned_only: bool = False, border_bits: int = 1) -> SynthFrame:
    """border_bits: cells of black border the markers are drawn with (aruco::drawMarker's borderBits; the detector side is
    DetectorParameters::markerBorderBits, aruco_detect.cpp:718)."""
    rng = np.random.default_rng(seed)
    if K is None:
        K = K_DEFAULT.copy()
        K[0, 2] = width / 2.0
        K[1, 2] = height / 2.0
        K[0, 0] = K[1, 1] = 1400.0 * width / 1920.0
    f, cx0, cy0 = K[0, 0], K[0, 2], K[1, 2]

    # background
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    gdir = rng.uniform(0, 2 * np.pi)
    gamp = rng.uniform(10, 30)
    phase = rng.uniform(0, 2 * np.pi)
    img = 128.0 + gamp * np.sin((np.cos(gdir) * xx / width + np.sin(gdir) * yy / height) * np.pi + phase)
    img = img.astype(np.float32)

    # grid layout: as square as possible, jittered
    cols = int(np.ceil(np.sqrt(n_markers * width / height)))
    rows = int(np.ceil(n_markers / cols))
    cw, ch = width / cols, height / rows
    cells = rng.permutation(cols * rows)[:n_markers]
    if ids is None:
        pool = np.flatnonzero(d.pinned) if pinned_only else np.arange(d.n_markers)
        ids = rng.choice(pool, size=n_markers, replace=len(pool) < n_markers)
    ids = np.asarray(ids, dtype=np.int32)

    n = d.marker_size
    ncell = n + 2 * border_bits  # marker cells incl. black border
    q = 1.0  # quiet zone in cells
    gt = np.zeros((n_markers, 4, 2))
    rvecs = np.zeros((n_markers, 3))
    tvecs = np.zeros((n_markers, 3))
    max_side = min(cw, ch) * 0.62
    for mi in range(n_markers):
        cell = cells[mi]
        gx, gy = cell % cols, cell // cols
        side = rng.uniform(side_range[0], min(side_range[1], max_side))
        jx = (cw - side * 1.5) / 2 * rng.uniform(-0.6, 0.6)
        jy = (ch - side * 1.5) / 2 * rng.uniform(-0.6, 0.6)
        pcx = (gx + 0.5) * cw + max(jx, -cw / 2) * (1 if cw > side * 1.5 else 0)
        pcy = (gy + 0.5) * ch + max(jy, -ch / 2) * (1 if ch > side * 1.5 else 0)
        theta = rng.uniform(-np.pi, np.pi)
        tilt = np.deg2rad(rng.uniform(0, max_tilt_deg))
        tdir = rng.uniform(0, 2 * np.pi)
        # marker frame: x right, y up, z out of the marker (aruco_detect.cpp:151-161); a marker facing the
        # camera with canonical TL at the image top-left has R = diag(1,-1,-1)
        R = _rodrigues(np.array([np.cos(tdir), np.sin(tdir), 0.0]) * tilt) @ _rodrigues(np.array([0, 0, theta])) @ np.diag([1.0, -1.0, -1.0])
        Z = f * MARKER_LEN / side
        t = np.array([(pcx - cx0) / f * Z, (pcy - cy0) / f * Z, Z])
        rvecs[mi] = _rot_to_rvec(R)
        tvecs[mi] = t

        def proj(P):
            Pc = P @ R.T + t
            return np.stack([f * Pc[:, 0] / Pc[:, 2] + cx0, f * Pc[:, 1] / Pc[:, 2] + cy0], axis=1)

        L = MARKER_LEN
        obj = np.array([[-L / 2, L / 2, 0], [L / 2, L / 2, 0], [L / 2, -L / 2, 0], [-L / 2, -L / 2, 0]])
        gt[mi] = proj(obj)
        # homography image -> texture (cells), texture spans [-q, ncell+q] on both axes
        cellm = L / ncell
        ext = L / 2 + q * cellm
        objq = np.array([[-ext, ext, 0], [ext, ext, 0], [ext, -ext, 0], [-ext, -ext, 0]])
        pq = proj(objq)
        tex = np.array([[-q, -q], [ncell + q, -q], [ncell + q, ncell + q], [-q, ncell + q]], dtype=np.float64)
        A = []
        for (x, y), (u, v) in zip(pq, tex):
            A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        _, _, Vt = np.linalg.svd(np.array(A))
        H = Vt[-1].reshape(3, 3)
        x0 = int(max(0, np.floor(pq[:, 0].min()) - 1)); x1 = int(min(width, np.ceil(pq[:, 0].max()) + 2))
        y0 = int(max(0, np.floor(pq[:, 1].min()) - 1)); y1 = int(min(height, np.ceil(pq[:, 1].max()) + 2))
        if x1 <= x0 or y1 <= y0:
            continue
        ss = 3
        sub = (np.arange(ss) + 0.5) / ss - 0.5
        px = (np.arange(x0, x1)[:, None] + sub[None, :]).reshape(-1)
        py = (np.arange(y0, y1)[:, None] + sub[None, :]).reshape(-1)
        PX, PY = np.meshgrid(px, py)
        den = H[2, 0] * PX + H[2, 1] * PY + H[2, 2]
        U = (H[0, 0] * PX + H[0, 1] * PY + H[0, 2]) / den
        V = (H[1, 0] * PX + H[1, 1] * PY + H[1, 2]) / den
        inside = (U >= -q) & (U < ncell + q) & (V >= -q) & (V < ncell + q)
        tiny = np.full((ncell + 2, ncell + 2), 230.0, dtype=np.float32)  # quiet zone white
        m = draw_marker(d, int(ids[mi]), ncell, border_bits).astype(np.float32)
        tiny[1:-1, 1:-1] = np.where(m > 0, 230.0, 25.0)
        ui = np.clip(np.floor(U + q).astype(np.int64), 0, ncell + 1)
        vi = np.clip(np.floor(V + q).astype(np.int64), 0, ncell + 1)
        val = tiny[vi, ui]
        hh, ww = (y1 - y0), (x1 - x0)
        cov = inside.reshape(hh, ss, ww, ss).mean(axis=(1, 3)).astype(np.float32)
        col = (val * inside).reshape(hh, ss, ww, ss).sum(axis=(1, 3)).astype(np.float32)
        cnt = inside.reshape(hh, ss, ww, ss).sum(axis=(1, 3)).astype(np.float32)
        colm = np.where(cnt > 0, col / np.maximum(cnt, 1), 0)
        img[y0:y1, x0:x1] = img[y0:y1, x0:x1] * (1 - cov) + colm * cov

    img = _blur(img, 0.8)
    if noise_sigma > 0:
        img = img + rng.normal(0.0, noise_sigma, size=img.shape).astype(np.float32)
    out = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return SynthFrame(out, ids, gt, rvecs, tvecs)


def stag_code_locations() -> np.ndarray:
    """The 48 code points of an STag marker in marker coordinates [0, 1]^2 (Stag::fillCodeLocations, Stag.cpp:129-174)."""
    hp = 1.570796326794897
    inner = 0.4 * 0.9
    polar = [(0.088363142525988, 0.785398163397448), (0.206935928182607, 0.459275804122858), (0.206935928182607, hp - 0.459275804122858),
             (0.313672146827381, 0.200579720495241), (0.327493143484516, 0.591687617505840), (0.327493143484516, hp - 0.591687617505840),
             (0.313672146827381, hp - 0.200579720495241), (0.437421957035861, 0.145724938287167), (0.437226762361658, 0.433363129825345),
             (0.430628029742607, 0.785398163397448), (0.437226762361658, hp - 0.433363129825345), (0.437421957035861, hp - 0.145724938287167)]
    out = np.zeros((48, 2))
    for i in range(4):
        for k, (rad, ang) in enumerate(polar):
            a = ang + i * hp
            out[k + 12 * i] = (0.5 + np.cos(a) * rad * (inner / 0.5), 0.5 - np.sin(a) * rad * (inner / 0.5))
    return out


def make_stag_frame(codewords: np.ndarray, seed: int, width: int = 1920, height: int = 1080, n_markers: int = 20,
                    ids: np.ndarray | None = None, noise_sigma: float = 2.0, side_range=(110.0, 200.0),
                    max_tilt_deg: float = 30.0, dot_radius: float = 0.033) -> SynthFrame:
    """STag markers (BASELINE cfg 5): black square, white disc of radius 0.4, a black dot at code point i where bit i of the
    codeword is 1, white quiet zone; placed and posed like make_frame.  codewords: the library (uint64, rotation 0 block
    first), ids index its first quarter.  Ground truth corners: marker corners (0,0) (1,0) (1,1) (0,1)."""
    rng = np.random.default_rng(seed)
    nlib = len(codewords) // 4
    f = 1400.0 * width / 1920.0
    cx0, cy0 = width / 2.0, height / 2.0
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    gdir, gamp, phase = rng.uniform(0, 2 * np.pi), rng.uniform(10, 25), rng.uniform(0, 2 * np.pi)
    img = (170.0 + gamp * np.sin((np.cos(gdir) * xx / width + np.sin(gdir) * yy / height) * np.pi + phase)).astype(np.float32)
    cols = int(np.ceil(np.sqrt(n_markers * width / height)))
    rows = int(np.ceil(n_markers / cols))
    cw, ch = width / cols, height / rows
    cells = rng.permutation(cols * rows)[:n_markers]
    if ids is None:
        ids = rng.choice(nlib, size=n_markers, replace=nlib < n_markers)
    ids = np.asarray(ids, dtype=np.int32)
    locs = stag_code_locations()
    gt = np.zeros((n_markers, 4, 2))
    rvecs = np.zeros((n_markers, 3))
    tvecs = np.zeros((n_markers, 3))
    qz = 0.18  # quiet zone, marker units
    max_side = min(cw, ch) * 0.6
    for mi in range(n_markers):
        cell = cells[mi]
        gx, gy = cell % cols, cell // cols
        side = rng.uniform(min(side_range[0], max_side), min(side_range[1], max_side))
        pcx = (gx + 0.5) * cw + (cw - side * 1.6) / 2 * rng.uniform(-0.5, 0.5)
        pcy = (gy + 0.5) * ch + (ch - side * 1.6) / 2 * rng.uniform(-0.5, 0.5)
        theta = rng.uniform(-np.pi, np.pi)
        tilt = np.deg2rad(rng.uniform(0, max_tilt_deg))
        tdir = rng.uniform(0, 2 * np.pi)
        R = _rodrigues(np.array([np.cos(tdir), np.sin(tdir), 0.0]) * tilt) @ _rodrigues(np.array([0, 0, theta]))
        L = MARKER_LEN
        Z = f * L / side
        t = np.array([(pcx - cx0) / f * Z, (pcy - cy0) / f * Z, Z])
        rvecs[mi] = _rot_to_rvec(R)
        tvecs[mi] = t

        def proj(uv):  # marker units -> pixels; marker plane: x right, y down (image-like), centred
            P = np.concatenate([(uv - 0.5) * L, np.zeros((len(uv), 1))], axis=1)
            Pc = P @ R.T + t
            return np.stack([f * Pc[:, 0] / Pc[:, 2] + cx0, f * Pc[:, 1] / Pc[:, 2] + cy0], axis=1)

        gt[mi] = proj(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float))
        tex = np.array([[-qz, -qz], [1 + qz, -qz], [1 + qz, 1 + qz], [-qz, 1 + qz]], float)
        pq = proj(tex)
        A = []
        for (x, y), (u, v) in zip(pq, tex):
            A.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            A.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        _, _, Vt = np.linalg.svd(np.array(A))
        Hm = Vt[-1].reshape(3, 3)
        x0 = int(max(0, np.floor(pq[:, 0].min()) - 1)); x1 = int(min(width, np.ceil(pq[:, 0].max()) + 2))
        y0 = int(max(0, np.floor(pq[:, 1].min()) - 1)); y1 = int(min(height, np.ceil(pq[:, 1].max()) + 2))
        if x1 <= x0 or y1 <= y0:
            continue
        ss = 3
        sub = (np.arange(ss) + 0.5) / ss - 0.5
        px = (np.arange(x0, x1)[:, None] + sub[None, :]).reshape(-1)
        py = (np.arange(y0, y1)[:, None] + sub[None, :]).reshape(-1)
        PX, PY = np.meshgrid(px, py)
        den = Hm[2, 0] * PX + Hm[2, 1] * PY + Hm[2, 2]
        U = (Hm[0, 0] * PX + Hm[0, 1] * PY + Hm[0, 2]) / den
        V = (Hm[1, 0] * PX + Hm[1, 1] * PY + Hm[1, 2]) / den
        inside = (U >= -qz) & (U < 1 + qz) & (V >= -qz) & (V < 1 + qz)
        val = np.full(U.shape, 235.0, dtype=np.float32)
        sq = (U >= 0) & (U < 1) & (V >= 0) & (V < 1)
        val[sq] = 25.0
        val[(U - 0.5) ** 2 + (V - 0.5) ** 2 < 0.4 ** 2] = 235.0
        word = int(codewords[int(ids[mi])])
        for i in range(48):
            if (word >> i) & 1:
                val[(U - locs[i, 0]) ** 2 + (V - locs[i, 1]) ** 2 < dot_radius ** 2] = 25.0
        hh, ww = (y1 - y0), (x1 - x0)
        cnt = inside.reshape(hh, ss, ww, ss).sum(axis=(1, 3)).astype(np.float32)
        col = (val * inside).reshape(hh, ss, ww, ss).sum(axis=(1, 3)).astype(np.float32)
        cov = cnt / (ss * ss)
        colm = np.where(cnt > 0, col / np.maximum(cnt, 1), 0)
        img[y0:y1, x0:x1] = img[y0:y1, x0:x1] * (1 - cov) + colm * cov
    img = _blur(img, 0.8)
    if noise_sigma > 0:
        img = img + rng.normal(0.0, noise_sigma, size=img.shape).astype(np.float32)
    return SynthFrame(np.clip(np.rint(img), 0, 255).astype(np.uint8), ids, gt, rvecs, tvecs)


def make_batch(d: Dictionary, seeds, **kw) -> tuple[np.ndarray, list[SynthFrame]]:
    frames = [make_frame(d, int(s), **kw) for s in seeds]
    return np.stack([f.image for f in frames]), frames
