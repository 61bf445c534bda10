#!/usr/bin/env python3
"""Randomised JPEG parity sweep (run on the GPU box, needs Pillow): random sizes, contents, qualities, sampling layouts,
restart intervals, standard and optimised tables -- the device decode must equal libjpeg-turbo's (Pillow) byte for byte, as
BGR and as the detector's gray image; batches mix layouts.  Usage: python tools/gpu_jpeg_stress.py [n_cases] [seed]"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from fiducials_amd import jpeg as fj  # noqa: E402


def gray_of(bgr):
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    done = 0
    while done < n:
        w, h = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        nb = int(rng.integers(1, 5))
        files, refs = [], []
        for _ in range(nb):
            kind = rng.integers(0, 4)
            if kind == 0:
                a = rng.integers(0, 256, (h, w, 3))
            elif kind == 1:
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), (xx + yy) % 256], -1)
            elif kind == 2:
                a = np.full((h, w, 3), int(rng.integers(0, 256)))
                for _ in range(int(rng.integers(1, 30))):
                    x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
                    a[y:y + int(rng.integers(1, 80)), x:x + int(rng.integers(1, 80))] = rng.integers(0, 256, 3)
            else:
                a = np.clip(rng.normal(128, 40, (h, w, 3)), 0, 255)
            a = a.astype(np.uint8)
            kw = dict(quality=int(rng.integers(1, 101)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
            r = rng.integers(0, 4)
            if r == 1:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
            elif r == 2:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            im = Image.fromarray(a[..., 0] if rng.integers(0, 5) == 0 else a)
            b = io.BytesIO()
            try:
                im.save(b, "JPEG", **kw)
            except OSError:  # (the encoder refuses some option combinations on tiny images)
                b = io.BytesIO()
                im.save(b, "JPEG", quality=kw["quality"], subsampling=kw["subsampling"])
            files.append(b.getvalue())
            refs.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB"))[..., ::-1])
        dec = fj.JpegDecoder(max_width=w, max_height=h, max_batch=nb)
        try:
            got = dec.decode(files, "bgr8")
            gotg = dec.decode(files, "mono8")
        except fj.FidError as e:
            if e.status != 4:
                raise
            print("  capacity reported (noise at quality ~100 can exceed two bytes per pixel):", w, h, [len(f) for f in files], flush=True)
            done += nb
            continue
        finally:
            dec.close()
        for k in range(nb):
            ok = np.array_equal(got[k], refs[k]) and np.array_equal(gotg[k], gray_of(refs[k]))
            if not ok:
                bad += 1
                print("MISMATCH", w, h, k, fj.probe(files[k]), flush=True)
            done += 1
    print("jpeg stress:", done, "files,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
