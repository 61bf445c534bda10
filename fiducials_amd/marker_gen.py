"""Printable fiducial markers without cv2 / cairosvg / pdfunite -- the marker-generation row of the reference
(aruco_detect/scripts/create_markers.py:39-54, marker_generation/marker_gen.py:31-98).

The reference rasterises `aruco.drawMarker(dict, id, 2000)` to a PNG, embeds it in an SVG page template (140 mm marker, 4 mm
frame bars, cut marks 14 mm outside the corners, two measuring lines "should be exactly 14.0cm long", the label "<id> D<dicno>")
and converts / concatenates with cairosvg + pdfunite.  Here the page is the same template, but the marker is drawn as vector
cells straight from the dictionary bits (`Dictionary::drawMarker` semantics: (n + 2)^2 cells, one-cell black border, bit 1 =
white), as a self-contained SVG (`gen_svg`) or as one page of a multi-page vector PDF written directly (`write_pdf`).

    python -m fiducials_amd.marker_gen 100 112 markers.pdf [dictionary=7] [--paper-size letter|a4]
    python -m fiducials_amd.marker_gen 100 112 outdir/ --svg
"""
from __future__ import annotations

import argparse
import os
import sys

from .dictionary import Dictionary, get_predefined_dictionary

FID_LEN_MM = 140.0  # marker_gen.py:84 ("fid_len": 140.0): the printed marker is 14 cm, the node's default ~fiducial_len
PAPER = {"letter": (215.9, 279.4), "a4": (210.0, 297.0)}  # create_markers.py:41-44


def marker_cells(d: Dictionary, marker_id: int):
    """(n + 2) x (n + 2) matrix of 0 (black) / 1 (white): the cells of `Dictionary::drawMarker` with one border cell."""
    n = d.marker_size
    bits = d.bits(marker_id)
    cells = [[0] * (n + 2) for _ in range(n + 2)]
    for r in range(n):
        for c in range(n):
            cells[r + 1][c + 1] = int(bits[r, c])
    return cells


def page_layout(d: Dictionary, marker_id: int, paper_size):
    """The page as drawing primitives in millimetres (origin top left): black rects, white rects, hair lines, texts."""
    pw, ph = paper_size
    fl = FID_LEN_MM
    x0, y0 = (pw - fl) / 2, (ph - fl) / 2
    cells = marker_cells(d, marker_id)
    nc = len(cells)
    cs = fl / nc
    black = [(x0, y0, fl, fl)]  # the whole marker square, then the white cells on top
    white = [(x0 + c * cs, y0 + r * cs, cs, cs) for r in range(nc) for c in range(nc) if cells[r][c]]
    # the four 4 mm frame bars of the template (inside the marker's own black border)
    black += [(x0, y0, fl, 4.0), (x0, y0 + fl - 4.0, fl, 4.0), (x0, y0, 4.0, fl), (x0 + fl - 4.0, y0, 4.0, fl)]
    cut = max(fl / 10 * 1.4, 10)
    lines = []
    for sx, sy in ((-1, -1), (1, -1), (-1, 1), (1, 1)):  # cut marks at the four corners
        cx = pw / 2 + sx * (fl / 2 + cut)
        cy = ph / 2 + sy * (fl / 2 + cut)
        lines.append((cx, cy, cx - sx * 2, cy))
        lines.append((cx, cy, cx, cy - sy * 2))
    top, bottom = ph / 2 - fl / 2 - cut, ph / 2 + fl / 2 + cut
    left, right = pw / 2 - fl / 2 - cut, pw / 2 + fl / 2 + cut
    for yy in (top, bottom):  # measuring lines: exactly fid_len long
        lines.append((pw / 2 - fl / 2, yy, pw / 2 + fl / 2, yy))
    for xx in (left, right):
        lines.append((xx, ph / 2 - fl / 2, xx, ph / 2 + fl / 2))
    texts = [(pw / 2, top - 1, 8, "This line should be exactly %scm long." % (fl / 10)),
             (pw / 2, (ph + fl) / 2 + 30, 24, "%d D%d" % (marker_id, _dicno_of(d)))]
    return black, white, lines, texts


def _dicno_of(d: Dictionary) -> int:
    from .dictionary import PREDEFINED

    return PREDEFINED[d.name][0]


def _check_printable(d: Dictionary, marker_id: int, allow_fillers: bool) -> None:
    """A printed marker must interoperate with OpenCV-based detectors: ids whose codeword in the shipped table is a locally
    generated filler (Dictionary.pinned False) are refused unless the caller asks for them explicitly."""
    if not 0 <= marker_id < d.n_markers:
        raise ValueError(f"marker id {marker_id} not in {d.name}")
    if not d.pinned[marker_id] and not allow_fillers:
        raise ValueError(f"marker {marker_id} of {d.name}: the shipped table holds a locally generated filler codeword for this id, "
                         "not OpenCV's -- a print of it would not be read by the reference's detector (allow_fillers=True to print it anyway)")


def gen_svg(marker_id: int, dicno: int = 7, paper_size=PAPER["letter"], allow_fillers: bool = False) -> str:
    d = get_predefined_dictionary(dicno, allow_fillers=allow_fillers)
    _check_printable(d, marker_id, allow_fillers)
    black, white, lines, texts = page_layout(d, marker_id, paper_size)
    pw, ph = paper_size
    out = ['<svg width="%gmm" height="%gmm" viewBox="0 0 %g %g" version="1.1" xmlns="http://www.w3.org/2000/svg">' % (pw, ph, pw, ph)]
    for x, y, w, h in black:
        out.append('  <rect x="%.4f" y="%.4f" width="%.4f" height="%.4f" style="stroke:none; fill:black"/>' % (x, y, w, h))
    for x, y, w, h in white:
        out.append('  <rect class="cell" x="%.4f" y="%.4f" width="%.4f" height="%.4f" style="stroke:none; fill:white"/>' % (x, y, w, h))
    for x1, y1, x2, y2 in lines:
        out.append('  <line x1="%.4f" y1="%.4f" x2="%.4f" y2="%.4f" style="stroke:black; stroke-width:0.2"/>' % (x1, y1, x2, y2))
    for x, y, size, s in texts:
        out.append('  <text x="%.4f" y="%.4f" text-anchor="middle" style="font-family:arial; font-size:%gpt;">%s</text>' % (x, y, size * 25.4 / 72 * 0.75, s))
    out.append("</svg>")
    return "\n".join(out) + "\n"


def write_pdf(path: str, marker_ids, dicno: int = 7, paper_size=PAPER["letter"], allow_fillers: bool = False) -> None:
    """One marker per page, vector graphics, built-in Helvetica: a complete PDF 1.4 file written by hand."""
    d = get_predefined_dictionary(dicno, allow_fillers=allow_fillers)
    marker_ids = list(marker_ids)
    for mid in marker_ids:
        _check_printable(d, mid, allow_fillers)
    k = 72 / 25.4  # mm -> pt
    pw, ph = paper_size
    objs = []  # (object number -> bytes), numbered from 1

    def add(body: bytes) -> int:
        objs.append(body)
        return len(objs)

    font = add(b"<< /Type /Font /Subtype /Type1 /BaseFont /Helvetica >>")
    pages_id = 2 + 0  # reserved below
    add(b"")  # placeholder for /Pages (object 2)
    page_ids = []
    for mid in marker_ids:
        if not 0 <= mid < d.n_markers:
            raise ValueError(f"marker id {mid} not in {d.name}")
        black, white, lines, texts = page_layout(d, mid, paper_size)
        c = []

        def rect(x, y, w, h):
            c.append("%.3f %.3f %.3f %.3f re f" % (x * k, (ph - y - h) * k, w * k, h * k))

        c.append("0 g")
        for r in black:
            rect(*r)
        c.append("1 g")
        for r in white:
            rect(*r)
        c.append("0 G 0.5 w")
        for x1, y1, x2, y2 in lines:
            c.append("%.3f %.3f m %.3f %.3f l S" % (x1 * k, (ph - y1) * k, x2 * k, (ph - y2) * k))
        c.append("0 g")
        for x, y, size, s in texts:
            width = 0.5 * size * len(s)  # Helvetica averages half an em per character: centred well enough for a label
            c.append("BT /F1 %g Tf %.3f %.3f Td (%s) Tj ET" % (size, x * k - width / 2, (ph - y) * k, s.replace("(", "\\(").replace(")", "\\)")))
        stream = "\n".join(c).encode("ascii")
        content = add(b"<< /Length %d >>\nstream\n" % len(stream) + stream + b"\nendstream")
        page_ids.append(add(b"<< /Type /Page /Parent 2 0 R /MediaBox [0 0 %.3f %.3f] /Contents %d 0 R /Resources << /Font << /F1 %d 0 R >> >> >>"
                            % (pw * k, ph * k, content, font)))
    objs[pages_id - 1] = b"<< /Type /Pages /Kids [" + b" ".join(b"%d 0 R" % p for p in page_ids) + b"] /Count %d >>" % len(page_ids)
    catalog = add(b"<< /Type /Catalog /Pages 2 0 R >>")
    out = bytearray(b"%PDF-1.4\n%\xe2\xe3\xcf\xd3\n")
    offsets = []
    for i, body in enumerate(objs, 1):
        offsets.append(len(out))
        out += b"%d 0 obj\n" % i + body + b"\nendobj\n"
    xref = len(out)
    out += b"xref\n0 %d\n" % (len(objs) + 1) + b"0000000000 65535 f \n"
    for off in offsets:
        out += b"%010d 00000 n \n" % off
    out += b"trailer\n<< /Size %d /Root %d 0 R >>\nstartxref\n%d\n%%%%EOF\n" % (len(objs) + 1, catalog, xref)
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Generate Aruco Markers.")
    ap.add_argument("startId", type=int, help="start of marker range to generate")
    ap.add_argument("endId", type=int, help="end of marker range to generate")
    ap.add_argument("out", type=str, help="PDF file to store the markers in (or a directory with --svg)")
    ap.add_argument("dictionary", type=int, default=7, nargs="?", help="dictionary to generate from")
    ap.add_argument("--paper-size", dest="paper_size", default="letter", choices=sorted(PAPER), help="paper size to use (letter or a4)")
    ap.add_argument("--svg", action="store_true", help="one self-contained SVG per marker into the directory `out`")
    ap.add_argument("--allow-fillers", action="store_true",
                    help="also print ids whose codeword in the shipped table is a locally generated filler (NOT OpenCV's: such a print "
                         "is only read back by this repository's own tables)")
    a = ap.parse_args(argv)
    ids = list(range(a.startId, a.endId + 1))
    if a.svg:
        os.makedirs(a.out, exist_ok=True)
        for i in ids:
            with open(os.path.join(a.out, "marker%d.svg" % i), "w") as fh:
                fh.write(gen_svg(i, a.dictionary, PAPER[a.paper_size], a.allow_fillers))
    else:
        write_pdf(a.out, ids, a.dictionary, PAPER[a.paper_size], a.allow_fillers)
    print("After printing, please make sure that the long lines around the marker are EXACTLY 14.0cm long. "
          "This is required for accurate position estimation.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
