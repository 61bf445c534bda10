"""Marker generation (reference: aruco_detect/scripts/create_markers.py, marker_generation/marker_gen.py):
the emitted vector pages carry exactly the cells of Dictionary::drawMarker, print at 140 mm, and what they draw is read back
by the CPU oracle with the right id."""
import re

import numpy as np

import oracle
from fiducials_amd import marker_gen
from fiducials_amd.dictionary import draw_marker, get_predefined_dictionary


def _rasterise(svg, px_per_mm=2.0):
    m = re.search(r'viewBox="0 0 ([\d.]+) ([\d.]+)"', svg)
    pw, ph = float(m.group(1)), float(m.group(2))
    img = np.full((int(round(ph * px_per_mm)), int(round(pw * px_per_mm))), 255, np.uint8)
    for mm in re.finditer(r'<rect (?:class="cell" )?x="([\d.]+)" y="([\d.]+)" width="([\d.]+)" height="([\d.]+)" style="stroke:none; fill:(black|white)"', svg):
        x, y, w, h = (float(mm.group(i)) for i in range(1, 5))
        x0, y0, x1, y1 = (int(round(v * px_per_mm)) for v in (x, y, x + w, y + h))
        img[y0:y1, x0:x1] = 0 if mm.group(5) == "black" else 255
    return img, pw, ph


def test_svg_page_is_the_reference_template_with_vector_cells():
    d = get_predefined_dictionary(7)
    for mid, paper in ((403, "letter"), (1, "a4")):
        svg = marker_gen.gen_svg(mid, 7, marker_gen.PAPER[paper])
        pw, ph = marker_gen.PAPER[paper]
        assert f'width="{pw:g}mm"' in svg and ">%d D7<" % mid in svg and "exactly 14.0cm" in svg
        cells = np.array(marker_gen.marker_cells(d, mid))
        assert cells.shape == (7, 7) and not cells[0].any() and not cells[:, 0].any() and not cells[-1].any() and not cells[:, -1].any()
        assert np.array_equal(cells[1:-1, 1:-1], d.bits(mid))
        assert svg.count('class="cell"') == int(cells.sum())
        # the drawn marker == aruco::drawMarker at the same size (nearest-neighbour upscale of the cell matrix)
        img, _, _ = _rasterise(svg, px_per_mm=2.0)
        side = int(round(marker_gen.FID_LEN_MM * 2.0))
        x0, y0 = int(round((pw - 140) / 2 * 2.0)), int(round((ph - 140) / 2 * 2.0))
        assert np.array_equal(img[y0:y0 + side, x0:x0 + side], draw_marker(d, mid, side))
        # and the page is read back with the right id (white paper = quiet zone)
        ids, corners = oracle.detect(img, d)
        assert ids.tolist() == [mid]
        assert abs(np.linalg.norm(corners[0][0] - corners[0][1]) - side) < 1.5


def test_pdf_is_a_well_formed_multi_page_file(tmp_path):
    path = tmp_path / "markers.pdf"
    import pytest
    with pytest.raises(ValueError, match="filler"):  # ids 101, 102: the shipped table does not hold OpenCV's codeword for them
        marker_gen.main(["100", "103", str(path), "7", "--paper-size", "a4"])
    assert marker_gen.main(["100", "103", str(path), "7", "--paper-size", "a4", "--allow-fillers"]) == 0
    raw = path.read_bytes()
    assert raw.startswith(b"%PDF-1.4") and raw.rstrip().endswith(b"%%EOF")
    assert raw.count(b"/Type /Page ") == 4 and b"/Count 4" in raw and b"(102 D7) Tj" in raw
    # every xref offset points at the object it names
    xref = int(re.search(rb"startxref\n(\d+)\n", raw).group(1))
    assert raw[xref:xref + 4] == b"xref"
    n = int(re.search(rb"xref\n0 (\d+)\n", raw).group(1))
    rows = raw[xref:].split(b"\n")[2:2 + n]
    for i, row in enumerate(rows[1:], 1):
        off = int(row[:10])
        assert raw[off:off + len(b"%d 0 obj" % i)] == b"%d 0 obj" % i
    # the white cells of one page: (bits set) rectangles after the "1 g" switch
    d = get_predefined_dictionary(7)
    first = raw.split(b"stream\n")[1].split(b"\nendstream")[0].decode()
    white = first.split("\n1 g\n")[1].split("\n0 G")[0]
    assert white.count(" re f") == int(d.bits(100).sum())


def test_svg_mode_and_bad_ids(tmp_path):
    assert marker_gen.main(["5", "6", str(tmp_path / "svgs"), "0", "--svg"]) == 0
    assert sorted(p.name for p in (tmp_path / "svgs").iterdir()) == ["marker5.svg", "marker6.svg"]
    import pytest
    with pytest.raises(ValueError):
        marker_gen.gen_svg(50, 0)  # DICT_4X4_50 has ids 0..49
    with pytest.raises(ValueError, match="filler"):
        marker_gen.gen_svg(3, 0)  # id 3 of 4X4: a filler codeword in the shipped table
    with pytest.raises(ValueError, match="not available"):
        marker_gen.gen_svg(1, 10)  # DICT_6X6_250: filler-only family
    assert "1 D10" in marker_gen.gen_svg(1, 10, allow_fillers=True)
