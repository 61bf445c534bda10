// fid_draw.hip -- the overlay of /fiducial_images: what imageCallback draws on its BGR8 copy of the
// frame before image_pub.publish (aruco_detect.cpp:381-387).  Part of the fid_api.hip translation unit.
//
// Host code on purpose: the reference draws on the CPU too (aruco::drawDetectedMarkers on cv_ptr->image), the work is a few
// thousand pixels per frame, and the image is a host buffer on both sides of the call.
//
// What is restated, exactly: cv::line(img, Point(p0), Point(p1), borderColor, thickness 1, LINE_8, shift 0) for the four sides
// of every marker -- drawing.cpp line() -> ThickLine() -> Line() -> LineIterator(img, pt1, pt2, 8, left_to_right = true), with
// clipLine() for end points outside the image; Point2f -> Point is saturate_cast<int> = cvRound (round half to even).
// What is NOT drawn (and is not claimed): the LINE_AA rectangle on the first corner and putText("id=<n>", FONT_HERSHEY_SIMPLEX,
// 0.5, thickness 2) -- LineAA's filter / slope-correction tables and the Hershey glyph tables are third-party data that is not
// in the reference tree nor on this machine, so they could not be pinned; aruco::drawAxis (:431, 3-px thick lines =
// FillConvexPoly + circles) likewise.  FID_DRAW_FIRST_CORNER_LINE8 adds the first-corner square as four LINE_8 lines in
// cornerColor for callers that want the cue -- those pixels are NOT the reference's (its square is anti-aliased).
#include <math.h>

namespace {

struct DrawPt {
    long long x, y;
};

// drawing.cpp clipLine(Size2l, Point2l&, Point2l&)
bool draw_clip_line(long long W, long long H, DrawPt &p1, DrawPt &p2)
{
    const long long right = W - 1, bottom = H - 1;
    if (W <= 0 || H <= 0) return false;
    long long &x1 = p1.x, &y1 = p1.y, &x2 = p2.x, &y2 = p2.y;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (x2 - x1) / (y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (x2 - x1) / (y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (y2 - y1) / (x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (y2 - y1) / (x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// Line(): LineIterator(img, pt1, pt2, connectivity 8, left_to_right true), three bytes per pixel
void draw_line8(uint8_t *img, int W, int H, long long stride, DrawPt a, DrawPt b, const uint8_t color[3])
{
    if ((unsigned long long)a.x >= (unsigned long long)W || (unsigned long long)b.x >= (unsigned long long)W ||
        (unsigned long long)a.y >= (unsigned long long)H || (unsigned long long)b.y >= (unsigned long long)H)
        if (!draw_clip_line(W, H, a, b)) return;
    // clipLine() does not look at y again after the x clip; with end points near +-2^31 its double arithmetic is off by hundreds
    // of pixels and the reference would write outside its image there.  Such a line is left out instead.
    if ((unsigned long long)a.x >= (unsigned long long)W || (unsigned long long)b.x >= (unsigned long long)W ||
        (unsigned long long)a.y >= (unsigned long long)H || (unsigned long long)b.y >= (unsigned long long)H)
        return;
    long long dx = b.x - a.x, dy = b.y - a.y;
    long long s = dx < 0 ? -1 : 0;
    // left_to_right: the walk starts at the end point with the smaller x
    dx = (dx ^ s) - s;
    dy = (dy ^ s) - s;
    a.x ^= (a.x ^ b.x) & s;
    a.y ^= (a.y ^ b.y) & s;
    uint8_t *ptr = img + a.y * stride + a.x * 3;
    long long bt_pix = 3, istep = stride;
    s = dy < 0 ? -1 : 0;
    dy = (dy ^ s) - s;
    istep = (istep ^ s) - s;
    s = dy > dx ? -1 : 0;
    // conditional swaps
    dx ^= dy & s;
    dy ^= dx & s;
    dx ^= dy & s;
    bt_pix ^= istep & s;
    istep ^= bt_pix & s;
    bt_pix ^= istep & s;
    long long err = dx - (dy + dy);
    const long long plusDelta = dx + dx, minusDelta = -(dy + dy), plusStep = istep, minusStep = bt_pix;
    const long long count = dx + 1;
    for (long long i = 0; i < count; i++) {
        ptr[0] = color[0];
        ptr[1] = color[1];
        ptr[2] = color[2];
        const long long mask = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & mask);
        ptr += minusStep + (plusStep & mask);
    }
}

// Point2f -> Point: saturate_cast<int>(float) = cvRound = cvtss2si on the reference's x86-64 build: round to nearest even, and
// the "integer indefinite" INT_MIN for NaN, +-inf and everything outside the int range (CORNER_REFINE_CONTOUR crosses two
// fitted lines: near-parallel ones give such corners).  The result always fits 32 bits, so the 64-bit line arithmetic below
// (OpenCV's own Point2l) cannot overflow.
inline long long draw_cv_round(float v)
{
    if (!(v >= -2147483648.f && v < 2147483648.f)) return -2147483648LL;  // (NaN fails both comparisons)
    return (long long)lrintf(v);
}

}  // namespace

extern "C" {

fid_status fid_to_bgr(const uint8_t *img, int32_t width, int32_t height, int32_t stride, fid_encoding enc, uint8_t *out_bgr, int64_t out_bytes)
{
    if (!img || !out_bgr || width < 1 || height < 1) return FID_E_INVALID_ARG;
    const int bpp = enc == FID_ENC_MONO8 ? 1 : ((enc == FID_ENC_BGRA8 || enc == FID_ENC_RGBA8) ? 4 : 3);
    if (enc != FID_ENC_MONO8 && enc != FID_ENC_BGR8 && enc != FID_ENC_RGB8 && enc != FID_ENC_BGRA8 && enc != FID_ENC_RGBA8) return FID_E_INVALID_ARG;
    if ((int64_t)stride < (int64_t)width * bpp) return FID_E_INVALID_ARG;  // (64-bit: width * bpp wraps for absurd widths)
    if (out_bytes < (int64_t)width * height * 3) return FID_E_CAPACITY;
    const bool swap = enc == FID_ENC_RGB8 || enc == FID_ENC_RGBA8;
    for (int y = 0; y < height; y++) {
        const uint8_t *s = img + (size_t)y * stride;
        uint8_t *o = out_bgr + (size_t)y * width * 3;
        if (enc == FID_ENC_MONO8) {
            for (int x = 0; x < width; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = s[x];  // cvtColor(GRAY2BGR)
        } else {
            for (int x = 0; x < width; x++) {
                const uint8_t *p = s + (size_t)x * bpp;
                o[3 * x] = swap ? p[2] : p[0];
                o[3 * x + 1] = p[1];
                o[3 * x + 2] = swap ? p[0] : p[2];
            }
        }
    }
    return FID_OK;
}

fid_status fid_draw_detected_markers(uint8_t *bgr, int32_t width, int32_t height, int32_t stride, const fid_marker *markers, int32_t n,
                                     uint32_t flags)
{
    if (!bgr || width < 1 || height < 1 || (int64_t)stride < (int64_t)width * 3 || n < 0 || (n > 0 && !markers)) return FID_E_INVALID_ARG;
    if (flags & ~(uint32_t)FID_DRAW_FIRST_CORNER_LINE8) return FID_E_INVALID_ARG;
    // borderColor = Scalar(0, 255, 0) (the default imageCallback leaves in place); cornerColor = borderColor with val[1] and
    // val[2] swapped (drawDetectedMarkers; its own comment says "G and B", the code swaps G and R): (0, 0, 255) = red in BGR
    const uint8_t border[3] = {0, 255, 0}, corner[3] = {0, 0, 255};
    for (int i = 0; i < n; i++) {
        const float *c = markers[i].corners;
        for (int j = 0; j < 4; j++) {
            const int k = (j + 1) & 3;
            const DrawPt p0 = {draw_cv_round(c[2 * j]), draw_cv_round(c[2 * j + 1])}, p1 = {draw_cv_round(c[2 * k]), draw_cv_round(c[2 * k + 1])};
            draw_line8(bgr, width, height, stride, p0, p1, border);
        }
        if (flags & FID_DRAW_FIRST_CORNER_LINE8) {
            // rectangle(img, c0 - (3, 3), c0 + (3, 3), cornerColor, 1, LINE_AA) drawn with LINE_8 sides instead: NOT the reference's pixels
            const float x0 = c[0] - 3.f, y0 = c[1] - 3.f, x1 = c[0] + 3.f, y1 = c[1] + 3.f;
            const DrawPt q[4] = {{draw_cv_round(x0), draw_cv_round(y0)}, {draw_cv_round(x1), draw_cv_round(y0)}, {draw_cv_round(x1), draw_cv_round(y1)},
                                 {draw_cv_round(x0), draw_cv_round(y1)}};
            for (int j = 0; j < 4; j++) draw_line8(bgr, width, height, stride, q[j], q[(j + 1) & 3], corner);
        }
    }
    return FID_OK;
}

}  // extern "C"
