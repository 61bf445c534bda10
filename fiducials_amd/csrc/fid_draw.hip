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

// ---- cv_bridge::toCvCopy(msg, "bgr8") for the encodings a raw (uncompressed) camera driver publishes beside the five the device
// path takes itself (aruco_detect.cpp:348).  cv_bridge.cpp toCvCopy -> convertColor: a list of conversions by (source, target)
// format -- cvtColor for the colour layout, then, when the bit depths differ, Mat::convertTo(8U, 255. / 65535.).
//   * mono16 / bgr16 / rgb16 / bgra16 / rgba16: layout first (GRAY2BGR / RGB2BGR / BGRA2BGR / RGBA2BGR are pure channel moves),
//     then every sample through convertTo: cvtScale<ushort, uchar, float> = saturate_cast<uchar>(cvRound(v * (float)(255. / 65535.)))
//     with the product formed in float; big-endian messages are byte-swapped first (cv_bridge does so when the host differs);
//   * bayer_rggb8 / bayer_bggr8 / bayer_gbrg8 / bayer_grbg8: cv_bridge maps them to COLOR_BayerBG / RG / GR / GB2BGR (OpenCV names a
//     pattern by the pixels at (1,1),(1,2)); cv::demosaicing's bilinear Bayer2RGB_<uchar>: an interior pixel keeps its own colour,
//     the other two are (a + b + 1) >> 1 of the two or (a + b + c + d + 2) >> 2 of the four nearest samples of that colour; the
//     first / last column of an interior row repeat their neighbour, then the first / last row repeat theirs; an image of
//     height <= 2 comes out black.
// Restated from the published OpenCV 4.2 / cv_bridge (noetic) sources, which are not on this machine: PARITY UNPINNED (no
// reference fixture uses these encodings); tests/test_overlay.py checks them against an independent statement of the same rules.
//   * yuv422 (UYVY): cvtColor(COLOR_YUV2BGR_UYVY), BT.601 in 20-bit fixed point (below).
// The 16-bit Bayer encodings and everything else stay FID_E_UNSUPPORTED: the node reports them like the cv_bridge exception it
// would catch.
static inline uint8_t cvb_scale_16_to_8(uint16_t v)
{
    const float a = (float)(255. / 65535.);
    const float p = (float)v * a;
    const long r = lrintf(p);  // cvRound: nearest, ties to even
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

static void cvb_bayer_to_bgr(const uint8_t *src, int W, int H, long long step, int blue0, int green0, uint8_t *dst)
{
    const long long ds = (long long)W * 3;
    if (H <= 2 || W <= 2) {
        // (height <= 2: both border rows are written as zeros and nothing lies between them; width <= 2 is refused by the caller)
        memset(dst, 0, (size_t)ds * (size_t)H);
        return;
    }
    int blue = blue0, start_with_green = green0;
    for (int i = 0; i < H - 2; i++) {
        const uint8_t *bayer = src + (long long)i * step;
        const uint8_t *bayer_end = bayer + (W - 2);
        uint8_t *dst0 = dst + (long long)(i + 1) * ds + 3 + 1;  // pixel (i + 1, 1), its middle (green) channel
        uint8_t *d = dst0;
        const long long bs = step;
        int t0, t1;
        if (start_with_green) {
            t0 = (bayer[1] + bayer[bs * 2 + 1] + 1) >> 1;
            t1 = (bayer[bs] + bayer[bs + 2] + 1) >> 1;
            d[-blue] = (uint8_t)t0;
            d[0] = bayer[bs + 1];
            d[blue] = (uint8_t)t1;
            bayer++;
            d += 3;
        }
        for (; bayer <= bayer_end - 2; bayer += 2, d += 6) {
            t0 = (bayer[0] + bayer[2] + bayer[bs * 2] + bayer[bs * 2 + 2] + 2) >> 2;
            t1 = (bayer[1] + bayer[bs] + bayer[bs + 2] + bayer[bs * 2 + 1] + 2) >> 2;
            d[-blue] = (uint8_t)t0;
            d[0] = (uint8_t)t1;
            d[blue] = bayer[bs + 1];
            t0 = (bayer[2] + bayer[bs * 2 + 2] + 1) >> 1;
            t1 = (bayer[bs + 1] + bayer[bs + 3] + 1) >> 1;
            d[3 - blue] = (uint8_t)t0;  // (the green pixel beside it: its vertical pair is the colour this row does not carry,
            d[3] = bayer[bs + 2];       //  its horizontal pair the colour of the pixel before it)
            d[3 + blue] = (uint8_t)t1;
        }
        if (bayer < bayer_end) {
            t0 = (bayer[0] + bayer[2] + bayer[bs * 2] + bayer[bs * 2 + 2] + 2) >> 2;
            t1 = (bayer[1] + bayer[bs] + bayer[bs + 2] + bayer[bs * 2 + 1] + 2) >> 2;
            d[-blue] = (uint8_t)t0;
            d[0] = (uint8_t)t1;
            d[blue] = bayer[bs + 1];
        }
        // the first and the last pixel of the row repeat their neighbours
        dst0[-4] = dst0[-1];
        dst0[-3] = dst0[0];
        dst0[-2] = dst0[1];
        dst0[(W - 2) * 3 - 1] = dst0[(W - 2) * 3 - 4];
        dst0[(W - 2) * 3] = dst0[(W - 2) * 3 - 3];
        dst0[(W - 2) * 3 + 1] = dst0[(W - 2) * 3 - 2];
        blue = -blue;
        start_with_green = !start_with_green;
    }
    for (long long k = 0; k < ds; k++) {  // the first and the last row repeat theirs
        dst[k] = dst[k + ds];
        dst[k + (long long)(H - 1) * ds] = dst[k + (long long)(H - 2) * ds];
    }
}

fid_status fid_image_to_bgr8(const uint8_t *img, int32_t width, int32_t height, int32_t stride, const char *encoding, int32_t is_bigendian,
                             uint8_t *out_bgr, int64_t out_bytes)
{
    if (!img || !out_bgr || !encoding || width < 1 || height < 1 || width > 32767 || height > 32767) return FID_E_INVALID_ARG;
    if (out_bytes < (int64_t)width * height * 3) return FID_E_CAPACITY;
    const std::string e(encoding);
    static const struct { const char *name; fid_encoding enc; } eight[] = {
        {"mono8", FID_ENC_MONO8}, {"bgr8", FID_ENC_BGR8}, {"rgb8", FID_ENC_RGB8}, {"bgra8", FID_ENC_BGRA8}, {"rgba8", FID_ENC_RGBA8}};
    for (const auto &k : eight)
        if (e == k.name) return fid_to_bgr(img, width, height, stride, k.enc, out_bgr, out_bytes);
    int ch = 0, swap = 0;
    if (e == "mono16") ch = 1;
    else if (e == "bgr16") ch = 3;
    else if (e == "rgb16") ch = 3, swap = 1;
    else if (e == "bgra16") ch = 4;
    else if (e == "rgba16") ch = 4, swap = 1;
    if (ch) {
        if ((int64_t)stride < (int64_t)width * ch * 2) return FID_E_INVALID_ARG;
        const bool be = is_bigendian != 0;
        for (int y = 0; y < height; y++) {
            const uint8_t *s = img + (size_t)y * (size_t)stride;
            uint8_t *o = out_bgr + (size_t)y * (size_t)width * 3;
            for (int x = 0; x < width; x++) {
                uint8_t v[4] = {0, 0, 0, 0};
                for (int k = 0; k < ch; k++) {
                    const uint8_t *p = s + ((size_t)x * ch + k) * 2;
                    v[k] = cvb_scale_16_to_8(be ? (uint16_t)((p[0] << 8) | p[1]) : (uint16_t)(p[0] | (p[1] << 8)));
                }
                if (ch == 1) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = v[0];
                else {
                    o[3 * x] = swap ? v[2] : v[0];
                    o[3 * x + 1] = v[1];
                    o[3 * x + 2] = swap ? v[0] : v[2];
                }
            }
        }
        return FID_OK;
    }
    if (e == "yuv422") {
        // cv_bridge: YUV422 -> BGR8 is cvtColor(COLOR_YUV2BGR_UYVY): U0 Y0 V0 Y1 per pixel pair, ITU-R BT.601 in 20-bit fixed point
        // (imgproc color_yuv: ITUR_BT_601_CY 1220542, CUB 2116026, CUG -409993, CVG -852492, CVR 1673527; y = max(0, Y - 16) * CY;
        //  channel = saturate((y + (1 << 19) + coefficient sums of (U - 128), (V - 128)) >> 20)).  Width must be even.
        if ((width & 1) || (int64_t)stride < (int64_t)width * 2) return FID_E_INVALID_ARG;
        const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SH = 20;
        auto sat = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (int y = 0; y < height; y++) {
            const uint8_t *sp = img + (size_t)y * (size_t)stride;
            uint8_t *o = out_bgr + (size_t)y * (size_t)width * 3;
            for (int x = 0; x < width; x += 2) {
                const int u = sp[2 * x] - 128, y0 = sp[2 * x + 1], v = sp[2 * x + 2] - 128, y1 = sp[2 * x + 3];
                const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
                const int a0 = (y0 - 16 > 0 ? y0 - 16 : 0) * CY, a1 = (y1 - 16 > 0 ? y1 - 16 : 0) * CY;
                o[3 * x] = sat((a0 + buv) >> SH);
                o[3 * x + 1] = sat((a0 + guv) >> SH);
                o[3 * x + 2] = sat((a0 + ruv) >> SH);
                o[3 * x + 3] = sat((a1 + buv) >> SH);
                o[3 * x + 4] = sat((a1 + guv) >> SH);
                o[3 * x + 5] = sat((a1 + ruv) >> SH);
            }
        }
        return FID_OK;
    }
    int blue = 0, green = 0;
    if (e == "bayer_rggb8") blue = -1, green = 0;       // COLOR_BayerBG2BGR
    else if (e == "bayer_bggr8") blue = 1, green = 0;   // COLOR_BayerRG2BGR
    else if (e == "bayer_gbrg8") blue = 1, green = 1;   // COLOR_BayerGR2BGR
    else if (e == "bayer_grbg8") blue = -1, green = 1;  // COLOR_BayerGB2BGR
    if (blue) {
        if (stride < width || width < 3) return FID_E_INVALID_ARG;
        cvb_bayer_to_bgr(img, width, height, stride, blue, green, out_bgr);
        return FID_OK;
    }
    return FID_E_UNSUPPORTED;
}

fid_status fid_encoding_from_string(const char *encoding, int32_t is_bigendian, fid_encoding *out_enc, int32_t *bytes_per_pixel)
{
    if (!encoding || !out_enc) return FID_E_INVALID_ARG;
    static const struct { const char *name; int enc, bpp; } tab[] = {
        {"mono8", FID_ENC_MONO8, 1}, {"bgr8", FID_ENC_BGR8, 3}, {"rgb8", FID_ENC_RGB8, 3}, {"bgra8", FID_ENC_BGRA8, 4}, {"rgba8", FID_ENC_RGBA8, 4},
        {"bayer_rggb8", FID_ENC_BAYER_RGGB8, 1}, {"bayer_bggr8", FID_ENC_BAYER_BGGR8, 1}, {"bayer_gbrg8", FID_ENC_BAYER_GBRG8, 1},
        {"bayer_grbg8", FID_ENC_BAYER_GRBG8, 1}, {"mono16", FID_ENC_MONO16, 2}, {"bgr16", FID_ENC_BGR16, 6}, {"rgb16", FID_ENC_RGB16, 6},
        {"bgra16", FID_ENC_BGRA16, 8}, {"rgba16", FID_ENC_RGBA16, 8}, {"yuv422", FID_ENC_YUV422, 2}};
    for (const auto &t : tab)
        if (!strcmp(encoding, t.name)) {
            const bool wide = t.enc >= FID_ENC_MONO16 && t.enc <= FID_ENC_RGBA16;
            *out_enc = (fid_encoding)(t.enc | (wide && is_bigendian ? FID_ENC_BIGENDIAN : 0));
            if (bytes_per_pixel) *bytes_per_pixel = t.bpp;
            return FID_OK;
        }
    return FID_E_UNSUPPORTED;
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
