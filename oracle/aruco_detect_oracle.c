/*
 * aruco_detect_oracle.c -- CPU restatement of aruco::detectMarkers (OpenCV 4.2.0 semantics) as the
 * reference calls it at aruco_detect/src/aruco_detect.cpp:350.  TEST INFRASTRUCTURE ONLY
 * (see aruco_oracle.h).  Compile with -ffp-contract=off: results are meant to be the plain
 * IEEE-754 double/float sequence of operations of the restated algorithm.
 *
 * Stage map:
 *   a2  ora_to_gray              cv_bridge::toCvCopy(BGR8) + cvtColor(BGR2GRAY)
 *   a3  ora_adaptive_threshold   aruco.cpp _threshold -> imgproc adaptiveThreshold/boxFilter
 *   a4  ora_find_contours        imgproc contours.cpp (Suzuki-Abe, RETR_LIST, CHAIN_APPROX_NONE)
 *       ora_approx_poly_dp       imgproc approx.cpp approxPolyDP_<int>, closed
 *       find_marker_contours     aruco.cpp _findMarkerContours
 *   a5  reorder/too-close        aruco.cpp _reorderCandidatesCorners, _filterTooCloseCandidates
 *   a6  extract_bits             aruco.cpp _extractBits (getPerspectiveTransform, warpPerspective
 *                                NEAREST, meanStdDev, Otsu threshold)
 *   a7  identify                 aruco.cpp _getBorderErrors, dictionary.cpp Dictionary::identify
 *   a8  filter_detected          aruco.cpp _filterDetectedMarkers (pointPolygonTest)
 *   a9  ora_corner_subpix        imgproc cornersubpix.cpp + samplers.cpp getRectSubPix_8u32f
 *   a9' ora_refine_candidate_lines  aruco.cpp _refineCandidateLines (CORNER_REFINE_CONTOUR; core solve / hal::LU32f / Matx solve)
 */
#define _GNU_SOURCE /* mmap flags of ora_arena.h under -std=c11 */
#include "aruco_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ora_arena.h" /* per-thread arena behind malloc / free in this file (test infrastructure; -DORA_NO_ARENA: off) */

/* ------------------------------------------------------------------------------------------- */
void ora_default_params(ora_params *p)
{
    /* node defaults, aruco_detect.cpp:690-727 (NOT OpenCV's defaults) */
    p->adaptiveThreshConstant = 7;
    p->adaptiveThreshWinSizeMin = 3;
    p->adaptiveThreshWinSizeMax = 53;
    p->adaptiveThreshWinSizeStep = 4;
    p->cornerRefinementMethod = 1;
    p->cornerRefinementWinSize = 5;
    p->cornerRefinementMaxIterations = 30;
    p->cornerRefinementMinAccuracy = 0.01;
    p->errorCorrectionRate = 0.6;
    p->minCornerDistanceRate = 0.05;
    p->markerBorderBits = 1;
    p->maxErroneousBitsInBorderRate = 0.04;
    p->minDistanceToBorder = 3;
    p->minMarkerDistanceRate = 0.05;
    p->minMarkerPerimeterRate = 0.1;
    p->maxMarkerPerimeterRate = 4.0;
    p->minOtsuStdDev = 5.0;
    p->perspectiveRemoveIgnoredMarginPerCell = 0.13;
    p->perspectiveRemovePixelPerCell = 8;
    p->polygonalApproxAccuracyRate = 0.01;
}

/* cvRound(double): round half to even (lrint under the default rounding mode) */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}

/* ------------------------------------------------------------------------------------------- */
/* a2: 8-bit BGR->gray, OpenCV 4.x fixed point (color_rgb.simd.hpp RGB2Gray<uchar>):
 *     BY15 = 3735, GY15 = 19235, RY15 = 9798, shift 15, CV_DESCALE rounding.               */
int ora_to_gray(const uint8_t *img, int w, int h, int stride, int enc, uint8_t *gray)
{
    if (!img || !gray || w <= 0 || h <= 0) return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = img + (size_t)y * stride;
        uint8_t *d = gray + (size_t)y * w;
        if (enc == 0) {
            memcpy(d, s, (size_t)w);
        } else {
            for (int x = 0; x < w; x++) {
                int c0 = s[3 * x], c1 = s[3 * x + 1], c2 = s[3 * x + 2];
                int b = enc == 1 ? c0 : c2, r = enc == 1 ? c2 : c0, g = c1;
                d[x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a3: adaptiveThreshold(src, dst, 255, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV, win, C)
 *   mean = boxFilter(src, win x win, normalize, BORDER_REPLICATE) as u8
 *          u8 result = round-to-nearest of sum/win^2.  OpenCV has two code paths
 *          (ColumnSum<ushort,uchar> fixed-point for win^2<=256, ColumnSum<int,uchar> float/double
 *          scale otherwise); for odd win both equal exact rounding because sum/win^2 can never sit
 *          on a half (tests/test_oracle_units.py::test_box_mean_rounding checks the fixed-point
 *          formula exhaustively).
 *   idelta = type == THRESH_BINARY ? cvCeil(C) : cvFloor(C) (thresh.cpp adaptiveThreshold): aruco passes
 *   THRESH_BINARY_INV, so FLOOR;  dst = (src - mean <= -idelta) ? 255 : 0 (tab[i] = i - 255 <= -idelta)   */
int ora_adaptive_threshold(const uint8_t *gray, int w, int h, int win, double C, uint8_t *out)
{
    if (!gray || !out || w <= 0 || h <= 0 || win < 3) return -1;
    if (win % 2 == 0) win++; /* aruco.cpp _threshold */
    const int r = win / 2, area = win * win;
    const int idelta = cv_floor(C);
    /* integral image with replicated border, (h+1) x (w+1) of int64 would be wasteful: use
     * running column sums (exactly what boxFilter's RowSum/ColumnSum compute). */
    int32_t *rowsum = (int32_t *)malloc((size_t)w * h * sizeof(int32_t));
    if (!rowsum) return -2;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = gray + (size_t)y * w;
        int32_t *rs = rowsum + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x == 0) {
                for (int k = -r; k <= r; k++) {
                    int xx = k < 0 ? 0 : (k >= w ? w - 1 : k);
                    acc += s[xx];
                }
            } else {
                int xo = x - 1 - r, xi = x + r;
                xo = xo < 0 ? 0 : xo;
                xi = xi >= w ? w - 1 : xi;
                acc = rs[x - 1] - s[xo] + s[xi];
            }
            rs[x] = acc;
        }
    }
    int32_t *col = (int32_t *)calloc((size_t)w, sizeof(int32_t));
    if (!col) {
        free(rowsum);
        return -2;
    }
    for (int k = -r; k <= r; k++) {
        int yy = k < 0 ? 0 : (k >= h ? h - 1 : k);
        const int32_t *rs = rowsum + (size_t)yy * w;
        for (int x = 0; x < w; x++) col[x] += rs[x];
    }
    for (int y = 0; y < h; y++) {
        if (y > 0) {
            int yo = y - 1 - r, yi = y + r;
            yo = yo < 0 ? 0 : yo;
            yi = yi >= h ? h - 1 : yi;
            const int32_t *ro = rowsum + (size_t)yo * w, *ri = rowsum + (size_t)yi * w;
            for (int x = 0; x < w; x++) col[x] += ri[x] - ro[x];
        }
        const uint8_t *s = gray + (size_t)y * w;
        uint8_t *d = out + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int mean = (2 * col[x] + area) / (2 * area); /* round(sum/area), never a tie */
            d[x] = (s[x] - mean <= -idelta) ? 255 : 0;
        }
    }
    free(col);
    free(rowsum);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a4: findContours(RETR_LIST, CHAIN_APPROX_NONE) -- contours.cpp cvFindNextContour +
 * icvFetchContour.  The image is copied into a (w+2)x(h+2) frame of zeros (copyMakeBorder in
 * cv::findContours) and binarised to 0/1.  Pixel marks: nbd=2 ("visited"), nbd|-128 = -126
 * ("right bound passed").                                                                      */
typedef struct {
    int32_t *pts;
    int64_t n, cap;
    int overflow;
} ptbuf;

static void pb_push(ptbuf *b, int x, int y)
{
    if (b->n + 1 > b->cap) {
        b->overflow = 1;
        return;
    }
    b->pts[2 * b->n] = x;
    b->pts[2 * b->n + 1] = y;
    b->n++;
}

static void fetch_contour(int8_t *ptr, int step, int px, int py, int is_hole, ptbuf *out)
{
    static const int dx8[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int dy8[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int deltas[16];
    for (int i = 0; i < 8; i++) deltas[i] = deltas[i + 8] = dy8[i] * step + dx8[i];
    const int8_t nbd = 2;
    int8_t *i0 = ptr, *i1, *i3, *i4 = 0;
    int s, s_end;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) { /* single pixel domain */
        *i0 = (int8_t)(nbd | -128);
        pb_push(out, px, py);
        return;
    }
    i3 = i0;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        /* check "right" bound */
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (int8_t)(nbd | -128);
        else if (*i3 == 1)
            *i3 = nbd;
        pb_push(out, px, py); /* CHAIN_APPROX_NONE: every visited pixel */
        px += dx8[s];
        py += dy8[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

int ora_find_contours(const uint8_t *mask, int w, int h, int32_t *pts, int64_t cap_pts,
                      int64_t *offsets, int32_t *is_hole_out, int cap_contours, int *n_contours)
{
    if (!mask || !pts || !offsets || !n_contours) return -1;
    const int step = w + 2;
    int8_t *img = (int8_t *)calloc((size_t)step * (h + 2), 1);
    if (!img) return -2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = mask[(size_t)y * w + x] ? 1 : 0;

    ptbuf pb = {pts, 0, cap_pts, 0};
    int n = 0, overflow = 0;
    offsets[0] = 0;
    for (int y = 1; y <= h && !overflow; y++) {
        int8_t *row = img + (size_t)y * step;
        int prev = 0;
        for (int x = 1; x <= w; x++) { /* cvFindNextContour: x in [1, img_size.width) */
            int p = row[x];
            if (p == prev) continue;
            int hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) {
                    prev = p;
                    continue;
                }
                hole = 1;
            }
            if (n >= cap_contours) {
                overflow = 1;
                break;
            }
            /* origin in source-image coordinates: padded (x - hole, y) minus the (1,1) offset */
            fetch_contour(row + x - hole, step, x - hole - 1, y - 1, hole, &pb);
            if (is_hole_out) is_hole_out[n] = hole;
            n++;
            offsets[n] = pb.n;
            prev = row[x]; /* p re-read after marking (scanner resumes with prev = img[x]) */
        }
        /* the scan stops at padded x = w (img_size.width = size.width - 1): the frame column is never read as p */
    }
    free(img);
    if (overflow || pb.overflow) return -1;
    /* cv::findContours returns the RETR_LIST sequence newest-first (cvInsertNodeIntoTree links every
     * new contour as the first child of the frame): reverse the discovery order in place. */
    if (n > 1) {
        int32_t *tmp = (int32_t *)malloc((size_t)pb.n * 2 * sizeof(int32_t));
        int64_t *off2 = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
        int32_t *h2 = (int32_t *)malloc((size_t)n * sizeof(int32_t));
        if (!tmp || !off2 || !h2) {
            free(tmp);
            free(off2);
            free(h2);
            return -2;
        }
        int64_t pos = 0;
        off2[0] = 0;
        for (int i = 0; i < n; i++) {
            int src = n - 1 - i;
            int64_t len = offsets[src + 1] - offsets[src];
            memcpy(tmp + 2 * pos, pts + 2 * offsets[src], (size_t)len * 2 * sizeof(int32_t));
            pos += len;
            off2[i + 1] = pos;
            h2[i] = is_hole_out ? is_hole_out[src] : 0;
        }
        memcpy(pts, tmp, (size_t)pb.n * 2 * sizeof(int32_t));
        memcpy(offsets, off2, (size_t)(n + 1) * sizeof(int64_t));
        if (is_hole_out) memcpy(is_hole_out, h2, (size_t)n * sizeof(int32_t));
        free(tmp);
        free(off2);
        free(h2);
    }
    *n_contours = n;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* approx.cpp approxPolyDP_<int>(closed = true) */
typedef struct {
    int start, end;
} range_t;

int ora_approx_poly_dp(const int32_t *src, int count0, double eps, int32_t *dst_out, int cap)
{
    if (count0 <= 0) return 0;
    const int count = count0;
    int new_count = 0;
    int32_t *dst = (int32_t *)malloc((size_t)count * 2 * sizeof(int32_t));
    size_t stacksz = (size_t)count + 16;
    range_t *stack = (range_t *)malloc(stacksz * sizeof(range_t));
    if (!dst || !stack) {
        free(dst);
        free(stack);
        return -2;
    }
    size_t top = 0;
    range_t slice = {0, 0}, right_slice = {0, 0};
    int sx = -1000000, sy = -1000000, ex = 0, ey = 0, px = 0, py = 0;
    int pos = 0, le_eps = 0;
#define READ_PT(X, Y, P)     \
    do {                     \
        X = src[2 * (P)];    \
        Y = src[2 * (P) + 1]; \
        if (++(P) >= count) (P) = 0; \
    } while (0)
#define PUSH_SLICE(S)                                                        \
    do {                                                                     \
        if (top >= stacksz) {                                                \
            stacksz = stacksz * 3 / 2;                                       \
            stack = (range_t *)realloc(stack, stacksz * sizeof(range_t));    \
        }                                                                    \
        stack[top++] = (S);                                                  \
    } while (0)
#define WRITE_PT(X, Y)              \
    do {                            \
        dst[2 * new_count] = (X);   \
        dst[2 * new_count + 1] = (Y); \
        new_count++;                \
    } while (0)

    eps *= eps;
    /* 1. find approximately two farthest points of the contour */
    right_slice.start = 0;
    for (int i = 0; i < 3; i++) {
        double dist, max_dist = 0;
        pos = (pos + right_slice.start) % count;
        READ_PT(sx, sy, pos);
        for (int j = 1; j < count; j++) {
            double dx, dy;
            READ_PT(px, py, pos);
            dx = px - sx;
            dy = py - sy;
            dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right_slice.start = j;
            }
        }
        le_eps = max_dist <= eps;
    }
    /* 2. initialise the stack */
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        PUSH_SLICE(right_slice);
        PUSH_SLICE(slice);
    } else
        WRITE_PT(sx, sy);

    /* 3. run recursive process */
    while (top > 0) {
        slice = stack[--top];
        ex = src[2 * slice.end];
        ey = src[2 * slice.end + 1];
        pos = slice.start;
        READ_PT(sx, sy, pos);
        if (pos != slice.end) {
            double dx, dy, dist, max_dist = 0;
            dx = ex - sx;
            dy = ey - sy;
            while (pos != slice.end) {
                READ_PT(px, py, pos);
                dist = fabs((py - sy) * dx - (px - sx) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right_slice.start = (pos + count - 1) % count;
                }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = 1;
            sx = src[2 * slice.start];
            sy = src[2 * slice.start + 1];
        }
        if (le_eps) {
            WRITE_PT(sx, sy);
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            PUSH_SLICE(right_slice);
            PUSH_SLICE(slice);
        }
    }

    /* last stage: remove extra points on the [almost] straight lines */
    {
        const int cnt = new_count;
        int wpos, i;
        pos = cnt - 1;
#define READ_DST_PT(X, Y, P)   \
    do {                       \
        X = dst[2 * (P)];      \
        Y = dst[2 * (P) + 1];  \
        if (++(P) >= cnt) (P) = 0; \
    } while (0)
        READ_DST_PT(sx, sy, pos);
        wpos = pos;
        READ_DST_PT(px, py, pos);
        for (i = 0; i < cnt && new_count > 2; i++) {
            double dx, dy, dist, sip;
            READ_DST_PT(ex, ey, pos);
            dx = ex - sx;
            dy = ey - sy;
            dist = fabs((px - sx) * dy - (py - sy) * dx);
            sip = (double)(px - sx) * (ex - px) + (double)(py - sy) * (ey - py);
            if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
                new_count--;
                dst[2 * wpos] = sx = ex;
                dst[2 * wpos + 1] = sy = ey;
                if (++wpos >= cnt) wpos = 0;
                READ_DST_PT(px, py, pos);
                i++;
                continue;
            }
            dst[2 * wpos] = sx = px;
            dst[2 * wpos + 1] = sy = py;
            if (++wpos >= cnt) wpos = 0;
            px = ex;
            py = ey;
        }
    }
    int ret = new_count;
    if (dst_out) {
        int m = new_count < cap ? new_count : cap;
        memcpy(dst_out, dst, (size_t)m * 2 * sizeof(int32_t));
    }
    free(dst);
    free(stack);
    return ret;
#undef READ_PT
#undef PUSH_SLICE
#undef WRITE_PT
#undef READ_DST_PT
}

/* convhull.cpp isContourConvex_<int> */
static int is_contour_convex(const int32_t *p, int n)
{
    int prevx = p[2 * ((n - 2 + n) % n)], prevy = p[2 * ((n - 2 + n) % n) + 1];
    int curx = p[2 * (n - 1)], cury = p[2 * (n - 1) + 1];
    int dx0 = curx - prevx, dy0 = cury - prevy;
    int orientation = 0;
    for (int i = 0; i < n; i++) {
        int dxdy0, dydx0, dx, dy;
        prevx = curx;
        prevy = cury;
        curx = p[2 * i];
        cury = p[2 * i + 1];
        dx = curx - prevx;
        dy = cury - prevy;
        dxdy0 = dx * dy0;
        dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return 0;
        dx0 = dx;
        dy0 = dy;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
typedef struct {
    ora_candidate *v;
    int32_t **pts; /* per candidate: its contour (x, y pairs, contour_size of them) or NULL -- kept only for
                      CORNER_REFINE_CONTOUR, the `contours` vector aruco.cpp carries beside `candidates` */
    int n, cap;
} candvec;

static int cv_push(candvec *c, const ora_candidate *x, int32_t *pts)
{
    if (c->n >= c->cap) {
        int nc = c->cap ? c->cap * 2 : 256;
        ora_candidate *nv = (ora_candidate *)realloc(c->v, (size_t)nc * sizeof(ora_candidate));
        if (!nv) return -1;
        c->v = nv;
        int32_t **np = (int32_t **)realloc(c->pts, (size_t)nc * sizeof(int32_t *));
        if (!np) return -1;
        c->pts = np;
        c->cap = nc;
    }
    c->pts[c->n] = pts;
    c->v[c->n++] = *x;
    return 0;
}

static void cv_free(candvec *c, int free_points)
{
    if (free_points)
        for (int i = 0; i < c->n; i++) free(c->pts[i]);
    free(c->pts);
    free(c->v);
    c->v = NULL;
    c->pts = NULL;
    c->n = c->cap = 0;
}

/* aruco.cpp _findMarkerContours on one thresholded image */
static int find_marker_contours(const uint8_t *mask, int w, int h, int scale, const ora_params *p,
                                candvec *out)
{
    const int maxdim = w > h ? w : h;
    unsigned int minPerimeterPixels = (unsigned int)(p->minMarkerPerimeterRate * maxdim);
    unsigned int maxPerimeterPixels = (unsigned int)(p->maxMarkerPerimeterRate * maxdim);

    int64_t cap_pts = (int64_t)w * h * 2 + 16; /* a pixel is emitted at most 4 times; grow on demand */
    int cap_c = w * h / 2 + 16;
    int32_t *pts = NULL;
    int64_t *off = NULL;
    int32_t *holes = NULL;
    int nc = 0, rc;
    for (;;) {
        pts = (int32_t *)malloc((size_t)cap_pts * 2 * sizeof(int32_t));
        off = (int64_t *)malloc((size_t)(cap_c + 1) * sizeof(int64_t));
        holes = (int32_t *)malloc((size_t)cap_c * sizeof(int32_t));
        if (!pts || !off || !holes) {
            free(pts);
            free(off);
            free(holes);
            return -2;
        }
        rc = ora_find_contours(mask, w, h, pts, cap_pts, off, holes, cap_c, &nc);
        if (rc == 0) break;
        free(pts);
        free(off);
        free(holes);
        if (rc == -2) return -2;
        cap_pts *= 2;
        cap_c *= 2;
    }
    for (int i = 0; i < nc; i++) {
        size_t sz = (size_t)(off[i + 1] - off[i]);
        if (sz < minPerimeterPixels || sz > maxPerimeterPixels) continue;
        const int32_t *c = pts + 2 * off[i];
        int32_t ap[8];
        int na = ora_approx_poly_dp(c, (int)sz, (double)sz * p->polygonalApproxAccuracyRate, ap, 4);
        if (na != 4 || !is_contour_convex(ap, 4)) continue;
        double minDistSq = (double)maxdim * maxdim;
        for (int j = 0; j < 4; j++) {
            double d = (double)(ap[2 * j] - ap[2 * ((j + 1) % 4)]) * (double)(ap[2 * j] - ap[2 * ((j + 1) % 4)]) +
                       (double)(ap[2 * j + 1] - ap[2 * ((j + 1) % 4) + 1]) *
                           (double)(ap[2 * j + 1] - ap[2 * ((j + 1) % 4) + 1]);
            minDistSq = minDistSq < d ? minDistSq : d;
        }
        double minCornerDistancePixels = (double)sz * p->minCornerDistanceRate;
        if (minDistSq < minCornerDistancePixels * minCornerDistancePixels) continue;
        int tooNear = 0;
        for (int j = 0; j < 4; j++) {
            if (ap[2 * j] < p->minDistanceToBorder || ap[2 * j + 1] < p->minDistanceToBorder ||
                ap[2 * j] > w - 1 - p->minDistanceToBorder || ap[2 * j + 1] > h - 1 - p->minDistanceToBorder)
                tooNear = 1;
        }
        if (tooNear) continue;
        ora_candidate cd;
        cd.scale = scale;
        cd.contour_size = (int32_t)sz;
        cd.start_x = c[0];
        cd.start_y = c[1];
        cd.is_hole = holes[i];
        for (int j = 0; j < 8; j++) cd.corners[j] = (float)ap[j];
        int32_t *keep = NULL;
        if (p->cornerRefinementMethod == 2) { /* contoursOut.push_back(contours[i]) */
            keep = (int32_t *)malloc(sz * 2 * sizeof(int32_t));
            if (keep) memcpy(keep, c, sz * 2 * sizeof(int32_t));
        }
        if ((p->cornerRefinementMethod == 2 && !keep) || cv_push(out, &cd, keep)) {
            free(keep);
            free(pts);
            free(off);
            free(holes);
            return -2;
        }
    }
    free(pts);
    free(off);
    free(holes);
    return 0;
}

/* aruco.cpp _reorderCandidatesCorners */
static void reorder_corners(ora_candidate *c)
{
    double dx1 = c->corners[2] - c->corners[0];
    double dy1 = c->corners[3] - c->corners[1];
    double dx2 = c->corners[4] - c->corners[0];
    double dy2 = c->corners[5] - c->corners[1];
    double crossProduct = (dx1 * dy2) - (dy1 * dx2);
    if (crossProduct < 0.0) {
        float tx = c->corners[2], ty = c->corners[3];
        c->corners[2] = c->corners[6];
        c->corners[3] = c->corners[7];
        c->corners[6] = tx;
        c->corners[7] = ty;
    }
}

/* aruco.cpp _filterTooCloseCandidates (4.2.0; detectInvertedMarker = false) */
static int filter_too_close(const candvec *in, candvec *out, double minMarkerDistanceRate)
{
    const int n = in->n;
    int npairs = 0, cappairs = 1024;
    int *pairs = (int *)malloc((size_t)cappairs * 2 * sizeof(int));
    uint8_t *toRemove = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!pairs || !toRemove) {
        free(pairs);
        free(toRemove);
        return -2;
    }
    for (int i = 0; i < n; i++) {
        for (int j = i + 1; j < n; j++) {
            int minimumPerimeter = in->v[i].contour_size < in->v[j].contour_size ? in->v[i].contour_size
                                                                                  : in->v[j].contour_size;
            for (int fc = 0; fc < 4; fc++) {
                double distSq = 0;
                for (int c = 0; c < 4; c++) {
                    int modC = (c + fc) % 4;
                    /* Point2f arithmetic: the subtractions and products are float, summed into a double */
                    float ax = in->v[i].corners[2 * modC] - in->v[j].corners[2 * c];
                    float ay = in->v[i].corners[2 * modC + 1] - in->v[j].corners[2 * c + 1];
                    distSq += ax * ax + ay * ay;
                }
                distSq /= 4.;
                double minMarkerDistancePixels = (double)minimumPerimeter * minMarkerDistanceRate;
                if (distSq < minMarkerDistancePixels * minMarkerDistancePixels) {
                    if (npairs >= cappairs) {
                        cappairs *= 2;
                        pairs = (int *)realloc(pairs, (size_t)cappairs * 2 * sizeof(int));
                    }
                    pairs[2 * npairs] = i;
                    pairs[2 * npairs + 1] = j;
                    npairs++;
                    break;
                }
            }
        }
    }
    for (int k = 0; k < npairs; k++) {
        int a = pairs[2 * k], b = pairs[2 * k + 1];
        if (toRemove[a] || toRemove[b]) continue;
        size_t perimeter1 = (size_t)in->v[a].contour_size;
        size_t perimeter2 = (size_t)in->v[b].contour_size;
        if (perimeter1 > perimeter2)
            toRemove[b] = 1;
        else
            toRemove[a] = 1;
    }
    for (int i = 0; i < n; i++)
        if (!toRemove[i] && cv_push(out, &in->v[i], in->pts[i])) { /* (the points stay owned by `in`) */
            free(pairs);
            free(toRemove);
            return -2;
        }
    free(pairs);
    free(toRemove);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* hal LU64f (matrix_decomp / lapack.cpp LUImpl<double>) with one right-hand side */
static int lu_solve8(double A[8][8], double b[8])
{
    const int m = 8;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j][i]) > fabs(A[k][i])) k = j;
        if (fabs(A[k][i]) < DBL_EPSILON * 100) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) {
                double t = A[i][j];
                A[i][j] = A[k][j];
                A[k][j] = t;
            }
            double t = b[i];
            b[i] = b[k];
            b[k] = t;
        }
        double d = -1 / A[i][i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j][i] * d;
            for (k = i + 1; k < m; k++) A[j][k] += alpha * A[i][k];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= A[i][k] * b[k];
        b[i] = s / A[i][i];
    }
    return 1;
}

/* imgwarp.cpp getPerspectiveTransform(src, dst, DECOMP_LU) */
static int get_perspective_transform(const float src[8], const float dst[8], double M[9])
{
    double a[8][8], b[8];
    for (int i = 0; i < 4; ++i) {
        float sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        a[i][0] = a[i + 4][3] = sx;
        a[i][1] = a[i + 4][4] = sy;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = (float)(-sx * dx); /* Point2f products are float before widening */
        a[i][7] = (float)(-sy * dx);
        a[i + 4][6] = (float)(-sx * dy);
        a[i + 4][7] = (float)(-sy * dy);
        b[i] = dx;
        b[i + 4] = dy;
    }
    if (!lu_solve8(a, b)) return 0;
    for (int i = 0; i < 8; i++) M[i] = b[i];
    M[8] = 1.;
    return 1;
}

/* matrix.cpp cv::invert, 3x3 double fast path */
static int invert3(const double S[9], double t[9])
{
#define Sd(r, c) S[(r)*3 + (c)]
    double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) -
               Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
               Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    if (d == 0.) return 0;
    d = 1. / d;
    t[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
    t[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
    t[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
    t[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
    t[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
    t[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
    t[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
    t[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
    t[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
#undef Sd
    return 1;
}

static inline int sat_int(double v)
{
    if (v != v) return INT_MIN; /* cvRound(NaN) on x86 = INT_MIN */
    if (v <= (double)INT_MIN) return INT_MIN;
    if (v >= (double)INT_MAX) return INT_MAX;
    return cv_round(v);
}

/* imgwarp.cpp warpPerspective(INTER_NEAREST, BORDER_CONSTANT 0), dst S x S, S <= 64 so a block row
 * starts at x = 0 (WarpPerspectiveInvoker with bw0 = S). */
static void warp_perspective_nearest(const uint8_t *gray, int w, int h, const double Mfwd[9], int S,
                                     uint8_t *dst)
{
    double M[9];
    if (!invert3(Mfwd, M)) {
        memset(dst, 0, (size_t)S * S);
        return;
    }
    for (int y = 0; y < S; y++) {
        double X0 = M[0] * 0 + M[1] * y + M[2];
        double Y0 = M[3] * 0 + M[4] * y + M[5];
        double W0 = M[6] * 0 + M[7] * y + M[8];
        for (int x1 = 0; x1 < S; x1++) {
            double W = W0 + M[6] * x1;
            W = W ? 1. / W : 0;
            double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + M[0] * x1) * W));
            double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + M[3] * x1) * W));
            int X = sat_int(fX), Y = sat_int(fY);
            /* saturate_cast<short> then remapNearest with constant border */
            X = X < SHRT_MIN ? SHRT_MIN : (X > SHRT_MAX ? SHRT_MAX : X);
            Y = Y < SHRT_MIN ? SHRT_MIN : (Y > SHRT_MAX ? SHRT_MAX : Y);
            uint8_t v = 0;
            if ((unsigned)X < (unsigned)w && (unsigned)Y < (unsigned)h) v = gray[(size_t)Y * w + X];
            dst[y * S + x1] = v;
        }
    }
}

/* thresh.cpp getThreshVal_Otsu_8u */
static double otsu_8u(const uint8_t *img, int total)
{
    const int N = 256;
    int hst[256] = {0};
    for (int i = 0; i < total; i++) hst[img[i]]++;
    double mu = 0, scale = 1. / (total);
    for (int i = 0; i < N; i++) mu += i * (double)hst[i];
    mu *= scale;
    double mu1 = 0, q1 = 0;
    double max_sigma = 0, max_val = 0;
    for (int i = 0; i < N; i++) {
        double p_i, q2, mu2, sigma;
        p_i = hst[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    return max_val;
}

/* aruco.cpp _extractBits */
static void extract_bits(const uint8_t *gray, int w, int h, const float corners[8], int markerSize,
                         int markerBorderBits, int cellSize, double cellMarginRate, double minStdDevOtsu,
                         uint8_t *bits)
{
    int msb = markerSize + 2 * markerBorderBits;
    int cellMarginPixels = (int)(cellMarginRate * cellSize);
    int S = msb * cellSize;
    float dstc[8] = {0, 0, (float)S - 1, 0, (float)S - 1, (float)S - 1, 0, (float)S - 1};
    double M[9];
    uint8_t *res = (uint8_t *)malloc((size_t)S * S);
    if (!get_perspective_transform(corners, dstc, M)) {
        /* singular system: cv::solve leaves X zeroed -> M = [0..0,1]; invert fails -> dst untouched
         * (zeros from Mat::create).  Degenerate quads never pass the convexity gate, keep it simple. */
        memset(res, 0, (size_t)S * S);
    } else {
        warp_perspective_nearest(gray, w, h, M, S, res);
    }
    /* meanStdDev on the inner region (cellSize/2 inset) */
    int in0 = cellSize / 2, in1 = S - cellSize / 2;
    int64_t s = 0, sq = 0;
    int nz = 0;
    for (int y = in0; y < in1; y++)
        for (int x = in0; x < in1; x++) {
            int v = res[y * S + x];
            s += v;
            sq += v * v;
            nz++;
        }
    double scale = nz ? 1. / nz : 0.;
    double mean = s * scale;
    double stddev = sqrt(fmax(sq * scale - mean * mean, 0.));
    if (stddev < minStdDevOtsu) {
        memset(bits, mean > 127 ? 1 : 0, (size_t)msb * msb);
        free(res);
        return;
    }
    double thr = otsu_8u(res, S * S);
    int ithr = cv_floor(thr);
    memset(bits, 0, (size_t)msb * msb);
    int cs = cellSize - 2 * cellMarginPixels;
    for (int y = 0; y < msb; y++)
        for (int x = 0; x < msb; x++) {
            int Xs = x * cellSize + cellMarginPixels, Ys = y * cellSize + cellMarginPixels;
            size_t nZ = 0;
            for (int yy = 0; yy < cs; yy++)
                for (int xx = 0; xx < cs; xx++) nZ += res[(Ys + yy) * S + Xs + xx] > ithr;
            if (nZ > (size_t)(cs * cs) / 2) bits[y * msb + x] = 1;
        }
    free(res);
}

/* aruco.cpp _getBorderErrors */
static int border_errors(const uint8_t *bits, int markerSize, int borderSize)
{
    int sizeWithBorders = markerSize + 2 * borderSize;
    int totalErrors = 0;
    for (int y = 0; y < sizeWithBorders; y++) {
        for (int k = 0; k < borderSize; k++) {
            if (bits[y * sizeWithBorders + k] != 0) totalErrors++;
            if (bits[y * sizeWithBorders + sizeWithBorders - 1 - k] != 0) totalErrors++;
        }
    }
    for (int x = borderSize; x < sizeWithBorders - borderSize; x++) {
        for (int k = 0; k < borderSize; k++) {
            if (bits[k * sizeWithBorders + x] != 0) totalErrors++;
            if (bits[(sizeWithBorders - 1 - k) * sizeWithBorders + x] != 0) totalErrors++;
        }
    }
    return totalErrors;
}

static int popcount8(unsigned v)
{
    int c = 0;
    while (v) {
        c += v & 1;
        v >>= 1;
    }
    return c;
}

/* dictionary.cpp Dictionary::getByteListFromBits (rotation 0 only is needed for the candidate) and
 * Dictionary::identify */
static int dict_identify(const ora_dict *d, const uint8_t *onlyBits, double maxCorrectionRate, int *idx,
                         int *rotation)
{
    int ms = d->marker_size;
    int nbytes = (ms * ms + 8 - 1) / 8;
    uint8_t cand[8] = {0};
    int currentBit = 0, currentByte = 0;
    for (int row = 0; row < ms; row++)
        for (int col = 0; col < ms; col++) {
            cand[currentByte] = (uint8_t)(cand[currentByte] << 1);
            cand[currentByte] |= onlyBits[row * ms + col];
            currentBit++;
            if (currentBit == 8) {
                currentBit = 0;
                currentByte++;
            }
        }
    int maxCorrectionRecalculed = (int)((double)d->max_correction_bits * maxCorrectionRate);
    *idx = -1;
    for (int m = 0; m < d->n_markers; m++) {
        int currentMinDistance = ms * ms + 1;
        int currentRotation = -1;
        for (unsigned r = 0; r < 4; r++) {
            const uint8_t *t = d->bytes + ((size_t)m * 4 + r) * nbytes;
            int ham = 0;
            for (int k = 0; k < nbytes; k++) ham += popcount8(t[k] ^ cand[k]);
            if (ham < currentMinDistance) {
                currentMinDistance = ham;
                currentRotation = (int)r;
            }
        }
        if (currentMinDistance <= maxCorrectionRecalculed) {
            *idx = m;
            *rotation = currentRotation;
            break;
        }
    }
    return *idx != -1;
}

/* aruco.cpp _identifyOneCandidate; corners are rotated in place on success */
int ora_identify(const uint8_t *gray, int w, int h, const ora_params *p, const ora_dict *d,
                 const float corners_in[8], uint8_t *bits_out, int *rotation_out)
{
    int ms = d->marker_size, bb = p->markerBorderBits, msb = ms + 2 * bb;
    uint8_t bits[16 * 16];
    if (msb > 16) return -1;
    extract_bits(gray, w, h, corners_in, ms, bb, p->perspectiveRemovePixelPerCell,
                 p->perspectiveRemoveIgnoredMarginPerCell, p->minOtsuStdDev, bits);
    if (bits_out) memcpy(bits_out, bits, (size_t)msb * msb);
    if (rotation_out) *rotation_out = -1;
    int maximumErrorsInBorder = (int)(ms * ms * p->maxErroneousBitsInBorderRate);
    int borderErrors = border_errors(bits, ms, bb);
    if (borderErrors > maximumErrorsInBorder) return -1;
    uint8_t only[8 * 8];
    for (int y = 0; y < ms; y++)
        for (int x = 0; x < ms; x++) only[y * ms + x] = bits[(y + bb) * msb + x + bb];
    int idx, rot;
    if (!dict_identify(d, only, p->errorCorrectionRate, &idx, &rot)) return -1;
    if (rotation_out) *rotation_out = rot;
    return idx;
}

/* geometry.cpp pointPolygonTest(contour of Point2f, pt, measureDist=false) */
static double point_polygon_test4(const float *cnt, float ptx, float pty)
{
    int counter = 0;
    float vx = cnt[6], vy = cnt[7], v0x, v0y;
    for (int i = 0; i < 4; i++) {
        double dist;
        v0x = vx;
        v0y = vy;
        vx = cnt[2 * i];
        vy = cnt[2 * i + 1];
        if ((v0y <= pty && vy <= pty) || (v0y > pty && vy > pty) || (v0x < ptx && vx < ptx)) {
            if (pty == vy && (ptx == vx || (pty == v0y && ((v0x <= ptx && ptx <= vx) || (vx <= ptx && ptx <= v0x)))))
                return 0;
            continue;
        }
        dist = (double)(pty - v0y) * (vx - v0x) - (double)(ptx - v0x) * (vy - v0y);
        if (dist == 0) return 0;
        if (vy < v0y) dist = -dist;
        counter += dist > 0;
    }
    return counter % 2 == 0 ? -1 : 1;
}

/* ------------------------------------------------------------------------------------------- */
/* a9: cornersubpix.cpp cornerSubPix + samplers.cpp getRectSubPix (8u -> 32f) */
static void get_rect_subpix_8u32f(const uint8_t *src, int src_step, int sw, int sh, float *dst, int dst_step,
                                  int win_w, int win_h, float cx0, float cy0)
{
    float cx = cx0, cy = cy0;
    cx -= (win_w - 1) * 0.5f;
    cy -= (win_h - 1) * 0.5f;
    int ipx = cv_floor(cx), ipy = cv_floor(cy);
#ifdef ORA_SUBPIX_GENERIC
    if (0) {
#else
    if (0 <= ipx && ipx + win_w < sw && 0 <= ipy && ipy + win_h < sh && win_w > 0 && win_h > 0) {
#endif
        /* getRectSubPix_8u32f fast path */
        float a = cx - ipx;
        float b = cy - ipy;
        a = a > 0.0001f ? a : 0.0001f;
        float a12 = a * (1.f - b);
        float a22 = a * b;
        float b1 = 1.f - b;
        float b2 = b;
        double s = (1. - a) / a;
        src += ipy * src_step + ipx;
        for (int i = 0; i < win_h; i++, src += src_step, dst += dst_step) {
            float prev = (1 - a) * (b1 * src[0] + b2 * src[src_step]);
            for (int j = 0; j < win_w; j++) {
                float t = a12 * src[j + 1] + a22 * src[j + 1 + src_step];
                dst[j] = prev + t;
                prev = (float)(t * s);
            }
        }
        return;
    }
    /* getRectSubPix_Cn_<uchar,float,float,nop,nop> (window not fully inside: replicate border) */
    {
        float a = cx - ipx, b = cy - ipy;
        float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        float b1 = 1.f - b, b2 = b;
        if (0 <= ipx && ipx < sw - win_w && 0 <= ipy && ipy < sh - win_h) {
            const uint8_t *sp = src + ipy * src_step + ipx;
            for (int i = 0; i < win_h; i++, sp += src_step, dst += dst_step)
                for (int j = 0; j < win_w; j++)
                    dst[j] = sp[j] * a11 + sp[j + 1] * a12 + sp[j + src_step] * a21 + sp[j + src_step + 1] * a22;
            return;
        }
        /* adjustRect */
        int rx, ry, rw, rh;
        const uint8_t *sp;
        {
            int x = ipx, y = ipy;
            const uint8_t *base = src;
            if (y >= 0)
                base += y * src_step, ry = 0;
            else
                ry = -y < win_h ? -y : win_h;
            if (y + win_h < sh)
                rh = win_h;
            else {
                rh = sh - y - 1;
                if (rh < 0) {
                    base += rh * src_step;
                    rh = 0;
                }
            }
            if (x >= 0)
                base += x, rx = 0;
            else {
                rx = -x < win_w ? -x : win_w;
            }
            if (x + win_w < sw)
                rw = win_w;
            else {
                rw = sw - x - 1;
                if (rw < 0) {
                    base += rw;
                    rw = 0;
                }
            }
            sp = base - rx;
        }
        for (int i = 0; i < win_h; i++, dst += dst_step) {
            const uint8_t *src2 = sp + src_step;
            if (i < ry || i >= rh) src2 -= src_step;
            int j;
            float s0;
            s0 = sp[rx] * b1 + src2[rx] * b2;
            for (j = 0; j < rx; j++) dst[j] = s0;
            for (; j < rw; j++)
                dst[j] = sp[j] * a11 + sp[j + 1] * a12 + src2[j] * a21 + src2[j + 1] * a22;
            s0 = sp[rw] * b1 + src2[rw] * b2;
            for (; j < win_w; j++) dst[j] = s0;
            if (i < rh) sp = src2;
        }
    }
}

int ora_corner_subpix(const uint8_t *gray, int w, int h, float *pts, int n, int win, int max_iter_in,
                      double eps_in)
{
    const int MAX_ITERS = 100;
    if (win <= 0 || win > 15) return -1;
    int win_w = win * 2 + 1, win_h = win * 2 + 1;
    int max_iters = max_iter_in < 1 ? 1 : (max_iter_in > MAX_ITERS ? MAX_ITERS : max_iter_in);
    double eps = eps_in > 0 ? eps_in : 0.;
    eps *= eps;
    if (n == 0) return 0;
    float mask[31 * 31], subpix_buf[33 * 33];
    for (int i = 0; i < win_h; i++) {
        float y = (float)(i - win) / win;
        float vy = expf(-y * y);
        for (int j = 0; j < win_w; j++) {
            float x = (float)(j - win) / win;
            mask[i * win_w + j] = (float)(vy * expf(-x * x));
        }
    }
    for (int pt_i = 0; pt_i < n; pt_i++) {
        float cTx = pts[2 * pt_i], cTy = pts[2 * pt_i + 1], cIx = cTx, cIy = cTy;
        int iter = 0;
        double err = 0;
        do {
            float cI2x, cI2y;
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_subpix_8u32f(gray, w, w, h, subpix_buf, win_w + 2, win_w + 2, win_h + 2, cIx, cIy);
            const float *subpix = subpix_buf + (win_w + 2) + 1;
            for (int i = 0, k = 0; i < win_h; i++, subpix += win_w + 2) {
                double py = i - win;
                for (int j = 0; j < win_w; j++, k++) {
                    double m = mask[k];
                    double tgx = subpix[j + 1] - subpix[j - 1];
                    double tgy = subpix[j + win_w + 2] - subpix[j - win_w - 2];
                    double gxx = tgx * tgx * m;
                    double gxy = tgx * tgy * m;
                    double gyy = tgy * tgy * m;
                    double px = j - win;
                    a += gxx;
                    b += gxy;
                    c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            cI2x = (float)(cIx + c * scale * bb1 - b * scale * bb2);
            cI2y = (float)(cIy - b * scale * bb1 + a * scale * bb2);
            err = (cI2x - cIx) * (cI2x - cIx) + (cI2y - cIy) * (cI2y - cIy);
            cIx = cI2x;
            cIy = cI2y;
            if (cIx < 0 || cIx >= w || cIy < 0 || cIy >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabs(cIx - cTx) > win || fabs(cIy - cTy) > win) {
            cIx = cTx;
            cIy = cTy;
        }
        pts[2 * pt_i] = cIx;
        pts[2 * pt_i + 1] = cIy;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a9', CORNER_REFINE_CONTOUR (aruco.cpp 4.2.0 _refineCandidateLines, _interpolate2Dline, _getCrossPoint): what the node selects
 * with doCornerRefinement = true, cornerRefinementSubPix = false (aruco_detect.cpp:274-283 dynamic_reconfigure, :700-711
 * rosparam; aruco_detect/cfg/DetectorParams.cfg:40-41).  The node calls detectMarkers without a camera matrix (:350), so the
 * undistort / re-distort branches of _refineCandidateLines are never taken and are not restated.
 * PARITY UNPINNED vs OpenCV: no reference fixture uses this method; the arithmetic below follows the published sources --
 *   cv::solve(A, B, C, DECOMP_NORMAL) on CV_32F: A^T A by mulTransposed (MulTransposedR<float, float>: double accumulators,
 *   one rounding to float per entry), A^T B by gemm(GEMM_1_T) (GEMMSingleMul<float, double>: likewise), then hal::LU32f
 *   (LUImpl<float>, eps = FLT_EPSILON * 10) on the 2 x 2 system; m == n (a side of exactly two points) skips the normal
 *   equations; a singular system leaves C = 0; fewer equations than unknowns (a side of ONE point) is CV_Error(StsBadArg) --
 *   the node's catch(cv::Exception&) logs it and publishes nothing for the frame (aruco_detect.cpp:391-393);
 *   Matx22f::solve = Matx_FastSolveOp<float, 2, 2, 1> (float determinant, d == 0 -> zeros).
 * The sums are sums of products of pixel coordinates: exact in double whatever the order, so the one thing a BLAS-backed
 * gemm (cv_hal_gemm32f with LAPACK, taken for >= 100 rows) could change is the rounding of A^T B -- stated, not pinned. */
static int lu32f_2x2(float A[2][2], float b[2])
{
    const float eps = FLT_EPSILON * 10;
    /* i = 0 */
    int k = fabsf(A[1][0]) > fabsf(A[0][0]) ? 1 : 0;
    if (fabsf(A[k][0]) < eps) return 0;
    if (k != 0) {
        for (int j = 0; j < 2; j++) {
            float t = A[0][j];
            A[0][j] = A[1][j];
            A[1][j] = t;
        }
        float t = b[0];
        b[0] = b[1];
        b[1] = t;
    }
    float d = -1 / A[0][0];
    float alpha = A[1][0] * d;
    A[1][1] += alpha * A[0][1];
    b[1] += alpha * b[0];
    /* i = 1 */
    if (fabsf(A[1][1]) < eps) return 0;
    /* back substitution */
    b[1] = b[1] / A[1][1];
    float s = b[0];
    s -= A[0][1] * b[1];
    b[0] = s / A[0][0];
    return 1;
}

/* _interpolate2Dline on the points of one side.  sums: n, sum x, sum y, sum xx, sum yy, sum xy, and the bounding box */
typedef struct {
    long long n, sx, sy, sxx, syy, sxy;
    int minx, maxx, miny, maxy;
    int x0, y0, x1, y1; /* the first two points (the m == n road needs them in order) */
} side_sums;

static void side_add(side_sums *s, int x, int y)
{
    if (s->n == 0) {
        s->minx = s->maxx = x;
        s->miny = s->maxy = y;
        s->x0 = x;
        s->y0 = y;
    } else {
        if (s->n == 1) {
            s->x1 = x;
            s->y1 = y;
        }
        if (x < s->minx) s->minx = x;
        if (x > s->maxx) s->maxx = x;
        if (y < s->miny) s->miny = y;
        if (y > s->maxy) s->maxy = y;
    }
    s->n++;
    s->sx += x;
    s->sy += y;
    s->sxx += (long long)x * x;
    s->syy += (long long)y * y;
    s->sxy += (long long)x * y;
}

static int interpolate_2d_line(const side_sums *s, float line[3])
{
    if (s->n < 2) return -1; /* n == 1: cv::solve throws (m < n); n == 0: nContours[0] on an empty vector */
    const int x_major = (float)s->maxx - (float)s->minx > (float)s->maxy - (float)s->miny;
    float A[2][2], b[2];
    if (s->n == 2) { /* m == n: is_normal = false, LU on the system itself */
        A[0][0] = (float)(x_major ? s->x0 : s->y0);
        A[0][1] = 1.f;
        A[1][0] = (float)(x_major ? s->x1 : s->y1);
        A[1][1] = 1.f;
        b[0] = (float)(x_major ? s->y0 : s->x0);
        b[1] = (float)(x_major ? s->y1 : s->x1);
    } else {
        const long long st = x_major ? s->sx : s->sy, stt = x_major ? s->sxx : s->syy, sv = x_major ? s->sy : s->sx;
        A[0][0] = (float)(double)stt;
        A[0][1] = A[1][0] = (float)(double)st;
        A[1][1] = (float)(double)s->n;
        b[0] = (float)(double)s->sxy;
        b[1] = (float)(double)sv;
    }
    if (!lu32f_2x2(A, b)) b[0] = b[1] = 0.f; /* if( !result ) dst = Scalar(0) */
    if (x_major) {
        line[0] = b[0];
        line[1] = -1.f;
        line[2] = b[1];
    } else {
        line[0] = -1.f;
        line[1] = b[0];
        line[2] = b[1];
    }
    return 0;
}

static void get_cross_point(const float l1[3], const float l2[3], float out[2])
{
    const float a00 = l1[0], a01 = l1[1], a10 = l2[0], a11 = l2[1];
    const float b0 = -l1[2], b1 = -l2[2];
    float d = a00 * a11 - a01 * a10;
    if (d == 0) {
        out[0] = out[1] = 0.f;
        return;
    }
    d = 1 / d;
    out[0] = (b0 * a11 - b1 * a01) * d;
    out[1] = (b1 * a00 - b0 * a10) * d;
}

int ora_refine_candidate_lines(const int32_t *pts, int n, float corners[8])
{
    /* cntPts[5]: one group per corner + the points in front of the first corner found */
    side_sums g[5];
    memset(g, 0, sizeof(g));
    int cornerIndex[4] = {-1}; /* as the reference writes it: {-1, 0, 0, 0} */
    int group = 4;
    for (int i = 0; i < n; i++) {
        const float px = (float)pts[2 * i], py = (float)pts[2 * i + 1];
        for (int j = 0; j < 4; j++)
            if (corners[2 * j] == px && corners[2 * j + 1] == py) {
                cornerIndex[j] = i;
                group = j;
            }
        side_add(&g[group], pts[2 * i], pts[2 * i + 1]);
    }
    if (group == 4) return -4; /* no corner on the contour: the reference appends to cntPts[4] while iterating it */
    if (g[4].n) { /* "saves extra group into corresponding": order-free for everything but the m == n road */
        side_sums *d = &g[group];
        const side_sums *e = &g[4];
        if (d->n == 0) {
            *d = *e;
        } else {
            if (d->n == 1) {
                d->x1 = e->x0;
                d->y1 = e->y0;
            }
            if (e->minx < d->minx) d->minx = e->minx;
            if (e->maxx > d->maxx) d->maxx = e->maxx;
            if (e->miny < d->miny) d->miny = e->miny;
            if (e->maxy > d->maxy) d->maxy = e->maxy;
            d->n += e->n;
            d->sx += e->sx;
            d->sy += e->sy;
            d->sxx += e->sxx;
            d->syy += e->syy;
            d->sxy += e->sxy;
        }
    }
    int inc = 1;
    inc = ((cornerIndex[0] > cornerIndex[1]) && (cornerIndex[3] > cornerIndex[0])) ? -1 : inc;
    inc = ((cornerIndex[2] > cornerIndex[3]) && (cornerIndex[1] > cornerIndex[2])) ? -1 : inc;
    float lines[4][3];
    for (int i = 0; i < 4; i++)
        if (interpolate_2d_line(&g[i], lines[i])) return -4;
    for (int i = 0; i < 4; i++) {
        if (inc < 0)
            get_cross_point(lines[i], lines[(i + 1) % 4], corners + 2 * i);
        else
            get_cross_point(lines[i], lines[(i + 3) % 4], corners + 2 * i);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
int ora_detect(const uint8_t *gray, int w, int h, const ora_params *p, const ora_dict *d, ora_marker *out,
               int cap, int *n_out, ora_trace *tr)
{
    if (!gray || !p || !d || !out || !n_out || w <= 0 || h <= 0) return -1;
    *n_out = 0;
    int rc = 0;
    /* _detectInitialCandidates */
    int nScales = (p->adaptiveThreshWinSizeMax - p->adaptiveThreshWinSizeMin) / p->adaptiveThreshWinSizeStep + 1;
    uint8_t *mask = (uint8_t *)malloc((size_t)w * h);
    candvec init = {0, 0, 0, 0}, filt = {0, 0, 0, 0};
    ora_marker *acc = NULL;
    int32_t **accpts = NULL;
    int *accn = NULL;
    uint8_t *toRemove = NULL;
    if (!mask) return -2;
    for (int i = 0; i < nScales && rc == 0; i++) {
        int currScale = p->adaptiveThreshWinSizeMin + i * p->adaptiveThreshWinSizeStep;
        rc = ora_adaptive_threshold(gray, w, h, currScale, p->adaptiveThreshConstant, mask);
        if (rc == 0) rc = find_marker_contours(mask, w, h, i, p, &init);
    }
    free(mask);
    if (rc) goto done;
    if (tr && tr->initial) {
        tr->n_initial = init.n;
        for (int i = 0; i < init.n && i < tr->cap_initial; i++) tr->initial[i] = init.v[i];
    }
    for (int i = 0; i < init.n; i++) reorder_corners(&init.v[i]);
    rc = filter_too_close(&init, &filt, p->minMarkerDistanceRate);
    if (rc) goto done;
    if (tr && tr->filtered) {
        tr->n_filtered = filt.n;
        for (int i = 0; i < filt.n && i < tr->cap_filtered; i++) tr->filtered[i] = filt.v[i];
    }
    /* _identifyCandidates */
    int ms = d->marker_size, msb = ms + 2 * p->markerBorderBits;
    int nacc = 0;
    const size_t nf1 = (size_t)(filt.n > 0 ? filt.n : 1);
    acc = (ora_marker *)malloc(nf1 * sizeof(ora_marker));
    accpts = (int32_t **)malloc(nf1 * sizeof(int32_t *));
    accn = (int *)malloc(nf1 * sizeof(int));
    if (!acc || !accpts || !accn) {
        rc = -2;
        goto done;
    }
    for (int i = 0; i < filt.n; i++) {
        uint8_t bits[256];
        int rot = -1;
        int id = ora_identify(gray, w, h, p, d, filt.v[i].corners, bits, &rot);
        if (tr && tr->bits && i < tr->cap_filtered) memcpy(tr->bits + (size_t)i * msb * msb, bits, (size_t)msb * msb);
        if (tr && tr->ident && i < tr->cap_filtered) {
            tr->ident[2 * i] = id;
            tr->ident[2 * i + 1] = rot;
        }
        if (id < 0) continue;
        ora_marker m;
        m.id = id;
        /* std::rotate(corners.begin(), corners.begin() + 4 - rotation, corners.end()) */
        for (int c = 0; c < 4; c++) {
            int srcc = (c + 4 - rot) % 4;
            m.corners[2 * c] = filt.v[i].corners[2 * srcc];
            m.corners[2 * c + 1] = filt.v[i].corners[2 * srcc + 1];
        }
        accpts[nacc] = filt.pts[i]; /* contours.push_back(_contours[i]) */
        accn[nacc] = filt.v[i].contour_size;
        acc[nacc++] = m;
    }
    /* _filterDetectedMarkers */
    toRemove = (uint8_t *)calloc((size_t)(nacc > 0 ? nacc : 1), 1);
    if (!toRemove) {
        rc = -2;
        goto done;
    }
    for (int i = 0; i + 1 < nacc; i++) {
        for (int j = i + 1; j < nacc; j++) {
            if (acc[i].id != acc[j].id) continue;
            int inside = 1;
            for (int q = 0; q < 4; q++)
                if (point_polygon_test4(acc[i].corners, acc[j].corners[2 * q], acc[j].corners[2 * q + 1]) < 0) {
                    inside = 0;
                    break;
                }
            if (inside) {
                toRemove[j] = 1;
                continue;
            }
            inside = 1;
            for (int q = 0; q < 4; q++)
                if (point_polygon_test4(acc[j].corners, acc[i].corners[2 * q], acc[i].corners[2 * q + 1]) < 0) {
                    inside = 0;
                    break;
                }
            if (inside) {
                toRemove[i] = 1;
                continue;
            }
        }
    }
    int n = 0;
    for (int i = 0; i < nacc; i++) {
        if (toRemove[i]) continue;
        if (n < cap) out[n] = acc[i];
        accpts[n] = accpts[i]; /* (n <= i) */
        accn[n] = accn[i];
        n++;
    }
    if (n > cap) {
        *n_out = cap;
        rc = -3;
        goto done;
    }
    if (tr && tr->presubpix) {
        tr->n_pre = n;
        for (int i = 0; i < n && i < tr->cap_pre; i++) tr->presubpix[i] = out[i];
    }
    /* corner refinement (CORNER_REFINE_SUBPIX) */
    if (p->cornerRefinementMethod == 1) {
        for (int i = 0; i < n; i++)
            ora_corner_subpix(gray, w, h, out[i].corners, 4, p->cornerRefinementWinSize,
                              p->cornerRefinementMaxIterations, p->cornerRefinementMinAccuracy);
    }
    /* corner refinement (CORNER_REFINE_CONTOUR): _refineCandidateLines(contours[i], candidates[i]) */
    if (p->cornerRefinementMethod == 2) {
        for (int i = 0; i < n && rc == 0; i++)
            if (ora_refine_candidate_lines(accpts[i], accn[i], out[i].corners)) rc = -4; /* cv::Exception */
        if (rc) {
            *n_out = 0; /* the node publishes nothing for this frame (aruco_detect.cpp:391-393) */
            goto done;
        }
    }
    *n_out = n;
done:
    free(toRemove);
    free(acc);
    free(accpts);
    free(accn);
    cv_free(&filt, 0);
    cv_free(&init, 1);
    return rc;
}

/* aruco_detect.cpp:164-200 dist + calcFiducialArea */
static double dist2f(float x1f, float y1f, float x2f, float y2f)
{
    double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
    double dx = x1 - x2, dy = y1 - y2;
    return sqrt(dx * dx + dy * dy);
}

double ora_fiducial_area(const float c[8])
{
    double a1 = dist2f(c[0], c[1], c[2], c[3]);
    double b1 = dist2f(c[0], c[1], c[6], c[7]);
    double c1 = dist2f(c[2], c[3], c[6], c[7]);
    double a2 = dist2f(c[2], c[3], c[4], c[5]);
    double b2 = dist2f(c[4], c[5], c[6], c[7]);
    double c2 = c1;
    double s1 = (a1 + b1 + c1) / 2.0;
    double s2 = (a2 + b2 + c2) / 2.0;
    a1 = sqrt(s1 * (s1 - a1) * (s1 - b1) * (s1 - c1));
    a2 = sqrt(s2 * (s2 - a2) * (s2 - b2) * (s2 - c2));
    return a1 + a2;
}
