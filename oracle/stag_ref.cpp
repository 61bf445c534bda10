// stag_ref.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle for the STag front-end).
//
// Builds the REFERENCE's own EDPF code where it lies under /root/reference (nothing is copied into this
// repository): the two reference translation units below are #included in place so that their file-static
// functions are reachable, and thin extern "C" entry points are put around them.  Output: oracle/_ref/
// libstag_ref.so (git-ignored; it travels to the GPU box like any built library).
//
//   ref_stag_gradient   = ComputeGradientMapByPrewitt      stag_detect/src/stag/ED/GradientOperators.cpp:77-136
//   ref_stag_anchors    = ComputeAnchorPoints              stag_detect/src/stag/ED/EDInternals.cpp:50-86
//                       + SortAnchorsByGradValue           stag_detect/src/stag/ED/EDInternals.cpp:146-186
//   ref_stag_route      = JoinAnchorPointsUsingSortedAnchors stag_detect/src/stag/ED/EDInternals.cpp:842-1448
//   ref_stag_validate   = ValidateEdgeSegments               stag_detect/src/stag/ED/ValidateEdgeSegments.cpp:365-413
//   ref_stag_smooth3    = SmoothImage(sigma = 1 / 2.5) of ED.cpp:176-177, restated like ref_stag_smooth5 ("parity unpinned"):
//                         cv::GaussianBlur(Size(0, 0), 0.4) -> ksize 3, 8.8 fixed-point kernel [10 236 10], one rounding
//   ref_stag_detect_quads = QuadDetector::detectQuads           stag_detect/src/stag/QuadDetector.cpp:12-66 (with Quad.cpp,
//                         EDInterface.cpp, utility.cpp compiled in place against oracle/cvshim: data types only)
//   ref_stag_detect_markers = Stag::detectMarkers               stag_detect/src/stag/Stag.cpp:24-51 (Stag.cpp, Decoder.cpp,
//                         Marker.cpp compiled in place; cv::threshold(OTSU) restated below; PoseRefiner stubbed until row s9)
//   ref_stag_smooth5    = what SmoothImage(..., sigma = 1.0) asks OpenCV for (ImageSmooth.cpp:43-55:
//                         cv::GaussianBlur(src, dst, Size(5, 5), 0, 0)) -- OpenCV is not installed here, so this one
//                         function is a RESTATEMENT ("parity unpinned"): for CV_8U and ksize 5 / sigma 0 OpenCV uses the
//                         fixed kernel [1 4 6 4 1] / 16 in 8.8 fixed point, BORDER_REFLECT_101, i.e.
//                         dst = (sum_ij k_i k_j src(y + i, x + j) + 128) >> 8.
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>

#include "src/stag/ED/GradientOperators.cpp"
#include "src/stag/ED/EDInternals.cpp"
#include "src/stag/ED/ValidateEdgeSegments.cpp"
// compiled as their own translation units by oracle/Makefile (target ref), straight from the reference tree:
//   src/stag/ED/ED.cpp  EDLines.cpp  LineSegment.cpp  NFA.cpp  MyMath.cpp
#include "stag/ED/ED.h"
#include "stag/ED/EDLines.h"
#include "stag/ED/ImageSmooth.h"
#include "stag/ED/NFA.h"
void SplitSegment2Lines(double *x, double *y, int noPixels, int segmentNo, EDLines *lines);
void JoinCollinearLines(EDLines *lines, double MAX_DISTANCE_BETWEEN_TWO_LINES, double MAX_ERROR);
void ValidateLineSegments(EdgeMap *map, unsigned char *srcImg, EDLines *lines, EDLines *invalidLines);
int ComputeMinLineLength(int width, int height);
double nfa(int n, int k, double p, double logNT);  // ED/NFA.cpp:155
// compiled from the reference tree against oracle/cvshim (data types only): Quad.cpp QuadDetector.cpp EDInterface.cpp utility.cpp
// likewise Stag.cpp Decoder.cpp Marker.cpp PoseRefiner.cpp Ellipse.cpp; Drawer (image output) is stubbed below.
// "private" is lifted for this wrapper only, so that a test can stop Stag::detectMarkers in front of the pose refinement
// (ref_stag_detect_markers with refine = 0 walks the same loop, Stag.cpp:36-47, with the reference's own member functions).
#include <bitset>
#include <string>
#include <vector>
#include <opencv2/opencv.hpp>
#define class struct  // the members in question are private by default, not by keyword
#include "stag/QuadDetector.h"
#include "stag/Stag.h"
#undef class

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

extern "C" {
int ref_stag_smooth5(const uint8_t *src, uint8_t *dst, int w, int h);
int ref_stag_smooth3(const uint8_t *src, uint8_t *dst, int w, int h);
}

// The one function of the ED library that calls OpenCV (ImageSmooth.cpp:43-55; not compiled): DetectEdgesByEDPF asks for
// sigma 1.0 (5x5) and then sigma 1 / 2.5 (3x3).  Restated, see the header of this file.
void SmoothImage(unsigned char *srcImg, unsigned char *smoothImg, int width, int height, double sigma)
{
    if (sigma == 1.0) ref_stag_smooth5(srcImg, smoothImg, width, height);
    else ref_stag_smooth3(srcImg, smoothImg, width, height);
}

// ---- stubs and restatements behind the reference's Stag.cpp
cv::Mat Drawer::drawMarkers(const string &, cv::Mat image, const vector<Marker> &) { return image; }  // drawing: out of scope
// PoseRefiner.cpp and Ellipse.cpp are compiled in place (oracle/Makefile).  The refiner's two OpenCV calls with numerical
// content are RESTATED here ("parity unpinned"):

// cv::Mat::inv() of a 3 x 3 double matrix (default DECOMP_LU): OpenCV's invert() takes the closed form for n <= 3 --
// determinant by the first row, cofactors times 1 / det
cv::Mat cv::Mat::inv() const
{
    const Mat &S = *this;
    Mat D(3, 3, CV_64FC1);
#define Sd(i, j) S.at<double>(i, j)
    double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) - Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
               Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    if (d != 0.) {
        d = 1. / d;
        double t[9];
        t[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
        t[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
        t[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
        t[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
        t[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
        t[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
        t[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
        t[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
        t[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
        for (int k = 0; k < 9; k++) D.at<double>(k) = t[k];
    }
#undef Sd
    return D;
}

// cv::DownhillSolver with its defaults (TermCriteria(MAX_ITER + EPS, 5000, 1e-6)): OpenCV's downhill_simplex.cpp -- initial
// simplex x - step / 2 and x + step_i / 2 along every axis; per iteration: worst / second worst / best vertex, stop when the
// simplex extent or the value spread is <= eps or 5000 evaluations are spent; reflect the worst vertex through the centroid
// (factor -1), if better than the best try factor -2 and keep the better of the two, if not better than the second worst
// try the half-way point (factor 0.5), else shrink every vertex half-way to the best one.
cv::Ptr<cv::DownhillSolver> cv::DownhillSolver::create() { return cv::Ptr<cv::DownhillSolver>(new cv::DownhillSolver()); }

namespace {
struct NM {
    const cv::MinProblemSolver::Function *f;
    int ndim;
    std::vector<double> p, sum, buf, y;  // p: (ndim + 1) x ndim
    int fcount;
    double *row(int i) { return &p[(size_t)i * ndim]; }
    void update_sum()
    {
        for (int j = 0; j < ndim; j++) sum[j] = 0.;
        for (int i = 0; i <= ndim; i++)
            for (int j = 0; j < ndim; j++) sum[j] += row(i)[j];
    }
    double try_point(int ihi, double alpha_)
    {
        const double alpha = (1.0 - alpha_) / ndim, beta = alpha - alpha_;
        for (int j = 0; j < ndim; j++) buf[j] = sum[j] * alpha - row(ihi)[j] * beta;
        fcount++;
        return f->calc(buf.data());
    }
    void replace_point(int ihi, double alpha_, double ytry)
    {
        const double alpha = (1.0 - alpha_) / ndim, beta = alpha - alpha_;
        for (int j = 0; j < ndim; j++) row(ihi)[j] = sum[j] * alpha - row(ihi)[j] * beta;
        y[ihi] = ytry;
        update_sum();
    }
    double run(double MinRange, double MinError, int nmax)
    {
        fcount = ndim + 1;
        for (int i = 0; i <= ndim; i++) y[i] = f->calc(row(i));
        update_sum();
        for (;;) {
            int ilo = 0, ihi, inhi;
            if (y[0] > y[1]) { ihi = 0; inhi = 1; } else { ihi = 1; inhi = 0; }
            for (int i = 0; i <= ndim; i++) {
                const double yval = y[i];
                if (yval <= y[ilo]) ilo = i;
                if (yval > y[ihi]) { inhi = ihi; ihi = i; }
                else if (yval > y[inhi] && i != ihi) inhi = i;
            }
            if (ilo == inhi || ilo == ihi) {
                for (int i = 0; i <= ndim; i++) {
                    const double yval = y[i];
                    if (yval == y[ilo] && i != ihi && i != inhi) { ilo = i; break; }
                }
            }
            const double error = fabs(y[ihi] - y[ilo]);
            double range = 0;
            for (int j = 0; j < ndim; j++) {
                double minval, maxval;
                minval = maxval = row(0)[j];
                for (int i = 1; i <= ndim; i++) {
                    const double pval = row(i)[j];
                    minval = std::min(minval, pval);
                    maxval = std::max(maxval, pval);
                }
                range = std::max(range, fabs(maxval - minval));
            }
            if (range <= MinRange || error <= MinError || fcount >= nmax) {
                std::swap(y[0], y[ilo]);
                for (int j = 0; j < ndim; j++) std::swap(row(0)[j], row(ilo)[j]);
                break;
            }
            const double y_lo = y[ilo], y_nhi = y[inhi], y_hi = y[ihi];
            double alpha = -1.0;
            double y_alpha = try_point(ihi, alpha);
            if (y_alpha < y_nhi) {
                if (y_alpha < y_lo) {
                    const double beta = -2.0;
                    const double y_beta = try_point(ihi, beta);
                    if (y_beta < y_alpha) { alpha = beta; y_alpha = y_beta; }
                }
                replace_point(ihi, alpha, y_alpha);
            } else {
                const double gamma = 0.5;
                const double y_gamma = try_point(ihi, gamma);
                if (y_gamma < y_hi) replace_point(ihi, gamma, y_gamma);
                else {
                    for (int i = 0; i <= ndim; i++) {
                        if (i != ilo) {
                            for (int j = 0; j < ndim; j++) row(i)[j] = 0.5 * (row(i)[j] + row(ilo)[j]);
                            y[i] = f->calc(row(i));
                        }
                    }
                    fcount += ndim;
                    update_sum();
                }
            }
        }
        return y[0];
    }
};
}  // namespace

double cv::DownhillSolver::minimize(cv::Mat &x)
{
    NM nm;
    nm.f = f_.get();
    nm.ndim = f_->getDims();
    const int n = nm.ndim;
    nm.p.assign((size_t)(n + 1) * n, 0.);
    nm.sum.assign(n, 0.);
    nm.buf.assign(n, 0.);
    nm.y.assign(n + 1, 0.);
    for (int j = 0; j < n; j++) nm.row(0)[j] = x.at<double>(j);
    for (int i = 1; i <= n; i++) {
        for (int j = 0; j < n; j++) nm.row(i)[j] = nm.row(0)[j];
        nm.row(i)[i - 1] += 0.5 * step_.at<double>(i - 1);
    }
    for (int j = 0; j < n; j++) nm.row(0)[j] -= 0.5 * step_.at<double>(j);
    const double res = nm.run(0.000001, 0.000001, 5000);
    for (int j = 0; j < n; j++) x.at<double>(j) = nm.row(0)[j];
    return res;
}

// cv::threshold(samples, samples, 0, 255, THRESH_OTSU + THRESH_BINARY_INV) on the 72 readings of Stag::readCode -- RESTATED
// ("parity unpinned" for this function; the same routine, OpenCV 4.2 getThreshVal_Otsu_8u, is pinned through the golden
// vectors of the aruco path in oracle/aruco_detect_oracle.c): histogram, between-class variance maximised in double,
// first maximum wins; then dst = src > thr ? 0 : maxval.
double cv::threshold(std::vector<unsigned char> &src, std::vector<unsigned char> &dst, double, double maxval, int)
{
    int h[256] = {0};
    const int N = (int)src.size();
    for (int i = 0; i < N; i++) h[src[i]]++;
    double mu = 0, scale = 1. / N;
    for (int i = 0; i < 256; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i, q2, mu2, sigma;
        p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    const int thr = (int)max_val;
    dst.resize(src.size());
    for (int i = 0; i < N; i++) dst[i] = src[i] > thr ? 0 : (unsigned char)maxval;
    return max_val;
}

static void export_lines(EDLines *lines, double *out, int cap, int *n_out)
{
    for (int i = 0; i < lines->noLines && i < cap; i++) {
        const LineSegment &l = lines->lines[i];
        double *o = out + 10 * i;
        o[0] = l.a; o[1] = l.b; o[2] = l.sx; o[3] = l.sy; o[4] = l.ex; o[5] = l.ey;
        o[6] = l.invert; o[7] = l.segmentNo; o[8] = l.firstPixelIndex; o[9] = l.len;
    }
    *n_out = lines->noLines;
}

extern "C" {

// The line-fitting part of DetectLinesByEDPF (EDLines.cpp:877-913) on a given EdgeMap: the same loop (copy the pixels of a
// segment into x / y, SplitSegment2Lines), then JoinCollinearLines(lines, 6.0, 1.50); with validate != 0 also
// ValidateLineSegments(map, src, lines, NULL) (:918).  lines_out: double [cap][10] = a b sx sy ex ey invert segmentNo
// firstPixelIndex len.
int ref_stag_fit_lines(const uint8_t *src, int w, int h, const int32_t *segpix, int n_pix, const int32_t *seg_in, int n_seg_in,
                       int validate, double *lines_out, int cap, int *n_out, int *min_line_len)
{
    EdgeMap *map = new EdgeMap(w, h);
    for (int i = 0; i < n_pix; i++) {
        map->pixels[i].r = segpix[2 * i];
        map->pixels[i].c = segpix[2 * i + 1];
    }
    for (int i = 0; i < n_seg_in; i++) {
        map->segments[i].pixels = map->pixels + seg_in[2 * i];
        map->segments[i].noPixels = seg_in[2 * i + 1];
    }
    map->noSegments = n_seg_in;
    EDLines *lines = new EDLines(w, h);
    lines->MIN_LINE_LEN = ComputeMinLineLength(w, h);
    if (lines->MIN_LINE_LEN < 9) lines->MIN_LINE_LEN = 9;
    *min_line_len = lines->MIN_LINE_LEN;
    for (int segmentNo = 0; segmentNo < map->noSegments; segmentNo++) {
        EdgeSegment *segment = &map->segments[segmentNo];
        for (int k = 0; k < segment->noPixels; k++) {
            lines->x[k] = segment->pixels[k].c;
            lines->y[k] = segment->pixels[k].r;
        }
        SplitSegment2Lines(lines->x, lines->y, segment->noPixels, segmentNo, lines);
    }
    JoinCollinearLines(lines, 6.0, 1.50);
    if (validate) ValidateLineSegments(map, const_cast<unsigned char *>(src), lines, NULL);
    export_lines(lines, lines_out, cap, n_out);
    const int rc = lines->noLines <= cap ? 0 : 1;
    delete lines;
    delete map;
    return rc;
}

// DetectLinesByEDPF (EDLines.cpp:849-941) end to end, as EDInterface::runEDPFandEDLines calls it (EDInterface.cpp:17-18)
int ref_stag_detect_lines(const uint8_t *src, int w, int h, double *lines_out, int cap, int *n_out, int32_t *seg_out, int cap_seg,
                          int *n_seg_out, int32_t *segpix_out, int cap_pix, int *n_pix_out)
{
    EdgeMap *map = NULL;
    EDLines *lines = DetectLinesByEDPF(map, const_cast<unsigned char *>(src), w, h, false, 0);
    export_lines(lines, lines_out, cap, n_out);
    int rc = lines->noLines <= cap ? 0 : 1;
    int total = 0;
    for (int i = 0; i < map->noSegments; i++) {
        const int off = (int)(map->segments[i].pixels - map->pixels), n = map->segments[i].noPixels;
        if (i < cap_seg) {
            seg_out[2 * i] = off;
            seg_out[2 * i + 1] = n;
        } else
            rc = 1;
        if (off + n > total) total = off + n;
    }
    if (total > cap_pix) rc = 1;
    for (int i = 0; i < total && i < cap_pix; i++) {
        segpix_out[2 * i] = map->pixels[i].r;
        segpix_out[2 * i + 1] = map->pixels[i].c;
    }
    *n_seg_out = map->noSegments;
    *n_pix_out = total;
    delete lines;
    delete map;
    return rc;
}

// QuadDetector::detectQuads (QuadDetector.cpp:12-66) end to end on a raw image (runs EDInterface::runEDPFandEDLines inside).
// quads_out: double [cap][12] = 8 corner coordinates, lineInf (3), projectiveDistortion.
int ref_stag_detect_quads(const uint8_t *src, int w, int h, double *quads_out, int cap, int *n_out, int *n_corner_groups)
{
    cv::Mat image(h, w, CV_8UC1, const_cast<uint8_t *>(src));
    EDInterface edi;
    QuadDetector qd(false);
    qd.detectQuads(image, &edi);
    const std::vector<Quad> &q = qd.getQuads();
    for (size_t i = 0; i < q.size() && (int)i < cap; i++) {
        double *o = quads_out + 12 * i;
        for (int k = 0; k < 4; k++) {
            o[2 * k] = q[i].corners[k].x;
            o[2 * k + 1] = q[i].corners[k].y;
        }
        o[8] = q[i].lineInf.x; o[9] = q[i].lineInf.y; o[10] = q[i].lineInf.z;
        o[11] = q[i].projectiveDistortion;
    }
    *n_out = (int)q.size();
    *n_corner_groups = (int)qd.getCornerGroups().size();
    delete edi.getEDLines();
    delete edi.getEdgeMap();
    return (int)q.size() <= cap ? 0 : 1;
}

// Stag::detectMarkers (Stag.cpp:24-51) end to end.  markers_out: double [cap][24] = id, 8 corner coordinates, center (2),
// H row-major (9), lineInf (3), projectiveDistortion.
int ref_stag_detect_markers(const uint8_t *src, int w, int h, int library_hd, int error_correction, int refine, double *markers_out, int cap,
                            int *n_out)
{
    cv::Mat image(h, w, CV_8UC1, const_cast<uint8_t *>(src));
    Stag stag(library_hd, error_correction, false);
    if (refine) {
        stag.detectMarkers(image);
    } else {  // Stag::detectMarkers without its last loop
        stag.markers.clear();
        stag.image = image;
        stag.quadDetector.detectQuads(stag.image, &stag.edInterface);
        std::vector<Quad> quads = stag.quadDetector.getQuads();
        for (size_t i = 0; i < quads.size(); ++i) {
            quads[i].estimateHomography();
            Codeword c = stag.readCode(quads[i]);
            int shift, id;
            if (stag.decoder.decode(c, stag.errorCorrection, id, shift)) {
                Marker marker(quads[i], id);
                marker.shiftCorners2(shift);
                stag.checkDuplicate(marker);
            }
        }
    }
    const std::vector<Marker> m = stag.getMarkerList();
    for (size_t i = 0; i < m.size() && (int)i < cap; i++) {
        double *o = markers_out + 24 * i;
        o[0] = m[i].id;
        for (int k = 0; k < 4; k++) {
            o[1 + 2 * k] = m[i].corners[k].x;
            o[2 + 2 * k] = m[i].corners[k].y;
        }
        o[9] = m[i].center.x; o[10] = m[i].center.y;
        for (int k = 0; k < 9; k++) o[11 + k] = m[i].H.at<double>(k / 3, k % 3);
        o[20] = m[i].lineInf.x; o[21] = m[i].lineInf.y; o[22] = m[i].lineInf.z;
        o[23] = m[i].projectiveDistortion;
    }
    *n_out = (int)m.size();
    return (int)m.size() <= cap ? 0 : 1;
}

// the reference's NFALUT for an image size (ED/EDLines.cpp:286-296: size (w + h) / 8, p = 0.125, logNT), its MIN_LINE_LEN and the
// sample points of Stag::fillCodeLocations
int ref_stag_host_tables(int w, int h, int32_t *lut, int cap, int *lut_size, int *min_line_len, double *locs /* [72][3] */)
{
    const double logNT = 2.0 * (log10((double)w) + log10((double)h));
    NFALUT *L = new NFALUT((w + h) / 8, 0.125, logNT);
    *lut_size = L->LUTSize;
    for (int i = 0; i < L->LUTSize && i < cap; i++) lut[i] = L->LUT[i];
    delete L;
    int m = ComputeMinLineLength(w, h);
    *min_line_len = m < 9 ? 9 : m;
    Stag stag(21, 7, false);
    for (int i = 0; i < 48; i++)
        for (int k = 0; k < 3; k++) locs[3 * i + k] = stag.codeLocs[i].at<double>(k);
    for (int i = 0; i < 12; i++)
        for (int k = 0; k < 3; k++) {
            locs[3 * (48 + i) + k] = stag.blackLocs[i].at<double>(k);
            locs[3 * (60 + i) + k] = stag.whiteLocs[i].at<double>(k);
        }
    return 0;
}

// nfa(n, k, p, logNT) >= 0 (checkValidationByNFA without the table, ED/NFA.cpp:49-51)
int ref_stag_nfa_valid(int n, int k, int w, int h)
{
    return nfa(n, k, 0.125, 2.0 * (log10((double)w) + log10((double)h))) >= 0.0;
}

int ref_stag_smooth5(const uint8_t *src, uint8_t *dst, int w, int h)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = -2; i <= 2; i++) {
                const uint8_t *row = src + (size_t)reflect101(y + i, h) * w;
                int racc = 0;
                for (int j = -2; j <= 2; j++) racc += k[j + 2] * row[reflect101(x + j, w)];
                acc += k[i + 2] * racc;
            }
            dst[(size_t)y * w + x] = (uint8_t)((acc + 128) >> 8);
        }
    return 0;
}

// grad: int16 [h][w]; dir: uint8 [h][w] (written only where grad >= thresh, as in the reference: the caller zero-fills)
int ref_stag_gradient(const uint8_t *smooth, int16_t *grad, uint8_t *dir, int w, int h, int grad_thresh)
{
    ComputeGradientMapByPrewitt(const_cast<unsigned char *>(smooth), grad, dir, w, h, grad_thresh);
    return 0;
}

// edge: uint8 [h][w] anchor map (ANCHOR_PIXEL where an anchor is); sorted: int32 [cap] anchor offsets in the order
// JoinAnchorPointsUsingSortedAnchors consumes them from the END (ascending gradient; EDInternals.cpp:857)
int ref_stag_anchors(const int16_t *grad, const uint8_t *dir, int w, int h, int grad_thresh, int anchor_thresh,
                     int scan_interval, uint8_t *edge, int32_t *sorted, int cap, int *n_out)
{
    EdgeMap *map = new EdgeMap(w, h);
    ComputeAnchorPoints(const_cast<short *>(grad), const_cast<unsigned char *>(dir), map, grad_thresh, anchor_thresh,
                        scan_interval);
    memcpy(edge, map->edgeImg, (size_t)w * h);
    int n = 0;
    int *A = SortAnchorsByGradValue(const_cast<short *>(grad), map, &n);
    for (int i = 0; i < n && i < cap; i++) sorted[i] = A[i];
    *n_out = n;
    delete[] A;
    delete map;
    return n <= cap ? 0 : 1;
}

// JoinAnchorPointsUsingSortedAnchors (EDInternals.cpp:842-1448) on a given anchor map.  edge: in = anchor map, out = the
// edge image after the routing.  segpix: int32 [cap_pix][2] = map->pixels as (r, c); seg: int32 [cap_seg][2] = (offset of
// segments[i].pixels inside map->pixels, noPixels).
int ref_stag_route(const int16_t *grad, const uint8_t *dir, int w, int h, int grad_thresh, int min_path_len, uint8_t *edge,
                   int32_t *segpix, int cap_pix, int32_t *seg, int cap_seg, int *n_seg, int *n_pix)
{
    EdgeMap *map = new EdgeMap(w, h);
    memcpy(map->edgeImg, edge, (size_t)w * h);
    memset(map->pixels, 0xff, sizeof(Pixel) * (size_t)w * h);  // (r, c) = (-1, -1) where the reference would read uninitialised memory
    JoinAnchorPointsUsingSortedAnchors(const_cast<short *>(grad), const_cast<unsigned char *>(dir), map, grad_thresh, min_path_len);
    memcpy(edge, map->edgeImg, (size_t)w * h);
    int total = 0, rc = 0;
    for (int i = 0; i < map->noSegments; i++) {
        const int off = (int)(map->segments[i].pixels - map->pixels), n = map->segments[i].noPixels;
        if (i < cap_seg) {
            seg[2 * i] = off;
            seg[2 * i + 1] = n;
        } else
            rc = 1;
        if (off + n > total) total = off + n;
    }
    if (total > cap_pix) rc = 1;
    for (int i = 0; i < total && i < cap_pix; i++) {
        segpix[2 * i] = map->pixels[i].r;
        segpix[2 * i + 1] = map->pixels[i].c;
    }
    *n_seg = map->noSegments;
    *n_pix = total;
    delete map;
    return rc;
}

int ref_stag_smooth3(const uint8_t *src, uint8_t *dst, int w, int h)
{
    static const unsigned k[3] = {10, 236, 10};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int i = -1; i <= 1; i++) {
                const uint8_t *row = src + (size_t)reflect101(y + i, h) * w;
                unsigned racc = 0;
                for (int j = -1; j <= 1; j++) racc += k[j + 1] * row[reflect101(x + j, w)];
                acc += k[i + 1] * racc;
            }
            dst[(size_t)y * w + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    return 0;
}

// ValidateEdgeSegments on a given EdgeMap (segments as (first pixel, length) pairs into segpix) and validation image.
// Out: edge image, validated segments as (first pixel, length) pairs into the same segpix.
int ref_stag_validate(const uint8_t *smooth2, int w, int h, const int32_t *segpix, int n_pix, const int32_t *seg_in, int n_seg_in,
                      double div, uint8_t *edge_out, int32_t *seg_out, int cap_seg, int *n_seg_out)
{
    EdgeMap *map = new EdgeMap(w, h);
    for (int i = 0; i < n_pix; i++) {
        map->pixels[i].r = segpix[2 * i];
        map->pixels[i].c = segpix[2 * i + 1];
    }
    for (int i = 0; i < n_seg_in; i++) {
        map->segments[i].pixels = map->pixels + seg_in[2 * i];
        map->segments[i].noPixels = seg_in[2 * i + 1];
    }
    map->noSegments = n_seg_in;
    ValidateEdgeSegments(map, const_cast<unsigned char *>(smooth2), div);
    memcpy(edge_out, map->edgeImg, (size_t)w * h);
    int rc = 0;
    for (int i = 0; i < map->noSegments; i++) {
        if (i >= cap_seg) { rc = 1; break; }
        seg_out[2 * i] = (int)(map->segments[i].pixels - map->pixels);
        seg_out[2 * i + 1] = map->segments[i].noPixels;
    }
    *n_seg_out = map->noSegments;
    delete map;
    return rc;
}

int ref_stag_constants(int *edge_vertical, int *edge_horizontal, int *anchor_pixel)
{
    *edge_vertical = EDGE_VERTICAL;
    *edge_horizontal = EDGE_HORIZONTAL;
    *anchor_pixel = ANCHOR_PIXEL;
    return 0;
}

}  // extern "C"
