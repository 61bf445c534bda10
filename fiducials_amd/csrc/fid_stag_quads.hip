// fid_stag_quads.hip -- STag rows s7, s8: quad detection, code reading and decoding.
// Part of the fid_stag.hip translation unit (included there; not compiled on its own).
// ------------------------------------------------------------------------------------------------ K14: quads
// QuadDetector::detectQuads (QuadDetector.cpp:12-66) behind EDLines: groupLines (:78-127) + EDInterface::correctLineDirection
// (EDInterface.cpp:25-142), detectCorners (:129-181), checkIfCornersFormQuad (:183-271) and the Quad constructor
// (Quad.cpp:8-12: line at infinity :55-130, projective distortion :132-148).  Lines of one edge segment form one group
// and groups do not interact: one wave per validated segment; the wave runs the group's (short) scalar logic uniformly
// and spreads only the "is the corner on the edge segment" scan over its lanes.  Quads land at the slot of their group's
// first line and are gathered in group order afterwards.
struct StagCorner {
    double x, y;
    int l1, l2;  // indices of the two lines; -1 = the placeholder lines of the "missing fourth corner"
};

__device__ __forceinline__ double sq_cross(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
__device__ __forceinline__ double sq_dist2(double ax, double ay, double bx, double by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

// EDInterface::intersectionOfLineSegments (EDInterface.cpp:144-184)
__device__ void sq_intersect(const fid_stag_line &l1, const fid_stag_line &l2, double *ox, double *oy)
{
    double aL1, bL1, aL2, bL2;
    if (l1.invert == 0) {
        aL1 = l1.b;
        bL1 = l1.a;
    } else {
        aL1 = 1 / l1.b;
        bL1 = -l1.a / l1.b;
    }
    if (l2.invert == 0) {
        aL2 = l2.b;
        bL2 = l2.a;
    } else {
        aL2 = 1 / l2.b;
        bL2 = -l2.a / l2.b;
    }
    double x = (bL2 - bL1) / (aL1 - aL2);
    double y = aL1 * x + bL1;
    if (l1.invert == 1 && l1.b == 0) {
        if (l2.invert == 0) y = l2.a + l2.b * l1.a;
        else y = (l1.a - l2.a) / l2.b;
        x = l1.a;
    } else if (l2.invert == 1 && l2.b == 0) {
        if (l1.invert == 0) y = l1.a + l1.b * l2.a;
        else y = (l2.a - l1.a) / l1.b;
        x = l2.a;
    }
    *ox = x;
    *oy = y;
}

// samples correctLineDirection takes along a line
__device__ __forceinline__ int sq_direction_samples(const fid_stag_line &ls)
{
    if (ls.invert == 0) return (int)(fmax(ls.sx, ls.ex) + 0.5) - (int)fmin(ls.sx, ls.ex) + 1;
    return (int)(fmax(ls.sy, ls.ey) + 0.5) - (int)fmin(ls.sy, ls.ey) + 1;
}
// EDInterface::correctLineDirection: going from start to end the darker side must be on the right
template <int LANES = 1>
__device__ void sq_correct_direction(const uint8_t *__restrict__ img, int W, int H, fid_stag_line &ls, int lane = 0)
{
    int n, mn;
    if (ls.invert == 0) {
        mn = (int)fmin(ls.sx, ls.ex);
        n = (int)(fmax(ls.sx, ls.ex) + 0.5) - mn + 1;
    } else {
        mn = (int)fmin(ls.sy, ls.ey);
        n = (int)(fmax(ls.sy, ls.ey) + 0.5) - mn + 1;
    }
    const double offset = 1;
    const bool fwd = ls.invert == 0 ? ls.sx < ls.ex : ls.sy < ls.ey;
    auto sample = [&](int i, int *rx, int *ry, int *lx, int *ly) {
        if (ls.invert == 0) {
            const double nx = mn + i, ny = ls.b * nx + ls.a;
            const int up = (int)round(ny - offset), dn = (int)round(ny + offset);
            *rx = (int)nx; *lx = (int)nx;
            *ry = fwd ? dn : up;
            *ly = fwd ? up : dn;
        } else {
            const double ny = mn + i, nx = ls.b * ny + ls.a;
            const int lo = (int)round(nx - offset), hi = (int)round(nx + offset);
            *ry = (int)ny; *ly = (int)ny;
            *rx = fwd ? lo : hi;
            *lx = fwd ? hi : lo;
        }
    };
    int rx0, ry0, lx0, ly0, rx1, ry1, lx1, ly1;
    sample(0, &rx0, &ry0, &lx0, &ly0);
    sample(n - 1, &rx1, &ry1, &lx1, &ly1);
    const int minX = min(min(rx0, rx1), min(lx0, lx1)), maxX = max(max(rx0, rx1), max(lx0, lx1));
    const int minY = min(min(ry0, ry1), min(ly0, ly1)), maxY = max(max(ry0, ry1), max(ly0, ly1));
    const bool safe = minX < 0 || maxX >= W || minY < 0 || maxY >= H;
    unsigned accR = 0, accL = 0;
    // LANES > 1 (round 6): the samples across the lanes of the wave, the two sums by wave reductions (integers: the same sums).  A
    // lane per line walked a marker's 450-pixel edge alone, two loads a step: the longest line was k_stag_quads' duration.
    for (int i = LANES > 1 ? lane : 0; i < n; i += LANES) {
        int rx, ry, lx, ly;
        sample(i, &rx, &ry, &lx, &ly);
        const bool rin = rx >= 0 && rx < W && ry >= 0 && ry < H, lin = lx >= 0 && lx < W && ly >= 0 && ly < H;
        // (without the safe read the reference reads unchecked; points between two in-range end points are in range)
        accR += rin ? img[ry * W + rx] : (safe ? 128u : 0u);
        accL += lin ? img[ly * W + lx] : (safe ? 128u : 0u);
    }
    if (LANES > 1) {
        accR = (unsigned)wave_sum_i32((int)accR);
        accL = (unsigned)wave_sum_i32((int)accL);
    }
    if (accL < accR) {
        const double t1 = ls.sx, t2 = ls.sy;
        ls.sx = ls.ex; ls.sy = ls.ey;
        ls.ex = t1; ls.ey = t2;
    }
}

struct StagQuadCtx {
    const fid_stag_line *L;
    const int *order;  // line index of the k-th line of the group
};

__device__ bool sq_quad_simple(const StagCorner c[4])
{
    const double v13x = c[2].x - c[0].x, v13y = c[2].y - c[0].y, v12x = c[1].x - c[0].x, v12y = c[1].y - c[0].y;
    const double v14x = c[3].x - c[0].x, v14y = c[3].y - c[0].y;
    if (sq_cross(v13x, v13y, v12x, v12y) * sq_cross(v13x, v13y, v14x, v14y) >= 0) return false;
    const double v24x = c[3].x - c[1].x, v24y = c[3].y - c[1].y, v21x = c[0].x - c[1].x, v21y = c[0].y - c[1].y;
    const double v23x = c[2].x - c[1].x, v23y = c[2].y - c[1].y;
    if (sq_cross(v24x, v24y, v21x, v21y) * sq_cross(v24x, v24y, v23x, v23y) >= 0) return false;
    return true;
}

// the end point of a corner's line that is farther from the corner, relative to the corner
__device__ void sq_far_point(const StagCorner &c, const fid_stag_line &l, double *px, double *py)
{
    if (sq_dist2(c.x, c.y, l.sx, l.sy) > sq_dist2(c.x, c.y, l.ex, l.ey)) {
        *px = l.sx - c.x;
        *py = l.sy - c.y;
    } else {
        *px = l.ex - c.x;
        *py = l.ey - c.y;
    }
}

__device__ bool sq_face_each_other(const fid_stag_line *L, const StagCorner &c1, const StagCorner &c2)
{
    double c1p1x, c1p1y, c1p2x, c1p2y, c2p1x, c2p1y, c2p2x, c2p2y;
    sq_far_point(c1, L[c1.l1], &c1p1x, &c1p1y);
    sq_far_point(c1, L[c1.l2], &c1p2x, &c1p2y);
    sq_far_point(c2, L[c2.l1], &c2p1x, &c2p1y);
    sq_far_point(c2, L[c2.l2], &c2p2x, &c2p2y);
    const double c1c2x = c2.x - c1.x, c1c2y = c2.y - c1.y, c2c1x = c1.x - c2.x, c2c1y = c1.y - c2.y;
    if (sq_cross(c1c2x, c1c2y, c1p1x, c1p1y) * sq_cross(c1c2x, c1c2y, c1p2x, c1p2y) >= 0) return false;
    if (sq_cross(c1p1x, c1p1y, c1c2x, c1c2y) * sq_cross(c1p1x, c1p1y, c1p2x, c1p2y) <= 0) return false;
    if (sq_cross(c2c1x, c2c1y, c2p1x, c2p1y) * sq_cross(c2c1x, c2c1y, c2p2x, c2p2y) >= 0) return false;
    if (sq_cross(c2p1x, c2p1y, c2c1x, c2c1y) * sq_cross(c2p1x, c2p1y, c2p2x, c2p2y) <= 0) return false;
    return true;
}

__device__ StagCorner sq_make_corner(const fid_stag_line *L, int la, int lb)
{
    StagCorner c;
    sq_intersect(L[la], L[lb], &c.x, &c.y);
    c.l1 = la;
    c.l2 = lb;
    return c;
}

// checkIfCornersFormQuad (QuadDetector.cpp:183-271), thresDist = 7
__device__ bool sq_form_quad(const fid_stag_line *L, StagCorner c[4])
{
    const double thresDist = 7;
    if (!sq_face_each_other(L, c[0], c[2])) return false;
    StagCorner e1 = sq_make_corner(L, c[0].l1, c[2].l1), e3 = sq_make_corner(L, c[0].l2, c[2].l2);
    StagCorner est[4] = {c[0], e1, c[2], e3};
    if (!sq_quad_simple(est)) {
        e1 = sq_make_corner(L, c[0].l1, c[2].l2);
        e3 = sq_make_corner(L, c[0].l2, c[2].l1);
        est[1] = e1;
        est[3] = e3;
    }
    if (!sq_quad_simple(est)) return false;
    const double d11 = sq_dist2(c[1].x, c[1].y, e1.x, e1.y), d13 = sq_dist2(c[1].x, c[1].y, e3.x, e3.y);
    const double d31 = sq_dist2(c[3].x, c[3].y, e1.x, e1.y), d33 = sq_dist2(c[3].x, c[3].y, e3.x, e3.y);
    const double t2 = thresDist * thresDist;
    if (d11 < d13 && d11 < d31 && d11 < d33 && d11 < t2) {
        if (!(d33 < t2)) c[3] = e3;
    } else if (d13 < d11 && d13 < d31 && d13 < d33 && d13 < t2) {
        if (!(d31 < t2)) c[3] = e1;
    } else if (d31 < d11 && d31 < d13 && d31 < d33 && d31 < t2) {
        if (!(d13 < t2)) c[1] = e3;
    } else if (d33 < d11 && d33 < d13 && d33 < d31 && d33 < t2) {
        if (!(d11 < t2)) c[1] = e1;
    } else
        return false;
    const double v13x = c[2].x - c[0].x, v13y = c[2].y - c[0].y, v12x = c[1].x - c[0].x, v12y = c[1].y - c[0].y;
    if (sq_cross(v13x, v13y, v12x, v12y) > 0) {
        const StagCorner t = c[1];
        c[1] = c[3];
        c[3] = t;
    }
    return true;
}

// Quad::calculateLineAtInfinity + calculateProjectiveDistortion (Quad.cpp:55-148)
__device__ void sq_make_quad(const StagCorner c[4], fid_stag_quad *q)
{
    for (int i = 0; i < 4; i++) {
        q->corners[2 * i] = c[i].x;
        q->corners[2 * i + 1] = c[i].y;
    }
    const double cross14 = sq_cross(c[0].x, c[0].y, c[3].x, c[3].y), cross23 = sq_cross(c[1].x, c[1].y, c[2].x, c[2].y);
    const double cross12 = sq_cross(c[0].x, c[0].y, c[1].x, c[1].y), cross34 = sq_cross(c[2].x, c[2].y, c[3].x, c[3].y);
    const double v23x = c[1].x - c[2].x, v23y = c[1].y - c[2].y, v14x = c[0].x - c[3].x, v14y = c[0].y - c[3].y;
    const double v34x = c[2].x - c[3].x, v34y = c[2].y - c[3].y, v12x = c[0].x - c[1].x, v12y = c[0].y - c[1].y;
    double i1x, i1y, i2x, i2y;
    const bool par1 = sq_cross(v14x, v14y, v23x, v23y) == 0, par2 = sq_cross(v12x, v12y, v34x, v34y) == 0;
    if (par1 && par2) {
        q->lineInf[0] = 0; q->lineInf[1] = 0; q->lineInf[2] = 1;
    } else {
        if (par1) {
            i2x = (cross12 * v34x - v12x * cross34) / (v12x * v34y - v12y * v34x);
            i2y = (cross12 * v34y - v12y * cross34) / (v12x * v34y - v12y * v34x);
            i1x = i2x + v14x;
            i1y = i2y + v14y;
        } else if (par2) {
            i1x = (cross14 * v23x - v14x * cross23) / (v14x * v23y - v14y * v23x);
            i1y = (cross14 * v23y - v14y * cross23) / (v14x * v23y - v14y * v23x);
            i2x = i1x + v12x;
            i2y = i1y + v12y;
        } else {
            i1x = (cross14 * v23x - v14x * cross23) / (v14x * v23y - v14y * v23x);
            i1y = (cross14 * v23y - v14y * cross23) / (v14x * v23y - v14y * v23x);
            i2x = (cross12 * v34x - v12x * cross34) / (v12x * v34y - v12y * v34x);
            i2y = (cross12 * v34y - v12y * cross34) / (v12x * v34y - v12y * v34x);
        }
        double l1 = i1y - i2y, l2 = i2x - i1x, l3 = i1x * i2y - i2x * i1y;
        const double nrm = sqrt(l1 * l1 + l2 * l2);
        l1 /= nrm;
        l2 /= nrm;
        l3 /= nrm;
        q->lineInf[0] = l1; q->lineInf[1] = l2; q->lineInf[2] = l3;
    }
    double cur = fabs(q->lineInf[0] * c[0].x + q->lineInf[1] * c[0].y + q->lineInf[2]);
    double mn = cur, mx = cur;
    for (int i = 1; i < 4; i++) {
        cur = fabs(q->lineInf[0] * c[i].x + q->lineInf[1] * c[i].y + q->lineInf[2]);
        if (cur < mn) mn = cur;
        if (cur > mx) mx = cur;
    }
    q->projectiveDistortion = mx / mn;
}

// first line and number of lines of every validated segment (lines are stored segment by segment)
__device__ __forceinline__ void k_stag_line_ranges_impl(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, int2 *__restrict__ range)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = *nlines;
    if (i >= n) return;
    const int sg = lines[i].segmentNo;
    if (i == 0 || lines[i - 1].segmentNo != sg) range[sg].x = i;
    if (i == n - 1 || lines[i + 1].segmentNo != sg) range[sg].y = i + 1;
}
__global__ __launch_bounds__(256) void k_stag_line_ranges(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, int2 *__restrict__ range)
{
    k_stag_line_ranges_impl(lines, nlines, range);
}
struct k_stag_line_ranges_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, int2 *__restrict__ range) const { k_stag_line_ranges_impl(lines, nlines, range); }
};

template <bool FM = false>
__device__ __forceinline__ void k_stag_quads_impl(fid_stag_line *__restrict__ lines, const int2 *__restrict__ range, const int *__restrict__ nsegs,
                                                    const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, const uint8_t *__restrict__ img, int W,
                                                    int H, StagCorner *__restrict__ corner_slots, int *__restrict__ order_slots,
                                                    fid_stag_quad *__restrict__ quad_slots, int *__restrict__ counts)
{
    const int seg = (int)STAG_BX<FM>() * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= *nsegs) return;
    const int lo = range[seg].x, n = range[seg].y - lo;
    if (lane == 0) counts[seg] = 0;
    if (range[seg].y == 0 || n < 4) return;  // groups need >= 4 lines of one edge segment
    // ---- groupLines: fix the direction of every line of the group (each lane one line), then the order of the group
    // (short lines a lane each; the long ones -- more samples than a wave has lanes -- one after the other by the whole wave)
    unsigned long long longm[2] = {0ull, 0ull};  // (groups of up to 128 lines keep their long ones as bits; beyond: a lane each)
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        bool is_long = false;
        if (k < n) {
            fid_stag_line l = lines[lo + k];
            is_long = k < 128 && sq_direction_samples(l) > 64;
            if (!is_long) {
                sq_correct_direction(img, W, H, l);
                lines[lo + k] = l;
            }
        }
        const unsigned long long m = __ballot(is_long);
        if (k0 < 128) longm[k0 >> 6] = m;
    }
    for (int w = 0; w < 2; w++)
        while (longm[w]) {
            const int k = 64 * w + __builtin_ctzll(longm[w]);
            longm[w] &= longm[w] - 1;
            fid_stag_line l = lines[lo + k];
            sq_correct_direction<64>(img, W, H, l, lane);
            if (lane == 0) lines[lo + k] = l;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const fid_stag_line *L = lines;
    bool rev;
    {
        double ix, iy;
        sq_intersect(L[lo], L[lo + 1], &ix, &iy);
        rev = fabs(L[lo].sx - ix) + fabs(L[lo].sy - iy) < fabs(L[lo].ex - ix) + fabs(L[lo].ey - iy);
    }
    int *order = order_slots + lo;
    for (int k = lane; k < n; k += 64) order[k] = rev ? lo + n - 1 - k : lo + k;
    // ---- detectCorners: consecutive lines that turn the right way and meet on the edge segment
    StagCorner *corners = corner_slots + lo;
    int nc = 0;
    const int2 *sp = pix + vsegs[seg].x;
    const int spn = vsegs[seg].y;
    for (int k = 0; k < n; k++) {
        const int a = rev ? lo + n - 1 - k : lo + k, kn = (k + 1) % n, b = rev ? lo + n - 1 - kn : lo + kn;
        const fid_stag_line &l1 = L[a], &l2 = L[b];
        if (sq_cross(l1.ex - l1.sx, l1.ey - l1.sy, l2.ex - l1.sx, l2.ey - l1.sy) <= 0) continue;
        double ix, iy;
        sq_intersect(l1, l2, &ix, &iy);
        const double thresManh = 7 * 1.41;
        bool on = false;
        for (int e0 = 0; e0 < spn && !on; e0 += 64) {
            const int e = e0 + lane;
            bool hit = false;
            if (e < spn) hit = fabs(sp[e].y - ix) + fabs(sp[e].x - iy) < thresManh;
            on = __ballot(hit) != 0ull;
        }
        if (!on) continue;
        if (lane == 0) {
            corners[nc].x = ix; corners[nc].y = iy; corners[nc].l1 = a; corners[nc].l2 = b;
        }
        nc++;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (nc < 3 || lane != 0) return;
    // ---- quads from the corner group (lane 0)
    fid_stag_quad *out = quad_slots + lo;
    int nq = 0;
    for (int ci = 0; ci < nc; ci++) {
        const int i1 = ci, i2 = (i1 + 1) % nc, i3 = (i1 + 2) % nc, i4 = (i1 + 3) % nc;
        StagCorner c[4] = {corners[i1], corners[i2], corners[i3], corners[i4]};
        if (i1 == i4) {
            c[3].x = INFINITY; c[3].y = INFINITY; c[3].l1 = c[3].l2 = -1;
        }
        if (!sq_form_quad(L, c)) continue;
        fid_stag_quad q;
        sq_make_quad(c, &q);
        if (q.projectiveDistortion > 1.5) continue;  // thresProjectiveDistortion
        out[nq++] = q;
        if (nc <= 4) break;
    }
    counts[seg] = nq;
}
__global__ __launch_bounds__(256) void k_stag_quads(fid_stag_line *__restrict__ lines, const int2 *__restrict__ range, const int *__restrict__ nsegs, const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, const uint8_t *__restrict__ img, int W, int H, StagCorner *__restrict__ corner_slots, int *__restrict__ order_slots, fid_stag_quad *__restrict__ quad_slots, int *__restrict__ counts)
{
    k_stag_quads_impl(lines, range, nsegs, vsegs, pix, img, W, H, corner_slots, order_slots, quad_slots, counts);
}
struct k_stag_quads_fn {
    static constexpr int kBounds = 256;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(fid_stag_line *__restrict__ lines, const int2 *__restrict__ range, const int *__restrict__ nsegs, const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, const uint8_t *__restrict__ img, int W, int H, StagCorner *__restrict__ corner_slots, int *__restrict__ order_slots, fid_stag_quad *__restrict__ quad_slots, int *__restrict__ counts) const { k_stag_quads_impl<true>(lines, range, nsegs, vsegs, pix, img, W, H, corner_slots, order_slots, quad_slots, counts); }
};

__device__ __forceinline__ void k_stag_gather_quads_impl(const int2 *__restrict__ range, const int *__restrict__ nsegs, const int *__restrict__ counts,
                                                          const int *__restrict__ total, const fid_stag_quad *__restrict__ slots,
                                                          fid_stag_quad *__restrict__ out)
{
    const int seg = blockIdx.x * 64 + threadIdx.x;
    const int ns = *nsegs;
    if (seg >= ns) return;
    const int o = counts[seg], n = (seg + 1 < ns ? counts[seg + 1] : *total) - o;
    const fid_stag_quad *Q = slots + range[seg].x;
    for (int j = 0; j < n; j++) out[o + j] = Q[j];
}
__global__ __launch_bounds__(64) void k_stag_gather_quads(const int2 *__restrict__ range, const int *__restrict__ nsegs, const int *__restrict__ counts, const int *__restrict__ total, const fid_stag_quad *__restrict__ slots, fid_stag_quad *__restrict__ out)
{
    k_stag_gather_quads_impl(range, nsegs, counts, total, slots, out);
}
struct k_stag_gather_quads_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(const int2 *__restrict__ range, const int *__restrict__ nsegs, const int *__restrict__ counts, const int *__restrict__ total, const fid_stag_quad *__restrict__ slots, fid_stag_quad *__restrict__ out) const { k_stag_gather_quads_impl(range, nsegs, counts, total, slots, out); }
};

// ------------------------------------------------------------------------------------------------ K15: decoding
// The loop of Stag::detectMarkers (Stag.cpp:36-48) per quad: Quad::estimateHomography (Quad.cpp:14-53), Stag::readCode
// (Stag.cpp:89-127: 48 code + 12 black + 12 white sample points through H, readPixelSafeBilinear utility.cpp:20-55 --
// weights are the DISTANCES to the four neighbours, as in the reference --, Otsu over the 72 readings, dark = 1),
// Decoder::decode (Decoder.cpp:45-56: first codeword within errorCorrection bits; id = i % n, shift = i / n),
// Marker::shiftCorners2 (Marker.cpp:27-52).  One wave per quad: a lane per sample point, the codeword search spread over
// the lanes.  Stag::checkDuplicate (Stag.cpp:57-72) then runs over the decoded quads in order (k_stag_dedup).
// The 72 sample points are made on the host with its libm, exactly as Stag::fillCodeLocations (Stag.cpp:129-277) does.
__device__ void sd_homography(const double cor[8], const double li[3], double H[9], double cen[2])
{
    double ax[4], ay[4];
    for (int i = 0; i < 4; i++) {
        ax[i] = cor[2 * i] / (li[0] * cor[2 * i] + li[1] * cor[2 * i + 1] + li[2]);
        ay[i] = cor[2 * i + 1] / (li[0] * cor[2 * i] + li[1] * cor[2 * i + 1] + li[2]);
    }
    double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, B[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    A[6] = -li[0] / li[2];
    A[7] = -li[1] / li[2];
    A[8] = 1 / li[2];
    B[0] = ax[1] - ax[0]; B[1] = ax[3] - ax[0]; B[2] = ax[0];
    B[3] = ay[1] - ay[0]; B[4] = ay[3] - ay[0]; B[5] = ay[0];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) H[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    const double c0 = H[0] * 0.5 + H[1] * 0.5 + H[2] * 1, c1 = H[3] * 0.5 + H[4] * 0.5 + H[5] * 1, c2 = H[6] * 0.5 + H[7] * 0.5 + H[8] * 1;
    cen[0] = c0 / c2;
    cen[1] = c1 / c2;
}

__device__ int sd_read_bilinear(const uint8_t *__restrict__ img, int W, int H, double px, double py)
{
    if (!(px >= 0 && px <= W - 1 && py >= 0 && py <= H - 1)) return 128;
    const int x1 = (int)floor(px), x2 = (int)ceil(px), y1 = (int)floor(py), y2 = (int)ceil(py);
    const double d1 = sqrt((x1 - px) * (x1 - px) + (y1 - py) * (y1 - py)), d2 = sqrt((x1 - px) * (x1 - px) + (y2 - py) * (y2 - py));
    const double d3 = sqrt((x2 - px) * (x2 - px) + (y1 - py) * (y1 - py)), d4 = sqrt((x2 - px) * (x2 - px) + (y2 - py) * (y2 - py));
    const double tot = d1 + d2 + d3 + d4;
    double acc = 0;
    acc += img[y1 * W + x1] * d1;
    acc += img[y2 * W + x1] * d2;
    acc += img[y1 * W + x2] * d3;
    acc += img[y2 * W + x2] * d4;
    if (tot == 0) return 0;  // a point on the pixel grid: 0 / 0 in the reference, which x86 converts to 0
    return (int)(acc / tot);
}

__device__ __forceinline__ void k_stag_decode_impl(const fid_stag_quad *__restrict__ quads, const int *__restrict__ nquads,
                                                     const uint8_t *__restrict__ img, int W, int H, const double *__restrict__ locs /* [72][3] */,
                                                     const unsigned long long *__restrict__ words, int nwords, int err_corr,
                                                     fid_stag_marker *__restrict__ cand, int *__restrict__ found)
{
    __shared__ int s_hist[4][256];
    const int wq = threadIdx.x >> 6, q = blockIdx.x * 4 + wq, lane = threadIdx.x & 63;
    if (q >= *nquads) return;
    const fid_stag_quad Q = quads[q];
    double Hm[9], cen[2];
    sd_homography(Q.corners, Q.lineInf, Hm, cen);
    int *hist = s_hist[wq];
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __builtin_amdgcn_wave_barrier();
    int smp[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        const int i = lane + 64 * k;
        if (i < 72) {
            const double *L = locs + 3 * i;
            const double p0 = Hm[0] * L[0] + Hm[1] * L[1] + Hm[2] * L[2], p1 = Hm[3] * L[0] + Hm[4] * L[1] + Hm[5] * L[2];
            const double p2 = Hm[6] * L[0] + Hm[7] * L[1] + Hm[8] * L[2];
            smp[k] = sd_read_bilinear(img, W, H, p0 / p2, p1 / p2) & 255;
            atomicAdd(&hist[smp[k]], 1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // Otsu threshold of the 72 readings (cv::threshold THRESH_OTSU: getThreshVal_Otsu_8u, the histogram form)
    int thr;
    {
        const double scale = 1. / 72;
        double mu = 0;
        for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
        mu *= scale;
        double mu1 = 0, q1 = 0, max_sigma = 0;
        int max_val = 0;
        // (round 6) 72 readings leave most of the 256 bins empty, and an empty bin only passes mu1 through (mu1 * q1) / q1.  Once that
        // returns the mu1 it was given, this bin and every empty bin behind it change nothing: same q1, same mu1, the same sigma again,
        // which is not above the maximum it already had its chance to set.  The walk jumps to the next non-empty bin there; where the
        // round trip still moves mu1 by a rounding it goes on bin by bin, as the reference does.  (Two divisions a bin, 256 bins, by
        // every lane alike: 30 of this kernel's 46 us.)
        unsigned long long nz[4];
#pragma unroll
        for (int w = 0; w < 4; w++) nz[w] = __ballot(hist[64 * w + lane] != 0);
        int i = 0;
        while (i < 256) {
            const int h = hist[i];
            const double p_i = h * scale;
            const double mu1_in = mu1;
            mu1 *= q1;
            q1 += p_i;
            const double q2 = 1. - q1;
            if (!(fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07)) {
                mu1 = (mu1 + i * p_i) / q1;
                const double mu2 = (mu - q1 * mu1) / q2;
                const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
                if (sigma > max_sigma) {
                    max_sigma = sigma;
                    max_val = i;
                }
            }
            if (h == 0 && mu1 == mu1_in) {
                int j = 256;
#pragma unroll
                for (int w = 3; w >= 0; w--) {
                    unsigned long long mm = nz[w];
                    if (w == ((i + 1) >> 6)) mm &= ~0ull << ((i + 1) & 63);
                    if (w >= ((i + 1) >> 6) && mm) j = 64 * w + (int)__builtin_ctzll(mm);
                }
                i = j;
            } else {
                i++;
            }
        }
        thr = max_val;
    }
    // THRESH_BINARY_INV: readings above the threshold -> 0, the others -> 255 -> bit 1
    const unsigned long long code = __ballot(lane < 48 && smp[0] <= thr);
    // Decoder::decode: the first codeword within err_corr bits
    int hit = -1;
    for (int base = 0; base < nwords && hit < 0; base += 64) {
        const int i = base + lane;
        const bool ok = i < nwords && __builtin_popcountll(code ^ words[i]) <= err_corr;
        const unsigned long long m = __ballot(ok);
        if (m) hit = base + __builtin_ctzll(m);
    }
    if (lane != 0) return;
    found[q] = hit >= 0 ? 1 : 0;
    if (hit < 0) return;
    const int n = nwords / 4, id = hit % n, shift = hit / n;
    fid_stag_marker M;
    M.id = id;
    M.shift = shift;
    for (int k = 0; k < 4; k++) {  // shiftCorners2: corner k <- corner (k + shift) % 4
        M.corners[2 * k] = Q.corners[2 * ((k + shift) & 3)];
        M.corners[2 * k + 1] = Q.corners[2 * ((k + shift) & 3) + 1];
    }
    for (int k = 0; k < 3; k++) M.lineInf[k] = Q.lineInf[k];
    M.projectiveDistortion = Q.projectiveDistortion;
    if (shift >= 1 && shift <= 3) sd_homography(M.corners, M.lineInf, M.H, M.center);
    else {
        for (int k = 0; k < 9; k++) M.H[k] = Hm[k];
        M.center[0] = cen[0];
        M.center[1] = cen[1];
    }
    M.code = code;
    cand[q] = M;
}
__global__ __launch_bounds__(256) void k_stag_decode(const fid_stag_quad *__restrict__ quads, const int *__restrict__ nquads, const uint8_t *__restrict__ img, int W, int H, const double *__restrict__ locs , const unsigned long long *__restrict__ words, int nwords, int err_corr, fid_stag_marker *__restrict__ cand, int *__restrict__ found)
{
    k_stag_decode_impl(quads, nquads, img, W, H, locs, words, nwords, err_corr, cand, found);
}
struct k_stag_decode_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const fid_stag_quad *__restrict__ quads, const int *__restrict__ nquads, const uint8_t *__restrict__ img, int W, int H, const double *__restrict__ locs , const unsigned long long *__restrict__ words, int nwords, int err_corr, fid_stag_marker *__restrict__ cand, int *__restrict__ found) const { k_stag_decode_impl(quads, nquads, img, W, H, locs, words, nwords, err_corr, cand, found); }
};

// Stag::checkDuplicate over the decoded quads in quad order: one marker per id, the least distorted one, at the position of
// the first quad that showed the id
__device__ __forceinline__ void k_stag_dedup_impl(const fid_stag_marker *__restrict__ cand, const int *__restrict__ found, const int *__restrict__ nquads,
                                                   fid_stag_marker *__restrict__ out, int *__restrict__ nout)
{
    if (blockIdx.x != 0) return;
    const int n = *nquads;
    const int lane = (int)threadIdx.x;
    // (round 6) by a wave: the flags and ids of 64 quads arrive with one load each, lane k keeps (id, distortion, source quad) of
    // output slot k, the decoded quads are taken in quad order as before -- one thread walked `found` one dependent load after the
    // other: 24 us for a frame's ~100 quads.  More than 64 distinct ids (never seen; the library has 12 .. 157): the loop below.
    {
        int nfound = 0;
        for (int q0 = 0; q0 < n; q0 += 64) nfound += (int)__builtin_popcountll(__ballot(q0 + lane < n && found[q0 + lane] != 0));
        if (nfound <= 64) {
            int m = 0, oid = -1, osrc = -1;
            double opd = 0;
            for (int q0 = 0; q0 < n; q0 += 64) {
                const int q = q0 + lane;
                const bool f = q < n && found[q] != 0;
                const int id = f ? cand[q].id : -1;
                const double pd = f ? cand[q].projectiveDistortion : 0.0;
                unsigned long long mask = __ballot(f);
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int idq = __builtin_amdgcn_readlane(id, l);
                    const double pdq = shfl_f64(pd, l);
                    const unsigned long long hit = __ballot(lane < m && oid == idq);
                    if (hit) {
                        if (lane < m && oid == idq && pdq < opd) {
                            opd = pdq;
                            osrc = q0 + l;
                        }
                    } else {
                        if (lane == m) {
                            oid = idq;
                            opd = pdq;
                            osrc = q0 + l;
                        }
                        m++;
                    }
                }
            }
            if (lane < m) out[lane] = cand[osrc];
            if (lane == 0) *nout = m;
            return;
        }
    }
    if (lane != 0) return;
    int m = 0;
    for (int q = 0; q < n; q++) {
        if (!found[q]) continue;
        bool notFound = true;
        for (int k = 0; k < m; k++) {
            if (out[k].id == cand[q].id) {
                notFound = false;
                if (cand[q].projectiveDistortion < out[k].projectiveDistortion) out[k] = cand[q];
            }
        }
        if (notFound) out[m++] = cand[q];
    }
    *nout = m;
}
__global__ __launch_bounds__(64) void k_stag_dedup(const fid_stag_marker *__restrict__ cand, const int *__restrict__ found, const int *__restrict__ nquads, fid_stag_marker *__restrict__ out, int *__restrict__ nout)
{
    k_stag_dedup_impl(cand, found, nquads, out, nout);
}
struct k_stag_dedup_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(const fid_stag_marker *__restrict__ cand, const int *__restrict__ found, const int *__restrict__ nquads, fid_stag_marker *__restrict__ out, int *__restrict__ nout) const { k_stag_dedup_impl(cand, found, nquads, out, nout); }
};
