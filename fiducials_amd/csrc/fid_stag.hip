// fid_stag.hip -- first kernels of the STag path: the EDPF edge-detection front end that
// Stag::detectMarkers -> QuadDetector::detectQuads -> EDInterface::runEDPFandEDLines -> DetectEdgesByEDPF
// (/root/reference/stag_detect/src/stag/ED/ED.cpp:144-187) runs before the sequential edge routing:
//
//   K9a k_stag_smooth_grad   SmoothImage(sigma = 1.0) = cv::GaussianBlur 5x5, sigma 0 (ImageSmooth.cpp:43-55) fused with
//                            ComputeGradientMapByPrewitt (GradientOperators.cpp:77-136): one LDS tile, 3-px halo
//   K9b k_stag_anchors       ComputeAnchorPoints (EDInternals.cpp:50-86) + the histogram of SortAnchorsByGradValue
//   K9c k_stag_bandsum, k_stag_bandscan, k_stag_scan, k_stag_place
//                            SortAnchorsByGradValue (EDInternals.cpp:146-186): counting sort by gradient value; inside a
//                            gradient value the reference's --C[grad] placement leaves the offsets in DESCENDING order.
//                            Counted per (row, value); slots handed out in that exact order, O(n), no sort
//
// Integer work throughout: results are bit-exact with the reference's own code (oracle/_ref, tests/test_gpu_stag.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fid_abi.h"

#define STAG_EDGE_VERTICAL 1
#define STAG_EDGE_HORIZONTAL 2
#define STAG_ANCHOR_PIXEL 254

__device__ __forceinline__ int stag_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// One workgroup = one SX x SY output tile.  src tile + 3 px halo (BORDER_REFLECT_101) -> LDS; horizontal [1 4 6 4 1] pass
// -> LDS u16; vertical pass + (x + 128) >> 8 -> smoothed tile + 1 px halo in LDS; Prewitt |gx| + |gy| and edge direction
// from LDS.  HBM traffic per pixel: 1 byte in (+ halo), 4 bytes out (smooth u8, grad i16, dir u8).
#define SX 64
#define SY 16
__global__ __launch_bounds__(256) void k_stag_smooth_grad(const uint8_t *__restrict__ src, int stride, int W, int H, int grad_thresh,
                                                           uint8_t *__restrict__ smooth, int16_t *__restrict__ grad,
                                                           uint8_t *__restrict__ dir)
{
    __shared__ uint8_t s_src[SY + 6][SX + 6 + 2];
    __shared__ uint16_t s_h[SY + 6][SX + 2];
    __shared__ uint8_t s_sm[SY + 2][SX + 2 + 2];
    const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
    const int tid = threadIdx.x;
    // source tile with halo 3
    for (int i = tid; i < (SY + 6) * (SX + 6); i += 256) {
        const int r = i / (SX + 6), c = i - r * (SX + 6);
        const int gy = stag_reflect101(y0 - 3 + r, H), gx = stag_reflect101(x0 - 3 + c, W);
        s_src[r][c] = src[(long long)gy * stride + gx];
    }
    __syncthreads();
    // horizontal pass for the smoothed region with halo 1: columns x0-1 .. x0+SX, all SY+6 rows
    for (int i = tid; i < (SY + 6) * (SX + 2); i += 256) {
        const int r = i / (SX + 2), c = i - r * (SX + 2);  // smoothed column x0 - 1 + c  <->  source column index c + 2
        const uint8_t *p = &s_src[r][c];
        s_h[r][c] = (uint16_t)(p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4]);
    }
    __syncthreads();
    // vertical pass: rows y0-1 .. y0+SY
    for (int i = tid; i < (SY + 2) * (SX + 2); i += 256) {
        const int r = i / (SX + 2), c = i - r * (SX + 2);  // smoothed row y0 - 1 + r  <->  s_h rows r .. r + 4
        const int acc = s_h[r][c] + 4 * s_h[r + 1][c] + 6 * s_h[r + 2][c] + 4 * s_h[r + 3][c] + s_h[r + 4][c];
        s_sm[r][c] = (uint8_t)((acc + 128) >> 8);
    }
    __syncthreads();
    // Prewitt on the smoothed image
    for (int i = tid; i < SY * SX; i += 256) {
        const int r = i / SX, c = i - r * SX;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        const long long idx = (long long)gy * W + gx;
        // NOTE: at image borders the halo of s_sm holds smoothed values of REFLECTED coordinates, which the reference never
        // reads: border pixels get the constant below
        smooth[idx] = s_sm[r + 1][c + 1];
        if (gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1) {
            grad[idx] = (int16_t)(grad_thresh - 1);
            dir[idx] = 0;  // (the reference leaves these unwritten)
            continue;
        }
        const int A = s_sm[r][c], B = s_sm[r][c + 1], C = s_sm[r][c + 2];
        const int D = s_sm[r + 1][c], E = s_sm[r + 1][c + 2];
        const int F = s_sm[r + 2][c], G = s_sm[r + 2][c + 1], Hh = s_sm[r + 2][c + 2];
        const int com1 = Hh - A, com2 = C - F;
        int gxv = com1 + com2 + (E - D), gyv = com1 - com2 + (G - B);
        gxv = gxv < 0 ? -gxv : gxv;
        gyv = gyv < 0 ? -gyv : gyv;
        const int sum = gxv + gyv;
        grad[idx] = (int16_t)sum;
        dir[idx] = sum >= grad_thresh ? (gxv >= gyv ? STAG_EDGE_VERTICAL : STAG_EDGE_HORIZONTAL) : 0;
    }
}

// The smoothed image is 8-bit, so |gx|, |gy| <= 3 * 255 and the gradient value never exceeds 1530: the reference's
// 128 * 256 counting-sort bins (SIZE in SortAnchorsByGradValue) are used only below STAG_BINS.
#define STAG_BINS 1536
#define STAG_BAND_ROWS 8  // rows per band of k_stag_place = waves per workgroup

// Anchor points: local gradient maxima across the edge normal (ANCHOR_THRESH, SCAN_INTERVAL as in the reference), counted
// per (row, gradient value) for the counting sort (global atomics: the anchors are sparse).
__global__ __launch_bounds__(256) void k_stag_anchors(const int16_t *__restrict__ grad, const uint8_t *__restrict__ dir, int W, int H,
                                                       int grad_thresh, int anchor_thresh, int scan_interval,
                                                       uint8_t *__restrict__ edge, unsigned *__restrict__ rowhist)
{
    const long long total = (long long)W * H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx / W), j = (int)(idx - (long long)i * W);
        uint8_t e = 0;
        if (i >= 2 && i < H - 2 && j >= 2 && j < W - 2) {
            // rows that are not a multiple of SCAN_INTERVAL are scanned at columns SCAN_INTERVAL, 2 SCAN_INTERVAL, ...
            const bool scanned = (i % scan_interval == 0) || (j >= scan_interval && j % scan_interval == 0);
            const int g = grad[idx];
            if (scanned && g >= grad_thresh) {
                int d1, d2;
                if (dir[idx] == STAG_EDGE_VERTICAL) {
                    d1 = g - grad[idx - 1];
                    d2 = g - grad[idx + 1];
                } else {
                    d1 = g - grad[idx - W];
                    d2 = g - grad[idx + W];
                }
                if (d1 >= anchor_thresh && d2 >= anchor_thresh) {
                    e = STAG_ANCHOR_PIXEL;
                    // SortAnchorsByGradValue only counts anchors with 1 <= i < H-1, 1 <= j < W-1: all of these qualify
                    atomicAdd(&rowhist[(size_t)i * STAG_BINS + g], 1u);
                }
            }
        }
        edge[idx] = e;
    }
}

// anchors per (band of STAG_BAND_ROWS rows, gradient value)
__global__ __launch_bounds__(256) void k_stag_bandsum(const unsigned *__restrict__ rowhist, int H, unsigned *__restrict__ bandhist)
{
    const int g = blockIdx.x * 256 + threadIdx.x, band = blockIdx.y;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < STAG_BAND_ROWS; r++) {
        const int row = band * STAG_BAND_ROWS + r;
        if (row < H) acc += rowhist[(size_t)row * STAG_BINS + g];
    }
    bandhist[(size_t)band * STAG_BINS + g] = acc;
}

// per gradient value: bands from the LAST to the first (the reference's --C[grad] placement leaves the offsets of one
// gradient value in descending order) -> start of each band inside the value's bucket; tot[g] = size of the bucket
__global__ __launch_bounds__(256) void k_stag_bandscan(unsigned *__restrict__ bandhist, int nbands, unsigned *__restrict__ tot)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    int b = nbands - 1;
    for (; b >= 3; b -= 4) {  // four independent loads in flight
        unsigned n[4];
#pragma unroll
        for (int k = 0; k < 4; k++) n[k] = bandhist[(size_t)(b - k) * STAG_BINS + g];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bandhist[(size_t)(b - k) * STAG_BINS + g] = acc;
            acc += n[k];
        }
    }
    for (; b >= 0; b--) {
        const unsigned n = bandhist[(size_t)b * STAG_BINS + g];
        bandhist[(size_t)b * STAG_BINS + g] = acc;
        acc += n;
    }
    tot[g] = acc;
}

// exclusive prefix sums over the gradient values (one workgroup): bstart[g] = number of anchors with a smaller gradient
__global__ __launch_bounds__(512) void k_stag_scan(const unsigned *__restrict__ tot, unsigned *__restrict__ bstart, unsigned *__restrict__ n_anchors)
{
    __shared__ unsigned s_part[512];
    const int tid = threadIdx.x;
    constexpr int PER = STAG_BINS / 512;
    unsigned loc[PER];
    unsigned acc = 0;
    for (int k = 0; k < PER; k++) {
        loc[k] = acc;
        acc += tot[tid * PER + k];
    }
    s_part[tid] = acc;
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {
        unsigned v = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const unsigned base = tid ? s_part[tid - 1] : 0u;
    for (int k = 0; k < PER; k++) bstart[tid * PER + k] = base + loc[k];
    if (tid == 511) *n_anchors = s_part[511];
}

// Placement: one workgroup per band, one wave per row.  LDS holds, per (row of the band, gradient value), the next free
// slot: bucket start + band start + anchors of that value in the LATER rows of the band.  Every wave then goes through
// its row from the last column to the first, 64 columns at a time; lanes with the same gradient value take consecutive
// slots in descending column order.  No sort, no atomics: the order is exactly the reference's.
__global__ __launch_bounds__(64 * STAG_BAND_ROWS) void k_stag_place(const int16_t *__restrict__ grad, const uint8_t *__restrict__ edge, int W, int H,
                                                                    const unsigned *__restrict__ rowhist, const unsigned *__restrict__ bandstart,
                                                                    const unsigned *__restrict__ bstart, int32_t *__restrict__ sorted)
{
    __shared__ unsigned s_slot[STAG_BAND_ROWS][STAG_BINS];
    const int band = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int g = tid; g < STAG_BINS; g += 64 * STAG_BAND_ROWS) {
        unsigned n[STAG_BAND_ROWS];
#pragma unroll
        for (int r = 0; r < STAG_BAND_ROWS; r++) {
            const int row = band * STAG_BAND_ROWS + r;
            n[r] = row < H ? rowhist[(size_t)row * STAG_BINS + g] : 0u;
        }
        unsigned acc = bstart[g] + bandstart[(size_t)band * STAG_BINS + g];
#pragma unroll
        for (int r = STAG_BAND_ROWS - 1; r >= 0; r--) {
            s_slot[r][g] = acc;
            acc += n[r];
        }
    }
    __syncthreads();
    const int row = band * STAG_BAND_ROWS + w;
    if (row >= H) return;
    const long long rbase = (long long)row * W;
    const unsigned long long above = lane == 63 ? 0ull : ~0ull << (lane + 1);
    unsigned *slot = s_slot[w];
    for (int c4 = ((W - 1) >> 8) << 8; c4 >= 0; c4 -= 256) {
        uint8_t e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int col = c4 + k * 64 + lane;
            e[k] = col < W ? edge[rbase + col] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const int col = c4 + k * 64 + lane;
            const bool is_anchor = e[k] == STAG_ANCHOR_PIXEL;
            unsigned long long pending = __ballot(is_anchor);
            if (!pending) continue;
            const int gv = is_anchor ? (int)grad[rbase + col] : -1;
            while (pending) {
                const int src = __builtin_ctzll(pending);
                const int g0 = __builtin_amdgcn_readlane(gv, src);
                const unsigned long long m = __ballot(gv == g0);
                const unsigned base = slot[g0];
                if (gv == g0) sorted[base + __builtin_popcountll(m & above)] = (int32_t)(rbase + col);
                if (lane == src) slot[g0] = base + (unsigned)__builtin_popcountll(m);
                pending &= ~m;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ K10: edge routing
// JoinAnchorPointsUsingSortedAnchors (EDInternals.cpp:842-1448): from every anchor that is still an anchor, strongest
// first, draw the edge through the gradient ridge in both directions; where the edge orientation flips, branch; keep the
// chain tree, emit its longest path as one segment and every remaining path of >= 10 pixels as further segments.  Every
// decision reads what earlier walks left in the edge image, so the order is part of the result: this first version keeps
// the reference's order by running ONE lane per frame (exact, slow); the walks of different connected components of
// {grad >= GRADIENT_THRESH} never meet, which is the parallelism the next version uses.
// The restatement is table-driven (one body for LEFT / RIGHT / UP / DOWN) and keeps the reference's array semantics where
// they are visible in the result: 16-bit chain fields, the scratch pixel array shared by all chains of one anchor, the
// contiguous output pixel array (a segment may look at the last pixel of the segment before it).
#define STAG_EDGE_PIXEL 255
#define STAG_MIN_PATH_LEN 10  // DoDetectEdgesByED, EDInternals.cpp:2604
enum { SR_LEFT = 0, SR_RIGHT = 1, SR_UP = 2, SR_DOWN = 3 };

struct StagChain {
    int16_t dir;
    uint16_t len;
    int16_t parent;
    int16_t child[2];
    int32_t pix;  // first pixel of the chain in the scratch pixel array
};

struct StagRoute {
    const int16_t *grad;
    const uint8_t *dir;
    uint8_t *edge;
    int W, H;
    int2 *pix;        // scratch: pixels of the chains of the current anchor (x = row, y = column)
    int4 *stack;      // scratch: pending branches (r, c, dir, parent); reused by the tree walk of longest()
    StagChain *chains;
    int *chainNos;
    int capPix, capStack, capChains, capNos;
    int2 *outpix;     // map->pixels
    int2 *segs;       // (first pixel, number of pixels) per segment
    int capOut, capSegs;
    int *counters;    // [0] segments [1] pixels used in outpix [2] overflow flags
};

__device__ __forceinline__ bool sr_near(int2 a, int2 b)
{
    int dr = a.x - b.x, dc = a.y - b.y;
    dr = dr < 0 ? -dr : dr;
    dc = dc < 0 ? -dc : dc;
    return dr <= 1 && dc <= 1;
}

struct StagRouter {
    StagRoute R;
    int noSegments, totalPixels, overflow;
    int segbase, nsp;  // the segment being assembled: first pixel in outpix, pixels so far
    // component-parallel routing: outpix is the component's own arena and "the pixel in front of the block" is the last pixel
    // of the block the reference would have written just before this one -- known (and relevant) only if that block came from
    // the same component
    bool par = false, prev_valid = false;
    int blk0 = 0;  // where the current anchor's block starts in outpix
    int wlane = -1;  // >= 0: a whole wave runs the extraction (identical scalar work in every lane, copies spread over the lanes)
    int wl_len = 0, wl_dup = 0, wl_chains = 0;  // what walk_anchor() left behind

    __device__ int2 cpx(int ch, int i) const
    {
        const int k = R.chains[ch].pix + i;
        return k >= 0 ? R.pix[k] : make_int2(-1000, -1000);  // (the reference reads in front of its array there)
    }
    __device__ int2 seg(int i) const
    {
        const int k = segbase + i;
        if (par && k < blk0) return (prev_valid && k >= 0) ? R.outpix[k] : make_int2(-1000, -1000);  // in front of this anchor's block
        return k >= 0 ? R.outpix[k] : make_int2(-1000, -1000);
    }
    // append `count` pixels of chain cn, chain index first + step * k, to the segment
    __device__ void seg_copy(int cn, int first, int step, int count)
    {
        if (count <= 0) return;
        if (segbase + nsp + count > R.capOut) {
            overflow |= 1;
            nsp += count;
            return;
        }
        const int2 *src = R.pix + R.chains[cn].pix;
        int2 *dst = R.outpix + segbase + nsp;
        if (wlane >= 0) {
            for (int k = wlane; k < count; k += 64) dst[k] = src[first + step * k];
        } else {
            for (int k = 0; k < count; k++) dst[k] = src[first + step * k];
        }
        nsp += count;
    }
    __device__ void seg_put(int2 v)
    {
        const int k = segbase + nsp;
        if (k < R.capOut) R.outpix[k] = v;
        else overflow |= 1;
        nsp++;
    }
    // LongestChain (EDInternals.cpp:191-214): length of the longest root-to-leaf path; prunes the shorter child of every
    // chain it visits.  Chains of length 0 end the descent.
    __device__ int longest(int root)
    {
        StagChain *ch = R.chains;
        if (root == -1 || ch[root].len == 0) return 0;
        int sp = 0, ret = 0;
        R.stack[sp++] = make_int4(root, 0, 0, 0);
        while (sp > 0) {
            int4 e = R.stack[sp - 1];
            const int node = e.x;
            if (e.y == 0) {
                e.y = 1;
                R.stack[sp - 1] = e;
                const int c = ch[node].child[0];
                if (c != -1 && ch[c].len != 0) {
                    if (sp < R.capStack) R.stack[sp++] = make_int4(c, 0, 0, 0);
                    else { overflow |= 2; ret = 0; }
                    if (!(overflow & 2)) continue;
                }
                ret = 0;
            }
            if (e.y == 1) {
                e.z = ret;
                e.y = 2;
                R.stack[sp - 1] = e;
                const int c = ch[node].child[1];
                if (c != -1 && ch[c].len != 0) {
                    if (sp < R.capStack) { R.stack[sp++] = make_int4(c, 0, 0, 0); continue; }
                    overflow |= 2;
                }
                ret = 0;
            }
            const int len0 = e.z, len1 = ret;
            int mx;
            if (len0 >= len1) {
                mx = len0;
                ch[node].child[1] = -1;
            } else {
                mx = len1;
                ch[node].child[0] = -1;
            }
            ret = ch[node].len + mx;
            sp--;
        }
        return ret;
    }
    // RetrieveChainNos (EDInternals.cpp:219-234)
    __device__ int retrieve(int root)
    {
        int count = 0;
        while (root != -1) {
            if (count < R.capNos) R.chainNos[count] = root;
            else { overflow |= 4; break; }
            count++;
            root = R.chains[root].child[0] != -1 ? R.chains[root].child[0] : R.chains[root].child[1];
        }
        return count;
    }
    // drop pixels at the end of the segment that touch the pixel the next chain starts with
    __device__ void trim_tail(int2 f)
    {
        int index = nsp - 2;
        while (index >= 0) {
            if (!sr_near(f, seg(index))) break;
            nsp--;
            index--;
        }
    }
    __device__ void append_forward(int count)
    {
        StagChain *ch = R.chains;
        for (int k = 0; k < count; k++) {
            const int cn = R.chainNos[k];
            trim_tail(cpx(cn, 0));
            int start = 0;
            const int L = ch[cn].len;
            if (L > 1 && sr_near(cpx(cn, 1), seg(nsp - 1))) start = 1;
            seg_copy(cn, start, 1, L - start);
            ch[cn].len = 0;  // copied
        }
    }
    __device__ void close_segment(bool clean_first)
    {
        int first = segbase, n = nsp;
        totalPixels += nsp;
        if (clean_first && sr_near(seg(1), seg(nsp - 1))) {
            first++;
            n--;
        }
        if (noSegments < R.capSegs) R.segs[noSegments] = make_int2(first, n);
        else overflow |= 8;
        noSegments++;
    }

    __device__ void route_anchor(int r0, int c0, int grad_thresh)
    {
        if (walk_anchor(r0, c0, grad_thresh)) extract_anchor(wl_chains);
    }

    // the walk: true if the anchor produced a path that is kept (the chain tree is then in R.chains / R.pix)
    __device__ bool walk_anchor(int r0, int c0, int grad_thresh)
    {
        const int W = R.W;
        StagChain *ch = R.chains;
        ch[0].dir = 0; ch[0].len = 0; ch[0].parent = -1; ch[0].child[0] = ch[0].child[1] = -1; ch[0].pix = -1;
        int noChains = 1, len = 0, dup = 0, top = -1;
        if (R.dir[r0 * W + c0] == STAG_EDGE_VERTICAL) {
            R.stack[++top] = make_int4(r0, c0, SR_DOWN, 0);
            R.stack[++top] = make_int4(r0, c0, SR_UP, 0);
        } else {
            R.stack[++top] = make_int4(r0, c0, SR_RIGHT, 0);
            R.stack[++top] = make_int4(r0, c0, SR_LEFT, 0);
        }
        while (top >= 0) {
            const int4 e = R.stack[top--];
            int r = e.x, c = e.y;
            const int d = e.z, parent = e.w;
            if (noChains >= R.capChains || len + 2 >= R.capPix || top + 3 >= R.capStack) {
                overflow |= 16;
                break;
            }
            if (R.edge[r * W + c] != STAG_EDGE_PIXEL) dup++;
            const int cur = noChains;
            ch[cur].dir = (int16_t)d; ch[cur].parent = (int16_t)parent; ch[cur].child[0] = ch[cur].child[1] = -1; ch[cur].pix = len;
            int chainLen = 0;
            R.pix[len++] = make_int2(r, c);
            chainLen++;
            const bool horiz = d == SR_LEFT || d == SR_RIGHT;
            const int need = horiz ? STAG_EDGE_HORIZONTAL : STAG_EDGE_VERTICAL;
            const int ar = d == SR_UP ? -1 : d == SR_DOWN ? 1 : 0, ac = d == SR_LEFT ? -1 : d == SR_RIGHT ? 1 : 0;
            const int pr = horiz ? 1 : 0, pc = horiz ? 0 : 1;            // across the walking direction
            const int fs = (d == SR_LEFT || d == SR_UP) ? -1 : 1;         // which diagonal is looked at first
            const int slot = (d == SR_LEFT || d == SR_UP) ? 0 : 1;
            bool stopped = false;
            while (R.dir[r * W + c] == need) {
                R.edge[r * W + c] = STAG_EDGE_PIXEL;
                uint8_t *s1 = R.edge + (r + pr) * W + (c + pc), *s2 = R.edge + (r - pr) * W + (c - pc);
                if (*s1 == STAG_ANCHOR_PIXEL) *s1 = 0;
                if (*s2 == STAG_ANCHOR_PIXEL) *s2 = 0;
                const int nr = r + ar, nc = c + ac;
                if (R.edge[nr * W + nc] >= STAG_ANCHOR_PIXEL) {
                    r = nr; c = nc;
                } else if (R.edge[(nr + fs * pr) * W + nc + fs * pc] >= STAG_ANCHOR_PIXEL) {
                    r = nr + fs * pr; c = nc + fs * pc;
                } else if (R.edge[(nr - fs * pr) * W + nc - fs * pc] >= STAG_ANCHOR_PIXEL) {
                    r = nr - fs * pr; c = nc - fs * pc;
                } else {
                    const int A = R.grad[(nr - pr) * W + nc - pc], B = R.grad[nr * W + nc], Cg = R.grad[(nr + pr) * W + nc + pc];
                    int side = 0;
                    if (A > B) side = A > Cg ? -1 : 1;
                    else if (Cg > B) side = 1;
                    r = nr + side * pr; c = nc + side * pc;
                }
                if (R.edge[r * W + c] == STAG_EDGE_PIXEL || R.grad[r * W + c] < grad_thresh) {
                    ch[cur].len = (uint16_t)chainLen;
                    ch[parent].child[slot] = (int16_t)cur;
                    noChains++;
                    stopped = true;
                    break;
                }
                if (len + 2 >= R.capPix) { overflow |= 16; stopped = true; break; }
                R.pix[len++] = make_int2(r, c);
                chainLen++;
            }
            if (stopped) continue;
            // the edge turns here: branch both ways across, this chain ends in front of the turning pixel
            R.stack[++top] = make_int4(r, c, horiz ? SR_DOWN : SR_RIGHT, cur);
            R.stack[++top] = make_int4(r, c, horiz ? SR_UP : SR_LEFT, cur);
            len--;
            chainLen--;
            ch[cur].len = (uint16_t)chainLen;
            ch[parent].child[slot] = (int16_t)cur;
            noChains++;
        }
        wl_len = len;
        wl_dup = dup;
        wl_chains = noChains;
        if (len - dup < STAG_MIN_PATH_LEN) {
            for (int k = 0; k < len; k++) R.edge[R.pix[k].x * W + R.pix[k].y] = 0;
            return false;
        }
        return true;
    }

    // The same walk run by a whole wave: control flow and bookkeeping are wave-uniform (every lane computes them, lane 0
    // stores them), and what a step needs from memory -- edge / gradient / direction of the three pixels ahead and the edge
    // value of the two pixels beside -- is fetched by eleven lanes at once, one round trip per step instead of a chain of
    // dependent loads.  None of those eleven pixels is written in the same step (the current pixel and the two beside it
    // are not among the three ahead), so the fetch sees exactly what the sequential code would read.
    __device__ bool walk_anchor_wave(int r0, int c0, int grad_thresh, int lane)
    {
        const int W = R.W;
        const bool L0 = lane == 0;
        StagChain *ch = R.chains;
        if (L0) {
            ch[0].dir = 0; ch[0].len = 0; ch[0].parent = -1; ch[0].child[0] = ch[0].child[1] = -1; ch[0].pix = -1;
        }
        int noChains = 1, len = 0, dup = 0, top = -1;
        const bool vert0 = R.dir[r0 * W + c0] == STAG_EDGE_VERTICAL;
        if (L0) {
            R.stack[0] = make_int4(r0, c0, vert0 ? SR_DOWN : SR_RIGHT, 0);
            R.stack[1] = make_int4(r0, c0, vert0 ? SR_UP : SR_LEFT, 0);
        }
        top = 1;
        while (top >= 0) {
            const int4 e = R.stack[top--];
            int r = e.x, c = e.y;
            const int d = e.z, parent = e.w;
            if (noChains >= R.capChains || len + 2 >= R.capPix || top + 3 >= R.capStack) {
                overflow |= 16;
                break;
            }
            if (R.edge[r * W + c] != STAG_EDGE_PIXEL) dup++;
            const int cur = noChains;
            if (L0) {
                ch[cur].dir = (int16_t)d; ch[cur].parent = (int16_t)parent; ch[cur].child[0] = ch[cur].child[1] = -1; ch[cur].pix = len;
                R.pix[len] = make_int2(r, c);
            }
            len++;
            int chainLen = 1;
            const bool horiz = d == SR_LEFT || d == SR_RIGHT;
            const int need = horiz ? STAG_EDGE_HORIZONTAL : STAG_EDGE_VERTICAL;
            const int ar = d == SR_UP ? -1 : d == SR_DOWN ? 1 : 0, ac = d == SR_LEFT ? -1 : d == SR_RIGHT ? 1 : 0;
            const int pr = horiz ? 1 : 0, pc = horiz ? 0 : 1;
            const int fs = (d == SR_LEFT || d == SR_UP) ? -1 : 1;
            const int slot = (d == SR_LEFT || d == SR_UP) ? 0 : 1;
            bool stopped = false;
            int curdir = R.dir[r * W + c];
            while (curdir == need) {
                const int nr = r + ar, nc = c + ac;
                // lane -> (array, pixel): 0-2 edge, 3-5 grad, 6-8 dir of A = ahead - p, B = ahead, C = ahead + p; 9, 10 edge beside
                int v = 0;
                {
                    const int k = lane % 3, side = k - 1;  // A, B, C
                    const int qr = nr + side * pr, qc = nc + side * pc;
                    const int q = qr * W + qc;
                    if (lane < 3) v = R.edge[q];
                    else if (lane < 6) v = R.grad[q];
                    else if (lane < 9) v = R.dir[q];
                    else if (lane == 9) v = R.edge[(r + pr) * W + (c + pc)];
                    else if (lane == 10) v = R.edge[(r - pr) * W + (c - pc)];
                }
                const int eA = __builtin_amdgcn_readlane(v, 0), eB = __builtin_amdgcn_readlane(v, 1), eC = __builtin_amdgcn_readlane(v, 2);
                const int gA = __builtin_amdgcn_readlane(v, 3), gB = __builtin_amdgcn_readlane(v, 4), gC = __builtin_amdgcn_readlane(v, 5);
                const int dA = __builtin_amdgcn_readlane(v, 6), dB = __builtin_amdgcn_readlane(v, 7), dC = __builtin_amdgcn_readlane(v, 8);
                const int s1 = __builtin_amdgcn_readlane(v, 9), s2 = __builtin_amdgcn_readlane(v, 10);
                if (L0) {
                    R.edge[r * W + c] = STAG_EDGE_PIXEL;
                    if (s1 == STAG_ANCHOR_PIXEL) R.edge[(r + pr) * W + (c + pc)] = 0;
                    if (s2 == STAG_ANCHOR_PIXEL) R.edge[(r - pr) * W + (c - pc)] = 0;
                }
                const int eF1 = fs < 0 ? eA : eC, eF2 = fs < 0 ? eC : eA;  // the diagonal looked at first / second
                int side;
                if (eB >= STAG_ANCHOR_PIXEL) side = 0;
                else if (eF1 >= STAG_ANCHOR_PIXEL) side = fs;
                else if (eF2 >= STAG_ANCHOR_PIXEL) side = -fs;
                else {
                    side = 0;
                    if (gA > gB) side = gA > gC ? -1 : 1;
                    else if (gC > gB) side = 1;
                }
                r = nr + side * pr;
                c = nc + side * pc;
                const int en = side < 0 ? eA : side > 0 ? eC : eB, gn = side < 0 ? gA : side > 0 ? gC : gB;
                curdir = side < 0 ? dA : side > 0 ? dC : dB;
                if (en == STAG_EDGE_PIXEL || gn < grad_thresh) {
                    if (L0) {
                        ch[cur].len = (uint16_t)chainLen;
                        ch[parent].child[slot] = (int16_t)cur;
                    }
                    noChains++;
                    stopped = true;
                    break;
                }
                if (len + 2 >= R.capPix) { overflow |= 16; stopped = true; break; }
                if (L0) R.pix[len] = make_int2(r, c);
                len++;
                chainLen++;
            }
            if (stopped) continue;
            if (L0) {
                R.stack[top + 1] = make_int4(r, c, horiz ? SR_DOWN : SR_RIGHT, cur);
                R.stack[top + 2] = make_int4(r, c, horiz ? SR_UP : SR_LEFT, cur);
            }
            top += 2;
            len--;
            chainLen--;
            if (L0) {
                ch[cur].len = (uint16_t)chainLen;
                ch[parent].child[slot] = (int16_t)cur;
            }
            noChains++;
        }
        wl_len = len;
        wl_dup = dup;
        wl_chains = noChains;
        if (len - dup < STAG_MIN_PATH_LEN) {
            for (int k = lane; k < len; k += 64) R.edge[R.pix[k].x * W + R.pix[k].y] = 0;
            return false;
        }
        return true;
    }

    // the chain tree -> segments
    __device__ void extract_anchor(int noChains)
    {
        StagChain *ch = R.chains;
        blk0 = totalPixels;
        segbase = totalPixels;
        nsp = 0;
        int totalLen = longest(ch[0].child[1]);
        if (totalLen > 0) {  // the path behind the anchor, copied backwards so that the segment runs through the anchor
            const int count = retrieve(ch[0].child[1]);
            for (int k = count - 1; k >= 0; k--) {
                const int cn = R.chainNos[k];
                trim_tail(cpx(cn, ch[cn].len - 1));
                if (ch[cn].len > 1 && sr_near(cpx(cn, ch[cn].len - 2), seg(nsp - 1))) ch[cn].len--;
                seg_copy(cn, ch[cn].len - 1, -1, ch[cn].len);
                ch[cn].len = 0;
            }
        }
        totalLen = longest(ch[0].child[0]);
        if (totalLen > 1) {
            const int count = retrieve(ch[0].child[0]);
            const int first = R.chainNos[0];  // its first pixel is the anchor again
            ch[first].pix++;
            ch[first].len--;
            append_forward(count);
        }
        close_segment(true);
        for (int k = 2; k < noChains; k++) {  // what is left of the tree
            if (ch[k].len < 2) continue;
            totalLen = longest(k);
            if (totalLen >= 10) {
                segbase = totalPixels;
                nsp = 0;
                append_forward(retrieve(k));
                close_segment(false);
            }
        }
    }
};

__global__ __launch_bounds__(64) void k_stag_route_seq(StagRoute R, const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors,
                                                       int grad_thresh)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    StagRouter S;
    S.R = R;
    S.noSegments = S.totalPixels = S.overflow = 0;
    S.segbase = S.nsp = 0;
    const int n = (int)*n_anchors;
    for (int k = n - 1; k >= 0; k--) {
        const int off = sorted[k];
        if (R.edge[off] != STAG_ANCHOR_PIXEL) continue;
        S.route_anchor(off / R.W, off % R.W, grad_thresh);
        if (S.overflow) break;
    }
    R.counters[0] = S.noSegments;
    R.counters[1] = S.totalPixels;
    R.counters[2] = S.overflow;
}

// ---- component-parallel routing ---------------------------------------------------------------------------------
// A walk only ever stands on pixels with grad >= GRADIENT_THRESH (it stops in front of anything weaker) and only touches the
// edge image at those pixels and their walked neighbours' cross pixels, which are anchors, hence also >= the threshold: walks
// of different 8-connected components of {grad >= GRADIENT_THRESH} never read or write the same pixel.  So every component
// can process ITS anchors, strongest first, on its own -- the edge image and every chain tree come out as in the reference's
// single sequential loop.  What remains global is the ORDER of the output (segments are listed in the order their anchors
// were processed) and one quirk: when a block of segments starts, the reference peeks at the pixel in front of it in the
// contiguous pixel array, i.e. at the last pixel of the block before -- which can only matter (8-adjacency) if that block
// belongs to the same component.  Hence two passes:
//   k_stag_ccl_*          connected components by union-find with atomic hooking (labels = smallest pixel offset)
//   k_stag_comp_*         per component: pixels, anchors -> arenas (scratch pixels, stack, chains, output) by atomic cursors;
//                         its anchors gathered and sorted by rank (bitonic, one wave per component)
//   k_stag_route_walk     one LANE per component: the walks; chain trees of producing anchors stay in the arenas
//   k_stag_next_above     for every anchor rank, the nearest producing rank above it (decides the quirk)
//   k_stag_route_extract  one lane per component: chain trees -> blocks of segments in the component's output arena
//   k_stag_route_gather   blocks -> EdgeMap::pixels / segments in global anchor order (offsets from two scans)
struct StagComp {
    int root, size, nanch;
    int anch_base, anch_cap;      // slice of the anchor-rank array (padded to a power of two for the sort)
    int pix_base, pix_cap;        // scratch pixels (chain trees of the producing anchors are kept)
    int stack_base, stack_cap;
    int chain_base, chain_cap;
    int out_base, out_cap;        // output pixels of this component's blocks; chainNos live in the stack arena's tail
    int seg_base, seg_cap;
    int nrec;                     // producing anchors
};

struct StagRec {  // one producing anchor
    int rank;
    int pix_off, len;       // its chain-tree pixels inside the component's scratch arena
    int chain_off, nchains;
    int out_off, out_len;   // its block inside the component's output arena
    int seg_off, nsegs;     // its segments inside the component's segment arena
};

__device__ __forceinline__ int ccl_find(const int *L, int a)
{
    while (true) {
        const int p = L[a];
        if (p == a) return a;
        a = p;
    }
}

__global__ __launch_bounds__(256) void k_stag_ccl_init(const int16_t *__restrict__ grad, int n, int thresh, int *__restrict__ label)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) label[i] = grad[i] >= thresh ? i : -1;
}

__global__ __launch_bounds__(256) void k_stag_ccl_merge(int W, int H, int *label)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * H || label[i] < 0) return;
    const int r = i / W, c = i - r * W;
    // the four neighbours that precede the pixel in raster order (border pixels are background: grad = thresh - 1)
    const int nb[4] = {c > 0 ? i - 1 : -1, (r > 0 && c > 0) ? i - W - 1 : -1, r > 0 ? i - W : -1, (r > 0 && c < W - 1) ? i - W + 1 : -1};
    for (int k = 0; k < 4; k++) {
        int b = nb[k];
        if (b < 0 || label[b] < 0) continue;
        int a = i;
        while (true) {
            a = ccl_find(label, a);
            b = ccl_find(label, b);
            if (a == b) break;
            if (a < b) {
                const int t = a;
                a = b;
                b = t;
            }
            const int old = atomicMin(&label[a], b);  // hook the larger root under the smaller one
            if (old == a) break;
            a = old;
        }
    }
}

__global__ __launch_bounds__(256) void k_stag_ccl_flatten(int n, int *label, const uint8_t *__restrict__ anchors, int *__restrict__ csize,
                                                          int *__restrict__ canch)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || label[i] < 0) return;
    const int root = ccl_find(label, i);
    label[i] = root;
    atomicAdd(&csize[root], 1);
    if (anchors[i] == STAG_ANCHOR_PIXEL) atomicAdd(&canch[root], 1);
}

// cursors: [0] components [1] anchor slots [2] scratch pixels [3] stack [4] chains [5] output pixels [6] segments [7] overflow
//          [8] overflow flags of the routing kernels [9] most anchors in one component
__global__ __launch_bounds__(256) void k_stag_comp_alloc(int n, const int *__restrict__ label, const int *__restrict__ csize,
                                                         const int *__restrict__ canch, int *__restrict__ cursors, int max_comps, const int *caps,
                                                         StagComp *__restrict__ comps, int *__restrict__ cidmap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || label[i] != i) return;
    cidmap[i] = -1;
    const int na = canch[i], sz = csize[i];
    if (na == 0) return;
    atomicMax(&cursors[9], na);
    const int cid = atomicAdd(&cursors[0], 1);
    if (cid >= max_comps) {
        atomicOr(&cursors[7], 1);
        return;
    }
    StagComp C;
    C.root = i; C.size = sz; C.nanch = na; C.nrec = 0;
    int p2 = 1;
    while (p2 < na) p2 <<= 1;
    C.anch_cap = p2;
    C.pix_cap = 2 * sz + 12 * na + 64;
    C.stack_cap = (sz + 2 * na + 64) + (sz / 4 + 64);  // pending branches + (in its tail) the chain lists of the extraction
    C.chain_cap = sz + 2 * na + 64;
    C.out_cap = C.pix_cap;
    C.seg_cap = C.pix_cap / 8 + na + 8;
    C.anch_base = atomicAdd(&cursors[1], C.anch_cap);
    C.pix_base = atomicAdd(&cursors[2], C.pix_cap);
    C.stack_base = atomicAdd(&cursors[3], C.stack_cap);
    C.chain_base = atomicAdd(&cursors[4], C.chain_cap);
    C.out_base = atomicAdd(&cursors[5], C.out_cap);
    C.seg_base = atomicAdd(&cursors[6], C.seg_cap);
    if (C.anch_base + C.anch_cap > caps[1] || C.pix_base + C.pix_cap > caps[2] || C.stack_base + C.stack_cap > caps[3] ||
        C.chain_base + C.chain_cap > caps[4] || C.out_base + C.out_cap > caps[5] || C.seg_base + C.seg_cap > caps[6]) {
        atomicOr(&cursors[7], 2);
        C.nanch = 0;  // not processed; the call reports FID_E_CAPACITY
    }
    comps[cid] = C;
    cidmap[i] = cid;
}

__global__ __launch_bounds__(256) void k_stag_comp_fill(const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors,
                                                        const int *__restrict__ label, const int *__restrict__ cidmap, const StagComp *__restrict__ comps,
                                                        int *__restrict__ fill, int *__restrict__ aslots)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= (int)*n_anchors) return;
    const int cid = cidmap[label[sorted[r]]];
    if (cid < 0 || comps[cid].nanch == 0) return;
    const int pos = atomicAdd(&fill[cid], 1);
    aslots[comps[cid].anch_base + pos] = r;
}

// ranks of one component, descending: bitonic sort of the (padded, -1 filled) slice, one wave per component; slices of up to
// 2048 entries are sorted in LDS
#define STAG_SORT_LDS 2048
__global__ __launch_bounds__(256) void k_stag_comp_sort(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots)
{
    __shared__ int s_buf[4][STAG_SORT_LDS];
    const int wv = threadIdx.x >> 6, cid = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (cid >= cursors[0]) return;
    const StagComp C = comps[cid];
    if (C.nanch < 2) return;
    int *g = aslots + C.anch_base;
    const int P = C.anch_cap;
    const bool in_lds = P <= STAG_SORT_LDS;
    int *a = in_lds ? s_buf[wv] : g;
    if (in_lds) {
        for (int i = lane; i < P; i += 64) a[i] = g[i];
        __builtin_amdgcn_wave_barrier();
    }
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const int x = a[i], y = a[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? x < y : x > y) {
                        a[i] = y;
                        a[l] = x;
                    }
                }
            }
            if (in_lds) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
    }
    if (in_lds)
        for (int i = lane; i < P; i += 64) g[i] = a[i];
}

struct StagArenas {
    int2 *pix;
    int4 *stack;
    StagChain *chains;
    int2 *out;
    int2 *segs;
    StagRec *recs;  // indexed like the anchor slots
};

__device__ void stag_bind(StagRouter &S, const StagRoute &G, const StagArenas &A, const StagComp &C)
{
    S.R = G;
    S.R.pix = A.pix + C.pix_base;
    S.R.capPix = C.pix_cap;
    S.R.stack = A.stack + C.stack_base;
    S.R.capStack = C.stack_cap - (C.size / 4 + 64);
    S.R.chainNos = (int *)(A.stack + C.stack_base + S.R.capStack);  // int view of the arena's tail: 4 ints per entry
    S.R.capNos = (C.size / 4 + 64) * 4;
    S.R.chains = A.chains + C.chain_base;
    S.R.capChains = C.chain_cap < 32767 ? C.chain_cap : 32767;
    S.R.outpix = A.out + C.out_base;
    S.R.capOut = C.out_cap;
    S.R.segs = A.segs + C.seg_base;
    S.R.capSegs = C.seg_cap;
    S.par = true;
}

__global__ __launch_bounds__(256) void k_stag_route_walk(StagRoute G, StagArenas A, StagComp *__restrict__ comps, const int *__restrict__ cursors,
                                                         const int32_t *__restrict__ sorted, const int *__restrict__ aslots, int grad_thresh,
                                                         int *__restrict__ prodflag, int *__restrict__ ovf)
{
    const int cid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // one wave per component
    if (cid >= cursors[0]) return;
    StagComp C = comps[cid];
    if (C.nanch == 0) return;
    StagRouter S;
    stag_bind(S, G, A, C);
    S.noSegments = S.totalPixels = S.overflow = 0;
    S.segbase = S.nsp = 0;
    StagRec *recs = A.recs + C.anch_base;
    int nrec = 0, pix_used = 0, chain_used = 0;
    int2 *pix0 = S.R.pix;
    StagChain *chain0 = S.R.chains;
    const int capPix0 = S.R.capPix, capChain0 = C.chain_cap;
    for (int k0 = 0; k0 < C.nanch; k0 += 64) {
        // which of the next 64 anchors are still anchors?  (a walk can only turn anchors OFF, so a stale "on" is re-checked)
        const int kk = k0 + lane;
        int my_off = -1;
        if (kk < C.nanch) my_off = sorted[aslots[C.anch_base + kk]];
        unsigned long long live = __ballot(my_off >= 0 && G.edge[my_off] == STAG_ANCHOR_PIXEL);
        while (live) {
            const int j = __builtin_ctzll(live);
            live &= live - 1;
            const int rank = aslots[C.anch_base + k0 + j];
            const int off = sorted[rank];
            if (G.edge[off] != STAG_ANCHOR_PIXEL) continue;
            S.R.pix = pix0 + pix_used;
            S.R.capPix = capPix0 - pix_used;
            S.R.chains = chain0 + chain_used;
            const int left = capChain0 - chain_used;
            S.R.capChains = left < 32767 ? left : 32767;
            if (S.R.capPix < 16 || S.R.capChains < 4) {
                S.overflow |= 32;
                break;
            }
            const bool keep = S.walk_anchor_wave(off / G.W, off % G.W, grad_thresh, lane);
            if (S.overflow) break;
            if (keep) {
                if (lane == 0) {
                    StagRec r;
                    r.rank = rank; r.pix_off = pix_used; r.len = S.wl_len; r.chain_off = chain_used; r.nchains = S.wl_chains;
                    r.out_off = r.out_len = r.seg_off = r.nsegs = 0;
                    recs[nrec] = r;
                    prodflag[rank] = 1;
                }
                nrec++;
                pix_used += S.wl_len + 1;
                chain_used += S.wl_chains;
            }
        }
        if (S.overflow) break;
    }
    if (lane == 0) {
        comps[cid].nrec = nrec;
        if (S.overflow) atomicOr(ovf, S.overflow);
    }
}

// next[r] = the smallest producing rank > r, or -1 (one workgroup, chunks of 1024 from the top)
__global__ __launch_bounds__(1024) void k_stag_next_above(const int *__restrict__ prodflag, const unsigned *__restrict__ n_anchors, int *__restrict__ next)
{
    __shared__ int s[1024];
    __shared__ int s_carry;
    const int tid = threadIdx.x, n = (int)*n_anchors;
    if (tid == 0) s_carry = -1;
    __syncthreads();
    for (int top = n; top > 0; top -= 1024) {
        // thread t looks at rank r = top - 1 - t: ranks run downwards with t
        const int r = top - 1 - tid;
        const int v = (r >= 0 && prodflag[r]) ? r : -1;
        // for every t: the producing rank with the largest t' < t (= nearest above), i.e. an exclusive "last set" scan
        s[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int o = tid >= d ? s[tid - d] : -1;
            __syncthreads();
            if (s[tid] < 0) s[tid] = o;  // keep the nearest (largest t') set value: own slot wins, else what came from the left
            __syncthreads();
        }
        // s[t] = nearest producing rank at t' <= t; exclusive: t' < t
        const int incl_prev = tid > 0 ? s[tid - 1] : -1;
        const int carry = s_carry;
        if (r >= 0) next[r] = incl_prev >= 0 ? incl_prev : carry;
        __syncthreads();
        if (tid == 1023) s_carry = s[1023] >= 0 ? s[1023] : carry;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_stag_route_extract(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors,
                                                            const int *__restrict__ next, const unsigned *__restrict__ n_anchors,
                                                            int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where,
                                                            int *__restrict__ ovf)
{
    // one wave per component: every lane runs the same scalar steps (same values, same stores); pixel runs are copied by all lanes
    const int cid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (cid >= cursors[0]) return;
    const StagComp C = comps[cid];
    if (C.nanch == 0 || C.nrec == 0) return;
    StagRouter S;
    stag_bind(S, G, A, C);
    S.wlane = lane;
    S.noSegments = S.totalPixels = S.overflow = 0;
    S.segbase = S.nsp = 0;
    StagRec *recs = A.recs + C.anch_base;
    int2 *pix0 = S.R.pix;
    StagChain *chain0 = S.R.chains;
    const int n = (int)*n_anchors;
    int prev_rank = -1;
    for (int k = 0; k < C.nrec; k++) {
        StagRec r = recs[k];
        S.R.pix = pix0 + r.pix_off;
        S.R.chains = chain0 + r.chain_off;
        // the block the reference wrote just before this one: ours only if no other component produced in between
        S.prev_valid = k > 0 && next[r.rank] == prev_rank;
        const int seg0 = S.noSegments, out0 = S.totalPixels;
        S.extract_anchor(r.nchains);
        r.out_off = out0; r.out_len = S.totalPixels - out0;
        r.seg_off = seg0; r.nsegs = S.noSegments - seg0;
        recs[k] = r;
        const int q = n - 1 - r.rank;  // position in processing order
        blk_pix[q] = r.out_len;
        blk_segs[q] = r.nsegs;
        blk_where[q] = make_int2(cid, k);
        prev_rank = r.rank;
        if (S.overflow) break;
    }
    if (S.overflow && lane == 0) atomicOr(ovf, S.overflow);
}

// blk_pix / blk_segs hold exclusive prefix sums by now: copy every block to its place in the global order
__global__ __launch_bounds__(256) void k_stag_route_gather(StagArenas A, const StagComp *__restrict__ comps, const unsigned *__restrict__ n_anchors,
                                                           const int *__restrict__ prodflag, const int *__restrict__ blk_pix,
                                                           const int *__restrict__ blk_segs, const int2 *__restrict__ blk_where,
                                                           int2 *__restrict__ outpix, int2 *__restrict__ segs, int capOut, int capSegs, int *__restrict__ ovf)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = (int)*n_anchors;
    if (q >= n || !prodflag[n - 1 - q]) return;
    const int2 w = blk_where[q];
    const StagComp C = comps[w.x];
    const StagRec r = (A.recs + C.anch_base)[w.y];
    const int po = blk_pix[q], so = blk_segs[q];
    if (po + r.out_len > capOut || so + r.nsegs > capSegs) {
        if (lane == 0) atomicOr(ovf, 64);
        return;
    }
    const int2 *src = A.out + C.out_base + r.out_off;
    for (int i = lane; i < r.out_len; i += 64) outpix[po + i] = src[i];
    const int2 *sg = A.segs + C.seg_base + r.seg_off;
    for (int i = lane; i < r.nsegs; i += 64) segs[so + i] = make_int2(sg[i].x - r.out_off + po, sg[i].y);
}

// ------------------------------------------------------------------------------------------------ K11: segment validation
// ValidateEdgeSegments (ValidateEdgeSegments.cpp:365-413) after the second smoothing of DetectEdgesByEDPF
// (ED.cpp:176-178: SmoothImage(sigma = 1 / 2.5) = cv::GaussianBlur(Size(0, 0), 0.4): OpenCV picks ksize 3 and the 8.8
// fixed-point kernel [10 236 10] / 256, one rounding at the end -- restated, "parity unpinned").
//   k_stag_smooth3_prewitt   the 3x3 blur fused with ComputePrewitt3x3 (:63-115): gradient map + histogram
//   k_stag_valid_prob        H[g] = P(gradient >= g) (:107-111), np = sum len (len - 1) / 2 (:381-385)
//   k_stag_test_segments     TestSegment (:134-199), one wave per segment, the recursion on an explicit stack that lives
//                            in the scratch slots of the segment's own pixels
//   k_stag_extract           ExtractNewSegments (:319-360): runs of still-marked pixels of >= 10 (count pass, scan, write pass)
__global__ __launch_bounds__(256) void k_stag_smooth3_prewitt(const uint8_t *__restrict__ src, int stride, int W, int H,
                                                              uint8_t *__restrict__ smooth, int16_t *__restrict__ grad,
                                                              unsigned *__restrict__ hist)
{
    __shared__ uint8_t s_src[SY + 4][SX + 4 + 4];
    __shared__ uint16_t s_h[SY + 4][SX + 2];
    __shared__ uint8_t s_sm[SY + 2][SX + 2 + 2];
    __shared__ unsigned s_hist[STAG_BINS];
    const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
    const int tid = threadIdx.x;
    for (int i = tid; i < STAG_BINS; i += 256) s_hist[i] = 0;
    for (int i = tid; i < (SY + 4) * (SX + 4); i += 256) {
        const int r = i / (SX + 4), c = i - r * (SX + 4);
        const int gy = stag_reflect101(y0 - 2 + r, H), gx = stag_reflect101(x0 - 2 + c, W);
        s_src[r][c] = src[(long long)gy * stride + gx];
    }
    __syncthreads();
    for (int i = tid; i < (SY + 4) * (SX + 2); i += 256) {
        const int r = i / (SX + 2), c = i - r * (SX + 2);
        const uint8_t *p = &s_src[r][c];
        s_h[r][c] = (uint16_t)(10 * p[0] + 236 * p[1] + 10 * p[2]);
    }
    __syncthreads();
    for (int i = tid; i < (SY + 2) * (SX + 2); i += 256) {
        const int r = i / (SX + 2), c = i - r * (SX + 2);
        const unsigned acc = 10u * s_h[r][c] + 236u * s_h[r + 1][c] + 10u * s_h[r + 2][c];
        s_sm[r][c] = (uint8_t)((acc + 32768u) >> 16);
    }
    __syncthreads();
    for (int i = tid; i < SY * SX; i += 256) {
        const int r = i / SX, c = i - r * SX;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        const long long idx = (long long)gy * W + gx;
        smooth[idx] = s_sm[r + 1][c + 1];
        int g = 0;
        if (gy >= 1 && gy < H - 1 && gx >= 1 && gx < W - 1) {
            const int A = s_sm[r][c], B = s_sm[r][c + 1], C = s_sm[r][c + 2];
            const int D = s_sm[r + 1][c], E = s_sm[r + 1][c + 2];
            const int F = s_sm[r + 2][c], G = s_sm[r + 2][c + 1], Hh = s_sm[r + 2][c + 2];
            const int com1 = Hh - A, com2 = C - F;
            int gxv = com1 + com2 + (E - D), gyv = com1 - com2 + (G - B);
            gxv = gxv < 0 ? -gxv : gxv;
            gyv = gyv < 0 ? -gyv : gyv;
            g = gxv + gyv;
            atomicAdd(&s_hist[g], 1u);
        }
        grad[idx] = (int16_t)g;
    }
    __syncthreads();
    for (int i = tid; i < STAG_BINS; i += 256)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

// one workgroup: cumulative histogram from the top -> H[g]; np over the segments (32-bit int arithmetic as in the reference)
__global__ __launch_bounds__(512) void k_stag_valid_prob(const unsigned *__restrict__ hist, int W, int H, const int2 *__restrict__ segs,
                                                         const int *__restrict__ counters, double *__restrict__ prob, int *__restrict__ np_out)
{
    __shared__ unsigned s_part[512];
    const int tid = threadIdx.x;
    constexpr int PER = STAG_BINS / 512;
    // suffix sums: thread t owns bins [t * PER, t * PER + PER)
    unsigned loc[PER];
    unsigned acc = 0;
    for (int k = PER - 1; k >= 0; k--) {
        acc += hist[tid * PER + k];
        loc[k] = acc;
    }
    s_part[tid] = acc;
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {
        unsigned v = tid + d < 512 ? s_part[tid + d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const unsigned above = tid + 1 < 512 ? s_part[tid + 1] : 0u;
    const double size = (double)((W - 2) * (H - 2));
    for (int k = 0; k < PER; k++) prob[tid * PER + k] = (double)(int)(loc[k] + above) / size;
    __syncthreads();
    // np
    unsigned part = 0;
    const int ns = counters[0];
    for (int i = tid; i < ns; i += 512) {
        const int len = segs[i].y;
        part += (unsigned)((len * (len - 1)) / 2);
    }
    s_part[tid] = part;
    __syncthreads();
    for (int d = 256; d > 0; d >>= 1) {
        if (tid < d) s_part[tid] += s_part[tid + d];
        __syncthreads();
    }
    if (tid == 0) *np_out = (int)s_part[0];
}

__global__ __launch_bounds__(256) void k_stag_test_segments(const int2 *__restrict__ segs, const int *__restrict__ counters,
                                                            const int2 *__restrict__ pix, const int16_t *__restrict__ vgrad, int W,
                                                            const double *__restrict__ prob, const int *__restrict__ np_in, double div,
                                                            int2 *__restrict__ stackmem, uint8_t *__restrict__ edge)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= counters[0]) return;
    const int first = segs[seg].x, n = segs[seg].y;
    if (n < STAG_MIN_PATH_LEN) return;
    const int2 *p = pix + first;
    int2 *stk = stackmem + first;  // n entries: more than the recursion can hold (every entry spans >= 10 pixels)
    const int np = *np_in;
    // every lane keeps the same (wave-uniform) stack: each writes and reads back its own copy of the same words
    stk[0] = make_int2(0, n - 1);
    int sp = 1;
    while (sp > 0) {
        const int2 range = stk[sp - 1];
        sp--;
        const int i1 = range.x, i2 = range.y;
        const int chainLen = i2 - i1 + 1;
        if (chainLen < STAG_MIN_PATH_LEN) continue;
        // first index of the minimum gradient
        int best = 1 << 30, bidx = i2 + 1;
        for (int k = i1 + lane; k <= i2; k += 64) {
            const int2 q = p[k];
            const int g = vgrad[q.x * W + q.y];
            if (g < best) {
                best = g;
                bidx = k;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int ob = __shfl_xor(best, off, 64), oi = __shfl_xor(bidx, off, 64);
            if (ob < best || (ob == best && oi < bidx)) {
                best = ob;
                bidx = oi;
            }
        }
        // NFA (:120-126): np * prob^len, stopped as soon as it is <= 1
        double nfa = (double)np;
        {
            const double pr = prob[best];
            const int len = (int)((double)chainLen / div);
            for (int i = 0; i < len && nfa > 1.0; i++) nfa *= pr;
        }
        if (nfa <= 1.0) {
            for (int k = i1 + lane; k <= i2; k += 64) {
                const int2 q = p[k];
                edge[q.x * W + q.y] = 255;
            }
            continue;
        }
        // split at the minimum: skip the pixels around it that are not above it
        int end = bidx - 1;
        while (end > i1) {
            const int2 q = p[end];
            if (vgrad[q.x * W + q.y] <= best) end--;
            else break;
        }
        int start = bidx + 1;
        while (start < i2) {
            const int2 q = p[start];
            if (vgrad[q.x * W + q.y] <= best) start++;
            else break;
        }
        stk[sp] = make_int2(i1, end);
        stk[sp + 1] = make_int2(start, i2);
        sp += 2;
    }
}

// ExtractNewSegments: one wave per segment.  write = 0: counts[seg] = number of runs of >= 10 marked pixels; write = 1:
// the runs go to out[] from counts[seg] (exclusive prefix sums by then) on.
__global__ __launch_bounds__(256) void k_stag_extract(const int2 *__restrict__ segs, const int *__restrict__ counters,
                                                      const int2 *__restrict__ pix, const uint8_t *__restrict__ edge, int W,
                                                      int *__restrict__ counts, int2 *__restrict__ out, int write)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= counters[0]) return;
    const int first = segs[seg].x, n = segs[seg].y;
    const int2 *p = pix + first;
    int nout = 0, run_start = -1;
    const int obase = write ? counts[seg] : 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int k = c0 + lane;
        bool on = false;
        if (k < n) {
            const int2 q = p[k];
            on = edge[q.x * W + q.y] != 0;
        }
        unsigned long long m = __ballot(on);
        // walk the runs of this chunk (wave-uniform); lanes behind the segment's end read as unmarked
        int pos = 0;
        while (pos < 64) {
            if (run_start < 0) {
                const unsigned long long rest = m >> pos;
                if (!rest) break;
                pos += __builtin_ctzll(rest);
                run_start = c0 + pos;
            }
            const unsigned long long z = ~m >> pos;
            if (!z) break;  // the run goes on into the next chunk
            const int zl = __builtin_ctzll(z);
            const int run_end = c0 + pos + zl;  // first unmarked pixel
            if (run_end - run_start >= 10) {
                if (write && lane == 0) out[obase + nout] = make_int2(first + run_start, run_end - run_start);
                nout++;
            }
            run_start = -1;
            pos += zl + 1;
        }
    }
    if (run_start >= 0 && n - run_start >= 10) {
        if (write && lane == 0) out[obase + nout] = make_int2(first + run_start, n - run_start);
        nout++;
    }
    if (!write && lane == 0) counts[seg] = nout;
}

// exclusive prefix sums over the per-segment counts (one workgroup, serial over chunks of 1024)
__global__ __launch_bounds__(1024) void k_stag_scan_counts(int *__restrict__ counts, const int *__restrict__ counters, int *__restrict__ total)
{
    __shared__ int s[1024];
    __shared__ int s_carry;
    const int tid = threadIdx.x, n = counters[0];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int v = base + tid < n ? counts[base + tid] : 0;
        s[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int t = tid >= d ? s[tid - d] : 0;
            __syncthreads();
            s[tid] += t;
            __syncthreads();
        }
        const int carry = s_carry;
        if (base + tid < n) counts[base + tid] = carry + s[tid] - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s[1023];
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}

// ------------------------------------------------------------------------------------------------ K12: EDLines, line fitting
// DetectLinesByEDPF (EDLines.cpp:849-941) after the edge detection: SplitSegment2Lines (:162-268) cuts every validated
// segment into least-squares lines, JoinCollinearLines (:114-156) merges neighbours inside a segment.  Both are sequential
// inside a segment and independent between segments: one lane per segment (k_stag_split_lines), lines of segment i parked at
// slot first_pixel_i / 9 onwards (a line takes >= 9 pixels), then counted, scanned and compacted in segment order.
// The fits are sums of integer coordinates: prefix sums (exact in 64 bits) make every refit O(1) and give bit for bit the
// doubles the reference accumulates; the remaining double arithmetic keeps the reference's operation order (the TU is built
// with -ffp-contract=off).
struct StagPrefix {  // prefix sums over the pixels of one segment, index k = sum over pixels < k
    long long *x, *y, *xx, *yy, *xy;
};

__device__ double sl_min_dist(double x1, double y1, double a, double b, int invert, double *cx = nullptr, double *cy = nullptr)
{
    double x2, y2;
    if (invert == 0) {
        if (b == 0) {
            x2 = x1;
            y2 = a;
        } else {
            const double d = -1.0 / b;
            const double c = y1 - d * x1;
            x2 = (a - c) / (d - b);
            y2 = a + b * x2;
        }
    } else {
        if (b == 0) {
            x2 = a;
            y2 = y1;
        } else {
            const double d = -1.0 / b;
            const double c = x1 - d * y1;
            y2 = (a - c) / (d - b);
            x2 = a + b * y2;
        }
    }
    if (cx) {
        *cx = x2;
        *cy = y2;
    }
    return sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
}

// LineFit with a known orientation (LineSegment.cpp:703-733) over pixels [base, base + count)
__device__ void sl_fit_known(const StagPrefix &P, int base, int count, int invert, double *a, double *b)
{
    if (count < 2) return;
    const double S = count;
    double Sx = (double)(P.x[base + count] - P.x[base]), Sy = (double)(P.y[base + count] - P.y[base]);
    double Sxx, Sxy = (double)(P.xy[base + count] - P.xy[base]);
    if (invert) {
        const double t = Sx;
        Sx = Sy;
        Sy = t;
        Sxx = (double)(P.yy[base + count] - P.yy[base]);
    } else {
        Sxx = (double)(P.xx[base + count] - P.xx[base]);
    }
    const double D = S * Sxx - Sx * Sx;
    *a = (Sxx * Sy - Sx * Sxy) / D;
    *b = (S * Sxy - Sx * Sy) / D;
}

// LineFit with orientation choice and fitting error (LineSegment.cpp:628-697) over pixels [base, base + count)
__device__ void sl_fit_first(const StagPrefix &P, const int2 *px, int base, int count, double *a, double *b, double *e, int *invert)
{
    if (count < 2) return;
    const double Sx0 = (double)(P.x[base + count] - P.x[base]), Sy0 = (double)(P.y[base + count] - P.y[base]);
    const double mx = Sx0 / count, my = Sy0 / count;
    double dx = 0.0, dy = 0.0;
    for (int i = 0; i < count; i++) {
        const double xi = px[base + i].y, yi = px[base + i].x;
        dx += (xi - mx) * (xi - mx);
        dy += (yi - my) * (yi - my);
    }
    const int inv = dx < dy ? 1 : 0;
    *invert = inv;
    sl_fit_known(P, base, count, inv, a, b);
    double error = 0.0;
    if (*b == 0.0) {
        for (int i = 0; i < count; i++) {
            const double yi = inv ? px[base + i].y : px[base + i].x;
            error += fabs((*a) - yi);
        }
        *e = error / count;
    } else {
        for (int i = 0; i < count; i++) {
            const double xi = inv ? px[base + i].x : px[base + i].y, yi = inv ? px[base + i].y : px[base + i].x;
            const double d = -1.0 / (*b);
            const double c = yi - d * xi;
            const double x2 = ((*a) - c) / (d - (*b));
            const double y2 = (*a) + (*b) * x2;
            error += (xi - x2) * (xi - x2) + (yi - y2) * (yi - y2);
        }
        *e = sqrt(error / count);
    }
}

// UpdateLineParameters (LineSegment.cpp:563-591)
__device__ void sl_update_params(fid_stag_line *ls)
{
    const double dx = ls->ex - ls->sx, dy = ls->ey - ls->sy;
    if (fabs(dx) >= fabs(dy)) {
        ls->invert = 0;
        if (fabs(dy) < 1e-3) {
            ls->b = 0;
            ls->a = (ls->sy + ls->ey) / 2;
        } else {
            ls->b = dy / dx;
            ls->a = ls->sy - (ls->b) * ls->sx;
        }
    } else {
        ls->invert = 1;
        if (fabs(dx) < 1e-3) {
            ls->b = 0;
            ls->a = (ls->sx + ls->ex) / 2;
        } else {
            ls->b = dx / dy;
            ls->a = ls->sx - (ls->b) * ls->sy;
        }
    }
}

// TryToJoinTwoLineSegments (LineSegment.cpp:239-395)
__device__ bool sl_try_join(fid_stag_line *l1, const fid_stag_line *l2, double max_dist, double max_err)
{
    double dx = l1->sx - l2->sx, dy = l1->sy - l2->sy;
    double mn = sqrt(dx * dx + dy * dy);
    dx = l1->sx - l2->ex; dy = l1->sy - l2->ey;
    double d = sqrt(dx * dx + dy * dy);
    if (d < mn) mn = d;
    dx = l1->ex - l2->sx; dy = l1->ey - l2->sy;
    d = sqrt(dx * dx + dy * dy);
    if (d < mn) mn = d;
    dx = l1->ex - l2->ex; dy = l1->ey - l2->ey;
    d = sqrt(dx * dx + dy * dy);
    if (d < mn) mn = d;
    if (mn > max_dist) return false;
    dx = l1->sx - l1->ex; dy = l1->sy - l1->ey;
    const double prevLen = sqrt(dx * dx + dy * dy);
    dx = l2->sx - l2->ex; dy = l2->sy - l2->ey;
    const double nextLen = sqrt(dx * dx + dy * dy);
    const fid_stag_line *shorter = l1, *longer = l2;
    if (prevLen > nextLen) {
        shorter = l2;
        longer = l1;
    }
    double dist = sl_min_dist(shorter->sx, shorter->sy, longer->a, longer->b, longer->invert);
    dist += sl_min_dist((shorter->sx + shorter->ex) / 2.0, (shorter->sy + shorter->ey) / 2.0, longer->a, longer->b, longer->invert);
    dist += sl_min_dist(shorter->ex, shorter->ey, longer->a, longer->b, longer->invert);
    dist /= 3.0;
    if (dist > max_err) return false;
    // keep the two end points that are farthest apart (Manhattan)
    double mx = fabs(l1->sx - l2->sx) + fabs(l1->sy - l2->sy);
    int which = 1;
    d = fabs(l1->sx - l2->ex) + fabs(l1->sy - l2->ey);
    if (d > mx) { mx = d; which = 2; }
    d = fabs(l1->ex - l2->sx) + fabs(l1->ey - l2->sy);
    if (d > mx) { mx = d; which = 3; }
    d = fabs(l1->ex - l2->ex) + fabs(l1->ey - l2->ey);
    if (d > mx) { mx = d; which = 4; }
    if (which == 1) {
        l1->ex = l2->sx; l1->ey = l2->sy;
    } else if (which == 2) {
        l1->ex = l2->ex; l1->ey = l2->ey;
    } else if (which == 3) {
        l1->sx = l2->sx; l1->sy = l2->sy;
    } else {
        l1->sx = l1->ex; l1->sy = l1->ey;
        l1->ex = l2->ex; l1->ey = l2->ey;
    }
    if (l1->firstPixelIndex + l1->len + 5 >= l2->firstPixelIndex) l1->len += l2->len;
    else if (l2->len > l1->len) {
        l1->firstPixelIndex = l2->firstPixelIndex;
        l1->len = l2->len;
    }
    sl_update_params(l1);
    return true;
}

__device__ __forceinline__ long long wave_iscan_ll(long long v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// One wave per segment.  The state of SplitSegment2Lines is wave-uniform; three things are spread over the lanes without
// changing any result: the prefix sums (wave scan), the search for the first window of MIN_LINE_LEN pixels that fits a line
// (64 window positions at a time, each lane its own 9-pixel fit), and the point-to-line distances of the next 64 pixels under
// the CURRENT line -- the sequential good / bad bookkeeping then runs over the ballot until a refit really changes the line
// (every tenth good pixel), at which point the rest of the batch is thrown away and recomputed.
__global__ __launch_bounds__(256) void k_stag_split_lines(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int2 *__restrict__ pix,
                                                          StagPrefix PF, int min_line_len, double line_error, fid_stag_line *__restrict__ slots,
                                                          int *__restrict__ counts)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= *nsegs) return;
    const int first = segs[seg].x, n = segs[seg].y;
    const int2 *px = pix + first;
    // the prefix arrays of this segment live at [first + seg, first + seg + n]: one extra slot per segment
    StagPrefix P;
    const int pb = first + seg;
    P.x = PF.x + pb; P.y = PF.y + pb; P.xx = PF.xx + pb; P.yy = PF.yy + pb; P.xy = PF.xy + pb;
    {
        long long cx = 0, cy = 0, cxx = 0, cyy = 0, cxy = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            long long x = 0, y = 0;
            if (k < n) {
                x = px[k].y;
                y = px[k].x;
            }
            const long long ix = wave_iscan_ll(x, lane), iy = wave_iscan_ll(y, lane), ixx = wave_iscan_ll(x * x, lane),
                            iyy = wave_iscan_ll(y * y, lane), ixy = wave_iscan_ll(x * y, lane);
            if (k < n) {  // exclusive value at k
                P.x[k] = cx + ix - x; P.y[k] = cy + iy - y; P.xx[k] = cxx + ixx - x * x; P.yy[k] = cyy + iyy - y * y; P.xy[k] = cxy + ixy - x * y;
            }
            cx += __shfl(ix, 63, 64); cy += __shfl(iy, 63, 64); cxx += __shfl(ixx, 63, 64); cyy += __shfl(iyy, 63, 64); cxy += __shfl(ixy, 63, 64);
        }
        if (lane == 0) {
            P.x[n] = cx; P.y[n] = cy; P.xx[n] = cxx; P.yy[n] = cyy; P.xy[n] = cxy;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    fid_stag_line *L = slots + first / 9;
    int nl = 0;
    const int MLL = min_line_len;
    int base = 0, noPixels = n, firstPixelIndex = 0;
    while (noPixels >= MLL) {
        bool valid = false;
        double lastA = 0, lastB = 0, error = 0;
        int lastInvert = 0;
        // first window (sliding by one pixel) whose MLL-pixel fit has error <= 0.5: 64 positions per round
        while (noPixels >= MLL) {
            const int avail = noPixels - MLL + 1;  // window starts base .. base + avail - 1
            double a = 0, bq = 0, e = 1e300;
            int inv = 0;
            if (lane < avail) sl_fit_first(P, px, base + lane, MLL, &a, &bq, &e, &inv);
            const unsigned long long okm = __ballot(lane < avail && e <= 0.5);
            if (okm) {
                const int j = __builtin_ctzll(okm);
                lastA = __shfl(a, j, 64); lastB = __shfl(bq, j, 64); error = __shfl(e, j, 64); lastInvert = __shfl(inv, j, 64);
                noPixels -= j; base += j; firstPixelIndex += j;
                valid = true;
                break;
            }
            const int adv = avail < 64 ? avail : 64;
            noPixels -= adv; base += adv; firstPixelIndex += adv;
        }
        if (!valid) break;
        int index = MLL, len = MLL;
        while (index < noPixels) {
            const int startIndex = index;
            int lastGoodIndex = index - 1, good = 0, bad = 0;
            int fitCount = len;  // pixels behind the current line parameters
            bool broke = false;
            while (index < noPixels && !broke) {
                // distances of the next pixels under the current line
                const int k = index + lane;
                bool ok = false;
                if (k < noPixels) ok = sl_min_dist((double)px[base + k].y, (double)px[base + k].x, lastA, lastB, lastInvert) <= line_error;
                const unsigned long long gm = __ballot(ok);
                const int lim = noPixels - index < 64 ? noPixels - index : 64;
                int t = 0;
                for (; t < lim; t++) {
                    if ((gm >> t) & 1ull) {
                        lastGoodIndex = index;
                        good++;
                        bad = 0;
                    } else {
                        bad++;
                        if (bad >= 5) {
                            broke = true;  // (the reference leaves `index` on this pixel)
                            break;
                        }
                    }
                    bool refit = false;
                    if (good % 10 == 0) {
                        const int cnt = lastGoodIndex - startIndex + len + 1;
                        if (cnt != fitCount) {  // same pixels -> same parameters: nothing to do
                            sl_fit_known(P, base, cnt, lastInvert, &lastA, &lastB);
                            fitCount = cnt;
                            refit = true;
                        }
                    }
                    index++;
                    if (refit) break;  // the rest of the batch was measured against the old line
                }
            }
            if (good >= 2) {
                len += lastGoodIndex - startIndex + 1;
                sl_fit_known(P, base, len, lastInvert, &lastA, &lastB);
                index = lastGoodIndex + 1;
            }
            if (good < 2 || index >= noPixels) {
                double sx, sy, ex, ey;
                int idx = 0;
                while (idx < noPixels - 1 && sl_min_dist((double)px[base + idx].y, (double)px[base + idx].x, lastA, lastB, lastInvert) > line_error) idx++;
                sl_min_dist((double)px[base + idx].y, (double)px[base + idx].x, lastA, lastB, lastInvert, &sx, &sy);
                const int skipped = idx;
                idx = lastGoodIndex;
                while (idx > 0 && sl_min_dist((double)px[base + idx].y, (double)px[base + idx].x, lastA, lastB, lastInvert) > line_error) idx--;
                sl_min_dist((double)px[base + idx].y, (double)px[base + idx].x, lastA, lastB, lastInvert, &ex, &ey);
                if (lane == 0) {
                    fid_stag_line &o = L[nl];
                    o.a = lastA; o.b = lastB; o.invert = lastInvert; o.sx = sx; o.sy = sy; o.ex = ex; o.ey = ey;
                    o.segmentNo = seg; o.firstPixelIndex = firstPixelIndex + skipped; o.len = idx - skipped + 1;
                }
                nl++;
                len = idx + 1;
                break;
            }
        }
        noPixels -= len;
        base += len;
        firstPixelIndex += len;
    }
    // JoinCollinearLines (EDLines.cpp:114-156), MAX_DISTANCE_BETWEEN_TWO_LINES 6.0, MAX_ERROR 1.5 (:913): lane 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane != 0) return;
    if (nl > 0) {
        int last = 0;
        for (int j = 1; j < nl; j++) {
            if (!sl_try_join(&L[last], &L[j], 6.0, 1.50)) {
                last++;
                if (last != j) L[last] = L[j];
            }
        }
        if (last != 0 && sl_try_join(&L[0], &L[last], 6.0, 1.50)) last--;
        nl = last + 1;
    }
    counts[seg] = nl;
}

// the lines of every segment, one after the other in segment order (counts hold exclusive prefix sums by now)
__global__ __launch_bounds__(64) void k_stag_gather_lines(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int *__restrict__ counts,
                                                          const int *__restrict__ total, const fid_stag_line *__restrict__ slots,
                                                          fid_stag_line *__restrict__ out)
{
    const int seg = blockIdx.x * 64 + threadIdx.x;
    const int ns = *nsegs;
    if (seg >= ns) return;
    const int o = counts[seg], n = (seg + 1 < ns ? counts[seg + 1] : *total) - o;
    const fid_stag_line *L = slots + segs[seg].x / 9;
    for (int j = 0; j < n; j++) out[o + j] = L[j];
}

// ------------------------------------------------------------------------------------------------ K13: line validation
// ValidateLineSegments (EDLines.cpp:274-409): a line is kept if enough of its pixels have a gradient direction within
// 22.5 degrees of the line (Helmholtz principle, number of false alarms from a table).  Lines of >= 80 pixels pass untested,
// lines of <= 25 pixels are tested on all pixels of a 2-pixel-wide rectangle around them (EnumerateRectPoints, :417-600, the
// LSD rectangle iterator), the others on their own pixels first and on the rectangle if that fails.  One lane per line.
// Host-made tables (functions of the image size only, evaluated with the host's libm exactly as the reference does):
//   atan_lut[i] = atan(i / 1024)  (myAtan2, MyMath.cpp:12-72);  kmin[n] = the NFALUT entry (NFA.cpp:13-44).
struct StagLineTables {
    const double *atan_lut;  // 1025 entries
    const int *kmin;         // kmin[n]: smallest number of aligned pixels out of n that validates; n <= kmin_n
    int kmin_n;
};

__device__ double sl_my_atan2(const double *lut, double yy, double xx)
{
    const double PI = 3.14159265358979323846;
    double y = fabs(yy), x = fabs(xx);
    if (x < 0.0001) return y < 0.0001 ? 0.0 : PI / 2;
    bool invert = false;
    if (y > x) {
        const double t = x;
        x = y;
        y = t;
        invert = true;
    }
    const double ratio = y / x;
    double angle = lut[(int)(ratio * 1024)];
    if (xx >= 0) {
        if (yy >= 0) {
            if (invert) angle = PI / 2 - angle;
        } else {
            angle = invert ? PI / 2 + angle : PI - angle;
        }
    } else {
        if (yy >= 0) {
            angle = invert ? PI / 2 + angle : PI - angle;
        } else {
            if (invert) angle = PI / 2 - angle;
        }
    }
    return angle;
}

// is the gradient at (r, c) of the source image aligned with the line?  (-1: the pixel does not count)
__device__ int sl_aligned(const uint8_t *__restrict__ src, int W, int H, int r, int c, double lineAngle, const double *lut)
{
    const double PI = 3.14159265358979323846, prec = (22.5 / 180) * PI;
    if (r <= 0 || r >= H - 1 || c <= 0 || c >= W - 1) return -1;
    const int com1 = src[(r + 1) * W + c + 1] - src[(r - 1) * W + c - 1];
    const int com2 = src[(r - 1) * W + c + 1] - src[(r + 1) * W + c - 1];
    const int gx = com1 + com2 + src[r * W + c + 1] - src[r * W + c - 1];
    const int gy = com1 - com2 + src[(r + 1) * W + c] - src[(r - 1) * W + c];
    const double pixelAngle = sl_my_atan2(lut, (double)gx, (double)-gy);
    const double diff = fabs(lineAngle - pixelAngle);
    return (diff <= prec || diff >= PI - prec) ? 1 : 0;
}

// ValidateLineSegmentRect (EDLines.cpp:612-690) with the rectangle iterator of EnumerateRectPoints (:417-600) inlined
__device__ bool sl_validate_rect(const uint8_t *__restrict__ src, int W, int H, const fid_stag_line &ls, double lineAngle, const StagLineTables &T)
{
    const double x1 = ls.sx, y1 = ls.sy, x2 = ls.ex, y2 = ls.ey, width = 2;
    double dx = x2 - x1, dy = y2 - y1;
    const double vLen = sqrt(dx * dx + dy * dy);
    dx = dx / vLen;
    dy = dy / vLen;
    double vxT[4], vyT[4], vx[4], vy[4];
    vxT[0] = x1 - dy * width / 2.0; vyT[0] = y1 + dx * width / 2.0;
    vxT[1] = x2 - dy * width / 2.0; vyT[1] = y2 + dx * width / 2.0;
    vxT[2] = x2 + dy * width / 2.0; vyT[2] = y2 - dx * width / 2.0;
    vxT[3] = x1 + dy * width / 2.0; vyT[3] = y1 - dx * width / 2.0;
    int offset;
    if (x1 < x2 && y1 <= y2) offset = 0;
    else if (x1 >= x2 && y1 < y2) offset = 1;
    else if (x1 > x2 && y1 >= y2) offset = 2;
    else offset = 3;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        vx[n] = vxT[(offset + n) % 4];
        vy[n] = vyT[(offset + n) % 4];
    }
    int x = (int)ceil(vx[0]) - 1, y = (int)ceil(vy[0]);
    double ys = -1.7976931348623157e308, ye = -1.7976931348623157e308;
    int noPoints = 0, count = 0, aligned = 0;
    const int maxNoOfPoints = (int)(fabs(ls.sx - ls.ex) + fabs(ls.sy - ls.ey)) * 4;
    while (noPoints < maxNoOfPoints) {
        y++;
        while (y > ye && x <= vx[2]) {
            x++;
            if (x > vx[2]) break;
            if ((double)x < vx[3]) {
                if (fabs(vx[0] - vx[3]) <= 0.01) {
                    if (vy[0] < vy[3]) ys = vy[0];
                    else if (vy[0] > vy[3]) ys = vy[3];
                    else ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
                } else
                    ys = vy[0] + (x - vx[0]) * (vy[3] - vy[0]) / (vx[3] - vx[0]);
            } else {
                if (fabs(vx[3] - vx[2]) <= 0.01) {
                    if (vy[3] < vy[2]) ys = vy[3];
                    else if (vy[3] > vy[2]) ys = vy[2];
                    else ys = vy[3] + (x - vx[3]) * (y2 - vy[3]) / (vx[2] - vx[3]);  // (y2, as in the reference)
                } else
                    ys = vy[3] + (x - vx[3]) * (vy[2] - vy[3]) / (vx[2] - vx[3]);
            }
            if ((double)x < vx[1]) {
                if (fabs(vx[0] - vx[1]) <= 0.01) {
                    if (vy[0] < vy[1]) ye = vy[1];
                    else if (vy[0] > vy[1]) ye = vy[0];
                    else ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
                } else
                    ye = vy[0] + (x - vx[0]) * (vy[1] - vy[0]) / (vx[1] - vx[0]);
            } else {
                if (fabs(vx[1] - vx[2]) <= 0.01) {
                    if (vy[1] < vy[2]) ye = vy[2];
                    else if (vy[1] > vy[2]) ye = vy[1];
                    else ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
                } else
                    ye = vy[1] + (x - vx[1]) * (vy[2] - vy[1]) / (vx[2] - vx[1]);
            }
            y = (int)ceil(ys);
        }
        if (x > vx[2]) break;
        noPoints++;
        const int al = sl_aligned(src, W, H, y, x, lineAngle, T.atan_lut);
        if (al >= 0) {
            count++;
            aligned += al;
        }
    }
    return count <= T.kmin_n ? aligned >= T.kmin[count] : false;
}

__global__ __launch_bounds__(64) void k_stag_validate_lines(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines,
                                                            const uint8_t *__restrict__ src, int W, int H, const int2 *__restrict__ vsegs,
                                                            const int2 *__restrict__ pix, StagLineTables T, int *__restrict__ flags)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= *nlines) return;
    const double PI = 3.14159265358979323846;
    const fid_stag_line ls = lines[i];
    double lineAngle = ls.invert == 0 ? atan(ls.b) : atan(1.0 / ls.b);
    if (lineAngle < 0) lineAngle += PI;
    bool valid;
    if (ls.len >= 80) {
        valid = true;
    } else if (ls.len <= 25) {
        valid = sl_validate_rect(src, W, H, ls, lineAngle, T);
    } else {
        const int2 *p = pix + vsegs[ls.segmentNo].x + ls.firstPixelIndex;
        int count = 0, aligned = 0;
        for (int j = 0; j < ls.len; j++) {
            const int al = sl_aligned(src, W, H, p[j].x, p[j].y, lineAngle, T.atan_lut);
            if (al >= 0) {
                count++;
                aligned += al;
            }
        }
        valid = count <= T.kmin_n ? aligned >= T.kmin[count] : false;
        if (!valid) valid = sl_validate_rect(src, W, H, ls, lineAngle, T);
    }
    flags[i] = valid ? 1 : 0;
}

// keep the valid lines, in order (flags hold exclusive prefix sums by now)
__global__ __launch_bounds__(256) void k_stag_compact_lines(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines,
                                                            const int *__restrict__ pos, const int *__restrict__ total,
                                                            fid_stag_line *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = *nlines;
    if (i >= n) return;
    const int next = i + 1 < n ? pos[i + 1] : *total;
    if (next != pos[i]) out[pos[i]] = lines[i];
}

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

// EDInterface::correctLineDirection: going from start to end the darker side must be on the right
__device__ void sq_correct_direction(const uint8_t *__restrict__ img, int W, int H, fid_stag_line &ls)
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
    for (int i = 0; i < n; i++) {
        int rx, ry, lx, ly;
        sample(i, &rx, &ry, &lx, &ly);
        const bool rin = rx >= 0 && rx < W && ry >= 0 && ry < H, lin = lx >= 0 && lx < W && ly >= 0 && ly < H;
        // (without the safe read the reference reads unchecked; points between two in-range end points are in range)
        accR += rin ? img[ry * W + rx] : (safe ? 128u : 0u);
        accL += lin ? img[ly * W + lx] : (safe ? 128u : 0u);
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
__global__ __launch_bounds__(256) void k_stag_line_ranges(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, int2 *__restrict__ range)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = *nlines;
    if (i >= n) return;
    const int sg = lines[i].segmentNo;
    if (i == 0 || lines[i - 1].segmentNo != sg) range[sg].x = i;
    if (i == n - 1 || lines[i + 1].segmentNo != sg) range[sg].y = i + 1;
}

__global__ __launch_bounds__(256) void k_stag_quads(fid_stag_line *__restrict__ lines, const int2 *__restrict__ range, const int *__restrict__ nsegs,
                                                    const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, const uint8_t *__restrict__ img, int W,
                                                    int H, StagCorner *__restrict__ corner_slots, int *__restrict__ order_slots,
                                                    fid_stag_quad *__restrict__ quad_slots, int *__restrict__ counts)
{
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg >= *nsegs) return;
    const int lo = range[seg].x, n = range[seg].y - lo;
    if (lane == 0) counts[seg] = 0;
    if (range[seg].y == 0 || n < 4) return;  // groups need >= 4 lines of one edge segment
    // ---- groupLines: fix the direction of every line of the group (each lane one line), then the order of the group
    for (int k = lane; k < n; k += 64) {
        fid_stag_line l = lines[lo + k];
        sq_correct_direction(img, W, H, l);
        lines[lo + k] = l;
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

__global__ __launch_bounds__(64) void k_stag_gather_quads(const int2 *__restrict__ range, const int *__restrict__ nsegs, const int *__restrict__ counts,
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

__global__ __launch_bounds__(256) void k_stag_decode(const fid_stag_quad *__restrict__ quads, const int *__restrict__ nquads,
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
        for (int i = 0; i < 256; i++) {
            const double p_i = hist[i] * scale;
            mu1 *= q1;
            q1 += p_i;
            const double q2 = 1. - q1;
            if (fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07) continue;
            mu1 = (mu1 + i * p_i) / q1;
            const double mu2 = (mu - q1 * mu1) / q2;
            const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
            if (sigma > max_sigma) {
                max_sigma = sigma;
                max_val = i;
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

// Stag::checkDuplicate over the decoded quads in quad order: one marker per id, the least distorted one, at the position of
// the first quad that showed the id
__global__ __launch_bounds__(64) void k_stag_dedup(const fid_stag_marker *__restrict__ cand, const int *__restrict__ found, const int *__restrict__ nquads,
                                                   fid_stag_marker *__restrict__ out, int *__restrict__ nout)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = *nquads;
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

// ------------------------------------------------------------------------------------------------ K16: pose refinement
// PoseRefiner::refineMarkerPose (PoseRefiner.cpp:12-190), one wave per marker:
//   (1) pick the edge segment that is the image of the marker's circular border: a closed loop of >= 20 pixels inside the
//       quad whose back-projection stays within 0.1 of the circle of radius 0.4 and whose distances to 36 points of that circle
//       sum to < 1.8 (lanes take pixels; minima are order-free, the sum over the 36 points runs in order);
//   (2) fit an ellipse to it: customEllipse(pix*, n) (Ellipse.cpp:296-473) = Fitzgibbon's direct least squares through the
//       reference's own small linear algebra (scatter matrix summed in pixel order -- one lane per matrix entry --, choldc,
//       Gauss-Jordan inverse, Jacobi eigenvalues, all 1-based like the original);
//   (3) move the 9 entries of H with Nelder-Mead so that H^T C H is the circle (0.5, 0.5, r 0.4): cv::DownhillSolver with its
//       defaults, restated (see oracle/stag_ref.cpp for the same restatement on the checker's side); cost Refine::calc (:224-258);
//   (4) corners and centre from the new H.
// atan / sin / cos come from the device's math library here and from glibc in the reference: the results agree to rounding,
// the optimiser then follows a path that can differ in the last bits -- this row's parity bar is a tolerance (corners to
// 1e-3 px), not equality.
struct SrEllipse {
    double A1, B1, C1, D1, E1, F1, cX, cY, a, b;
};

// the conic -> ellipse conversion shared by both customEllipse constructors (Ellipse.cpp:394-454, :668-728); coefficients
// come in unnormalised
__device__ void sr_conic_to_ellipse(double A1, double B1, double C1, double D1, double E1, double F1, SrEllipse *e)
{
    B1 /= A1; C1 /= A1; D1 /= A1; E1 /= A1; F1 /= A1; A1 /= A1;
    double A2, C2, D2, E2, F2, rotation = 0, sr = 0, cr = 1;  // (the reference leaves rotation unset when B1 == 0)
    if (B1 == 0) {
        A2 = A1; C2 = C1; D2 = D1; E2 = E1; F2 = F1;
    } else {
        rotation = atan(B1 / (A1 - C1)) / 2;
        double s2, c2;
        sincos(2 * rotation, &s2, &c2);
        sincos(rotation, &sr, &cr);
        A2 = 0.5 * (A1 * (1 + c2 + B1 * s2 + C1 * (1 - c2)));
        C2 = 0.5 * (A1 * (1 - c2 - B1 * s2 + C1 * (1 + c2)));
        D2 = D1 * cr + E1 * sr;
        E2 = -D1 * sr + E1 * cr;
        F2 = F1;
    }
    const double D3 = D2 / A2, E3 = E2 / C2;
    double cX = -(D3 / 2), cY = -(E3 / 2);
    const double F3 = A2 * (cX * cX) + C2 * (cY * cY) - F2;
    e->a = sqrt(F3 / A2);
    e->b = sqrt(F3 / C2);
    if (rotation != 0) {
        const double tx = cX, ty = cY;
        cX = tx * cr - ty * sr;
        cY = tx * sr + ty * cr;
    }
    e->cX = cX; e->cY = cY;
    e->A1 = A1; e->B1 = B1; e->C1 = C1; e->D1 = D1; e->E1 = E1; e->F1 = F1;
}

// 1-based 7 x 7 scratch matrices as in the reference
typedef double SrM[7][7];

__device__ void sr_jacobi(SrM a, double d[7], SrM v)
{
    const int n = 6;
    double b[7], z[7];
    for (int ip = 1; ip <= n; ip++) {
        for (int iq = 1; iq <= n; iq++) v[ip][iq] = 0.0;
        v[ip][ip] = 1.0;
    }
    for (int ip = 1; ip <= n; ip++) {
        b[ip] = d[ip] = a[ip][ip];
        z[ip] = 0.0;
    }
    auto rot = [](SrM m, int i, int j, int k, int l, double tau, double s) {
        const double g = m[i][j], h = m[k][l];
        m[i][j] = g - s * (h + g * tau);
        m[k][l] = h + s * (g - h * tau);
    };
    for (int i = 1; i <= 50; i++) {
        double sm = 0.0;
        for (int ip = 1; ip <= n - 1; ip++)
            for (int iq = ip + 1; iq <= n; iq++) sm += fabs(a[ip][iq]);
        if (sm == 0.0) return;
        const double tresh = i < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int ip = 1; ip <= n - 1; ip++) {
            for (int iq = ip + 1; iq <= n; iq++) {
                const double g = 100.0 * fabs(a[ip][iq]);
                if (i > 4 && g == 0.0) a[ip][iq] = 0.0;
                else if (fabs(a[ip][iq]) > tresh) {
                    double h = d[iq] - d[ip], t;
                    if (g == 0.0) t = (a[ip][iq]) / h;
                    else {
                        const double theta = 0.5 * h / (a[ip][iq]);
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / sqrt(1 + t * t), sn = t * c, tau = sn / (1.0 + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h;
                    a[ip][iq] = 0.0;
                    for (int j = 1; j <= ip - 1; j++) rot(a, j, ip, j, iq, tau, sn);
                    for (int j = ip + 1; j <= iq - 1; j++) rot(a, ip, j, j, iq, tau, sn);
                    for (int j = iq + 1; j <= n; j++) rot(a, ip, j, iq, j, tau, sn);
                    for (int j = 1; j <= n; j++) rot(v, j, ip, j, iq, tau, sn);
                }
            }
        }
        for (int ip = 1; ip <= n; ip++) {
            b[ip] += z[ip];
            d[ip] = b[ip];
            z[ip] = 0.0;
        }
    }
}

// customEllipse(pix*, n) from the scatter matrix S (1-based, full) on: returns false if the inverse fails
__device__ bool sr_fit_from_scatter(SrM S, SrEllipse *e)
{
    const int n = 6;
    SrM L, invL, temp, C, V, sol, Const;
    double d[7], p[7];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) L[i][j] = invL[i][j] = temp[i][j] = C[i][j] = V[i][j] = sol[i][j] = Const[i][j] = 0.0;
    for (int i = 0; i < 7; i++) d[i] = p[i] = 0.0;
    Const[1][3] = -2; Const[2][2] = 1; Const[3][1] = -2;  // FPF mode
    // choldc
    for (int i = 1; i <= n; i++) {
        for (int j = i; j <= n; j++) {
            double sum = S[i][j];
            for (int k = i - 1; k >= 1; k--) sum -= S[i][k] * S[j][k];
            if (i == j) {
                if (sum > 0.0) p[i] = sqrt(sum);
            } else
                S[j][i] = sum / p[i];
        }
    }
    for (int i = 1; i <= n; i++)
        for (int j = i; j <= n; j++) {
            if (i == j) L[i][i] = p[i];
            else {
                L[j][i] = S[j][i];
                L[i][j] = 0.0;
            }
        }
    // inverse(L): Gauss-Jordan with row pivoting on [L | I]
    {
        double A[7][14];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 14; j++) A[k][j] = 0.0;
        for (int k = 1; k <= n; k++) {
            for (int j = 1; j <= n; j++) A[k][j] = L[k][j];  // (column n + 1 stays 0 as in the reference)
            A[k][k - 1 + n + 2] = 1;
        }
        for (int k = 1; k <= n; k++) {
            double maxpivot = fabs(A[k][k]);
            int npivot = k;
            for (int i = k; i <= n; i++)
                if (maxpivot < fabs(A[i][k])) {
                    maxpivot = fabs(A[i][k]);
                    npivot = i;
                }
            if (!(maxpivot >= 10e-20)) return false;
            if (npivot != k)
                for (int j = k; j <= 2 * n + 1; j++) {
                    const double t = A[npivot][j];
                    A[npivot][j] = A[k][j];
                    A[k][j] = t;
                }
            const double Dv = A[k][k];
            for (int j = 2 * n + 1; j >= k; j--) A[k][j] = A[k][j] / Dv;
            for (int i = 1; i <= n; i++)
                if (i != k) {
                    const double mult = A[i][k];
                    for (int j = 2 * n + 1; j >= k; j--) A[i][j] = A[i][j] - mult * A[k][j];
                }
        }
        for (int k = 1; k <= n; k++)
            for (int j = n + 2, q = 1; j <= 2 * n + 1; j++, q++) invL[k][q] = A[k][j];
    }
    // temp = Const * invL^T, C = invL * temp
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            temp[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) temp[pp][q] = temp[pp][q] + Const[pp][l] * invL[q][l];
        }
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            C[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) C[pp][q] = C[pp][q] + invL[pp][l] * temp[l][q];
        }
    sr_jacobi(C, d, V);
    // sol = invL^T * V
    for (int pp = 1; pp <= n; pp++)
        for (int q = 1; q <= n; q++) {
            sol[pp][q] = 0.0;
            for (int l = 1; l <= n; l++) sol[pp][q] = sol[pp][q] + invL[l][pp] * V[l][q];
        }
    for (int j = 1; j <= n; j++) {
        double mod = 0.0;
        for (int i = 1; i <= n; i++) mod += sol[i][j] * sol[i][j];
        for (int i = 1; i <= n; i++) sol[i][j] /= sqrt(mod);
    }
    int solind = 0;
    for (int i = 1; i <= n; i++)
        if (d[i] < 0 && fabs(d[i]) > 10e-20) solind = i;
    sr_conic_to_ellipse(sol[1][solind], sol[2][solind], sol[3][solind], sol[4][solind], sol[5][solind], sol[6][solind], e);
    return true;
}

__device__ void sr_mul3(const double a[9], const double b[9], double d[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// cv::Mat::inv() of a 3 x 3 (closed form, as OpenCV's invert() does for n <= 3)
__device__ void sr_inv3(const double S[9], double D[9])
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    for (int k = 0; k < 9; k++) D[k] = 0.0;
    if (d != 0.) {
        d = 1. / d;
        D[0] = (S[4] * S[8] - S[5] * S[7]) * d; D[1] = (S[2] * S[7] - S[1] * S[8]) * d; D[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        D[3] = (S[5] * S[6] - S[3] * S[8]) * d; D[4] = (S[0] * S[8] - S[2] * S[6]) * d; D[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        D[6] = (S[3] * S[7] - S[4] * S[6]) * d; D[7] = (S[1] * S[6] - S[0] * S[7]) * d; D[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    }
}

// Refine::calc (PoseRefiner.cpp:224-258): x[i + 3 j] = H(i, j)
__device__ double sr_cost(const double x[9], const double Cm[9])
{
    double H[9], HT[9], T[9], P[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            H[3 * i + j] = x[i + j * 3];
            HT[3 * j + i] = x[i + j * 3];
        }
    sr_mul3(HT, Cm, T);
    sr_mul3(T, H, P);
    SrEllipse e;
    sr_conic_to_ellipse(P[0], -P[1] * 2, P[4], P[2] * 2, -P[5] * 2, P[8], &e);
    double acc = 0;
    acc += fabs(e.a - 0.4);
    acc += fabs(e.b - 0.4);
    acc += fabs(e.cX - 0.5);
    acc += fabs(-e.cY - 0.5);
    return acc;
}

// cv::DownhillSolver::minimize with the default TermCriteria(MAX_ITER + EPS, 5000, 1e-6), 9 dimensions, run by one wave with
// the simplex in LDS.  The three candidate points of an iteration -- reflection (-1), expansion (-2), contraction (0.5) --
// depend only on the current simplex: lanes 0, 1, 2 evaluate them side by side and the solver's decision sequence then
// picks what it would have evaluated one after the other (the evaluation counter advances as in the sequential solver).
// The ten vertices of the start simplex and of a shrink step are evaluated by ten lanes.  Same arithmetic per point as the
// sequential form (column sums in vertex order, the same alpha / beta expressions).
struct SrSimplex {
    double p[10][9], y[10], sum[9];
};

#define SR_LDS_SYNC()                                          \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

__device__ void sr_downhill(double x[9], const double step[9], const double Cm[9], int lane, SrSimplex *S)
{
    const int nd = 9;
    if (lane <= nd) {
        for (int j = 0; j < nd; j++) {
            double v = x[j];
            if (lane == 0) v -= 0.5 * step[j];
            else if (lane - 1 == j) v += 0.5 * step[j];
            S->p[lane][j] = v;
        }
    }
    SR_LDS_SYNC();
    auto eval_rows = [&](int skip) {  // y[i] = f(p[i]) for every vertex but `skip`, one lane per vertex
        const int i = lane <= nd ? lane : nd;
        double row[9];
        for (int j = 0; j < nd; j++) row[j] = S->p[i][j];
        const double v = sr_cost(row, Cm);
        if (lane <= nd && lane != skip) S->y[lane] = v;
        SR_LDS_SYNC();
    };
    auto update_sum = [&]() {
        if (lane < nd) {
            double acc = 0.;
            for (int i = 0; i <= nd; i++) acc += S->p[i][lane];
            S->sum[lane] = acc;
        }
        SR_LDS_SYNC();
    };
    auto replace_point = [&](int ihi, double alpha_, double ytry) {
        const double alpha = (1.0 - alpha_) / nd, beta = alpha - alpha_;
        if (lane < nd) S->p[ihi][lane] = S->sum[lane] * alpha - S->p[ihi][lane] * beta;
        if (lane == 0) S->y[ihi] = ytry;
        SR_LDS_SYNC();
        update_sum();
    };
    int fcount = nd + 1;
    eval_rows(-1);
    update_sum();
    for (;;) {
        double y[10];
        for (int i = 0; i <= nd; i++) y[i] = S->y[i];
        int ilo = 0, ihi, inhi;
        if (y[0] > y[1]) { ihi = 0; inhi = 1; } else { ihi = 1; inhi = 0; }
        for (int i = 0; i <= nd; i++) {
            const double yv = y[i];
            if (yv <= y[ilo]) ilo = i;
            if (yv > y[ihi]) { inhi = ihi; ihi = i; }
            else if (yv > y[inhi] && i != ihi) inhi = i;
        }
        if (ilo == inhi || ilo == ihi)
            for (int i = 0; i <= nd; i++)
                if (y[i] == y[ilo] && i != ihi && i != inhi) { ilo = i; break; }
        const double error = fabs(y[ihi] - y[ilo]);
        double range = 0;
        {
            double r = 0;
            if (lane < nd) {
                double mn = S->p[0][lane], mx = mn;
                for (int i = 1; i <= nd; i++) {
                    mn = fmin(mn, S->p[i][lane]);
                    mx = fmax(mx, S->p[i][lane]);
                }
                r = fabs(mx - mn);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) r = fmax(r, __shfl_xor(r, off, 64));
            range = __shfl(r, 0, 64);
        }
        if (range <= 0.000001 || error <= 0.000001 || fcount >= 5000) {
            for (int j = 0; j < nd; j++) x[j] = S->p[ilo][j];
            return;
        }
        const double y_lo = y[ilo], y_nhi = y[inhi], y_hi = y[ihi];
        double buf[9];
        {
            const double a_ = lane == 0 ? -1.0 : lane == 1 ? -2.0 : 0.5;
            const double alpha = (1.0 - a_) / nd, beta = alpha - a_;
            for (int j = 0; j < nd; j++) buf[j] = S->sum[j] * alpha - S->p[ihi][j] * beta;
        }
        const double yl = sr_cost(buf, Cm);
        const double y_refl = __shfl(yl, 0, 64), y_exp = __shfl(yl, 1, 64), y_con = __shfl(yl, 2, 64);
        fcount++;
        double alpha = -1.0, y_alpha = y_refl;
        if (y_alpha < y_nhi) {
            if (y_alpha < y_lo) {
                fcount++;
                if (y_exp < y_alpha) { alpha = -2.0; y_alpha = y_exp; }
            }
            replace_point(ihi, alpha, y_alpha);
        } else {
            fcount++;
            if (y_con < y_hi) replace_point(ihi, 0.5, y_con);
            else {
                if (lane <= nd && lane != ilo)
                    for (int j = 0; j < nd; j++) S->p[lane][j] = 0.5 * (S->p[lane][j] + S->p[ilo][j]);
                SR_LDS_SYNC();
                eval_rows(ilo);
                fcount += nd;
                update_sum();
            }
        }
    }
}

// PoseRefiner::checkIfPointInQuad (PoseRefiner.cpp:200-222)
__device__ bool sr_in_quad(const double c[8], double px, double py)
{
    const double c1c2x = c[2] - c[0], c1c2y = c[3] - c[1], c1c4x = c[6] - c[0], c1c4y = c[7] - c[1];
    const double c3c2x = c[2] - c[4], c3c2y = c[3] - c[5], c3c4x = c[6] - c[4], c3c4y = c[7] - c[5];
    const double c1px = px - c[0], c1py = py - c[1], c3px = px - c[4], c3py = py - c[5];
    if (sq_cross(c1px, c1py, c1c2x, c1c2y) * sq_cross(c1px, c1py, c1c4x, c1c4y) >= 0) return false;
    if (sq_cross(c1c2x, c1c2y, c1px, c1py) * sq_cross(c1c2x, c1c2y, c1c4x, c1c4y) <= 0) return false;
    if (sq_cross(c3px, c3py, c3c2x, c3c2y) * sq_cross(c3px, c3py, c3c4x, c3c4y) >= 0) return false;
    if (sq_cross(c3c2x, c3c2y, c3px, c3py) * sq_cross(c3c2x, c3c2y, c3c4x, c3c4y) <= 0) return false;
    return true;
}

__global__ __launch_bounds__(64) void k_stag_refine(fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers,
                                                    const int2 *__restrict__ vsegs, const int *__restrict__ nsegs, const int2 *__restrict__ pix,
                                                    int *__restrict__ chosen_out)
{
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= *nmarkers) return;
    fid_stag_marker M = markers[m];
    const double sinVals[36] = {0.000000,  0.173648,  0.342020,  0.500000,  0.642788,  0.766044,  0.866025,  0.939693,  0.984808,
                                1.000000,  0.984808,  0.939693,  0.866025,  0.766044,  0.642788,  0.500000,  0.342020,  0.173648,
                                0.000000,  -0.173648, -0.342020, -0.500000, -0.642788, -0.766044, -0.866025, -0.939693, -0.984808,
                                -1.000000, -0.984808, -0.939693, -0.866025, -0.766044, -0.642788, -0.500000, -0.342020, -0.173648};
    double Hinv[9];
    sr_inv3(M.H, Hinv);
    // ---- (1) the edge segment of the circular border
    int chosen = -1;
    double minAcc = INFINITY;
    const int ns = *nsegs;
    for (int sg = 0; sg < ns; sg++) {
        const int first = vsegs[sg].x, n = vsegs[sg].y;
        if (n < 20) continue;
        const int2 *p = pix + first;
        if (sq_dist2((double)p[0].y, (double)p[0].x, (double)p[n - 1].y, (double)p[n - 1].x) > 7.0 * 7.0) continue;
        bool outside = false;
        for (int k = 0; k < n && !outside; k += 20)
            if (!sr_in_quad(M.corners, (double)p[k].y, (double)p[k].x)) outside = true;
        if (outside) continue;
        // back-projection; per sample point the minimum over the pixels, per pixel the minimum over the sample points
        bool bad = false;
        double sampleErr[36];
        for (int s = 0; s < 36; s++) sampleErr[s] = INFINITY;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            double qx = 0, qy = 0;
            const bool act = k < n;
            if (act) {
                const double ex = p[k].y, ey = p[k].x;
                const double a0 = Hinv[0] * ex + Hinv[1] * ey + Hinv[2] * 1, a1 = Hinv[3] * ex + Hinv[4] * ey + Hinv[5] * 1;
                const double a2 = Hinv[6] * ex + Hinv[7] * ey + Hinv[8] * 1;
                qx = a0 / a2;
                qy = a1 / a2;
            }
            double pixErr = INFINITY;
            for (int s = 0; s < 36; s++) {
                const double sx = 0.5 + 0.4 * sinVals[(s + 9) % 36], sy = 0.5 + 0.4 * sinVals[s];
                const double d = act ? sqrt((qx - sx) * (qx - sx) + (qy - sy) * (qy - sy)) : INFINITY;
                if (d < pixErr) pixErr = d;
                double mn = d;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
                if (mn < sampleErr[s]) sampleErr[s] = mn;
            }
            if (__ballot(act && pixErr > 0.1)) bad = true;
        }
        if (bad) continue;
        double errSum = 0;
        for (int s = 0; s < 36; s++) errSum += sampleErr[s];
        if (errSum < minAcc && errSum < 36 * 0.05) {
            minAcc = errSum;
            chosen = sg;
        }
    }
    if (lane == 0) chosen_out[m] = chosen;
    if (chosen < 0) return;
    // ---- (2) ellipse through the chosen segment: scatter matrix, one lane per entry (p <= q), summed in pixel order
    __shared__ double s_S[7][7];
    {
        const int first = vsegs[chosen].x, n = vsegs[chosen].y;
        const int2 *p = pix + first;
        if (lane < 49) s_S[lane / 7][lane % 7] = 0.0;
        __builtin_amdgcn_wave_barrier();
        int pi = 0, qi = 0, idx = lane;
        bool mine = false;
        for (int a = 1; a <= 6 && !mine; a++)
            for (int b = a; b <= 6; b++) {
                if (idx == 0) { pi = a; qi = b; mine = true; break; }
                idx--;
            }
        if (mine) {
            double acc = 0.0;
            for (int l = 0; l < n; l++) {
                const double tx = (double)p[l].y, ty = (double)(-p[l].x);
                const double Dl[7] = {0, tx * tx, tx * ty, ty * ty, tx, ty, 1.0};
                acc = acc + Dl[pi] * Dl[qi];
            }
            s_S[pi][qi] = acc;
            s_S[qi][pi] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        if (n < 6) return;
    }
    // from here on every lane carries the same values (the fit is small scalar work; the simplex search spreads its function
    // evaluations over the lanes)
    __shared__ SrSimplex s_simplex;
    SrM S;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) S[i][j] = s_S[i][j];
    SrEllipse E;
    if (!sr_fit_from_scatter(S, &E)) return;
    double Cm[9];
    Cm[0] = E.A1; Cm[1] = Cm[3] = -E.B1 / 2; Cm[4] = E.C1; Cm[2] = Cm[6] = E.D1 / 2; Cm[5] = Cm[7] = -E.E1 / 2; Cm[8] = E.F1;
    // ---- (3) Nelder-Mead over the entries of H
    double x[9], step[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) x[i + j * 3] = M.H[3 * i + j];
    for (int k = 0; k < 9; k++) step[k] = fabs(0.001 * x[k]);
    sr_downhill(x, step, Cm, lane, &s_simplex);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M.H[3 * i + j] = x[i + j * 3];
    // ---- (4) points from the refined H
    auto project = [&](double px, double py, double *ox, double *oy) {
        const double a0 = M.H[0] * px + M.H[1] * py + M.H[2] * 1, a1 = M.H[3] * px + M.H[4] * py + M.H[5] * 1, a2 = M.H[6] * px + M.H[7] * py + M.H[8] * 1;
        *ox = a0 / a2;
        *oy = a1 / a2;
    };
    project(0.5, 0.5, &M.center[0], &M.center[1]);
    project(0, 0, &M.corners[0], &M.corners[1]);
    project(1, 0, &M.corners[2], &M.corners[3]);
    project(1, 1, &M.corners[4], &M.corners[5]);
    project(0, 1, &M.corners[6], &M.corners[7]);
    if (lane == 0) markers[m] = M;
}

// ------------------------------------------------------------------------------------------------ K17: marker pose
// StagNode::imageCallback -> Common::solvePnpSingle (stag_detect.cpp:140-165, common.hpp:34-46): cv::solvePnP (ITERATIVE) on
// FIVE coplanar points, the marker centre (0, 0, 0) and the four corners (-h, h) (h, h) (h, -h) (-h, -h), h = float(marker_size /
// 2).  Same scheme as the aruco pose kernel (fid_kernels.hip K8): closed-form start from the four corners, then the reference's
// Levenberg-Marquardt (CvLevMarq: <= 20 iterations, lambda 1e-3 x 10^k, same accept / reject rule) on the reprojection error
// of all five points with distortion; a 16-lane group per marker, lane g < 10 owns residual g.  Tolerance row (the reference
// starts from a 5-point DLT + refinement; both land on the same minimum).
__device__ __forceinline__ double grp_sum16(double v)
{
    v += shfl_xor_f64(v, 1);
    v += shfl_xor_f64(v, 2);
    v += shfl_xor_f64(v, 4);
    v += shfl_xor_f64(v, 8);
    return v;
}

__device__ void sp_undistort(const double K[9], const double kd[5], double u, double v, double *ox, double *oy)
{
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1) / (1 + ((kd[4] * r2 + kd[1]) * r2 + kd[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * kd[2] * x * y + kd[3] * (r2 + 2 * x * x);
        const double deltaY = kd[2] * (r2 + 2 * y * y) + 2 * kd[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = x;
    *oy = y;
}

__global__ __launch_bounds__(64) void k_stag_pose(const fid_stag_marker *__restrict__ markers, const int *__restrict__ nmarkers, PoseCam cam,
                                                  double marker_size, fid_stag_pose_out *__restrict__ out)
{
    const int item = blockIdx.x * 4 + (threadIdx.x >> 4), g = threadIdx.x & 15;
    if (item >= *nmarkers) return;  // group-uniform
    const fid_stag_marker mk = markers[item];
    const double *K = cam.K, *kd = cam.D;
    const float halff = (float)(marker_size / 2.0);
    const double hx = (double)halff;
    const bool act = g < 10;
    const int pi = act ? g >> 1 : 0, sel = g & 1;
    // object point of this lane: 0 centre, 1..4 corners
    double M[3] = {0., 0., 0.};
    if (pi >= 1) {
        M[0] = (pi == 2 || pi == 3) ? hx : -hx;
        M[1] = (pi <= 2) ? hx : -hx;
    }
    const double mobs = pi == 0 ? mk.center[sel] : mk.corners[2 * (pi - 1) + sel];
    double param[6];
    {
        double mnx[4], mny[4];
        for (int i = 0; i < 4; i++) {
            double x, y;
            sp_undistort(K, kd, mk.corners[2 * i], mk.corners[2 * i + 1], &x, &y);
            mnx[i] = x;
            mny[i] = y;
        }
        // homography marker plane -> normalised image through the four corners (unit square -> quad, composed with
        // (X, Y) -> ((X + h) / 2h, (h - Y) / 2h)), then R, t from its columns
        const double x0 = mnx[0], y0 = mny[0], x1 = mnx[1], y1 = mny[1], x2 = mnx[2], y2 = mny[2], x3 = mnx[3], y3 = mny[3];
        const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
        const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
        const double den = dx1 * dy2 - dy1 * dx2;
        double h[9];
        bool okh = den != 0.;
        if (okh) {
            const double gg = (sx * dy2 - sy * dx2) / den, hh = (dx1 * sy - dy1 * sx) / den;
            const double a = x1 - x0 + gg * x1, b = x3 - x0 + hh * x3, c = x0;
            const double d = y1 - y0 + gg * y1, e = y3 - y0 + hh * y3, ff = y0;
            const double sc0 = 1. / (2. * hx);
            h[0] = a * sc0;  h[1] = -b * sc0;  h[2] = 0.5 * a + 0.5 * b + c;
            h[3] = d * sc0;  h[4] = -e * sc0;  h[5] = 0.5 * d + 0.5 * e + ff;
            h[6] = gg * sc0; h[7] = -hh * sc0; h[8] = 0.5 * gg + 0.5 * hh + 1.;
            okh = h[8] != 0.;
            if (okh) {
                const double sc = 1. / h[8];
                for (int i = 0; i < 9; i++) h[i] *= sc;
            }
        }
        double R[9];
        param[3] = param[4] = param[5] = 0.;
        if (okh) {
            const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / fmax(h1n, DBL_EPSILON), s2 = 1. / fmax(h2n, DBL_EPSILON), stt = 2. / fmax(h1n + h2n, DBL_EPSILON);
            param[3] = h[2] * stt; param[4] = h[5] * stt; param[5] = h[8] * stt;
            h[0] *= s1; h[3] *= s1; h[6] *= s1;
            h[1] *= s2; h[4] *= s2; h[7] *= s2;
            h[2] = h[3] * h[7] - h[6] * h[4];
            h[5] = h[6] * h[1] - h[0] * h[7];
            h[8] = h[0] * h[4] - h[3] * h[1];
            double rtmp[3], dummy[27];
            rodrigues_m2v(h, rtmp);
            rodrigues_v2m(rtmp, R, dummy, false);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        }
        rodrigues_m2v(R, param);
    }
    // ---- CvLevMarq over the 10 residuals
    double prevParam[6], S[21], gJ[6], Jrow[6] = {0, 0, 0, 0, 0, 0};
    double err = 0, prevErrNorm = 0, errNorm = 0;
    int lambdaLg10 = -3, iters = 0, state = 1;
    const double LOG10 = log(10.);
    for (int i = 0; i < 6; i++) prevParam[i] = param[i];
    for (;;) {
        bool needJ = false, needErr = false;
        if (state == 1) {
            needJ = needErr = true;
            state = 2;
        } else if (state == 2) {
            int idx = 0;
            for (int a = 0; a < 6; a++) {
                for (int b = a; b < 6; b++) S[idx++] = grp_sum16(Jrow[a] * Jrow[b]);
                gJ[a] = grp_sum16(Jrow[a] * err);
            }
            for (int i = 0; i < 6; i++) prevParam[i] = param[i];
            double xs[6];
            solve6_spd(S, gJ, exp(lambdaLg10 * LOG10), xs);
            for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
            if (iters == 0) prevErrNorm = sqrt(grp_sum16(err * err));
            needErr = true;
            state = 3;
        } else {
            errNorm = sqrt(grp_sum16(err * err));
            bool retry = false;
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    double xs[6];
                    solve6_spd(S, gJ, exp(lambdaLg10 * LOG10), xs);
                    for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
                    needErr = true;
                    state = 3;
                    retry = true;
                }
            }
            if (!retry) {
                lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
                double dn = 0, pn = 0;
                for (int i = 0; i < 6; i++) {
                    dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
                    pn += prevParam[i] * prevParam[i];
                }
                const double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
                if (++iters >= 20 || rel < FLT_EPSILON) break;
                prevErrNorm = errNorm;
                needJ = needErr = true;
                state = 2;
            }
        }
        if (!needErr) break;
        const double pr = project_one(M, param, K, kd, sel, Jrow, needJ);
        err = act ? pr - mobs : 0.;
        if (!act)
            for (int i = 0; i < 6; i++) Jrow[i] = 0.;
    }
    if (g == 0) {
        fid_stag_pose_out o;
        o.id = mk.id;
        for (int i = 0; i < 3; i++) {
            o.rvec[i] = param[i];
            o.tvec[i] = param[3 + i];
        }
        double dummy[27];
        rodrigues_v2m(param, o.R, dummy, false);  // cv::Rodrigues(rVec, rMat) of solvePnpSingle
        out[item] = o;
    }
}

// ------------------------------------------------------------------------------------------------ C-ABI
// ---- host side of the line validation: the number-of-false-alarms table.  nfa() restates NFA.cpp:155-239 (the LSD
// binomial-tail bound: log-gamma by Windschitl / Lanczos, series with a 10 % truncation tolerance); the table is what
// NFALUT::NFALUT builds (NFA.cpp:13-44), continued past LUTSize where the reference calls nfa(n, k) directly (assumes, like
// the table itself, that nfa grows with k).
static double stag_log_gamma(double x)
{
    if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0.0;
    for (int n = 0; n < 7; n++) {
        a -= log(x + (double)n);
        b += q[n] * pow(x, (double)n);
    }
    return a + log(b);
}

static bool stag_double_equal(double a, double b)
{
    if (a == b) return true;
    const double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = aa > bb ? aa : bb;
    if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;
    return (abs_diff / abs_max) <= (100.0 * 2.2204460492503131e-16);
}

static double stag_nfa(int n, int k, double p, double logNT)
{
    const double tolerance = 0.1, LN10 = 2.30258509299404568402;
    if (n < 0 || k < 0 || k > n || p <= 0.0 || p >= 1.0) return -1.0;
    if (n == 0 || k == 0) return -logNT;
    if (n == k) return -logNT - (double)n * log10(p);
    const double p_term = p / (1.0 - p);
    const double log1term = stag_log_gamma((double)n + 1.0) - stag_log_gamma((double)k + 1.0) - stag_log_gamma((double)(n - k) + 1.0) +
                            (double)k * log(p) + (double)(n - k) * log(1.0 - p);
    double term = exp(log1term);
    if (stag_double_equal(term, 0.0)) {
        if ((double)k > (double)n * p) return -log1term / LN10 - logNT;
        return -logNT;
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; i++) {
        const double bin_term = (double)(n - i + 1) * (1.0 / (double)i);
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1.0) {
            const double err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
            if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - logNT;
}

static void stag_build_kmin(int W, int H, std::vector<int> &kmin)
{
    const double prob = 0.125, logNT = 2.0 * (log10((double)W) + log10((double)H));
    const int lutSize = (W + H) / 8, nmax = 4 * (W + H) + 8;
    kmin.assign(nmax + 1, 0);
    kmin[0] = 1;
    int j = 1;
    for (int i = 1; i <= nmax; i++) {
        kmin[i] = (i < lutSize ? lutSize : i) + 1;  // "never": more aligned pixels than there are pixels
        double ret = stag_nfa(i, j, prob, logNT);
        if (ret < 0) {
            while (j < i) {
                j++;
                ret = stag_nfa(i, j, prob, logNT);
                if (ret >= 0) break;
            }
            if (ret < 0) continue;
        }
        kmin[i] = j;
    }
}

// Stag::fillCodeLocations (Stag.cpp:129-277) + createMatFromPolarCoords (:279-286): the 48 code points on three rings inside
// the circle, 12 points on the black border, 12 outside it, in marker coordinates [0, 1]^2, homogeneous.
static void stag_fill_code_locations(double *locs /* [72][3] */)
{
    const double HALF_PI = 1.570796326794897;
    const double outerCircleRadius = 0.4;
    const double innerCircleRadius = outerCircleRadius * 0.9;
    auto polar = [&](int idx, double radius, double radians) {
        locs[3 * idx + 0] = 0.5 + cos(radians) * radius * (innerCircleRadius / 0.5);
        locs[3 * idx + 1] = 0.5 - sin(radians) * radius * (innerCircleRadius / 0.5);
        locs[3 * idx + 2] = 1;
    };
    for (int i = 0; i < 4; i++) {
        polar(0 + i * 12, 0.088363142525988, 0.785398163397448 + i * HALF_PI);
        polar(1 + i * 12, 0.206935928182607, 0.459275804122858 + i * HALF_PI);
        polar(2 + i * 12, 0.206935928182607, HALF_PI - 0.459275804122858 + i * HALF_PI);
        polar(3 + i * 12, 0.313672146827381, 0.200579720495241 + i * HALF_PI);
        polar(4 + i * 12, 0.327493143484516, 0.591687617505840 + i * HALF_PI);
        polar(5 + i * 12, 0.327493143484516, HALF_PI - 0.591687617505840 + i * HALF_PI);
        polar(6 + i * 12, 0.313672146827381, HALF_PI - 0.200579720495241 + i * HALF_PI);
        polar(7 + i * 12, 0.437421957035861, 0.145724938287167 + i * HALF_PI);
        polar(8 + i * 12, 0.437226762361658, 0.433363129825345 + i * HALF_PI);
        polar(9 + i * 12, 0.430628029742607, 0.785398163397448 + i * HALF_PI);
        polar(10 + i * 12, 0.437226762361658, HALF_PI - 0.433363129825345 + i * HALF_PI);
        polar(11 + i * 12, 0.437421957035861, HALF_PI - 0.145724938287167 + i * HALF_PI);
    }
    const double b = 0.045;  // borderDist
    const double black[12][2] = {{b, b * 3}, {b * 2, b * 2}, {b * 3, b}, {1 - 3 * b, b}, {1 - 2 * b, b * 2}, {1 - b, b * 3},
                                 {1 - b, 1 - 3 * b}, {1 - 2 * b, 1 - 2 * b}, {1 - 3 * b, 1 - b}, {b * 3, 1 - b}, {b * 2, 1 - 2 * b}, {b, 1 - 3 * b}};
    const double white[12][2] = {{0.25, -b}, {0.5, -b}, {0.75, -b}, {1 + b, 0.25}, {1 + b, 0.5}, {1 + b, 0.75},
                                 {0.75, 1 + b}, {0.5, 1 + b}, {0.25, 1 + b}, {-b, 0.75}, {-b, 0.5}, {-b, 0.25}};
    for (int i = 0; i < 12; i++) {
        locs[3 * (48 + i) + 0] = black[i][0]; locs[3 * (48 + i) + 1] = black[i][1]; locs[3 * (48 + i) + 2] = 1;
        locs[3 * (60 + i) + 0] = white[i][0]; locs[3 * (60 + i) + 1] = white[i][1]; locs[3 * (60 + i) + 2] = 1;
    }
}

struct fid_stag_ctx {
    int device = 0, maxW = 0, maxH = 0, libraryHD = 0, errorCorrection = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_src = nullptr, *d_smooth = nullptr, *d_dir = nullptr, *d_edge = nullptr;
    int16_t *d_grad = nullptr;
    unsigned *d_rowhist = nullptr, *d_bandhist = nullptr, *d_tot = nullptr, *d_bstart = nullptr, *d_n = nullptr;
    int32_t *d_sorted = nullptr;
    uint8_t *d_edgeimg = nullptr;  // edge image of the routing (starts as a copy of the anchor map)
    int2 *d_rpix = nullptr, *d_outpix = nullptr, *d_segs = nullptr;
    int4 *d_rstack = nullptr;
    StagChain *d_chains = nullptr;
    int *d_chainnos = nullptr, *d_rcount = nullptr;
    int rcount[3] = {0, 0, 0};
    bool routed = false;
    // component-parallel routing
    int *d_label = nullptr, *d_csize = nullptr, *d_canch = nullptr, *d_cidmap = nullptr, *d_cursors = nullptr, *d_caps = nullptr;
    int *d_fill = nullptr, *d_aslots = nullptr, *d_prodflag = nullptr, *d_next = nullptr, *d_blkpix = nullptr, *d_blksegs = nullptr;
    int2 *d_blkwhere = nullptr, *d_apix = nullptr, *d_aout = nullptr, *d_asegs = nullptr;
    int4 *d_astack = nullptr;
    StagChain *d_achains = nullptr;
    StagComp *d_comps = nullptr;
    StagRec *d_recs = nullptr;
    int max_comps = 0, cap_aslots = 0, route_mode = 1, route_fallbacks = 0;
    // validation
    uint8_t *d_smooth2 = nullptr;
    int16_t *d_vgrad = nullptr;
    unsigned *d_vhist = nullptr;
    double *d_prob = nullptr;
    int *d_np = nullptr, *d_vcounts = nullptr, *d_vtotal = nullptr;
    int2 *d_vstack = nullptr, *d_vsegs = nullptr;
    int n_vsegs = 0, np = 0;
    bool validated = false;
    // EDLines
    long long *d_prefix = nullptr;  // 5 arrays of prefcap entries
    size_t prefcap = 0;
    fid_stag_line *d_lslots = nullptr, *d_lines = nullptr;
    int *d_lcounts = nullptr, *d_ltotal = nullptr;
    int n_lines = 0, min_line_len = 0;
    bool lined = false;
    // line validation
    double *d_atan_lut = nullptr;
    int *d_kmin = nullptr, *d_lflags = nullptr, *d_vltotal = nullptr;
    fid_stag_line *d_vlines = nullptr;
    int kmin_w = 0, kmin_h = 0, kmin_n = 0, n_vlines = 0;
    bool lines_validated = false;
    // quads
    int2 *d_lrange = nullptr;
    StagCorner *d_corners = nullptr;
    int *d_order = nullptr, *d_qcounts = nullptr, *d_qtotal = nullptr;
    fid_stag_quad *d_qslots = nullptr, *d_quads = nullptr;
    int n_quads = 0;
    bool quadded = false;
    // decoding
    double *d_locs = nullptr;
    unsigned long long *d_words = nullptr;
    int n_words = 0;
    fid_stag_marker *d_cand = nullptr, *d_markers = nullptr;
    int *d_found = nullptr, *d_nmarkers = nullptr;
    int n_markers = 0;
    bool decoded = false;
    int *d_chosen = nullptr;
    fid_stag_pose_out *d_poses = nullptr;
    int W = 0, H = 0;
    unsigned n_anchors = 0;
};

extern "C" {

fid_status fid_stag_create(int libraryHD, int errorCorrection, int max_width, int max_height, int device, fid_stag_ctx **out)
{
    if (!out || max_width < 8 || max_height < 8 || max_width > 8191 || max_height > 8191) return FID_E_INVALID_ARG;
    // Decoder.cpp:14-37: libraries HD11 ... HD23 (odd), errorCorrection <= (HD - 1) / 2
    if (libraryHD < 11 || libraryHD > 23 || !(libraryHD & 1) || errorCorrection < 0 || errorCorrection > (libraryHD - 1) / 2)
        return FID_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FID_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return FID_E_INVALID_ARG;
    fid_stag_ctx *c = new (std::nothrow) fid_stag_ctx();
    if (!c) return FID_E_OUT_OF_MEMORY;
    c->device = device;
    c->maxW = max_width;
    c->maxH = max_height;
    c->libraryHD = libraryHD;
    c->errorCorrection = errorCorrection;
    const size_t n = (size_t)max_width * max_height;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_src, n) == hipSuccess && hipMalloc((void **)&c->d_smooth, n) == hipSuccess &&
         hipMalloc((void **)&c->d_dir, n) == hipSuccess && hipMalloc((void **)&c->d_edge, n) == hipSuccess &&
         hipMalloc((void **)&c->d_grad, n * 2) == hipSuccess && hipMalloc((void **)&c->d_sorted, n * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_rowhist, (size_t)max_height * STAG_BINS * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_bandhist, (size_t)((max_height + STAG_BAND_ROWS - 1) / STAG_BAND_ROWS) * STAG_BINS * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_tot, STAG_BINS * 4) == hipSuccess && hipMalloc((void **)&c->d_bstart, STAG_BINS * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_n, 4) == hipSuccess;
    // routing: the reference sizes its scratch arrays for the worst case (width * height entries each, EDInternals.cpp:848-853)
    ok = ok && hipMalloc((void **)&c->d_edgeimg, n) == hipSuccess && hipMalloc((void **)&c->d_rpix, n * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_outpix, n * sizeof(int2)) == hipSuccess && hipMalloc((void **)&c->d_segs, (n / 8 + 16) * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_rstack, n * sizeof(int4)) == hipSuccess && hipMalloc((void **)&c->d_chains, 32767 * sizeof(StagChain)) == hipSuccess &&
         hipMalloc((void **)&c->d_chainnos, (size_t)(max_width + max_height) * 8 * sizeof(int)) == hipSuccess &&
         hipMalloc((void **)&c->d_rcount, 16) == hipSuccess;
    // component-parallel routing: labels, per-root counters, component table, arenas (sizes in entries; see k_stag_comp_alloc)
    c->max_comps = (int)(n / 8 + 64);
    c->cap_aslots = (int)(n / 2 + 64);
    ok = ok && hipMalloc((void **)&c->d_label, n * 4) == hipSuccess && hipMalloc((void **)&c->d_csize, n * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_canch, n * 4) == hipSuccess && hipMalloc((void **)&c->d_cidmap, n * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_cursors, 64) == hipSuccess && hipMalloc((void **)&c->d_caps, 64) == hipSuccess &&
         hipMalloc((void **)&c->d_fill, (size_t)c->max_comps * 4) == hipSuccess && hipMalloc((void **)&c->d_aslots, (size_t)c->cap_aslots * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_prodflag, n * 4) == hipSuccess && hipMalloc((void **)&c->d_next, n * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_blkpix, n * 4) == hipSuccess && hipMalloc((void **)&c->d_blksegs, n * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_blkwhere, n * sizeof(int2)) == hipSuccess && hipMalloc((void **)&c->d_apix, 3 * n * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_aout, 3 * n * sizeof(int2)) == hipSuccess && hipMalloc((void **)&c->d_asegs, (n / 2 + 64) * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_astack, 2 * n * sizeof(int4)) == hipSuccess && hipMalloc((void **)&c->d_achains, 2 * n * sizeof(StagChain)) == hipSuccess &&
         hipMalloc((void **)&c->d_comps, (size_t)c->max_comps * sizeof(StagComp)) == hipSuccess &&
         hipMalloc((void **)&c->d_recs, (size_t)c->cap_aslots * sizeof(StagRec)) == hipSuccess;
    if (ok) {
        const int caps[16] = {c->max_comps, c->cap_aslots, (int)(3 * n), (int)(2 * n), (int)(2 * n), (int)(3 * n), (int)(n / 2 + 64), 0};
        ok = hipMemcpy(c->d_caps, caps, sizeof(caps), hipMemcpyHostToDevice) == hipSuccess;
        const char *e = getenv("FID_STAG_ROUTE");
        c->route_mode = (e && !strcmp(e, "seq")) ? 0 : 1;
    }
    ok = ok && hipMalloc((void **)&c->d_smooth2, n) == hipSuccess && hipMalloc((void **)&c->d_vgrad, n * 2) == hipSuccess &&
         hipMalloc((void **)&c->d_vhist, STAG_BINS * 4) == hipSuccess && hipMalloc((void **)&c->d_prob, STAG_BINS * 8) == hipSuccess &&
         hipMalloc((void **)&c->d_np, 4) == hipSuccess && hipMalloc((void **)&c->d_vcounts, (n / 8 + 16) * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_vtotal, 4) == hipSuccess && hipMalloc((void **)&c->d_vstack, n * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_vsegs, (n / 8 + 16) * sizeof(int2)) == hipSuccess;
    c->prefcap = n + n / 8 + 64;
    ok = ok && hipMalloc((void **)&c->d_prefix, c->prefcap * 5 * sizeof(long long)) == hipSuccess &&
         hipMalloc((void **)&c->d_lslots, (n / 9 + 16) * sizeof(fid_stag_line)) == hipSuccess &&
         hipMalloc((void **)&c->d_lines, (n / 9 + 16) * sizeof(fid_stag_line)) == hipSuccess &&
         hipMalloc((void **)&c->d_lcounts, (n / 8 + 16) * 4) == hipSuccess && hipMalloc((void **)&c->d_ltotal, 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_atan_lut, 1025 * 8) == hipSuccess &&
         hipMalloc((void **)&c->d_kmin, (size_t)(4 * (max_width + max_height) + 16) * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_lflags, (n / 9 + 16) * 4) == hipSuccess && hipMalloc((void **)&c->d_vltotal, 4) == hipSuccess &&
         hipMalloc((void **)&c->d_vlines, (n / 9 + 16) * sizeof(fid_stag_line)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_lrange, (n / 8 + 16) * sizeof(int2)) == hipSuccess &&
         hipMalloc((void **)&c->d_corners, (n / 9 + 16) * sizeof(StagCorner)) == hipSuccess &&
         hipMalloc((void **)&c->d_order, (n / 9 + 16) * 4) == hipSuccess && hipMalloc((void **)&c->d_qcounts, (n / 8 + 16) * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_qtotal, 4) == hipSuccess && hipMalloc((void **)&c->d_qslots, (n / 9 + 16) * sizeof(fid_stag_quad)) == hipSuccess &&
         hipMalloc((void **)&c->d_quads, (n / 9 + 16) * sizeof(fid_stag_quad)) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_locs, 72 * 3 * 8) == hipSuccess && hipMalloc((void **)&c->d_cand, (n / 9 + 16) * sizeof(fid_stag_marker)) == hipSuccess &&
         hipMalloc((void **)&c->d_markers, (n / 9 + 16) * sizeof(fid_stag_marker)) == hipSuccess &&
         hipMalloc((void **)&c->d_found, (n / 9 + 16) * 4) == hipSuccess && hipMalloc((void **)&c->d_nmarkers, 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_chosen, (n / 9 + 16) * 4) == hipSuccess &&
         hipMalloc((void **)&c->d_poses, (n / 9 + 16) * sizeof(fid_stag_pose_out)) == hipSuccess;
    if (ok) {
        double locs[72 * 3];
        stag_fill_code_locations(locs);
        ok = hipMemcpy(c->d_locs, locs, sizeof(locs), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        double lut[1025];
        for (int i = 0; i <= 1024; i++) lut[i] = atan((double)i / 1024);
        ok = hipMemcpy(c->d_atan_lut, lut, sizeof(lut), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        fid_stag_destroy(c);
        return FID_E_OUT_OF_MEMORY;
    }
    *out = c;
    return FID_OK;
}

void fid_stag_destroy(fid_stag_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_src, c->d_smooth, c->d_dir, c->d_edge, c->d_grad, c->d_sorted, c->d_rowhist, c->d_bandhist, c->d_tot, c->d_bstart, c->d_n,
                   c->d_edgeimg, c->d_rpix, c->d_outpix, c->d_segs, c->d_rstack, c->d_chains, c->d_chainnos, c->d_rcount,
                   c->d_label, c->d_csize, c->d_canch, c->d_cidmap, c->d_cursors, c->d_caps, c->d_fill, c->d_aslots, c->d_prodflag, c->d_next,
                   c->d_blkpix, c->d_blksegs, c->d_blkwhere, c->d_apix, c->d_aout, c->d_asegs, c->d_astack, c->d_achains, c->d_comps, c->d_recs,
                   c->d_smooth2, c->d_vgrad, c->d_vhist, c->d_prob, c->d_np, c->d_vcounts, c->d_vtotal, c->d_vstack, c->d_vsegs,
                   c->d_prefix, c->d_lslots, c->d_lines, c->d_lcounts, c->d_ltotal,
                   c->d_atan_lut, c->d_kmin, c->d_lflags, c->d_vltotal, c->d_vlines,
                   c->d_lrange, c->d_corners, c->d_order, c->d_qcounts, c->d_qtotal, c->d_qslots, c->d_quads,
                   c->d_locs, c->d_words, c->d_cand, c->d_markers, c->d_found, c->d_nmarkers, c->d_chosen, c->d_poses};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

fid_status fid_stag_edge_frontend(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    if (!c || !gray || width < 8 || height < 8 || width > c->maxW || height > c->maxH || stride < width) return FID_E_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    hipStream_t st = c->stream;
    const int W = width, H = height;
    if (hipMemcpy2DAsync(c->d_src, (size_t)W, gray, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice, st) != hipSuccess) return FID_E_HIP;
    if (hipMemsetAsync(c->d_rowhist, 0, (size_t)H * STAG_BINS * 4, st) != hipSuccess) return FID_E_HIP;
    const int GRADIENT_THRESH = 16, ANCHOR_THRESH = 0, SCAN_INTERVAL = 1;  // DetectEdgesByEDPF, ED.cpp:155-169
    hipLaunchKernelGGL(k_stag_smooth_grad, dim3((W + SX - 1) / SX, (H + SY - 1) / SY), dim3(256), 0, st, c->d_src, W, W, H, GRADIENT_THRESH,
                       c->d_smooth, c->d_grad, c->d_dir);
    const int blocks = 2048, nbands = (H + STAG_BAND_ROWS - 1) / STAG_BAND_ROWS;
    hipLaunchKernelGGL(k_stag_anchors, dim3(blocks), dim3(256), 0, st, c->d_grad, c->d_dir, W, H, GRADIENT_THRESH, ANCHOR_THRESH, SCAN_INTERVAL,
                       c->d_edge, c->d_rowhist);
    hipLaunchKernelGGL(k_stag_bandsum, dim3(STAG_BINS / 256, nbands), dim3(256), 0, st, c->d_rowhist, H, c->d_bandhist);
    hipLaunchKernelGGL(k_stag_bandscan, dim3(STAG_BINS / 256), dim3(256), 0, st, c->d_bandhist, nbands, c->d_tot);
    hipLaunchKernelGGL(k_stag_scan, dim3(1), dim3(512), 0, st, c->d_tot, c->d_bstart, c->d_n);
    hipLaunchKernelGGL(k_stag_place, dim3(nbands), dim3(64 * STAG_BAND_ROWS), 0, st, c->d_grad, c->d_edge, W, H, c->d_rowhist, c->d_bandhist,
                       c->d_bstart, c->d_sorted);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_anchors, c->d_n, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->W = W;
    c->H = H;
    c->routed = false;
    return FID_OK;
}

// sequential routing: one lane for the whole frame (the reference's loop as it stands)
static fid_status stag_route_seq(fid_stag_ctx *c, const StagRoute &R)
{
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(k_stag_route_seq, dim3(1), dim3(64), 0, st, R, c->d_sorted, c->d_n, 16);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(c->rcount, c->d_rcount, 12, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    return c->rcount[2] ? FID_E_CAPACITY : FID_OK;
}

// component-parallel routing; FID_E_CAPACITY if an arena was too small (the caller then takes the sequential road)
static fid_status stag_route_par(fid_stag_ctx *c, const StagRoute &R)
{
    hipStream_t st = c->stream;
    const int W = c->W, H = c->H, n = W * H, na = (int)c->n_anchors;
    if (na == 0) {
        c->rcount[0] = c->rcount[1] = c->rcount[2] = 0;
        return hipMemsetAsync(c->d_rcount, 0, 12, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess ? FID_OK : FID_E_HIP;
    }
    const int nb = (n + 255) / 256;
    bool ok = hipMemsetAsync(c->d_csize, 0, (size_t)n * 4, st) == hipSuccess && hipMemsetAsync(c->d_canch, 0, (size_t)n * 4, st) == hipSuccess &&
              hipMemsetAsync(c->d_cursors, 0, 64, st) == hipSuccess && hipMemsetAsync(c->d_fill, 0, (size_t)c->max_comps * 4, st) == hipSuccess &&
              hipMemsetAsync(c->d_aslots, 0xff, (size_t)c->cap_aslots * 4, st) == hipSuccess &&
              hipMemsetAsync(c->d_prodflag, 0, (size_t)na * 4, st) == hipSuccess && hipMemsetAsync(c->d_blkpix, 0, (size_t)na * 4, st) == hipSuccess &&
              hipMemsetAsync(c->d_blksegs, 0, (size_t)na * 4, st) == hipSuccess &&
              hipMemsetAsync(c->d_aout, 0xff, (size_t)3 * n * sizeof(int2), st) == hipSuccess;
    if (!ok) return FID_E_HIP;
    hipLaunchKernelGGL(k_stag_ccl_init, dim3(nb), dim3(256), 0, st, c->d_grad, n, 16, c->d_label);
    hipLaunchKernelGGL(k_stag_ccl_merge, dim3(nb), dim3(256), 0, st, W, H, c->d_label);
    hipLaunchKernelGGL(k_stag_ccl_flatten, dim3(nb), dim3(256), 0, st, n, c->d_label, c->d_edge, c->d_csize, c->d_canch);
    hipLaunchKernelGGL(k_stag_comp_alloc, dim3(nb), dim3(256), 0, st, n, c->d_label, c->d_csize, c->d_canch, c->d_cursors, c->max_comps, c->d_caps,
                       c->d_comps, c->d_cidmap);
    hipLaunchKernelGGL(k_stag_comp_fill, dim3((na + 255) / 256), dim3(256), 0, st, c->d_sorted, c->d_n, c->d_label, c->d_cidmap, c->d_comps, c->d_fill,
                       c->d_aslots);
    int cur[10];
    if (hipMemcpyAsync(cur, c->d_cursors, 40, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    if (cur[7]) return FID_E_CAPACITY;
    // one component holding (nearly) all anchors -- a frame of noise -- leaves nothing to run side by side, and sorting its
    // anchors would cost more than the sequential road's single pass over the globally sorted list
    if (cur[9] > 65536) return FID_E_CAPACITY;
    const int nc = cur[0];
    StagArenas A;
    A.pix = c->d_apix; A.stack = c->d_astack; A.chains = c->d_achains; A.out = c->d_aout; A.segs = c->d_asegs; A.recs = c->d_recs;
    int *ovf = c->d_cursors + 8;
    if (nc > 0) {
        hipLaunchKernelGGL(k_stag_comp_sort, dim3((nc + 3) / 4), dim3(256), 0, st, c->d_comps, c->d_cursors, c->d_aslots);
        hipLaunchKernelGGL(k_stag_route_walk, dim3((nc + 3) / 4), dim3(256), 0, st, R, A, c->d_comps, c->d_cursors, c->d_sorted, c->d_aslots, 16,
                           c->d_prodflag, ovf);
    }
    hipLaunchKernelGGL(k_stag_next_above, dim3(1), dim3(1024), 0, st, c->d_prodflag, c->d_n, c->d_next);
    if (nc > 0)
        hipLaunchKernelGGL(k_stag_route_extract, dim3((nc + 3) / 4), dim3(256), 0, st, R, A, c->d_comps, c->d_cursors, c->d_next, c->d_n, c->d_blkpix,
                           c->d_blksegs, c->d_blkwhere, ovf);
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_blkpix, (const int *)c->d_n, c->d_rcount + 1);
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_blksegs, (const int *)c->d_n, c->d_rcount);
    hipLaunchKernelGGL(k_stag_route_gather, dim3((na + 3) / 4), dim3(256), 0, st, A, c->d_comps, c->d_n, c->d_prodflag, c->d_blkpix, c->d_blksegs,
                       c->d_blkwhere, c->d_outpix, c->d_segs, R.capOut, R.capSegs, ovf);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    int o = 0;
    if (hipMemcpyAsync(c->rcount, c->d_rcount, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&o, ovf, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return FID_E_HIP;
    c->rcount[2] = o;
    return o ? FID_E_CAPACITY : FID_OK;
}

fid_status fid_stag_detect_edges(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    fid_status rc = fid_stag_edge_frontend(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int W = c->W, H = c->H;
    const size_t n = (size_t)W * H;
    StagRoute R;
    R.grad = c->d_grad; R.dir = c->d_dir; R.edge = c->d_edgeimg; R.W = W; R.H = H;
    R.pix = c->d_rpix; R.stack = c->d_rstack; R.chains = c->d_chains; R.chainNos = c->d_chainnos;
    R.capPix = (int)((size_t)c->maxW * c->maxH); R.capStack = R.capPix; R.capChains = 32767; R.capNos = (c->maxW + c->maxH) * 8;
    R.outpix = c->d_outpix; R.segs = c->d_segs; R.capOut = R.capPix; R.capSegs = R.capPix / 8 + 16;
    R.counters = c->d_rcount;
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool par = c->route_mode == 1 && attempt == 0;
        if (!par && attempt == 0 && c->route_mode == 1) continue;
        if (hipMemcpyAsync(c->d_edgeimg, c->d_edge, n, hipMemcpyDeviceToDevice, st) != hipSuccess) return FID_E_HIP;
        // pixels the routing has not written read as (-1, -1) (the reference reads uninitialised memory there)
        if (hipMemsetAsync(c->d_outpix, 0xff, n * sizeof(int2), st) != hipSuccess) return FID_E_HIP;
        rc = par ? stag_route_par(c, R) : stag_route_seq(c, R);
        if (rc == FID_OK) break;
        if (!par || rc != FID_E_CAPACITY) return rc;
        c->route_fallbacks++;  // an arena of the parallel road was too small: same result by the sequential road
    }
    if (rc != FID_OK) return rc;
    c->routed = true;
    c->validated = false;
    return FID_OK;
}

fid_status fid_stag_detect_edges_validated(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    fid_status rc = fid_stag_detect_edges(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int W = c->W, H = c->H;
    const size_t n = (size_t)W * H;
    const int ns = c->rcount[0];
    // ValidateEdgeSegments starts from an empty edge image (ValidateEdgeSegments.cpp:370)
    if (hipMemsetAsync(c->d_edgeimg, 0, n, st) != hipSuccess || hipMemsetAsync(c->d_vhist, 0, STAG_BINS * 4, st) != hipSuccess) return FID_E_HIP;
    hipLaunchKernelGGL(k_stag_smooth3_prewitt, dim3((W + SX - 1) / SX, (H + SY - 1) / SY), dim3(256), 0, st, c->d_src, W, W, H, c->d_smooth2,
                       c->d_vgrad, c->d_vhist);
    hipLaunchKernelGGL(k_stag_valid_prob, dim3(1), dim3(512), 0, st, c->d_vhist, W, H, c->d_segs, c->d_rcount, c->d_prob, c->d_np);
    const int wg = (ns + 3) / 4;
    if (wg > 0) {
        hipLaunchKernelGGL(k_stag_test_segments, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_vgrad, W, c->d_prob, c->d_np,
                           2.25, c->d_vstack, c->d_edgeimg);
        hipLaunchKernelGGL(k_stag_extract, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_edgeimg, W, c->d_vcounts,
                           c->d_vsegs, 0);
    }
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_vcounts, c->d_rcount, c->d_vtotal);
    if (wg > 0)
        hipLaunchKernelGGL(k_stag_extract, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_edgeimg, W, c->d_vcounts,
                           c->d_vsegs, 1);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_vsegs, c->d_vtotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&c->np, c->d_np, 4, hipMemcpyDeviceToHost, st) != hipSuccess)
        return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->validated = true;
    c->lined = false;
    return FID_OK;
}

// ComputeMinLineLength (EDLines.cpp:694-703) and the floor of 9 of DetectLinesByEDPF (:888-892): a function of the image
// size alone, evaluated on the host like the reference does
static int stag_min_line_len(int W, int H)
{
    const double logNT = 2.0 * (log10((double)W) + log10((double)H));
    int m = (int)((-logNT / log10(0.125)) * 0.5 + 0.5);
    return m < 9 ? 9 : m;
}

fid_status fid_stag_detect_lines(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    fid_status rc = fid_stag_detect_edges_validated(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int ns = c->n_vsegs;
    c->min_line_len = stag_min_line_len(c->W, c->H);
    StagPrefix PF;
    PF.x = c->d_prefix; PF.y = PF.x + c->prefcap; PF.xx = PF.y + c->prefcap; PF.yy = PF.xx + c->prefcap; PF.xy = PF.yy + c->prefcap;
    const int wg = (ns + 63) / 64;
    if (wg > 0)
        hipLaunchKernelGGL(k_stag_split_lines, dim3((ns + 3) / 4), dim3(256), 0, st, c->d_vsegs, c->d_vtotal, c->d_outpix, PF, c->min_line_len, 1.0, c->d_lslots,
                           c->d_lcounts);
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_lcounts, c->d_vtotal, c->d_ltotal);
    if (wg > 0)
        hipLaunchKernelGGL(k_stag_gather_lines, dim3(wg), dim3(64), 0, st, c->d_vsegs, c->d_vtotal, c->d_lcounts, c->d_ltotal, c->d_lslots, c->d_lines);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_lines, c->d_ltotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->lined = true;
    c->lines_validated = false;
    return FID_OK;
}

fid_status fid_stag_detect_lines_validated(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    fid_status rc = fid_stag_detect_lines(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int W = c->W, H = c->H;
    if (c->kmin_w != W || c->kmin_h != H) {  // the false-alarm table is a function of the image size
        std::vector<int> kmin;
        stag_build_kmin(W, H, kmin);
        if (hipMemcpy(c->d_kmin, kmin.data(), kmin.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return FID_E_HIP;
        c->kmin_w = W;
        c->kmin_h = H;
        c->kmin_n = (int)kmin.size() - 1;
    }
    StagLineTables T;
    T.atan_lut = c->d_atan_lut;
    T.kmin = c->d_kmin;
    T.kmin_n = c->kmin_n;
    const int nl = c->n_lines;
    if (nl > 0)
        hipLaunchKernelGGL(k_stag_validate_lines, dim3((nl + 63) / 64), dim3(64), 0, st, c->d_lines, c->d_ltotal, c->d_src, W, H, c->d_vsegs,
                           c->d_outpix, T, c->d_lflags);
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_lflags, c->d_ltotal, c->d_vltotal);
    if (nl > 0)
        hipLaunchKernelGGL(k_stag_compact_lines, dim3((nl + 255) / 256), dim3(256), 0, st, c->d_lines, c->d_ltotal, c->d_lflags, c->d_vltotal,
                           c->d_vlines);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_vlines, c->d_vltotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->lines_validated = true;
    c->quadded = false;
    return FID_OK;
}

fid_status fid_stag_detect_quads(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    fid_status rc = fid_stag_detect_lines_validated(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int W = c->W, H = c->H, ns = c->n_vsegs, nl = c->n_vlines;
    if (hipMemsetAsync(c->d_lrange, 0, (size_t)(ns + 1) * sizeof(int2), st) != hipSuccess) return FID_E_HIP;
    if (nl > 0) hipLaunchKernelGGL(k_stag_line_ranges, dim3((nl + 255) / 256), dim3(256), 0, st, c->d_vlines, c->d_vltotal, c->d_lrange);
    if (ns > 0)
        hipLaunchKernelGGL(k_stag_quads, dim3((ns + 3) / 4), dim3(256), 0, st, c->d_vlines, c->d_lrange, c->d_vtotal, c->d_vsegs, c->d_outpix, c->d_src,
                           W, H, c->d_corners, c->d_order, c->d_qslots, c->d_qcounts);
    hipLaunchKernelGGL(k_stag_scan_counts, dim3(1), dim3(1024), 0, st, c->d_qcounts, c->d_vtotal, c->d_qtotal);
    if (ns > 0)
        hipLaunchKernelGGL(k_stag_gather_quads, dim3((ns + 63) / 64), dim3(64), 0, st, c->d_lrange, c->d_vtotal, c->d_qcounts, c->d_qtotal, c->d_qslots,
                           c->d_quads);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_quads, c->d_qtotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->quadded = true;
    c->decoded = false;
    return FID_OK;
}

fid_status fid_stag_load_library(fid_stag_ctx *c, const uint64_t *codewords, int32_t n_codewords)
{
    if (!c || !codewords || n_codewords <= 0 || (n_codewords & 3)) return FID_E_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    if (c->d_words) (void)hipFree(c->d_words);
    c->d_words = nullptr;
    c->n_words = 0;
    if (hipMalloc((void **)&c->d_words, (size_t)n_codewords * 8) != hipSuccess) return FID_E_OUT_OF_MEMORY;
    if (hipMemcpy(c->d_words, codewords, (size_t)n_codewords * 8, hipMemcpyHostToDevice) != hipSuccess) return FID_E_HIP;
    c->n_words = n_codewords;
    return FID_OK;
}

fid_status fid_stag_detect_markers_unrefined(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    if (!c || !c->d_words) return FID_E_INVALID_ARG;  // no marker library loaded
    fid_status rc = fid_stag_detect_quads(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    const int nq = c->n_quads;
    if (nq > 0)
        hipLaunchKernelGGL(k_stag_decode, dim3((nq + 3) / 4), dim3(256), 0, st, c->d_quads, c->d_qtotal, c->d_src, c->W, c->H, c->d_locs, c->d_words,
                           c->n_words, c->errorCorrection, c->d_cand, c->d_found);
    hipLaunchKernelGGL(k_stag_dedup, dim3(1), dim3(64), 0, st, c->d_cand, c->d_found, c->d_qtotal, c->d_markers, c->d_nmarkers);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(&c->n_markers, c->d_nmarkers, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    c->decoded = true;
    return FID_OK;
}

fid_status fid_stag_detect_markers(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride, fid_stag_marker *out,
                                   int32_t cap, int32_t *n_out)
{
    fid_status rc = fid_stag_detect_markers_unrefined(c, gray, width, height, stride);
    if (rc != FID_OK) return rc;
    hipStream_t st = c->stream;
    if (c->n_markers > 0) {
        hipLaunchKernelGGL(k_stag_refine, dim3(c->n_markers), dim3(64), 0, st, c->d_markers, c->d_nmarkers, c->d_vsegs, c->d_vtotal, c->d_outpix,
                           c->d_chosen);
        if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    }
    if (n_out) *n_out = c->n_markers;
    if (out) {
        if (c->n_markers > cap) return FID_E_CAPACITY;
        if (c->n_markers > 0 &&
            hipMemcpyAsync(out, c->d_markers, (size_t)c->n_markers * sizeof(fid_stag_marker), hipMemcpyDeviceToHost, st) != hipSuccess)
            return FID_E_HIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return FID_E_HIP;
    return FID_OK;
}

fid_status fid_stag_pose_last(fid_stag_ctx *c, const double K[9], const double D[5], double marker_size, fid_stag_pose_out *out, int32_t cap,
                              int32_t *n_out)
{
    if (!c || !K || !out || !c->decoded || !(marker_size > 0)) return FID_E_INVALID_ARG;
    if (n_out) *n_out = c->n_markers;
    if (c->n_markers > cap) return FID_E_CAPACITY;
    if (c->n_markers == 0) return FID_OK;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    PoseCam cam;
    for (int i = 0; i < 9; i++) cam.K[i] = K[i];
    for (int i = 0; i < 5; i++) cam.D[i] = D ? D[i] : 0.0;
    cam.fiducial_len = marker_size;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(k_stag_pose, dim3((c->n_markers + 3) / 4), dim3(64), 0, st, c->d_markers, c->d_nmarkers, cam, marker_size, c->d_poses);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(out, c->d_poses, (size_t)c->n_markers * sizeof(fid_stag_pose_out), hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? FID_OK : FID_E_HIP;
}

int64_t fid_stag_tap_bytes(fid_stag_ctx *c, fid_stag_tap which)
{
    if (!c || c->W <= 0) return 0;
    const int64_t n = (int64_t)c->W * c->H;
    switch (which) {
    case FID_STAG_TAP_SMOOTH:
    case FID_STAG_TAP_DIR:
    case FID_STAG_TAP_ANCHORS: return n;
    case FID_STAG_TAP_GRAD: return n * 2;
    case FID_STAG_TAP_SORTED: return (int64_t)c->n_anchors * 4;
    case FID_STAG_TAP_EDGEIMG: return c->routed ? n : 0;
    case FID_STAG_TAP_SEGMENTS: return c->routed ? (int64_t)c->rcount[0] * 8 : 0;
    case FID_STAG_TAP_SEGPIX: return c->routed ? (int64_t)c->rcount[1] * 8 : 0;
    case FID_STAG_TAP_SMOOTH2: return c->validated ? n : 0;
    case FID_STAG_TAP_VGRAD: return c->validated ? n * 2 : 0;
    case FID_STAG_TAP_VPROB: return c->validated ? (int64_t)STAG_BINS * 8 : 0;
    case FID_STAG_TAP_VSEGMENTS: return c->validated ? (int64_t)c->n_vsegs * 8 : 0;
    case FID_STAG_TAP_LINES: return c->lined ? (int64_t)c->n_lines * (int64_t)sizeof(fid_stag_line) : 0;
    case FID_STAG_TAP_VLINES: return c->lines_validated ? (int64_t)c->n_vlines * (int64_t)sizeof(fid_stag_line) : 0;
    case FID_STAG_TAP_QUADS: return c->quadded ? (int64_t)c->n_quads * (int64_t)sizeof(fid_stag_quad) : 0;
    case FID_STAG_TAP_MARKERS: return c->decoded ? (int64_t)c->n_markers * (int64_t)sizeof(fid_stag_marker) : 0;
    }
    return 0;
}

fid_status fid_stag_tap_read(fid_stag_ctx *c, fid_stag_tap which, void *dst, int64_t dst_bytes)
{
    if (!c || !dst || c->W <= 0) return FID_E_INVALID_ARG;
    const int64_t need = fid_stag_tap_bytes(c, which);
    if (dst_bytes < need) return FID_E_CAPACITY;
    if (need == 0) return FID_OK;
    const void *src = nullptr;
    switch (which) {
    case FID_STAG_TAP_SMOOTH: src = c->d_smooth; break;
    case FID_STAG_TAP_GRAD: src = c->d_grad; break;
    case FID_STAG_TAP_DIR: src = c->d_dir; break;
    case FID_STAG_TAP_ANCHORS: src = c->d_edge; break;
    case FID_STAG_TAP_SORTED: src = c->d_sorted; break;
    case FID_STAG_TAP_EDGEIMG: src = c->d_edgeimg; break;
    case FID_STAG_TAP_SEGMENTS: src = c->d_segs; break;
    case FID_STAG_TAP_SEGPIX: src = c->d_outpix; break;
    case FID_STAG_TAP_SMOOTH2: src = c->d_smooth2; break;
    case FID_STAG_TAP_VGRAD: src = c->d_vgrad; break;
    case FID_STAG_TAP_VPROB: src = c->d_prob; break;
    case FID_STAG_TAP_VSEGMENTS: src = c->d_vsegs; break;
    case FID_STAG_TAP_LINES: src = c->d_lines; break;
    case FID_STAG_TAP_VLINES: src = c->d_vlines; break;
    case FID_STAG_TAP_QUADS: src = c->d_quads; break;
    case FID_STAG_TAP_MARKERS: src = c->d_markers; break;
    }
    if (!src) return FID_E_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    return hipMemcpy(dst, src, (size_t)need, hipMemcpyDeviceToHost) == hipSuccess ? FID_OK : FID_E_HIP;
}

}  // extern "C"
