// fid_stag.hip -- the STag path (stag_detect): this file holds the EDPF edge-detection front end
// (rows s2, s3) and, at the end, the host side + C-ABI; the stages in between are included from
//   fid_stag_route.hip  s4      edge routing             fid_stag_lines.hip  s5, s6  validation, EDLines
//   fid_stag_quads.hip  s7, s8  quads, decoding          fid_stag_pose.hip   s9, s10 pose refinement, marker pose
// Front end = what Stag::detectMarkers -> QuadDetector::detectQuads -> EDInterface::runEDPFandEDLines -> DetectEdgesByEDPF
// (/root/reference/stag_detect/src/stag/ED/ED.cpp:144-187) runs before the edge routing:
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

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

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
__device__ __forceinline__ void k_stag_smooth_grad_impl(const uint8_t *__restrict__ src, int stride, int W, int H, int grad_thresh,
                                                           uint8_t *__restrict__ smooth, int16_t *__restrict__ grad,
                                                           uint8_t *__restrict__ dir)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SY + 6][SX + 6 + 2];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[SY + 6][SX + 4];  // (68 columns: the four-pixel path keeps x0 - 2 .. x0 + 65)
    __shared__ __attribute__((aligned(16))) uint8_t s_sm[SY + 2][SX + 2 + 2];
    const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
    const int tid = threadIdx.x;
    // (round 6) INTERIOR tiles -- no reflected coordinate, no image-border pixel, rows that can be read and written as 32-bit words:
    // 90 % of a 1080p frame -- take four pixels a thread.  The byte-a-thread form below issued ~115 VALU instructions per 64 pixels
    // (five LDS bytes + four multiply-adds per tap pass, eight bytes per Prewitt, divisions by 70 and 66 for the tile indices) and
    // was bound by that, alone on the chip, not by HBM.  Here a tap pass is one v_dot4_u32_u8 per output on bytes shifted into place
    // by v_alignbyte (horizontal) or packed 16-bit arithmetic on two columns at once (vertical: the sums stay below 65 536), and
    // Prewitt works on the column sums of three rows.  The same integers as below, pixel for pixel.
    if (x0 >= 4 && x0 + SX + 4 <= W && y0 >= 3 && y0 + SY + 3 <= H && ((stride | W) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        // bytes x0 - 4 .. x0 + 67 of rows y0 - 3 .. y0 + 18, as words
        uint32_t(*w_src)[18] = reinterpret_cast<uint32_t(*)[18]>(&s_src[0][0]);  // 22 x 18 words = 1 584 bytes of the 1 584 there
        uint32_t(*w_h)[34] = reinterpret_cast<uint32_t(*)[34]>(&s_h[0][0]);       // 22 x 34 words: two smoothed columns a word
        uint32_t(*w_sm)[17] = reinterpret_cast<uint32_t(*)[17]>(&s_sm[0][0]);     // 18 x 17 words: four smoothed pixels a word
        for (int i = tid; i < 22 * 18; i += 256) {
            const int r = i / 18, c = i - r * 18;
            w_src[r][c] = *reinterpret_cast<const uint32_t *>(src + (long long)(y0 - 3 + r) * stride + (x0 - 4 + 4 * c));
        }
        __syncthreads();
        // horizontal [1 4 6 4 1]: item (row r, group j) = smoothed columns x0 - 2 + 4 j + k, k = 0 .. 3, from bytes k .. k + 4 of words j, j + 1
        for (int i = tid; i < 22 * 17; i += 256) {
            const int r = i / 17, j = i - r * 17;
            const uint32_t d0 = w_src[r][j], d1 = w_src[r][j + 1];
            const uint32_t h0 = __builtin_amdgcn_udot4(d0, 0x04060401u, d1 & 0xffu, false);
            const uint32_t h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), 0x04060401u, (d1 >> 8) & 0xffu, false);
            const uint32_t h2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), 0x04060401u, (d1 >> 16) & 0xffu, false);
            const uint32_t h3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), 0x04060401u, d1 >> 24, false);
            w_h[r][2 * j] = h0 | (h1 << 16);
            w_h[r][2 * j + 1] = h2 | (h3 << 16);
        }
        __syncthreads();
        // vertical pass, two columns a word: smoothed row y0 - 1 + r from rows r .. r + 4; (acc + 128) >> 8 per 16-bit half
        for (int i = tid; i < 18 * 17; i += 256) {
            const int r = i / 17, j = i - r * 17;
            uint32_t o[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t a = w_h[r][2 * j + q], b = w_h[r + 1][2 * j + q], c = w_h[r + 2][2 * j + q], d = w_h[r + 3][2 * j + q],
                               e = w_h[r + 4][2 * j + q];
                // a + e + 4 (b + d) + 6 c <= 16 * 4 080 = 65 280 in each half: no carry crosses the halves
                const uint32_t acc = (a + e) + 4u * (b + d) + 6u * c + 0x00800080u;
                o[q] = (acc >> 8) & 0x00ff00ffu;
            }
            w_sm[r][j] = (o[0] & 0xffu) | ((o[0] >> 8) & 0xff00u) | ((o[1] & 0xffu) << 16) | ((o[1] << 8) & 0xff000000u);
        }
        __syncthreads();
        // Prewitt: thread = four pixels x0 + 4 j + k of row y0 + r; their 3 x 3 neighbourhoods are bytes 1 + k .. 3 + k of words j, j + 1
        // (smoothed columns x0 - 2 + 4 j ...) of rows r .. r + 2.  gx = S(x + 1) - S(x - 1) with S = the column's sum over the three
        // rows, gy = (bottom - top) summed over the three columns: ComputeGradient's com1 / com2 form, regrouped (integers: exact)
        {
            const int r = tid >> 4, j = tid & 15;
            const uint32_t t0 = w_sm[r][j], t1 = w_sm[r][j + 1], m0 = w_sm[r + 1][j], m1 = w_sm[r + 1][j + 1], b0 = w_sm[r + 2][j],
                           b1 = w_sm[r + 2][j + 1];
            int S[8], D[8];
#pragma unroll
            for (int p = 1; p <= 6; p++) {
                const int tp = (int)(((p < 4 ? t0 : t1) >> (8 * (p & 3))) & 0xffu), mp = (int)(((p < 4 ? m0 : m1) >> (8 * (p & 3))) & 0xffu),
                          bp = (int)(((p < 4 ? b0 : b1) >> (8 * (p & 3))) & 0xffu);
                S[p] = tp + mp + bp;
                D[p] = bp - tp;
            }
            uint32_t gw[2] = {0u, 0u}, dw = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int gxv = S[3 + k] - S[1 + k], gyv = D[1 + k] + D[2 + k] + D[3 + k];
                gxv = gxv < 0 ? -gxv : gxv;
                gyv = gyv < 0 ? -gyv : gyv;
                const int sum = gxv + gyv;
                gw[k >> 1] |= (uint32_t)sum << (16 * (k & 1));
                dw |= (uint32_t)(sum >= grad_thresh ? (gxv >= gyv ? STAG_EDGE_VERTICAL : STAG_EDGE_HORIZONTAL) : 0) << (8 * k);
            }
            const long long idx = (long long)(y0 + r) * W + (x0 + 4 * j);
            *reinterpret_cast<uint32_t *>(smooth + idx) = __builtin_amdgcn_alignbyte(m1, m0, 2);  // bytes 2 .. 5: the four centres
            *reinterpret_cast<uint2 *>(grad + idx) = make_uint2(gw[0], gw[1]);
            *reinterpret_cast<uint32_t *>(dir + idx) = dw;
        }
        return;
    }
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
__global__ __launch_bounds__(256) void k_stag_smooth_grad(const uint8_t *__restrict__ src, int stride, int W, int H, int grad_thresh, uint8_t *__restrict__ smooth, int16_t *__restrict__ grad, uint8_t *__restrict__ dir)
{
    k_stag_smooth_grad_impl(src, stride, W, H, grad_thresh, smooth, grad, dir);
}
struct k_stag_smooth_grad_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const uint8_t *__restrict__ src, int stride, int W, int H, int grad_thresh, uint8_t *__restrict__ smooth, int16_t *__restrict__ grad, uint8_t *__restrict__ dir) const { k_stag_smooth_grad_impl(src, stride, W, H, grad_thresh, smooth, grad, dir); }
};

// The smoothed image is 8-bit, so |gx|, |gy| <= 3 * 255 and the gradient value never exceeds 1530: the reference's
// 128 * 256 counting-sort bins (SIZE in SortAnchorsByGradValue) are used only below STAG_BINS.
#define STAG_BINS 1536
#define STAG_VHIST_SLICES 16  // copies of the validation histogram (k_stag_smooth3_prewitt adds, k_stag_valid_prob sums)
#define STAG_BAND_ROWS 4  // rows per band of k_stag_place = waves per workgroup (round 6: 4, a 256-thread workgroup with 24 KB of LDS; 8 rows =
                          // 512 threads and 48 KB waited five times its own duration for room on a CU beside the other groups' kernels)

// Anchor points: local gradient maxima across the edge normal (ANCHOR_THRESH, SCAN_INTERVAL as in the reference), counted
// per (row, gradient value) for the counting sort (global atomics: the anchors are sparse; 16-bit counts, two a word).
__device__ __forceinline__ void k_stag_anchors_impl(const int16_t *__restrict__ grad, const uint8_t *__restrict__ dir, int W, int H,
                                                       int grad_thresh, int anchor_thresh, int scan_interval,
                                                       uint8_t *__restrict__ edge, unsigned *__restrict__ rowhist)
{
    // (row and column are carried along the grid-stride loop: one 32-bit division per thread instead of a 64-bit one per pixel --
    //  the images are < 2^31 pixels, fid_stag_create caps both sides at 8 191)
    const unsigned total = (unsigned)W * (unsigned)H, stride = gridDim.x * 256u;
    const unsigned first = blockIdx.x * 256u + threadIdx.x;
    const int si = (int)(stride / (unsigned)W), sj = (int)(stride - (unsigned)si * (unsigned)W);
    int i = (int)(first / (unsigned)W), j = (int)(first - (unsigned)i * (unsigned)W);
    for (unsigned idx = first; idx < total; idx += stride, i += si, j += sj) {
        if (j >= W) {
            j -= W;
            i++;
        }
        uint8_t e = 0;
        if (i >= 2 && i < H - 2 && j >= 2 && j < W - 2) {
            // rows that are not a multiple of SCAN_INTERVAL are scanned at columns SCAN_INTERVAL, 2 SCAN_INTERVAL, ...
            const bool scanned = scan_interval == 1 || (i % scan_interval == 0) || (j >= scan_interval && j % scan_interval == 0);
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
                    // (two 16-bit counts a word -- a row holds < 2^16 anchors: half the clearing, summing and placing traffic)
                    atomicAdd(&rowhist[(size_t)i * (STAG_BINS / 2) + (g >> 1)], 1u << ((g & 1) * 16));
                }
            }
        }
        edge[idx] = e;
    }
}
__global__ __launch_bounds__(256) void k_stag_anchors(const int16_t *__restrict__ grad, const uint8_t *__restrict__ dir, int W, int H, int grad_thresh, int anchor_thresh, int scan_interval, uint8_t *__restrict__ edge, unsigned *__restrict__ rowhist)
{
    k_stag_anchors_impl(grad, dir, W, H, grad_thresh, anchor_thresh, scan_interval, edge, rowhist);
}
struct k_stag_anchors_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int16_t *__restrict__ grad, const uint8_t *__restrict__ dir, int W, int H, int grad_thresh, int anchor_thresh, int scan_interval, uint8_t *__restrict__ edge, unsigned *__restrict__ rowhist) const { k_stag_anchors_impl(grad, dir, W, H, grad_thresh, anchor_thresh, scan_interval, edge, rowhist); }
};

// anchors per (band of STAG_BAND_ROWS rows, gradient value)
__device__ __forceinline__ void k_stag_bandsum_impl(const unsigned *__restrict__ rowhist, int H, unsigned *__restrict__ bandhist)
{
    const int g = blockIdx.x * 256 + threadIdx.x, band = blockIdx.y;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < STAG_BAND_ROWS; r++) {
        const int row = band * STAG_BAND_ROWS + r;
        if (row < H) acc += (rowhist[(size_t)row * (STAG_BINS / 2) + (g >> 1)] >> ((g & 1) * 16)) & 0xffffu;
    }
    bandhist[(size_t)band * STAG_BINS + g] = acc;
}
__global__ __launch_bounds__(256) void k_stag_bandsum(const unsigned *__restrict__ rowhist, int H, unsigned *__restrict__ bandhist)
{
    k_stag_bandsum_impl(rowhist, H, bandhist);
}
struct k_stag_bandsum_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const unsigned *__restrict__ rowhist, int H, unsigned *__restrict__ bandhist) const { k_stag_bandsum_impl(rowhist, H, bandhist); }
};

// per gradient value: bands from the LAST to the first (the reference's --C[grad] placement leaves the offsets of one
// gradient value in descending order) -> start of each band inside the value's bucket; tot[g] = size of the bucket
__device__ __forceinline__ void k_stag_bandscan_impl(unsigned *__restrict__ bandhist, int nbands, unsigned *__restrict__ tot)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    int b = nbands - 1;
    for (; b >= 15; b -= 16) {  // sixteen independent loads in flight (four until round 6: with 4-row bands a 1080p frame has 270 of them)
        unsigned n[16];
#pragma unroll
        for (int k = 0; k < 16; k++) n[k] = bandhist[(size_t)(b - k) * STAG_BINS + g];
#pragma unroll
        for (int k = 0; k < 16; k++) {
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
__global__ __launch_bounds__(256) void k_stag_bandscan(unsigned *__restrict__ bandhist, int nbands, unsigned *__restrict__ tot)
{
    k_stag_bandscan_impl(bandhist, nbands, tot);
}
struct k_stag_bandscan_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(unsigned *__restrict__ bandhist, int nbands, unsigned *__restrict__ tot) const { k_stag_bandscan_impl(bandhist, nbands, tot); }
};

// exclusive prefix sums over the gradient values (one workgroup): bstart[g] = number of anchors with a smaller gradient
__device__ __forceinline__ void k_stag_scan_impl(const unsigned *__restrict__ tot, unsigned *__restrict__ bstart, unsigned *__restrict__ n_anchors)
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
__global__ __launch_bounds__(512) void k_stag_scan(const unsigned *__restrict__ tot, unsigned *__restrict__ bstart, unsigned *__restrict__ n_anchors)
{
    k_stag_scan_impl(tot, bstart, n_anchors);
}
struct k_stag_scan_fn {
    static constexpr int kBounds = 512;
    __device__ __forceinline__ void operator()(const unsigned *__restrict__ tot, unsigned *__restrict__ bstart, unsigned *__restrict__ n_anchors) const { k_stag_scan_impl(tot, bstart, n_anchors); }
};

// Placement: one workgroup per band, one wave per row.  LDS holds, per (row of the band, gradient value), the next free
// slot: bucket start + band start + anchors of that value in the LATER rows of the band.  Every wave then goes through
// its row from the last column to the first, 64 columns at a time; lanes with the same gradient value take consecutive
// slots in descending column order.  No sort, no atomics: the order is exactly the reference's.
__device__ __forceinline__ void k_stag_place_impl(const int16_t *__restrict__ grad, const uint8_t *__restrict__ edge, int W, int H,
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
            n[r] = row < H ? (rowhist[(size_t)row * (STAG_BINS / 2) + (g >> 1)] >> ((g & 1) * 16)) & 0xffffu : 0u;
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
__global__ __launch_bounds__(64 * STAG_BAND_ROWS) void k_stag_place(const int16_t *__restrict__ grad, const uint8_t *__restrict__ edge, int W, int H, const unsigned *__restrict__ rowhist, const unsigned *__restrict__ bandstart, const unsigned *__restrict__ bstart, int32_t *__restrict__ sorted)
{
    k_stag_place_impl(grad, edge, W, H, rowhist, bandstart, bstart, sorted);
}
struct k_stag_place_fn {
    static constexpr int kBounds = 64 * STAG_BAND_ROWS;
    __device__ __forceinline__ void operator()(const int16_t *__restrict__ grad, const uint8_t *__restrict__ edge, int W, int H, const unsigned *__restrict__ rowhist, const unsigned *__restrict__ bandstart, const unsigned *__restrict__ bstart, int32_t *__restrict__ sorted) const { k_stag_place_impl(grad, edge, W, H, rowhist, bandstart, bstart, sorted); }
};

#include "fid_stag_route.hip"
#include "fid_stag_lines.hip"
#include "fid_stag_quads.hip"
#include "fid_stag_pose.hip"
#include "fid_stag_batch.h"
thread_local StagRecorder *g_stag_rec = nullptr;
std::vector<StagHostAlias> g_stag_aliases;
std::mutex g_stag_alias_mutex;

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

#define STAG_PIN_MARKERS 512  // results of a frame staged through pinned memory (more than that: a blocking copy)
// What the launches of a frame are sized by when the frame is queued ahead of its own counts: anchors, components, output-arena
// pixels, most anchors in one component, largest LDS tile, edge segments, validated segments, lines, validated lines, quads, markers
struct StagPred {
    bool valid = false;
    int W = 0, H = 0;
    int na = 0, nc = 0, aout = 0, most = 0, tile = 0, ns = 0, nvs = 0, nl = 0, nvl = 0, nq = 0, nm = 0;
};

struct fid_stag_ctx {
    int device = 0, maxW = 0, maxH = 0, libraryHD = 0, errorCorrection = 0;
    hipStream_t stream = nullptr;
    hipStream_t group_stream = nullptr;  // group mode (fid_stag_batch.h): the stream of the group this context's frame travels with
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
    int4 *d_cbox = nullptr;  // per root: bounding box of the component's pixels
    int *d_label = nullptr, *d_csize = nullptr, *d_canch = nullptr, *d_cidmap = nullptr, *d_cursors = nullptr, *d_caps = nullptr;
    uint8_t *d_tilefg = nullptr;  // per 64 x 16 tile of the connected-component passes: does it hold a foreground pixel?
    int *d_roots = nullptr;   // the roots of the frame's connected components (k_stag_ccl_flatten's list; k_stag_comp_alloc goes by it)
    int max_roots = 0;
    int *d_corder = nullptr;  // the components longest-first (k_stag_comp_tilemax; the walk and the extraction go by it)
    int *d_fill = nullptr, *d_aslots = nullptr, *d_prodflag = nullptr, *d_next = nullptr, *d_blkpix = nullptr, *d_blksegs = nullptr;
    int2 *d_blkwhere = nullptr, *d_apix = nullptr, *d_aout = nullptr, *d_asegs = nullptr;
    int4 *d_astack = nullptr;
    StagChain *d_achains = nullptr;
    StagComp *d_comps = nullptr;
    StagRec *d_recs = nullptr;
    int max_comps = 0, cap_aslots = 0, route_mode = 1, route_fallbacks = 0, route_tile = 1;
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
    // pinned landing area of the asynchronous device -> host read-backs (a copy into pageable memory would make the
    // "asynchronous" copy wait for the stream, and the segments of different contexts could not overlap)
    struct Pinned {
        unsigned n_anchors;
        int cur[11], rcount[3], ovf, n_vsegs, np, n_lines, n_vlines, n_quads, n_markers;
        int spec_bad;  // (a frame queued ahead: a count exceeded what its launches were sized for)
        fid_stag_marker markers[STAG_PIN_MARKERS];
        fid_stag_pose_out poses[STAG_PIN_MARKERS];
    } *hp = nullptr;
    uint8_t *h_src = nullptr;  // pinned staging of the input frame (host rows -> here -> one asynchronous DMA)
    // a frame QUEUED AHEAD (round 5): the counts of the last frame this context finished size the next frame's launches, so that
    // the whole frame is enqueued without a host wait (stag_advance_impl)
    StagPred pred;
    int *d_specbad = nullptr;
    int spec_frames = 0, spec_misses = 0;
    // back-off of the queue-ahead road: a stream whose counts jump from frame to frame (a moving scene) misses most predictions, and
    // every miss is a whole extra pass.  Two misses among the last four queued frames switch the road off for spec_pause frames
    // (16, doubling up to 256 while the misses go on; one fit after a pause brings it back to 16).  Results are the same either way.
    unsigned spec_hist = 0;  // the last queued frames, one bit each (1 = missed), newest in bit 0
    int spec_pause = 16, spec_skip = 0, spec_backoffs = 0;
    char *d_slab = nullptr;  // every device buffer above is a piece of this one allocation (same offsets in every context of a size)
    size_t slab_bytes = 0;
    unsigned long long *d_words_slab = nullptr;  // the slab's room for the marker library (d_words points here when it fits)
    char *pin_alias = nullptr;  // device alias of the pinned block hp
    int tile_kb_env = 0, no_sparse = 0;  // FID_STAG_TILE_KB (LDS a component's walk may ask for), FID_STAG_SPARSE=0
    int split_lds_env = -1;              // FID_STAG_SPLIT_LDS (pixels per wave of k_stag_split_lines that live in LDS)
};

#define STAG_WORDS_RESERVE (1u << 20)  // bytes of the slab kept for the marker library (HD11, the largest: 22 309 words)
struct StagSlab {
    struct Item {
        void **p;
        size_t off;
    };
    std::vector<Item> items;
    size_t bytes = 0;
    bool take(void **p, size_t b)
    {
        items.push_back({p, bytes});
        bytes += (b + 255) & ~(size_t)255;  // (hipMalloc's own alignment, which the kernels' 16-byte accesses rely on)
        return true;
    }
    bool commit(char **base, size_t *total)
    {
        if (hipMalloc((void **)base, bytes) != hipSuccess) return false;
        for (const Item &i : items) *i.p = *base + i.off;
        *total = bytes;
        return true;
    }
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
    // (the context's stream is made when it is first needed -- stag_stream(): a pool of 64 frame slots works on four streams, and
    //  64 idle streams are 64 claims on the hardware queues this process shares with every other user of the GPU)
    bool ok = hipSetDevice(device) == hipSuccess;
    // Every device buffer of the context is a piece of ONE slab, laid out in the order of the takes below: two contexts made for the
    // same image size have every buffer at the same OFFSET, so that frame f's pointer is frame 0's plus (slab_f - slab_0) -- which is
    // what lets a group's launch carry frame 0's arguments once and 32 bytes per further frame (fid_stag_batch.h).
    StagSlab slab;
    int caps_host[16] = {0};
    ok = ok && slab.take((void **)&c->d_src, n) && slab.take((void **)&c->d_smooth, n) &&
         slab.take((void **)&c->d_dir, n) && slab.take((void **)&c->d_edge, n) &&
         slab.take((void **)&c->d_grad, n * 2) && slab.take((void **)&c->d_sorted, n * 4) &&
         slab.take((void **)&c->d_rowhist, (size_t)max_height * STAG_BINS * 4) &&
         slab.take((void **)&c->d_bandhist, (size_t)((max_height + STAG_BAND_ROWS - 1) / STAG_BAND_ROWS) * STAG_BINS * 4) &&
         slab.take((void **)&c->d_tot, STAG_BINS * 4) && slab.take((void **)&c->d_bstart, STAG_BINS * 4) &&
         slab.take((void **)&c->d_n, 4);
    // routing: the reference sizes its scratch arrays for the worst case (width * height entries each, EDInternals.cpp:848-853)
    ok = ok && slab.take((void **)&c->d_edgeimg, n) && slab.take((void **)&c->d_rpix, n * sizeof(int2)) &&
         slab.take((void **)&c->d_outpix, n * sizeof(int2)) && slab.take((void **)&c->d_segs, (n / 8 + 16) * sizeof(int2)) &&
         slab.take((void **)&c->d_rstack, n * sizeof(int4)) && slab.take((void **)&c->d_chains, 32767 * sizeof(StagChain)) &&
         slab.take((void **)&c->d_chainnos, (size_t)(max_width + max_height) * 8 * sizeof(int)) &&
         slab.take((void **)&c->d_rcount, 16);
    // component-parallel routing: labels, per-root counters, component table, arenas (sizes in entries; see k_stag_comp_alloc)
    c->max_comps = (int)(n / 8 + 64);
    c->cap_aslots = (int)(n / 2 + 64);
    c->max_roots = ((max_width + 1) / 2) * ((max_height + 1) / 2) + 64;  // (8-connected components of an image: no more than that)
    ok = ok && slab.take((void **)&c->d_label, n * 4) && slab.take((void **)&c->d_csize, n * 4) &&
         slab.take((void **)&c->d_canch, n * 4) && slab.take((void **)&c->d_cidmap, n * 4) && slab.take((void **)&c->d_cbox, n * sizeof(int4)) &&
         slab.take((void **)&c->d_cursors, 64) && slab.take((void **)&c->d_caps, 64) &&
         slab.take((void **)&c->d_corder, (size_t)c->max_comps * 4) && slab.take((void **)&c->d_roots, (size_t)c->max_roots * 4) &&
         slab.take((void **)&c->d_tilefg, (size_t)((max_width + CCL_TW - 1) / CCL_TW) * ((max_height + CCL_TH - 1) / CCL_TH)) && slab.take((void **)&c->d_fill, (size_t)c->max_comps * 4) && slab.take((void **)&c->d_aslots, (size_t)c->cap_aslots * 4) &&
         slab.take((void **)&c->d_prodflag, n * 4) && slab.take((void **)&c->d_next, n * 4) &&
         slab.take((void **)&c->d_blkpix, n * 4) && slab.take((void **)&c->d_blksegs, n * 4) &&
         slab.take((void **)&c->d_blkwhere, n * sizeof(int2)) && slab.take((void **)&c->d_apix, 3 * n * sizeof(int2)) &&
         slab.take((void **)&c->d_aout, 3 * n * sizeof(int2)) && slab.take((void **)&c->d_asegs, (n / 2 + 64) * sizeof(int2)) &&
         slab.take((void **)&c->d_astack, 2 * n * sizeof(int4)) && slab.take((void **)&c->d_achains, 2 * n * sizeof(StagChain)) &&
         slab.take((void **)&c->d_comps, (size_t)c->max_comps * sizeof(StagComp)) &&
         slab.take((void **)&c->d_recs, (size_t)c->cap_aslots * sizeof(StagRec));
    if (ok) {
        const int caps[16] = {c->max_comps, c->cap_aslots, (int)(3 * n), (int)(2 * n), (int)(2 * n), (int)(3 * n), (int)(n / 2 + 64), 0};
        memcpy(caps_host, caps, sizeof(caps));  // (uploaded below, when the slab exists)
        const char *e = getenv("FID_STAG_ROUTE");
        c->route_mode = (e && !strcmp(e, "seq")) ? 0 : 1;
        c->route_tile = (e && !strcmp(e, "notile")) ? 0 : 1;  // "notile": component-parallel, walks in global memory
        // (read per context: the tests run the roads side by side in one process)
        const char *tk = getenv("FID_STAG_TILE_KB"), *sp = getenv("FID_STAG_SPARSE");
        c->tile_kb_env = tk ? atoi(tk) : 0;
        c->no_sparse = sp && atoi(sp) == 0 ? 1 : 0;
        const char *sl = getenv("FID_STAG_SPLIT_LDS");
        c->split_lds_env = sl ? atoi(sl) : -1;
        ok = ok && hipFuncSetAttribute((const void *)k_stag_route_walk, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
             hipFuncSetAttribute((const void *)k_stag_comp_sort_big, hipFuncAttributeMaxDynamicSharedMemorySize, STAG_SORT_BIG * 4) == hipSuccess &&
             // (and their group-mode trampolines, fid_stag_batch.h)
             hipFuncSetAttribute((const void *)k_stag_batch<k_stag_route_walk_fn>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
             hipFuncSetAttribute((const void *)k_stag_batch<k_stag_comp_sort_big_fn>, hipFuncAttributeMaxDynamicSharedMemorySize, STAG_SORT_BIG * 4) == hipSuccess &&
             hipFuncSetAttribute((const void *)k_stag_split_lines, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES(1024)) == hipSuccess &&
             hipFuncSetAttribute((const void *)k_stag_batch<k_stag_split_lines_fn>, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES(1024)) == hipSuccess;
        if (getenv("FID_VERBOSE"))
            fprintf(stderr, "fid stag: frames per merged launch: route_walk %d, route_extract %d, quads %d, decode %d, smooth_grad %d\n",
                    StagTab<k_stag_route_walk_fn>::kMax, StagTab<k_stag_route_extract_fn>::kMax, StagTab<k_stag_quads_fn>::kMax,
                    StagTab<k_stag_decode_fn>::kMax, StagTab<k_stag_smooth_grad_fn>::kMax);
    }
    ok = ok && slab.take((void **)&c->d_smooth2, n) && slab.take((void **)&c->d_vgrad, n * 2) &&
         slab.take((void **)&c->d_vhist, STAG_BINS * 4 * STAG_VHIST_SLICES) && slab.take((void **)&c->d_prob, STAG_BINS * 8) &&
         slab.take((void **)&c->d_np, 4) && slab.take((void **)&c->d_vcounts, (n / 8 + 16) * 4) &&
         slab.take((void **)&c->d_vtotal, 4) && slab.take((void **)&c->d_vstack, n * sizeof(int2)) &&
         slab.take((void **)&c->d_vsegs, (n / 8 + 16) * sizeof(int2));
    c->prefcap = n + n / 8 + 64;
    ok = ok && slab.take((void **)&c->d_prefix, c->prefcap * 5 * sizeof(long long)) &&
         slab.take((void **)&c->d_lslots, (n / 9 + 16) * sizeof(fid_stag_line)) &&
         slab.take((void **)&c->d_lines, (n / 9 + 16) * sizeof(fid_stag_line)) &&
         slab.take((void **)&c->d_lcounts, (n / 8 + 16) * 4) && slab.take((void **)&c->d_ltotal, 4);
    ok = ok && slab.take((void **)&c->d_atan_lut, 1025 * 8) &&
         slab.take((void **)&c->d_kmin, (size_t)(4 * (max_width + max_height) + 16) * 4) &&
         slab.take((void **)&c->d_lflags, (n / 9 + 16) * 4) && slab.take((void **)&c->d_vltotal, 4) &&
         slab.take((void **)&c->d_vlines, (n / 9 + 16) * sizeof(fid_stag_line));
    ok = ok && slab.take((void **)&c->d_lrange, (n / 8 + 16) * sizeof(int2)) &&
         slab.take((void **)&c->d_corners, (n / 9 + 16) * sizeof(StagCorner)) &&
         slab.take((void **)&c->d_order, (n / 9 + 16) * 4) && slab.take((void **)&c->d_qcounts, (n / 8 + 16) * 4) &&
         slab.take((void **)&c->d_qtotal, 4) && slab.take((void **)&c->d_qslots, (n / 9 + 16) * sizeof(fid_stag_quad)) &&
         slab.take((void **)&c->d_quads, (n / 9 + 16) * sizeof(fid_stag_quad));
    ok = ok && slab.take((void **)&c->d_locs, 72 * 3 * 8) && slab.take((void **)&c->d_cand, (n / 9 + 16) * sizeof(fid_stag_marker)) &&
         slab.take((void **)&c->d_markers, (n / 9 + 16) * sizeof(fid_stag_marker)) &&
         slab.take((void **)&c->d_found, (n / 9 + 16) * 4) && slab.take((void **)&c->d_nmarkers, 4);
    ok = ok && slab.take((void **)&c->d_specbad, 16) && slab.take((void **)&c->d_words_slab, STAG_WORDS_RESERVE);
    ok = ok && slab.take((void **)&c->d_chosen, (n / 9 + 16) * 4) &&
         slab.take((void **)&c->d_poses, (n / 9 + 16) * sizeof(fid_stag_pose_out));
    ok = ok && slab.commit(&c->d_slab, &c->slab_bytes);
    ok = ok && hipMemcpy(c->d_caps, caps_host, sizeof(caps_host), hipMemcpyHostToDevice) == hipSuccess && hipMemset(c->d_specbad, 0, 16) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->hp, sizeof(fid_stag_ctx::Pinned), hipHostMallocDefault) == hipSuccess &&
         hipHostMalloc((void **)&c->h_src, n, hipHostMallocDefault) == hipSuccess;
    if (ok) {  // group mode writes the per-segment counters straight into the pinned block (fid_stag_batch.h)
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, c->hp, 0) == hipSuccess && dev) {
            std::lock_guard<std::mutex> g(g_stag_alias_mutex);
            g_stag_aliases.push_back({(const char *)c->hp, sizeof(fid_stag_ctx::Pinned), (char *)dev});
            c->pin_alias = (char *)dev;
        }
    }
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
    if (c->d_words && c->d_words != c->d_words_slab) (void)hipFree(c->d_words);  // (a library larger than the slab's room for it)
    if (c->d_slab) (void)hipFree(c->d_slab);
    if (c->hp) {
        std::lock_guard<std::mutex> g(g_stag_alias_mutex);
        for (size_t k = 0; k < g_stag_aliases.size(); k++)
            if (g_stag_aliases[k].host == (const char *)c->hp) {
                g_stag_aliases.erase(g_stag_aliases.begin() + (long)k);
                break;
            }
    }
    if (c->hp) (void)hipHostFree(c->hp);
    if (c->h_src) (void)hipHostFree(c->h_src);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C" (the pipeline driver below is internal)

// ---- the pipeline of one frame on one context, cut into SEGMENTS at the points where the host needs a count the device has
// produced (to size the next launches).  A segment first waits for the context's stream (the counts of the previous segment
// are then in host memory), does the host-side bookkeeping and launches the next piece without waiting for it.  Run back to
// back on one context this is the frame-at-a-time path (every staged entry point = the segments up to its stage).  Several
// frames: fid_stag_detect_markers_batch carries GROUPS of frames through the same segments in lockstep, their launches recorded
// and issued once per group (fid_stag_batch.h; a host thread per group); FID_STAG_BATCH=contexts keeps round 2's road, the
// contexts dealt out to FID_STAG_THREADS host threads (default 4), each context on its own stream.
enum StagStage { SS_FRONTEND = 0, SS_EDGES, SS_EDGES_VALIDATED, SS_LINES, SS_LINES_VALIDATED, SS_QUADS, SS_UNREFINED, SS_MARKERS, SS_POSE };
enum { RS_NONE = 0, RS_PAR_A, RS_PAR_B, RS_SEQ, RS_EMPTY };

struct StagJob {
    // what is asked for
    const uint8_t *gray = nullptr;
    int width = 0, height = 0, stride = 0, last = SS_MARKERS;
    fid_stag_marker *out = nullptr;
    int cap = 0;
    int32_t *n_out = nullptr;
    const double *K = nullptr, *D = nullptr;
    double marker_size = 0;
    fid_stag_pose_out *poses = nullptr;
    int pose_cap = 0;
    // where it stands
    int seg = 0, rstate = RS_NONE;
    bool done = false;
    fid_status rc = FID_OK;
    int cur[11] = {0}, ovf = 0;
    StagRoute R;
    // queued ahead: every launch of the frame sized by j.use (the context's last counts with a margin), one wait at the end
    bool spec = false, nospec = false;
    bool staged = false;  // the frame's rows are in the context's pinned staging buffer already (the group driver copies a group's frames with several threads)
    StagPred use;
    int lds_cap = 0;
};

static hipStream_t stag_stream(fid_stag_ctx *c)
{
    if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) c->stream = nullptr;  // (nullptr: the default stream)
    return c->stream;
}

// ComputeMinLineLength (EDLines.cpp:694-703) and the floor of 9 of DetectLinesByEDPF (:888-892): a function of the image
// size alone, evaluated on the host like the reference does
static int stag_min_line_len(int W, int H)
{
    const double logNT = 2.0 * (log10((double)W) + log10((double)H));
    int m = (int)((-logNT / log10(0.125)) * 0.5 + 0.5);
    return m < 9 ? 9 : m;
}


static fid_status stag_finish(StagJob &j, fid_status rc)
{
    j.done = true;
    j.rc = rc;
    return rc;
}

// the sequential road of the routing: one lane for the whole frame (the reference's loop as it stands)
static bool stag_launch_route_seq(fid_stag_ctx *c, StagJob &j)
{
    hipStream_t st = c->group_stream ? c->group_stream : stag_stream(c);
    const size_t n = (size_t)c->W * c->H;
    if (STAG_MEMCPY(c->d_edgeimg, c->d_edge, n, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    // pixels the routing has not written read as (-1, -1) (the reference reads uninitialised memory there)
    if (STAG_MEMSET(c->d_outpix, 0xff, n * sizeof(int2), st) != hipSuccess) return false;
    STAG_LAUNCH(k_stag_route_seq, dim3(1), dim3(64), 0, st, j.R, c->d_sorted, c->d_n, 16);
    if (hipGetLastError() != hipSuccess) return false;
    if (STAG_MEMCPY(c->hp->rcount, c->d_rcount, 12, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    j.rstate = RS_SEQ;
    return true;
}

// ---- frames queued ahead (round 5).  The nine waits of a frame exist because the host sizes the next launches by counts the
// device has just produced.  Every kernel behind such a count reads it from device memory and returns beyond it, so the grid only
// has to be LARGE ENOUGH: a context that has finished a frame sizes the next frame's launches (grids, fills, clears, the LDS tile,
// the result copies) by that frame's counts plus a margin and enqueues the WHOLE frame -- segments 0 to 7 in one go, no wait in
// between; k_stag_spec_guard stands where each read-back stood (fid_stag_route.hip).  One wait at the end; the host then checks
// every true count against what it had assumed (and the device flag): a frame that did not fit is run again on the counted road
// from its first segment -- same result either way, the counted road is what the staged entry points and every first frame use.
// FID_STAG_SPEC=0 keeps every frame on the counted road.
// Default (FID_STAG_SPEC unset): frame-at-a-time calls are queued ahead, GROUPS stay on the counted road -- measured on the
// cfg 5 batch (256 frames, 128 slots, runtime-default hardware queues; tools/gpu_r5_stag.sh): counted 5 030 / 4 510 / 5 050
// frames/s against 4 950 / 4 400 / 4 860 queued ahead.  A group's waits were never the limit there (eight groups on eight host
// threads: one group's wait runs under the others' kernels; the step is the sum of the latency-bound kernels, two to three of
// which run side by side), and a frame that outgrows its slot's last counts costs a second pass.  FID_STAG_SPEC=1: both; 0: neither.
static bool stag_spec_enabled(bool grouped)
{
    const char *e = getenv("FID_STAG_SPEC");  // (read per frame: the tests switch roads inside one process)
    if (!e) return !grouped;
    return atoi(e) != 0;
}
static bool stag_spec_no_backoff()
{
    const char *e = getenv("FID_STAG_SPEC_BACKOFF");  // (0: queue ahead whenever a prediction exists, round 5's behaviour)
    return e && atoi(e) == 0;
}
static void stag_plan(const fid_stag_ctx *c, StagJob &j)
{
    const StagPred &p = c->pred;
    const long long n = (long long)c->maxW * c->maxH;
    auto up = [](long long v, long long slack, long long cap) { const long long x = v + v / 2 + slack; return (int)(x < cap ? x : cap); };
    StagPred &u = j.use;
    u = p;
    u.na = up(p.na, 1024, n);
    u.nc = up(p.nc, 64, c->max_comps);
    u.aout = up(p.aout, 8192, 3 * n);
    u.most = p.most * 2 < 65536 ? p.most * 2 : 65536;  // (above STAG_SORT_WAVE: the workgroup sort is launched as well)
    u.tile = p.tile + p.tile / 4;                      // (capped by the road's LDS limit where it is used; too small only costs time)
    u.ns = up(p.ns, 256, n / 8 + 16);
    u.nvs = up(p.nvs, 256, n / 8 + 15);
    u.nl = up(p.nl, 256, n / 9 + 16);
    u.nvl = up(p.nvl, 256, n / 9 + 16);
    u.nq = up(p.nq, 64, n / 9 + 16);
    u.nm = up(p.nm, 16, STAG_PIN_MARKERS);
}
// the counts of a frame that went through on the component-parallel road become the next frame's sizes
static void stag_learn(fid_stag_ctx *c, const StagJob &j, bool parallel_road)
{
    StagPred &p = c->pred;
    p.valid = parallel_road && c->n_anchors > 0 && c->n_markers <= STAG_PIN_MARKERS;
    p.W = c->W; p.H = c->H;
    p.na = (int)c->n_anchors; p.nc = j.cur[0]; p.aout = j.cur[5]; p.most = j.cur[9]; p.tile = j.cur[10];
    p.ns = c->rcount[0]; p.nvs = c->n_vsegs; p.nl = c->n_lines; p.nvl = c->n_vlines; p.nq = c->n_quads; p.nm = c->n_markers;
}
// (queued ahead: the guard also carries the true counts to the pinned block -- M0..M2 = {host field, device source, ints}; the
//  counted road keeps its copies)
struct StagMirror {
    void *dst;
    const void *src;
    int n;
};
static bool stag_guard_fill(fid_stag_ctx *c, int reset, StagGuard &g, StagMirror m0, StagMirror m1 = {nullptr, nullptr, 0}, StagMirror m2 = {nullptr, nullptr, 0})
{
    const StagMirror m[3] = {m0, m1, m2};
    g.reset = reset;
    g.bad = c->d_specbad;
    g.bad_host = (int *)stag_device_alias(&c->hp->spec_bad, 4);
    if (!g.bad_host) return false;
    for (int k = 0; k < 3; k++) {
        g.mdst[k] = m[k].dst ? (int *)stag_device_alias(m[k].dst, (size_t)m[k].n * 4) : nullptr;
        g.msrc[k] = (const int *)m[k].src;
        g.mn[k] = m[k].n;
        if (m[k].dst && !g.mdst[k]) return false;
    }
    return true;
}
// (the launch stands AT the call site: merged launches go out in site order, and a site inside a helper above its callers would
//  come back to a lower site number every time -- stag_order_guard would then issue everything recorded so far, unmerged)
#define STAG_GUARD_OR_FAIL(reset_, g_, ...)                                                                \
    do {                                                                                                   \
        StagGuard gg_ = g_;                                                                                \
        if (!stag_guard_fill(c, reset_, gg_, __VA_ARGS__)) return stag_finish(j, FID_E_HIP);               \
        STAG_LAUNCH(k_stag_spec_guard, dim3(1), dim3(64), 0, st, gg_);                                     \
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);                             \
    } while (0)

// one segment of the job; FID_OK while the job is under way or has finished well (j.done tells which)
#ifdef FID_DEBUG_STATS
static std::atomic<long long> g_stag_ns_sync(0), g_stag_ns_seg[12];
#define STAG_NOW() std::chrono::steady_clock::now()
#define STAG_NS(a, b) std::chrono::duration_cast<std::chrono::nanoseconds>((b) - (a)).count()
#endif
static fid_status stag_advance_impl(fid_stag_ctx *c, StagJob &j);
static fid_status stag_advance(fid_stag_ctx *c, StagJob &j)
{
    const bool was_spec = j.spec;
#ifdef FID_DEBUG_STATS
    const int seg0 = j.seg;
    const auto t0 = STAG_NOW();
    const fid_status rc = stag_advance_impl(c, j);
    g_stag_ns_seg[seg0 < 11 ? seg0 : 11] += STAG_NS(t0, STAG_NOW());
#else
    const fid_status rc = stag_advance_impl(c, j);
#endif
    // A frame queued ahead raises the stage flags while it ENQUEUES (its counts are adopted at the one wait at the end).  If it ends
    // in an error on the way, the context would keep "routed / validated / lined ..." beside the counts of the frame BEFORE it, and
    // fid_stag_tap_* would size their reads by those over half-written buffers: nothing of this frame is to be looked at.
    if (j.done && j.rc != FID_OK && j.rc != FID_E_CAPACITY && (was_spec || j.spec)) {
        c->routed = c->validated = c->lined = c->lines_validated = c->quadded = c->decoded = false;
        c->pred.valid = false;
    }
    return rc;
}
static fid_status stag_advance_impl(fid_stag_ctx *c, StagJob &j)
{
    if (j.done) return j.rc;
    hipStream_t st = c->group_stream ? c->group_stream : stag_stream(c);
    const bool grouped = c->group_stream != nullptr;  // (the group driver has waited for the group's stream already)
    // threads of the one-workgroup scans (block-size generic kernels): a frame on its own has the chip to itself and wants the scan
    // short (30 k anchors: 4 rounds of 1 024 threads against 15 of 256); in a group a 1 024-thread workgroup waits for sixteen free
    // wave slots on one CU longer than the scan takes
    const int scan_threads = grouped ? 256 : 1024;
    if (hipSetDevice(c->device) != hipSuccess) return stag_finish(j, FID_E_HIP);
#ifdef FID_DEBUG_STATS
    {
        const auto t0 = STAG_NOW();
        if (j.seg > 0 && !grouped && hipStreamSynchronize(st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        g_stag_ns_sync += STAG_NS(t0, STAG_NOW());
    }
#else
    if (j.seg > 0 && !grouped && hipStreamSynchronize(st) != hipSuccess) return stag_finish(j, FID_E_HIP);
#endif
    const int GRADIENT_THRESH = 16, ANCHOR_THRESH = 0, SCAN_INTERVAL = 1;  // DetectEdgesByEDPF, ED.cpp:155-169
    for (;;) {  // (a frame queued ahead passes all its segments in this one call: "continue" where the counted road returns)
    switch (j.seg) {
    case 0: {  // ---- smoothing, gradient, anchors, anchor sort
        if (!j.gray || j.width < 8 || j.height < 8 || j.width > c->maxW || j.height > c->maxH || j.stride < j.width) return stag_finish(j, FID_E_INVALID_ARG);
        if (j.last >= SS_UNREFINED && !c->d_words) return stag_finish(j, FID_E_INVALID_ARG);  // no marker library loaded
        const int W = j.width, H = j.height;
        j.spec = stag_spec_enabled(grouped) && !j.nospec && j.last >= SS_MARKERS && c->route_mode == 1 && c->pred.valid && c->pred.W == W && c->pred.H == H &&
                 (!j.out || j.cap > 0) && stag_device_alias(c->hp, sizeof(fid_stag_ctx::Pinned)) != nullptr;
        if (j.spec && c->spec_skip > 0 && !stag_spec_no_backoff()) {  // (backed off: this frame takes the counted road)
            c->spec_skip--;
            j.spec = false;
        }
        if (j.spec) stag_plan(c, j);
        if (!j.staged)
            for (int y = 0; y < H; y++) memcpy(c->h_src + (size_t)y * W, j.gray + (size_t)y * j.stride, (size_t)W);
        if (STAG_MEMCPY(c->d_src, c->h_src, (size_t)W * H, hipMemcpyHostToDevice, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (STAG_MEMSET(c->d_rowhist, 0, (size_t)H * STAG_BINS * 2, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        STAG_LAUNCH(k_stag_smooth_grad, dim3((W + SX - 1) / SX, (H + SY - 1) / SY), dim3(256), 0, st, c->d_src, W, W, H, GRADIENT_THRESH,
                           c->d_smooth, c->d_grad, c->d_dir);
        const int blocks = 2048, nbands = (H + STAG_BAND_ROWS - 1) / STAG_BAND_ROWS;
        STAG_LAUNCH(k_stag_anchors, dim3(blocks), dim3(256), 0, st, c->d_grad, c->d_dir, W, H, GRADIENT_THRESH, ANCHOR_THRESH, SCAN_INTERVAL,
                           c->d_edge, c->d_rowhist);
        STAG_LAUNCH(k_stag_bandsum, dim3(STAG_BINS / 256, nbands), dim3(256), 0, st, c->d_rowhist, H, c->d_bandhist);
        STAG_LAUNCH(k_stag_bandscan, dim3(STAG_BINS / 256), dim3(256), 0, st, c->d_bandhist, nbands, c->d_tot);
        STAG_LAUNCH(k_stag_scan, dim3(1), dim3(512), 0, st, c->d_tot, c->d_bstart, c->d_n);
        STAG_LAUNCH(k_stag_place, dim3(nbands), dim3(64 * STAG_BAND_ROWS), 0, st, c->d_grad, c->d_edge, W, H, c->d_rowhist, c->d_bandhist,
                           c->d_bstart, c->d_sorted);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(&c->hp->n_anchors, c->d_n, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{(int *)c->d_n}, {j.use.na}, {(int *)c->d_n}, 1, 1};
            STAG_GUARD_OR_FAIL(1, g, {&c->hp->n_anchors, c->d_n, 1});
        }
        j.seg = 1;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 1: {  // ---- routing: the component-parallel road up to the component table (or the sequential road)
        if (!j.spec) c->n_anchors = c->hp->n_anchors;  // (queued ahead: the true counts are taken over at the frame's one wait)
        c->W = j.width;
        c->H = j.height;
        // (a frame that is refused further down must not leave the stages of the frame before readable: taps, fid_stag_pose_last)
        c->routed = c->validated = c->lined = c->lines_validated = c->quadded = c->decoded = false;
        if (j.last == SS_FRONTEND) return stag_finish(j, FID_OK);
        const int W = c->W, H = c->H, n = W * H, na = j.spec ? j.use.na : (int)c->n_anchors;
        StagRoute &R = j.R;
        R.grad = c->d_grad; R.dir = c->d_dir; R.edge = c->d_edgeimg; R.W = W; R.H = H;
        R.pix = c->d_rpix; R.stack = c->d_rstack; R.chains = c->d_chains; R.chainNos = c->d_chainnos;
        R.capPix = (int)((size_t)c->maxW * c->maxH); R.capStack = R.capPix; R.capChains = 32767; R.capNos = (c->maxW + c->maxH) * 8;
        R.outpix = c->d_outpix; R.segs = c->d_segs; R.capOut = R.capPix; R.capSegs = R.capPix / 8 + 16;
        R.counters = c->d_rcount;
        j.seg = 2;
        if (c->route_mode != 1) return stag_launch_route_seq(c, j) ? FID_OK : stag_finish(j, FID_E_HIP);
        if (STAG_MEMCPY(c->d_edgeimg, c->d_edge, (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        // (EdgeMap::pixels needs no clearing on this road: k_stag_route_gather writes every entry below the final count from
        //  the cleared arenas; the sequential road clears it itself -- 16.6 MB of writes per 1080p frame less)
        if (na == 0) {
            c->rcount[0] = c->rcount[1] = c->rcount[2] = 0;
            j.rstate = RS_EMPTY;
            return STAG_MEMSET(c->d_rcount, 0, 12, st) == hipSuccess ? FID_OK : stag_finish(j, FID_E_HIP);
        }
        // (the per-root counters are zeroed by k_stag_ccl_init where a root can be; the output arena is cleared once its used
        // size is known: clearing the whole allocations cost 70 MB of writes per frame)
        bool ok = true;
        {
            // (all of them are hipMalloc'ed, i.e. 256-byte aligned, and sized in whole 16-byte words or rounded up inside their allocation)
            StagFills F;
            void *ptr[6] = {c->d_cursors, c->d_fill, c->d_aslots, c->d_prodflag, c->d_blkpix, c->d_blksegs};
            const size_t bytes[6] = {64, (size_t)c->max_comps * 4, (size_t)c->cap_aslots * 4, (size_t)na * 4, (size_t)na * 4, (size_t)na * 4};
            const unsigned val[6] = {0u, 0u, 0xffffffffu, 0u, 0u, 0u};
            unsigned most = 0;
            for (int k = 0; k < 6; k++) {
                F.p[k] = (uint4 *)ptr[k];
                const unsigned n16 = (unsigned)((bytes[k] + 15) / 16);
                if (k < 3) F.n16[k] = n16;
                else F.na16 = n16;  // (the three of them are sized by the frame's anchors)
                F.v[k] = val[k];
                most = n16 > most ? n16 : most;
            }
            STAG_LAUNCH(k_stag_fills, dim3((most + 255) / 256), dim3(256), 0, st, F);
        }
        if (!ok) return stag_finish(j, FID_E_HIP);
        {
            const dim3 tiles((W + CCL_TW - 1) / CCL_TW, (H + CCL_TH - 1) / CCL_TH);
            STAG_LAUNCH(k_stag_ccl_tile, tiles, dim3(256), 0, st, c->d_grad, W, H, 16, c->d_label, c->d_csize, c->d_canch, c->d_cbox, c->d_tilefg);
            STAG_LAUNCH(k_stag_ccl_border, tiles, dim3(128), 0, st, W, H, c->d_label);
            STAG_LAUNCH(k_stag_ccl_flatten, tiles, dim3(256), 0, st, W, H, c->d_label, c->d_edge, c->d_csize, c->d_canch, c->d_cbox, c->d_roots, c->d_cursors, c->d_tilefg);
        }
        // (one thread per ROOT of k_stag_ccl_flatten's list, 64 workgroups going through it: a frame has a few hundred to a few thousand roots)
        STAG_LAUNCH(k_stag_comp_alloc, dim3(64), dim3(256), 0, st, c->d_roots, c->d_csize, c->d_canch, c->d_cbox,
                           c->d_cursors, c->max_comps, c->d_caps, c->d_comps, c->d_cidmap);
        STAG_LAUNCH(k_stag_comp_fill, dim3((na + 255) / 256), dim3(256), 0, st, c->d_sorted, c->d_n, c->d_label, c->d_cidmap, c->d_comps, c->d_fill,
                           c->d_aslots);
        // the largest LDS tile a component's walk may ask for.  One frame at a time: 150 of the 160 KB of a CU (every component of a
        // marker frame walks in LDS).  A group of frames: 40 KB -- the walk kernel allocates the group's largest tile for every
        // workgroup, and with ~80 KB marker tiles a CU held one workgroup (3.0 k frames/s; 64 KB: 3.3 k, 40 KB: 3.5 k, 24 KB: 3.3 k);
        // the components above the cap take the global-memory walk, same result.  FID_STAG_TILE_KB overrides.
        // (round 6, groups of 32 on 256 slots: 48 - 50 KB + the walk's 2.5 KB of stack = three workgroups per CU: 7.47 - 7.56 k frames/s against
        //  7.29 - 7.33 k at 37 KB (four per CU), 7.1 - 7.2 k at 24 / 60 / 76 KB, 6.1 k at 150: more of the marker components keep their tile in
        //  LDS, and the walk's registers hold it to four workgroups per CU anyway)
        const int tile_kb = c->tile_kb_env > 0 ? c->tile_kb_env : (grouped ? 50 : 150);
        const int LDS_CAP = (tile_kb < 8 ? 8 : (tile_kb > 150 ? 150 : tile_kb)) * 1024;
        j.lds_cap = LDS_CAP;
        STAG_LAUNCH(k_stag_comp_tilemax, dim3((c->max_comps + 255) / 256), dim3(256), 0, st, c->d_comps, c->d_cursors, LDS_CAP, c->d_corder, c->max_comps);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(c->hp->cur, c->d_cursors, 44, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            // (components, output-arena pixels, the allocation overflow flag, most anchors in one component against the sort that is launched)
            const StagGuard g = {{c->d_cursors, c->d_cursors + 5, c->d_cursors + 7, c->d_cursors + 9},
                                 {j.use.nc, j.use.aout, 0, j.use.most > STAG_SORT_WAVE ? 65536 : STAG_SORT_WAVE},
                                 {c->d_cursors, (int *)c->d_n}, 4, 2};
            STAG_GUARD_OR_FAIL(0, g, {c->hp->cur, c->d_cursors, 11});
        }
        j.rstate = RS_PAR_A;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 2: {  // ---- routing, second half; then edge validation
        if (j.rstate == RS_PAR_A) {
            if (j.spec) {
                memset(j.cur, 0, sizeof(j.cur));
                j.cur[0] = j.use.nc;
                j.cur[5] = j.use.aout;
                j.cur[9] = j.use.most;
                j.cur[10] = j.use.tile < j.lds_cap ? j.use.tile : j.lds_cap;
            } else {
                memcpy(j.cur, c->hp->cur, sizeof(j.cur));
            }
            const int *cur = j.cur;
            // an arena too small, or one component holding (nearly) all anchors -- a frame of noise -- which leaves nothing to
            // run side by side (sorting its anchors would cost more than the sequential road's single pass over the globally
            // sorted list): same result by the sequential road
            if (cur[7] || cur[9] > 65536) {
                c->route_fallbacks++;
                return stag_launch_route_seq(c, j) ? FID_OK : stag_finish(j, FID_E_HIP);
            }
            const int na = j.spec ? j.use.na : (int)c->n_anchors;
            // pixels of the output arena the extraction does not write read as (-1, -1), like the reference's untouched array
            if (cur[5] > 0 && STAG_MEMSET(c->d_aout, 0xff, (size_t)cur[5] * sizeof(int2), st) != hipSuccess) return stag_finish(j, FID_E_HIP);
            const int nc = cur[0];
            StagArenas A;
            A.pix = c->d_apix; A.stack = c->d_astack; A.chains = c->d_achains; A.out = c->d_aout; A.segs = c->d_asegs; A.recs = c->d_recs;
            int *ovf = c->d_cursors + 8;
            if (nc > 0) {
                STAG_LAUNCH(k_stag_comp_sort, dim3((nc + 3) / 4), dim3(256), 0, st, c->d_comps, c->d_cursors, c->d_aslots);
                if (cur[9] > STAG_SORT_WAVE)  // (cur[9]: most anchors in one component)
                {
                    // LDS and threads by the frame's largest slice (cur[9] anchors, padded to a power of two like k_stag_comp_alloc pads it):
                    // a marker frame's slices are 1 - 4 K entries, and a 1 024-thread workgroup with the full 64 KB waited five times
                    // its own duration for room on a CU beside the other groups' kernels.  A frame queued ahead knows only a PREDICTED
                    // count (its guard admits up to 65 536): the full size there.
                    int p2 = 1;
                    while (p2 < cur[9] && p2 < STAG_SORT_BIG) p2 <<= 1;
                    const int slice = j.spec ? STAG_SORT_BIG : p2;
                    STAG_LAUNCH(k_stag_comp_sort_big, dim3(!grouped || nc < 24 ? nc : 24), dim3(slice <= 4096 ? 256 : 1024), (size_t)slice * 4, st, c->d_comps,
                                       c->d_cursors, c->d_aslots);
                }
                // LDS per workgroup = the largest tile a component of this frame asks for (components whose box does not fit 150 KB
                // walk in global memory): frames of small components keep many workgroups per CU
                const int no_sparse = c->no_sparse;  // (FID_STAG_SPARSE=0: no blocks, the walk in global memory)
                // (a group: every frame asks for the cap -- one value for the whole launch, which allocates the largest request anyway,
                //  and at 99 VGPRs the kernel holds four workgroups per CU whether they own 8 or 37 KB)
                const int lds = !c->route_tile ? 0 : (grouped ? j.lds_cap : ((cur[10] + 1023) / 1024) * 1024);
                if (grouped && c->route_tile && lds > STAG_SMALL_TILE) {
                    // (two launches by class: see k_stag_route_walk_impl; both grids cover every rank, a workgroup beyond its class returns)
                    STAG_LAUNCH(k_stag_route_walk, dim3(nc), dim3(256), (size_t)lds, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder, c->d_sorted,
                                       c->d_aslots, c->d_label, 16, lds | no_sparse, c->d_prodflag, ovf, 1);
                    STAG_LAUNCH(k_stag_route_walk, dim3(nc), dim3(256), (size_t)STAG_SMALL_TILE, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder,
                                       c->d_sorted, c->d_aslots, c->d_label, 16, STAG_SMALL_TILE | no_sparse, c->d_prodflag, ovf, 2);
                } else {
                    STAG_LAUNCH(k_stag_route_walk, dim3(nc), dim3(256), (size_t)lds, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder, c->d_sorted,
                                       c->d_aslots, c->d_label, 16, lds | no_sparse, c->d_prodflag, ovf, 0);
                }
            }
            STAG_LAUNCH(k_stag_next_above, dim3(1), dim3(scan_threads), 0, st, c->d_prodflag, c->d_n, c->d_next);
            if (nc > 0 && grouped) {  // (two launches by class, like the walk: the big components, then the small ones four workgroups to a CU)
                STAG_LAUNCH(k_stag_route_extract_big, dim3(nc), dim3(64), 0, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder, c->d_next,
                                   c->d_n, c->d_blkpix, c->d_blksegs, c->d_blkwhere, ovf, 1);
                STAG_LAUNCH(k_stag_route_extract_small, dim3((nc + 3) / 4), dim3(256), 0, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder,
                                   c->d_next, c->d_n, c->d_blkpix, c->d_blksegs, c->d_blkwhere, ovf, 2);
            } else if (nc > 0) {
                STAG_LAUNCH(k_stag_route_extract, dim3((nc + 3) / 4), dim3(256), 0, st, j.R, A, c->d_comps, c->d_cursors, c->d_corder, c->d_next,
                                   c->d_n, c->d_blkpix, c->d_blksegs, c->d_blkwhere, ovf, 0);
            }
            {
                StagScanJobs sj;
                sj.counts[0] = c->d_blkpix; sj.total[0] = c->d_rcount + 1;
                sj.counts[1] = c->d_blksegs; sj.total[1] = c->d_rcount;
                STAG_LAUNCH(k_stag_scan_counts_n, dim3(2), dim3(scan_threads), 0, st, sj, (const int *)c->d_n);
            }
            // (anchors per wave: 64 in a group -- a lane asks, the wave copies what produced, one after the other; 1 for a frame on its own,
            //  where the chip is empty and the copies of a wave's producing anchors should not queue behind each other)
            const int gpw = grouped ? 64 : 1;
            STAG_LAUNCH(k_stag_route_gather, dim3((na + 4 * gpw - 1) / (4 * gpw)), dim3(256), 0, st, gpw, A, c->d_comps, c->d_n, c->d_prodflag, c->d_blkpix, c->d_blksegs,
                               c->d_blkwhere, c->d_outpix, c->d_segs, j.R.capOut, j.R.capSegs, ovf);
            if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
            if (!j.spec && (STAG_MEMCPY(c->hp->rcount, c->d_rcount, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                            STAG_MEMCPY(&c->hp->ovf, ovf, 4, hipMemcpyDeviceToHost, st) != hipSuccess))
                return stag_finish(j, FID_E_HIP);
            if (j.spec) {
                const StagGuard g = {{c->d_rcount, ovf}, {j.use.ns, 0}, {c->d_rcount, c->d_rcount + 1}, 2, 2};
                STAG_GUARD_OR_FAIL(0, g, {c->hp->rcount, c->d_rcount, 2}, {&c->hp->ovf, ovf, 1});
            }
            j.rstate = RS_PAR_B;
            if (!j.spec) return FID_OK;  // (this segment again, with the second half's counts)
            continue;
        }
        if (j.rstate == RS_PAR_B && !j.spec) {
            j.ovf = c->hp->ovf;
            c->rcount[0] = c->hp->rcount[0];
            c->rcount[1] = c->hp->rcount[1];
            c->rcount[2] = j.ovf;
            if (j.ovf) {  // an arena of the parallel road was too small: same result by the sequential road
                c->route_fallbacks++;
                return stag_launch_route_seq(c, j) ? FID_OK : stag_finish(j, FID_E_HIP);
            }
        } else if (j.rstate == RS_SEQ) {
            c->rcount[0] = c->hp->rcount[0];
            c->rcount[1] = c->hp->rcount[1];
            c->rcount[2] = c->hp->rcount[2];
            if (c->rcount[2]) return stag_finish(j, FID_E_CAPACITY);
        }
        c->routed = true;
        c->validated = false;
        if (j.last == SS_EDGES) return stag_finish(j, FID_OK);
        const int W = c->W, H = c->H;
        const size_t n = (size_t)W * H;
        const int ns = j.spec ? j.use.ns : c->rcount[0];
        // ValidateEdgeSegments starts from an empty edge image (ValidateEdgeSegments.cpp:370)
        if (STAG_MEMSET(c->d_edgeimg, 0, n, st) != hipSuccess || STAG_MEMSET(c->d_vhist, 0, STAG_BINS * 4 * STAG_VHIST_SLICES, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        STAG_LAUNCH(k_stag_smooth3_prewitt, dim3((W + SX - 1) / SX, (H + SY - 1) / SY), dim3(256), 0, st, c->d_src, W, W, H, c->d_smooth2,
                           c->d_vgrad, c->d_vhist);
        STAG_LAUNCH(k_stag_valid_prob, dim3(1), dim3(512), 0, st, c->d_vhist, W, H, c->d_segs, c->d_rcount, c->d_prob, c->d_np);
        const int wg = (ns + 3) / 4;
        if (wg > 0) {
            STAG_LAUNCH(k_stag_test_segments, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_vgrad, W, c->d_prob, c->d_np,
                               2.25, c->d_vstack, c->d_edgeimg);
            STAG_LAUNCH(k_stag_extract, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_edgeimg, W, c->d_vcounts,
                               c->d_vsegs, 0);
        }
        STAG_LAUNCH(k_stag_scan_counts, dim3(1), dim3(scan_threads), 0, st, c->d_vcounts, c->d_rcount, c->d_vtotal);
        if (wg > 0)
            STAG_LAUNCH(k_stag_extract, dim3(wg), dim3(256), 0, st, c->d_segs, c->d_rcount, c->d_outpix, c->d_edgeimg, W, c->d_vcounts,
                               c->d_vsegs, 1);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && (STAG_MEMCPY(&c->hp->n_vsegs, c->d_vtotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        STAG_MEMCPY(&c->hp->np, c->d_np, 4, hipMemcpyDeviceToHost, st) != hipSuccess))
            return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{c->d_vtotal}, {j.use.nvs}, {c->d_vtotal}, 1, 1};
            STAG_GUARD_OR_FAIL(0, g, {&c->hp->n_vsegs, c->d_vtotal, 1}, {&c->hp->np, c->d_np, 1});
        }
        j.seg = 3;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 3: {  // ---- EDLines: split, join, gather
        if (!j.spec) {
            c->n_vsegs = c->hp->n_vsegs;
            c->np = c->hp->np;
        }
        c->validated = true;
        c->lined = false;
        if (j.last == SS_EDGES_VALIDATED) return stag_finish(j, FID_OK);
        const int ns = j.spec ? j.use.nvs : c->n_vsegs;
        c->min_line_len = stag_min_line_len(c->W, c->H);
        StagPrefix PF;
        PF.x = c->d_prefix; PF.y = PF.x + c->prefcap; PF.xx = PF.y + c->prefcap; PF.yy = PF.xx + c->prefcap; PF.xy = PF.yy + c->prefcap;
        const int wg = (ns + 63) / 64;
        if (wg > 0) {
            // pixels per wave that live in LDS (fid_stag_lines.hip): a frame on its own has the CUs to itself, a group keeps four
            // workgroups per CU resident.  FID_STAG_SPLIT_LDS overrides (0: global memory throughout).
            int lds_pix = c->split_lds_env >= 0 ? c->split_lds_env : (grouped ? 256 : 1024);
            lds_pix = lds_pix > 1024 ? 1024 : lds_pix;
            if (grouped && c->split_lds_env < 0) {
                // (two launches of one-wave workgroups by segment length: see k_stag_split_lines_impl)
                STAG_LAUNCH(k_stag_split_lines, dim3(ns), dim3(64), (size_t)sl_lds_wave_bytes(1024), st, c->d_vsegs, c->d_vtotal, c->d_outpix, PF,
                                   c->min_line_len, 1.0, c->d_lslots, c->d_lcounts, 1024, 256, 0x7fffffff);
                STAG_LAUNCH(k_stag_split_lines, dim3(ns), dim3(64), (size_t)sl_lds_wave_bytes(256), st, c->d_vsegs, c->d_vtotal, c->d_outpix, PF,
                                   c->min_line_len, 1.0, c->d_lslots, c->d_lcounts, 256, -1, 256);
            } else {
                STAG_LAUNCH(k_stag_split_lines, dim3((ns + 3) / 4), dim3(256), (size_t)SL_LDS_BYTES(lds_pix), st, c->d_vsegs, c->d_vtotal, c->d_outpix, PF,
                                   c->min_line_len, 1.0, c->d_lslots, c->d_lcounts, lds_pix, -1, 0x7fffffff);
            }
        }
        STAG_LAUNCH(k_stag_scan_counts, dim3(1), dim3(scan_threads), 0, st, c->d_lcounts, c->d_vtotal, c->d_ltotal);
        if (wg > 0)
            STAG_LAUNCH(k_stag_gather_lines, dim3(wg), dim3(64), 0, st, c->d_vsegs, c->d_vtotal, c->d_lcounts, c->d_ltotal, c->d_lslots, c->d_lines);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(&c->hp->n_lines, c->d_ltotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{c->d_ltotal}, {j.use.nl}, {c->d_ltotal}, 1, 1};
            STAG_GUARD_OR_FAIL(0, g, {&c->hp->n_lines, c->d_ltotal, 1});
        }
        j.seg = 4;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 4: {  // ---- line validation
        if (!j.spec) c->n_lines = c->hp->n_lines;
        c->lined = true;
        c->lines_validated = false;
        if (j.last == SS_LINES) return stag_finish(j, FID_OK);
        const int W = c->W, H = c->H;
        if (c->kmin_w != W || c->kmin_h != H) {  // the false-alarm table is a function of the image size
            std::vector<int> kmin;
            stag_build_kmin(W, H, kmin);
            if (hipMemcpy(c->d_kmin, kmin.data(), kmin.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return stag_finish(j, FID_E_HIP);
            c->kmin_w = W;
            c->kmin_h = H;
            c->kmin_n = (int)kmin.size() - 1;
        }
        StagLineTables T;
        T.atan_lut = c->d_atan_lut;
        T.kmin = c->d_kmin;
        T.kmin_n = c->kmin_n;
        const int nl = j.spec ? j.use.nl : c->n_lines;
        if (nl > 0)
            STAG_LAUNCH(k_stag_validate_lines, dim3((nl + 3) / 4), dim3(256), 0, st, c->d_lines, c->d_ltotal, c->d_src, W, H, c->d_vsegs,
                               c->d_outpix, T, c->d_lflags);
        STAG_LAUNCH(k_stag_scan_counts, dim3(1), dim3(scan_threads), 0, st, c->d_lflags, c->d_ltotal, c->d_vltotal);
        if (nl > 0)
            STAG_LAUNCH(k_stag_compact_lines, dim3((nl + 255) / 256), dim3(256), 0, st, c->d_lines, c->d_ltotal, c->d_lflags, c->d_vltotal,
                               c->d_vlines);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(&c->hp->n_vlines, c->d_vltotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{c->d_vltotal}, {j.use.nvl}, {c->d_vltotal}, 1, 1};
            STAG_GUARD_OR_FAIL(0, g, {&c->hp->n_vlines, c->d_vltotal, 1});
        }
        j.seg = 5;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 5: {  // ---- quads
        if (!j.spec) c->n_vlines = c->hp->n_vlines;
        c->lines_validated = true;
        c->quadded = false;
        if (j.last == SS_LINES_VALIDATED) return stag_finish(j, FID_OK);
        const int W = c->W, H = c->H, ns = j.spec ? j.use.nvs : c->n_vsegs, nl = j.spec ? j.use.nvl : c->n_vlines;
        if (STAG_MEMSET(c->d_lrange, 0, (size_t)(ns + 1) * sizeof(int2), st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (nl > 0) STAG_LAUNCH(k_stag_line_ranges, dim3((nl + 255) / 256), dim3(256), 0, st, c->d_vlines, c->d_vltotal, c->d_lrange);
        if (ns > 0)
            STAG_LAUNCH(k_stag_quads, dim3((ns + 3) / 4), dim3(256), 0, st, c->d_vlines, c->d_lrange, c->d_vtotal, c->d_vsegs, c->d_outpix, c->d_src,
                               W, H, c->d_corners, c->d_order, c->d_qslots, c->d_qcounts);
        STAG_LAUNCH(k_stag_scan_counts, dim3(1), dim3(scan_threads), 0, st, c->d_qcounts, c->d_vtotal, c->d_qtotal);
        if (ns > 0)
            STAG_LAUNCH(k_stag_gather_quads, dim3((ns + 63) / 64), dim3(64), 0, st, c->d_lrange, c->d_vtotal, c->d_qcounts, c->d_qtotal, c->d_qslots,
                               c->d_quads);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(&c->hp->n_quads, c->d_qtotal, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{c->d_qtotal}, {j.use.nq}, {c->d_qtotal}, 1, 1};
            STAG_GUARD_OR_FAIL(0, g, {&c->hp->n_quads, c->d_qtotal, 1});
        }
        j.seg = 6;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 6: {  // ---- code reading, decoding, duplicates
        if (!j.spec) c->n_quads = c->hp->n_quads;
        c->quadded = true;
        c->decoded = false;
        if (j.last == SS_QUADS) return stag_finish(j, FID_OK);
        const int nq = j.spec ? j.use.nq : c->n_quads;
        if (nq > 0)
            STAG_LAUNCH(k_stag_decode, dim3((nq + 3) / 4), dim3(256), 0, st, c->d_quads, c->d_qtotal, c->d_src, c->W, c->H, c->d_locs, c->d_words,
                               c->n_words, c->errorCorrection, c->d_cand, c->d_found);
        STAG_LAUNCH(k_stag_dedup, dim3(1), dim3(64), 0, st, c->d_cand, c->d_found, c->d_qtotal, c->d_markers, c->d_nmarkers);
        if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (!j.spec && STAG_MEMCPY(&c->hp->n_markers, c->d_nmarkers, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return stag_finish(j, FID_E_HIP);
        if (j.spec) {
            const StagGuard g = {{c->d_nmarkers}, {j.use.nm}, {c->d_nmarkers}, 1, 1};
            STAG_GUARD_OR_FAIL(0, g, {&c->hp->n_markers, c->d_nmarkers, 1});
        }
        j.seg = 7;
        if (!j.spec) return FID_OK;
        continue;
    }
    case 7: {  // ---- pose refinement of the markers (and, if asked for, the 5-point pose right behind it)
        if (j.spec) {
            // queued ahead: as many workgroups and result slots as the last frame's markers (with the margin); the counts, the
            // capacity checks and the hand-over wait for the frame's one wait (default:)
            const int nm = j.use.nm;
            STAG_LAUNCH(k_stag_refine, dim3(nm), dim3(64), 0, st, c->d_markers, c->d_nmarkers, c->d_vsegs, c->d_vtotal, c->d_outpix, c->d_chosen);
            if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
            if (j.out && STAG_MEMCPY(c->hp->markers, c->d_markers, (size_t)nm * sizeof(fid_stag_marker), hipMemcpyDeviceToHost, st) != hipSuccess)
                return stag_finish(j, FID_E_HIP);
            if (j.last == SS_POSE) {
                PoseCam cam;
                for (int i = 0; i < 9; i++) cam.K[i] = j.K[i];
                for (int i = 0; i < 5; i++) cam.D[i] = j.D ? j.D[i] : 0.0;
                cam.fiducial_len = j.marker_size;
                STAG_LAUNCH(k_stag_pose, dim3((nm + 3) / 4), dim3(64), 0, st, c->d_markers, c->d_nmarkers, cam, j.marker_size, c->d_poses);
                if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
                if (STAG_MEMCPY(c->hp->poses, c->d_poses, (size_t)nm * sizeof(fid_stag_pose_out), hipMemcpyDeviceToHost, st) != hipSuccess)
                    return stag_finish(j, FID_E_HIP);
            }
            // (the frame's flag is in the pinned block already: every k_stag_spec_guard writes it there)
            j.seg = 8;
            return FID_OK;  // (the frame's one wait)
        }
        c->n_markers = c->hp->n_markers;
        c->decoded = true;
        if (j.last == SS_UNREFINED) return stag_finish(j, FID_OK);
        if (c->n_markers > 0) {
            STAG_LAUNCH(k_stag_refine, dim3(c->n_markers), dim3(64), 0, st, c->d_markers, c->d_nmarkers, c->d_vsegs, c->d_vtotal, c->d_outpix,
                               c->d_chosen);
            if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
        }
        // the capacity checks come first: on FID_E_CAPACITY nothing is copied, so the count handed back must be 0 (a caller of
        // the batch entry point walks n_per_frame[f] entries of its cap_per_frame slot; c->n_markers keeps what was found)
        if ((j.out && c->n_markers > j.cap) || (j.last == SS_POSE && c->n_markers > 0 && c->n_markers > j.pose_cap)) {
            if (j.n_out) *j.n_out = 0;
            return stag_finish(j, FID_E_CAPACITY);
        }
        if (j.n_out) *j.n_out = c->n_markers;
        const bool pin = c->n_markers <= STAG_PIN_MARKERS;  // staged through pinned memory, handed over in the last segment
        if (j.out) {
            if (c->n_markers > 0 && STAG_MEMCPY(pin ? c->hp->markers : j.out, c->d_markers, (size_t)c->n_markers * sizeof(fid_stag_marker),
                                                   hipMemcpyDeviceToHost, st) != hipSuccess)
                return stag_finish(j, FID_E_HIP);
        }
        if (j.last == SS_POSE && c->n_markers > 0) {
            PoseCam cam;
            for (int i = 0; i < 9; i++) cam.K[i] = j.K[i];
            for (int i = 0; i < 5; i++) cam.D[i] = j.D ? j.D[i] : 0.0;
            cam.fiducial_len = j.marker_size;
            STAG_LAUNCH(k_stag_pose, dim3((c->n_markers + 3) / 4), dim3(64), 0, st, c->d_markers, c->d_nmarkers, cam, j.marker_size, c->d_poses);
            if (hipGetLastError() != hipSuccess) return stag_finish(j, FID_E_HIP);
            if (STAG_MEMCPY(pin ? c->hp->poses : j.poses, c->d_poses, (size_t)c->n_markers * sizeof(fid_stag_pose_out), hipMemcpyDeviceToHost, st) !=
                hipSuccess)
                return stag_finish(j, FID_E_HIP);
        }
        j.seg = 8;
        return FID_OK;
    }
    default:  // the copies of segment 7 have landed
        if (j.spec) {
            // the frame's one wait is over: every true count against what the launches were sized for
            const fid_stag_ctx::Pinned &h = *c->hp;
            const StagPred &u = j.use;
            const bool fits = !h.spec_bad && h.n_anchors > 0 && h.n_anchors <= (unsigned)u.na && h.cur[0] <= u.nc && h.cur[5] <= u.aout && !h.cur[7] &&
                              h.cur[9] <= (u.most > STAG_SORT_WAVE ? 65536 : STAG_SORT_WAVE) && !h.ovf && h.rcount[0] <= u.ns && h.n_vsegs <= u.nvs &&
                              h.n_lines <= u.nl && h.n_vlines <= u.nvl && h.n_quads <= u.nq && h.n_markers <= u.nm;
            c->spec_frames++;
            c->spec_hist = (c->spec_hist << 1) | (fits ? 0u : 1u);
            if (fits) c->spec_pause = 16;
            if (!fits) {  // the counted road from the first segment: the same result, nine waits
                c->spec_misses++;
                if (__builtin_popcount(c->spec_hist & 0xfu) >= 2) {
                    c->spec_skip = c->spec_pause;
                    c->spec_pause = c->spec_pause < 256 ? c->spec_pause * 2 : 256;
                    c->spec_hist = 0;
                    c->spec_backoffs++;
                }
                c->pred.valid = false;
                j.spec = false;
                j.nospec = true;
                j.seg = 0;
                j.rstate = RS_NONE;
                continue;
            }
            c->n_anchors = h.n_anchors;
            memcpy(j.cur, h.cur, sizeof(j.cur));
            c->rcount[0] = h.rcount[0]; c->rcount[1] = h.rcount[1]; c->rcount[2] = 0;
            c->n_vsegs = h.n_vsegs; c->np = h.np; c->n_lines = h.n_lines; c->n_vlines = h.n_vlines; c->n_quads = h.n_quads; c->n_markers = h.n_markers;
            c->routed = c->validated = c->lined = c->lines_validated = c->quadded = c->decoded = true;
            stag_learn(c, j, true);
            if ((j.out && c->n_markers > j.cap) || (j.last == SS_POSE && c->n_markers > 0 && c->n_markers > j.pose_cap)) {
                if (j.n_out) *j.n_out = 0;
                return stag_finish(j, FID_E_CAPACITY);
            }
            if (j.n_out) *j.n_out = c->n_markers;
            if (c->n_markers > 0) {
                if (j.out) memcpy(j.out, h.markers, (size_t)c->n_markers * sizeof(fid_stag_marker));
                if (j.last == SS_POSE && j.poses) memcpy(j.poses, h.poses, (size_t)c->n_markers * sizeof(fid_stag_pose_out));
            }
            return stag_finish(j, FID_OK);
        }
        if (c->n_markers > 0 && c->n_markers <= STAG_PIN_MARKERS) {
            if (j.out) memcpy(j.out, c->hp->markers, (size_t)c->n_markers * sizeof(fid_stag_marker));
            if (j.last == SS_POSE && j.poses) memcpy(j.poses, c->hp->poses, (size_t)c->n_markers * sizeof(fid_stag_pose_out));
        }
        stag_learn(c, j, j.rstate == RS_PAR_B);
        return stag_finish(j, FID_OK);
    }
    }  // for (;;)
}

// one frame, every segment back to back
static fid_status stag_run(fid_stag_ctx *c, StagJob &j)
{
    if (!c) return FID_E_INVALID_ARG;
    while (!j.done) (void)stag_advance(c, j);
    return j.rc;
}

static fid_status stag_run_to(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride, int last)
{
    StagJob j;
    j.gray = gray; j.width = width; j.height = height; j.stride = stride; j.last = last;
    return stag_run(c, j);
}

extern "C" {

fid_status fid_stag_edge_frontend(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_FRONTEND);
}

fid_status fid_stag_detect_edges(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_EDGES);
}

fid_status fid_stag_detect_edges_validated(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_EDGES_VALIDATED);
}

fid_status fid_stag_detect_lines(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_LINES);
}

fid_status fid_stag_detect_lines_validated(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_LINES_VALIDATED);
}

fid_status fid_stag_detect_quads(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    return stag_run_to(c, gray, width, height, stride, SS_QUADS);
}

// the host-made tables of the STag path (functions of the image size alone), without a device: for tests and diagnostics
fid_status fid_stag_host_tables(int32_t width, int32_t height, int32_t *kmin, int32_t kmin_cap, int32_t *kmin_n, int32_t *lut_size,
                                double *code_locations, int32_t *min_line_len)
{
    if (width < 8 || height < 8) return FID_E_INVALID_ARG;
    std::vector<int> k;
    stag_build_kmin(width, height, k);
    if (kmin_n) *kmin_n = (int32_t)k.size();
    if (lut_size) *lut_size = (width + height) / 8;
    if (kmin) {
        if ((int)k.size() > kmin_cap) return FID_E_CAPACITY;
        for (size_t i = 0; i < k.size(); i++) kmin[i] = k[i];
    }
    if (code_locations) stag_fill_code_locations(code_locations);
    if (min_line_len) *min_line_len = stag_min_line_len(width, height);
    return FID_OK;
}

fid_status fid_stag_load_library(fid_stag_ctx *c, const uint64_t *codewords, int32_t n_codewords)
{
    if (!c || !codewords || n_codewords <= 0 || (n_codewords & 3)) return FID_E_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    if (c->d_words && c->d_words != c->d_words_slab) (void)hipFree(c->d_words);
    c->d_words = nullptr;
    c->n_words = 0;
    if ((size_t)n_codewords * 8 <= STAG_WORDS_RESERVE) c->d_words = c->d_words_slab;  // (inside the slab: a group's launches merge)
    else if (hipMalloc((void **)&c->d_words, (size_t)n_codewords * 8) != hipSuccess) return FID_E_OUT_OF_MEMORY;
    if (hipMemcpy(c->d_words, codewords, (size_t)n_codewords * 8, hipMemcpyHostToDevice) != hipSuccess) return FID_E_HIP;
    c->n_words = n_codewords;
    return FID_OK;
}

fid_status fid_stag_detect_markers_unrefined(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride)
{
    if (!c || !c->d_words) return FID_E_INVALID_ARG;  // no marker library loaded
    return stag_run_to(c, gray, width, height, stride, SS_UNREFINED);
}

fid_status fid_stag_detect_markers(fid_stag_ctx *c, const uint8_t *gray, int32_t width, int32_t height, int32_t stride, fid_stag_marker *out,
                                   int32_t cap, int32_t *n_out)
{
    if (!c || !c->d_words) return FID_E_INVALID_ARG;
    StagJob j;
    j.gray = gray; j.width = width; j.height = height; j.stride = stride; j.last = SS_MARKERS;
    j.out = out; j.cap = cap; j.n_out = n_out;
    return stag_run(c, j);
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
    hipStream_t st = stag_stream(c);
    hipLaunchKernelGGL(k_stag_pose, dim3((c->n_markers + 3) / 4), dim3(64), 0, st, c->d_markers, c->d_nmarkers, cam, marker_size, c->d_poses);
    if (hipGetLastError() != hipSuccess) return FID_E_HIP;
    if (hipMemcpyAsync(out, c->d_poses, (size_t)c->n_markers * sizeof(fid_stag_pose_out), hipMemcpyDeviceToHost, st) != hipSuccess) return FID_E_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? FID_OK : FID_E_HIP;
}

// Frames as a grid dimension (round 3; fid_stag_batch.h).  The contexts are cut into GROUPS of up to STAG_MAXF; a group takes
// that many frames and carries them through the segments of stag_advance in LOCKSTEP on one stream: per segment one wait for the
// group's stream, the host part of every frame (its launches recorded), then every launch site once for the whole group.  One
// host thread goes round the groups, so the wait of one group overlaps the kernels of the others.  A group takes its next
// frames when all of its frames are through (frames that need an extra segment -- a routing fallback -- hold their group for
// that round).  Results are those of frame-by-frame calls: same kernels bodies, same per-frame buffers, no shared state.
static fid_status stag_batch_groups(fid_stag_ctx *const *ctxs, int32_t nctx, const uint8_t *frames, int32_t nframes, int32_t width, int32_t height,
                                    int32_t stride, int64_t frame_stride, const double K[9], const double D[5], double marker_size,
                                    fid_stag_marker *markers, fid_stag_pose_out *poses, int32_t cap_per_frame, int32_t *n_per_frame)
{
    // a group is at most as large as the smallest argument table (the routing kernels carry ~250 bytes of arguments per frame and
    // kernel-argument memory is 4 KB: a group one frame larger would launch them twice)
    constexpr int kGroupMax = std::min({StagTab<k_stag_route_walk_fn>::kMax, StagTab<k_stag_route_extract_fn>::kMax, StagTab<k_stag_route_extract_small_fn>::kMax, StagTab<k_stag_route_extract_big_fn>::kMax, StagTab<k_stag_route_gather_fn>::kMax,
                                        StagTab<k_stag_quads_fn>::kMax, StagTab<k_stag_decode_fn>::kMax, StagTab<k_stag_validate_lines_fn>::kMax,
                                        StagTab<k_stag_split_lines_fn>::kMax, StagTab<k_stag_refine_fn>::kMax, (int)STAG_MAXF});
    // Group size.  More frames per launch make the whole-image passes cheaper per frame (a group of 16 costs 305 us of kernel time a
    // frame, 32: 212, 64: 164 -- profiles/r06_stag_group_sizes.txt) and the latency-bound kernels carry more frames for the same
    // duration while workgroup slots last; more GROUPS keep more kernels in flight, one's long walks under another's image passes.
    // Measured on 256 slots / 1 024 frames: 16 x 16: 6.5 k frames/s, 8 x 32: 7.3 k, 4 x 64: 6.9 - 7.1 k, 2 x 112: 5.6 k; on 128 slots:
    // 8 x 16: 5.5 k, 4 x 32: 6.4 - 6.6 k, 2 x 64: 5.4 - 6.1 k.  So: 32 where the pool has at least two such groups, else two groups.
    int gs = nctx >= 64 ? 32 : (nctx >= 4 ? nctx / 2 : nctx);
    if (const char *e = getenv("FID_STAG_GROUP")) gs = atoi(e);
    gs = gs < 1 ? 1 : (gs > kGroupMax ? kGroupMax : (gs > nctx ? nctx : gs));
    const int ngroups = nctx / gs;
    // a host thread per group (FID_STAG_THREADS caps it; 1 = one thread goes round the groups): a group's host round -- its
    // frames' bookkeeping, 2 MB of staging per new frame -- then runs beside the other groups' rounds and kernels
    int nthreads = ngroups;
    if (const char *e = getenv("FID_STAG_THREADS")) nthreads = atoi(e);
    nthreads = nthreads < 1 ? 1 : (nthreads > ngroups ? ngroups : nthreads);
    // helper threads that stage a group's new frames (FID_STAG_STAGE_THREADS; default: the host's cores shared out among the groups, 1 - 8)
    int stage_threads = (int)std::thread::hardware_concurrency() / (nthreads > 0 ? nthreads : 1);
    if (const char *e = getenv("FID_STAG_STAGE_THREADS")) stage_threads = atoi(e);
    stage_threads = stage_threads < 1 ? 1 : (stage_threads > 8 ? 8 : stage_threads);
    std::atomic<int> next(0);
    std::atomic<bool> stop(false), hip_failed(false);
    std::mutex err_mutex;
    fid_status first_err = FID_OK;
    const bool verbose = getenv("FID_VERBOSE") != nullptr;
    std::atomic<long long> ns_wait(0), ns_host(0), ns_flush(0), n_rec(0), n_iss(0), n_unm(0);
    auto worker = [&](int tid) {
        struct Group {
            std::vector<StagJob> jobs;
            std::vector<int> frame_of;
            int live = 0, g = 0;
            hipStream_t stream = nullptr;
        };
        std::vector<Group> groups;
        for (int g = tid; g < ngroups; g += nthreads) {
            Group G;
            G.g = g;
            G.jobs.resize((size_t)gs);
            G.frame_of.assign((size_t)gs, -1);
            G.stream = stag_stream(ctxs[g * gs]);
            groups.push_back(std::move(G));
        }
        StagRecorder R;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto nsec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
        };
        for (;;) {
            bool any = false;
            for (Group &G : groups) {
                const int g = G.g;
                if (hipSetDevice(ctxs[g * gs]->device) != hipSuccess) {
                    hip_failed = true;
                    return;
                }
                if (G.live == 0) {
                    if (stop.load(std::memory_order_relaxed)) continue;
                    const int f0 = next.fetch_add(gs, std::memory_order_relaxed);  // the group's next frames, all at once
                    if (f0 >= nframes) continue;
                    for (int k = 0; k < gs && f0 + k < nframes; k++) {
                        const int f = f0 + k;
                        StagJob j;
                        j.gray = frames + (size_t)f * frame_stride; j.width = width; j.height = height; j.stride = stride;
                        j.out = markers + (size_t)f * cap_per_frame; j.cap = cap_per_frame; j.n_out = n_per_frame + f;
                        j.last = (K && poses) ? SS_POSE : SS_MARKERS;
                        j.K = K; j.D = D; j.marker_size = marker_size;
                        j.poses = poses ? poses + (size_t)f * cap_per_frame : nullptr; j.pose_cap = cap_per_frame;
                        G.jobs[k] = j;
                        G.frame_of[k] = f;
                        G.live++;
                    }
                    // the frames' rows into the contexts' pinned staging buffers, side by side: 2 MB of memcpy per 1080p frame is
                    // ~0.1 ms, and one thread doing it for a group of 64 kept the group's stream empty for 6 ms of every cycle
                    if (stage_threads > 1 && G.live > 1 && width >= 8 && height >= 8 && width <= ctxs[g * gs]->maxW && height <= ctxs[g * gs]->maxH &&
                        stride >= width) {
                        const int nlive = G.live;
                        auto stage = [&](int t, int T) {
                            for (int k = t; k < nlive; k += T) {
                                fid_stag_ctx *c = ctxs[g * gs + k];
                                const uint8_t *src = G.jobs[k].gray;
                                if ((size_t)stride == (size_t)width) memcpy(c->h_src, src, (size_t)width * height);
                                else
                                    for (int y = 0; y < height; y++) memcpy(c->h_src + (size_t)y * width, src + (size_t)y * stride, (size_t)width);
                                G.jobs[k].staged = true;
                            }
                        };
                        const int T = stage_threads < nlive ? stage_threads : nlive;
                        std::vector<std::thread> helpers;
                        for (int t = 1; t < T; t++) helpers.emplace_back(stage, t, T);
                        stage(0, T);
                        for (auto &h : helpers) h.join();
                    }
                } else {
                    const auto t0 = now();
                    if (hipStreamSynchronize(G.stream) != hipSuccess) {
                        hip_failed = true;
                        return;
                    }
                    if (verbose) ns_wait += nsec(t0, now());
                }
                any = true;
                // ---- one segment round: the host part of every frame, launches recorded; then every site once
                const auto t1 = now();
                R.on = true;
                R.stream = G.stream;
                g_stag_rec = &R;
                for (int k = 0; k < gs; k++) {
                    if (G.frame_of[k] < 0) continue;
                    fid_stag_ctx *c = ctxs[g * gs + k];
                    c->group_stream = G.stream;
                    R.frame_last_site = -1;  // (a new frame's round: its sites must come in increasing order, stag_order_guard)
                    R.slab = c->d_slab; R.slab_bytes = c->slab_bytes;  // (what this frame's pointers are measured against)
                    R.pin = c->pin_alias; R.pin_bytes = sizeof(fid_stag_ctx::Pinned);
                    (void)stag_advance(c, G.jobs[k]);
                }
                R.on = false;
                g_stag_rec = nullptr;
                const auto t2 = now();
                const bool ok = stag_flush(R);
                if (verbose) {
                    ns_host += nsec(t1, t2);
                    ns_flush += nsec(t2, now());
                }
                for (int k = 0; k < gs; k++) {
                    if (G.frame_of[k] < 0 || !G.jobs[k].done) continue;
                    ctxs[g * gs + k]->group_stream = nullptr;
                    if (G.jobs[k].rc != FID_OK) {
                        std::lock_guard<std::mutex> lk(err_mutex);
                        if (first_err == FID_OK) first_err = G.jobs[k].rc;                    // the first failure is the call's status ...
                        if (G.jobs[k].rc != FID_E_CAPACITY) stop.store(true, std::memory_order_relaxed);  // ... a frame that did not fit costs only that frame
                    }
                    G.frame_of[k] = -1;
                    G.live--;
                }
                if (!ok) {
                    hip_failed = true;
                    stop = true;
                }
            }
            if (!any) break;
        }
        n_rec += R.recorded_launches;
        n_iss += R.merged_launches;
        n_unm += R.unmergeable;
    };
    if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(worker, t);
        worker(0);
        for (auto &th : pool) th.join();
    }
    if (hip_failed.load()) {
        for (int g = 0; g < ngroups; g++) (void)hipStreamSynchronize(ctxs[g * gs]->stream);
        for (int t = 0; t < nctx; t++) ctxs[t]->group_stream = nullptr;
        return FID_E_HIP;
    }
    if (verbose)
        fprintf(stderr, "fid stag batch: %d frames, %d group(s) of %d (at most %d), %d host thread(s): %lld launches recorded, %lld issued (%lld times a frame's arguments did not fit the launch open at its site); host ms (all threads): waiting %.2f, frames' host parts %.2f, issuing %.2f\n",
                nframes, ngroups, gs, kGroupMax, nthreads, n_rec.load(), n_iss.load(), n_unm.load(), ns_wait.load() * 1e-6, ns_host.load() * 1e-6, ns_flush.load() * 1e-6);
    return first_err;
}

// Frames over several contexts: every context carries one frame at a time through the segments of stag_advance; a host
// thread goes round ITS contexts, one segment each, so the waits of its contexts overlap.  A context that finishes a frame
// takes the next one off a shared counter.  One thread keeps about a thousand frames a second going (a frame is ~60 launches,
// ~15 copies and a 2 MB staging memcpy: 0.9 ms of host time), so the contexts are dealt out to FID_STAG_THREADS threads
// (default 4, at most one per context).  Results are those of frame-by-frame calls (a frame never sees another frame's data).
fid_status fid_stag_detect_markers_batch(fid_stag_ctx *const *ctxs, int32_t nctx, const uint8_t *frames, int32_t nframes, int32_t width, int32_t height,
                                         int32_t stride, int64_t frame_stride, const double K[9], const double D[5], double marker_size,
                                         fid_stag_marker *markers, fid_stag_pose_out *poses, int32_t cap_per_frame, int32_t *n_per_frame)
{
    if (!ctxs || nctx <= 0 || !frames || nframes < 0 || !markers || !n_per_frame || cap_per_frame <= 0) return FID_E_INVALID_ARG;
    for (int t = 0; t < nctx; t++)
        if (!ctxs[t] || !ctxs[t]->d_words) return FID_E_INVALID_ARG;
    if (K && poses && !(marker_size > 0)) return FID_E_INVALID_ARG;
    for (int f = 0; f < nframes; f++) n_per_frame[f] = 0;  // every count is defined whatever happens to a frame
    // frames as a grid dimension (default); FID_STAG_BATCH=contexts: round 2's road, a stream per context and host threads
    if (!(getenv("FID_STAG_BATCH") && !strcmp(getenv("FID_STAG_BATCH"), "contexts")))
        return stag_batch_groups(ctxs, nctx, frames, nframes, width, height, stride, frame_stride, K, D, marker_size, markers, poses, cap_per_frame,
                                 n_per_frame);
    int nthreads = 4;
    if (const char *e = getenv("FID_STAG_THREADS")) nthreads = atoi(e);
    nthreads = nthreads < 1 ? 1 : (nthreads > nctx ? nctx : nthreads);
    std::atomic<int> next(0);
    std::atomic<bool> stop(false);  // a failure other than "this frame did not fit" ends the call
    std::mutex err_mutex;
    fid_status first_err = FID_OK;
    auto worker = [&](int tid) {
        std::vector<int> mine;  // this thread's contexts
        for (int t = tid; t < nctx; t += nthreads) mine.push_back(t);
        std::vector<StagJob> jobs(mine.size());
        std::vector<int> frame_of(mine.size(), -1);
        int live = 0;
        bool dry = false;
        for (;;) {
            for (size_t k = 0; k < mine.size(); k++) {
                if (frame_of[k] < 0 && !dry && !stop.load(std::memory_order_relaxed)) {  // idle context: next frame
                    const int f = next.fetch_add(1, std::memory_order_relaxed);
                    if (f >= nframes) {
                        dry = true;
                    } else {
                        StagJob j;
                        j.gray = frames + (size_t)f * frame_stride; j.width = width; j.height = height; j.stride = stride;
                        j.out = markers + (size_t)f * cap_per_frame; j.cap = cap_per_frame; j.n_out = n_per_frame + f;
                        j.last = (K && poses) ? SS_POSE : SS_MARKERS;
                        j.K = K; j.D = D; j.marker_size = marker_size;
                        j.poses = poses ? poses + (size_t)f * cap_per_frame : nullptr; j.pose_cap = cap_per_frame;
                        jobs[k] = j;
                        frame_of[k] = f;
                        live++;
                    }
                }
                if (frame_of[k] < 0) continue;
                (void)stag_advance(ctxs[mine[k]], jobs[k]);
                if (jobs[k].done) {
                    if (jobs[k].rc != FID_OK) {
                        std::lock_guard<std::mutex> g(err_mutex);
                        if (first_err == FID_OK) first_err = jobs[k].rc;               // the first failure is the call's status ...
                        if (jobs[k].rc != FID_E_CAPACITY) stop.store(true, std::memory_order_relaxed);  // ... a frame that did not fit costs only that frame
                    }
                    frame_of[k] = -1;
                    live--;
                }
            }
            if (live == 0 && (dry || stop.load(std::memory_order_relaxed))) break;
        }
    };
    if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(worker, t);
        worker(0);
        for (auto &th : pool) th.join();
    }
#ifdef FID_DEBUG_STATS
    fprintf(stderr, "stag batch: %d frames, %d threads; host ms per frame: in sync %.3f; segments (incl. their sync)", nframes, nthreads,
            g_stag_ns_sync.exchange(0) * 1e-6 / (nframes > 0 ? nframes : 1));
    for (int k = 0; k < 12; k++) fprintf(stderr, " %.3f", g_stag_ns_seg[k].exchange(0) * 1e-6 / (nframes > 0 ? nframes : 1));
    fprintf(stderr, "\n");
#endif
    return first_err;
}

fid_status fid_stag_queue_stats(const fid_stag_ctx *c, int32_t *queued, int32_t *rerun)
{
    if (!c) return FID_E_INVALID_ARG;
    if (queued) *queued = c->spec_frames;
    if (rerun) *rerun = c->spec_misses;
    return FID_OK;
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
