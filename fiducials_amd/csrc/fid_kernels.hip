// fid_kernels.hip -- hand-written gfx950 kernels for the aruco detection hot path
// (reference boundary: aruco::detectMarkers / cv::solvePnP as called from
//  /root/reference/aruco_detect/src/aruco_detect.cpp:350 and :247; stage map in SURVEY.md §8a).
//
//   K0 k_to_gray        a2  bgr8/rgb8 -> gray (15-bit fixed point), or stride compaction
//   K1 k_threshold_stream (node window table) / k_threshold (any table)   a3  the 13 adaptive-threshold scales from one
//                       pass over the image, bit-packed tiled masks out; k_threshold_fixed = the round-1 tile kernel (FID_THR=tile)
//   K2 k_find_starts    a4  border-following start points and tracing seeds by bit-parallel tests on mask words
//   K3 k_probe<6>, <32> a4  one lane per start: a few steps of Suzuki-Abe border following sieve the starts
//      k_walk_full<1>, <2>  persistent walkers on private LDS windows: seeds follow their segment, probe survivors walk to the
//                       first seed (<0>: whole borders, FID_TRACE=legacy); points go to pool chunks as they are found
//      k_seg_link / k_seg_chain / k_seg_copy   segment chains -> accepted contours -> dense point arrays
//   K4 k_approx         a4  one wave per accepted contour: approxPolyDP in OpenCV's slice order, quad gates
//   K5 k_sort_cands / k_near / k_resolve   a5  OpenCV order, corner reorder, too-close filter
//   K6 k_identify       a6/a7 one wave per candidate: homography (LU on 64 lanes), unwarp, Otsu, bits, Hamming
//   K7 k_filter_markers / k_subpix        a8/a9
//   K8 k_pose           a11-a13 eight lanes per marker: planar init + Levenberg-Marquardt
//
// Wavefront = 64 everywhere.  Integer stages are bit-exact by construction; floating-point stages
// replay the reference's operation order (this TU is built with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>

#include <type_traits>
#include <utility>

#include "fid_device.h"
#include "../../include/fid_abi.h"

#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ unsigned long long ballot64(int pred) { return __ballot(pred); }

// Wave-wide scans and reductions on DPP row operations (no LDS crossbar: ds_bpermute costs an LDS round trip per step,
// and these sit on the latency path of one-wave-per-object kernels).  All 64 lanes must be active.
#define FID_DPP(OLD, V, CTRL, RMASK) __builtin_amdgcn_update_dpp((OLD), (V), (CTRL), (RMASK), 0xf, false)
// row_shr:1, 2, 4, 8, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
#define FID_DPP_SCAN_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)

// inclusive wave scan
__device__ __forceinline__ int wave_iscan(int v)
{
#define S_(C, M) v += FID_DPP(0, v, C, M);
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    return v;
}

// wave max of a 64-bit key (hi, lo), in every lane
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
#define S_(C, M)                                                                      \
    {                                                                                 \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, M), oh = (unsigned)FID_DPP(0, (int)hi, C, M); \
        const bool gt = oh > hi || (oh == hi && ol > lo);                             \
        lo = gt ? ol : lo;                                                            \
        hi = gt ? oh : hi;                                                            \
    }
    FID_DPP_SCAN_STEPS(S_)  // lane 63 ends up with the max over all lanes (keys are >= 0: the fill value 0 is neutral)
#undef S_
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return ((unsigned long long)hi << 32) | lo;
}

// arg-max with first-index tie-break, in every lane: md = the largest d over the lanes, mi = the smallest idx among the lanes that
// hold it (a lane with nothing to offer passes d = 0, idx = 0xffffffff).  Two 32-bit reductions whose steps are single DPP
// instructions (v_max_u32 / v_min_u32 with a DPP operand): ~ 16 instructions where the 64-bit key (d, ~idx) took ~ 72 -- and
// k_approx does one per farthest-point pass and slice, thirteen per contour.
__device__ __forceinline__ void wave_argmax_first(unsigned d, unsigned idx, unsigned &md, unsigned &mi)
{
    unsigned v = d;
#define S_(C, M)                                                  \
    {                                                             \
        const unsigned o = (unsigned)FID_DPP(0, (int)v, C, M);    \
        v = o > v ? o : v;                                        \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    md = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    unsigned w = d == md ? idx : 0xffffffffu;
#define S_(C, M)                                                  \
    {                                                             \
        const unsigned o = (unsigned)FID_DPP(-1, (int)w, C, M);   \
        w = o < w ? o : w;                                        \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    mi = (unsigned)__builtin_amdgcn_readlane((int)w, 63);
}

// wave max of doubles that are >= 0 in every lane (the fill value 0 of a DPP step is then neutral), in every lane
__device__ __forceinline__ double wave_max_nonneg_f64(double v)
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
#define S_(C, M)                                                                                           \
    {                                                                                                      \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, M), oh = (unsigned)FID_DPP(0, (int)hi, C, M); \
        const bool gt = oh > hi || (oh == hi && ol > lo); /* (the bit patterns of doubles >= 0 order like the values) */ \
        lo = gt ? ol : lo;                                                                                 \
        hi = gt ? oh : hi;                                                                                 \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#define S_(C, M)                                  \
    {                                             \
        const int o = FID_DPP(INT_MAX, v, C, M);  \
        v = o < v ? o : v;                        \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32_dpp(int v) { return wave_min_i32(v); }

__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_iscan(v), 63); }

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)((unsigned long long)v >> 32);
#define S_(C, M)                                                                      \
    {                                                                                 \
        const unsigned ol = (unsigned)FID_DPP(0, (int)lo, C, M), oh = (unsigned)FID_DPP(0, (int)hi, C, M); \
        const unsigned long long t = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)oh << 32) | ol); \
        lo = (unsigned)t;                                                             \
        hi = (unsigned)(t >> 32);                                                     \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// the value of lane `src` in every lane, `src` wave-uniform: two v_readlane (no LDS crossbar round trip)
__device__ __forceinline__ double bcast_f64(double v, int src)
{
    const unsigned long long u = __double_as_longlong(v);
    const int l = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double shfl_f64(double v, int src)
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = __shfl((unsigned)u, src, WAVE);
    unsigned hi = __shfl((unsigned)(u >> 32), src, WAVE);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------------
// K0: cv_bridge::toCvCopy(BGR8) + cvtColor(BGR2GRAY)  (aruco_detect.cpp:348; OpenCV 4.x RGB2Gray<uchar>:
//     (B*3735 + G*19235 + R*9798 + 2^14) >> 15).  Also used to compact a strided mono8 frame.
__global__ __launch_bounds__(256) void k_to_gray(const uint8_t *__restrict__ src, int stride, long long fstride,
                                                  int enc, uint8_t *__restrict__ dst, int W, int H, int F)
{
    long long total = (long long)W * H * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W);
        long long t = i / W;
        int y = (int)(t % H);
        int f = (int)(t / H);
        const uint8_t *s = src + f * fstride + (long long)y * stride;
        uint8_t v;
        if (enc == FID_ENC_MONO8) {
            v = s[x];
        } else {
            const int px = (enc == FID_ENC_BGRA8 || enc == FID_ENC_RGBA8) ? 4 : 3;  // (an alpha channel is dropped)
            int c0 = s[px * x], c1 = s[px * x + 1], c2 = s[px * x + 2];
            const bool bfirst = enc == FID_ENC_BGR8 || enc == FID_ENC_BGRA8;
            int b = bfirst ? c0 : c2, r = bfirst ? c2 : c0;
            v = (uint8_t)((b * 3735 + c1 * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
        dst[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K0 for what raw camera drivers publish (ABI 7, round 6): cv_bridge::toCvCopy(msg, "bgr8") (aruco_detect.cpp:348) followed by
// detectMarkers' BGR2GRAY, in ONE pass over the message's own bytes -- the 8-bit Bayer mosaics, the 16-bit gray / colour layouts
// and UYVY.  The arithmetic is fid_image_to_bgr8's (fid_draw.hip: the host statement of the same rules, now the checker) pixel
// for pixel:
//   * Bayer: OpenCV's bilinear Bayer2RGB_<uchar> under cv_bridge's pattern mapping.  An interior pixel (1 <= y <= H-2,
//     1 <= x <= W-2) keeps its own colour; a green site gets (up + down + 1) >> 1 and (left + right + 1) >> 1 for the two others,
//     a red / blue site (4 edge neighbours + 2) >> 2 for green and (4 diagonals + 2) >> 2 for the opposite colour; which of
//     B / R the horizontal (own) value is alternates row by row (`blue`), green sites alternate along a row starting with
//     `green` in column 1.  Border columns, then border rows repeat their neighbours: pixel (y, x) is pixel (clamp(y, 1, H-2),
//     clamp(x, 1, W-2)).
//   * 16 bit: cvRound(float(v) * float(255. / 65535.)) per sample (nearest, ties to even), big-endian samples swapped first.
//   * UYVY: BT.601 in 20-bit fixed point.
// Grid (ceil(W / 256), ceil(H / 4), F), block (64, 4): a wave per row, a lane per four pixels (one 32-bit store where the row
// start allows).  HBM-bound: reads the message once (1 - 8 B per pixel), writes 1 B per pixel.
__device__ __forceinline__ int k0_gray(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }
__device__ __forceinline__ int k0_scale16(const uint8_t *p, bool be)
{
    const unsigned v = be ? ((unsigned)p[0] << 8) | p[1] : (unsigned)p[0] | ((unsigned)p[1] << 8);
    const float a = (float)(255. / 65535.);
    const int r = __float2int_rn((float)v * a);
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}
__device__ __forceinline__ int k0_sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// one demosaiced pixel from its 3 x 3 neighbourhood (row-major n[9]); hv: +1 / -1 (the host code's `blue` for this row)
__device__ __forceinline__ int k0_bayer_gray(const int n[9], bool is_green, int hv)
{
    int own, other, green;
    if (is_green) {
        green = n[4];
        other = (n[1] + n[7] + 1) >> 1;  // vertical pair   -> channel 1 - hv
        own = (n[3] + n[5] + 1) >> 1;    // horizontal pair -> channel 1 + hv
    } else {
        own = n[4];
        green = (n[1] + n[3] + n[5] + n[7] + 2) >> 2;
        other = (n[0] + n[2] + n[6] + n[8] + 2) >> 2;
    }
    const int b = hv > 0 ? other : own, r = hv > 0 ? own : other;  // channel 0 = B, 2 = R
    return k0_gray(b, green, r);
}
__global__ __launch_bounds__(256) void k_raw_to_gray(const uint8_t *__restrict__ src, int stride, long long fstride, int enc,
                                                      uint8_t *__restrict__ dst, int W, int H)
{
    const int x0 = ((int)blockIdx.x * 64 + (int)threadIdx.x) * 4, y = (int)blockIdx.y * 4 + (int)threadIdx.y, f = blockIdx.z;
    if (x0 >= W || y >= H) return;
    const uint8_t *img = src + (long long)f * fstride;
    uint8_t *o = dst + ((long long)f * H + y) * W + x0;
    const int n = W - x0 < 4 ? W - x0 : 4;
    const int base = enc & 0xff;
    const bool be = (enc & FID_ENC_BIGENDIAN) != 0;
    int g[4] = {0, 0, 0, 0};
    if (base >= FID_ENC_BAYER_RGGB8 && base <= FID_ENC_BAYER_GRBG8) {
        // (rggb: blue -1, green 0; bggr: +1, 0; gbrg: +1, 1; grbg: -1, 1 -- fid_image_to_bgr8's table)
        const int blue0 = (base == FID_ENC_BAYER_RGGB8 || base == FID_ENC_BAYER_GRBG8) ? -1 : 1;
        const int green0 = (base == FID_ENC_BAYER_GBRG8 || base == FID_ENC_BAYER_GRBG8) ? 1 : 0;
        const int yy = y < 1 ? 1 : (y > H - 2 ? H - 2 : y);
        const int hv = ((yy - 1) & 1) ? -blue0 : blue0;
        const int gs = green0 ^ ((yy - 1) & 1);
        const uint8_t *r0 = img + (long long)(yy - 1) * stride, *r1 = r0 + stride, *r2 = r1 + stride;
        if (x0 >= 1 && x0 + 4 <= W - 1) {
            // the six columns x0 - 1 ... x0 + 4 of three rows serve four interior pixels
            int a0[6], a1[6], a2[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                a0[k] = r0[x0 - 1 + k];
                a1[k] = r1[x0 - 1 + k];
                a2[k] = r2[x0 - 1 + k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int nb[9] = {a0[k], a0[k + 1], a0[k + 2], a1[k], a1[k + 1], a1[k + 2], a2[k], a2[k + 1], a2[k + 2]};
                g[k] = k0_bayer_gray(nb, (gs ^ ((x0 + k - 1) & 1)) != 0, hv);
            }
        } else {
            for (int k = 0; k < n; k++) {  // a row's first and last lanes: the clamped column, nine loads a pixel
                const int x = x0 + k, xx = x < 1 ? 1 : (x > W - 2 ? W - 2 : x);
                const int nb[9] = {r0[xx - 1], r0[xx], r0[xx + 1], r1[xx - 1], r1[xx], r1[xx + 1], r2[xx - 1], r2[xx], r2[xx + 1]};
                g[k] = k0_bayer_gray(nb, (gs ^ ((xx - 1) & 1)) != 0, hv);
            }
        }
    } else if (base == FID_ENC_MONO16) {
        const uint8_t *s = img + (long long)y * stride + 2 * x0;
        for (int k = 0; k < n; k++) g[k] = k0_scale16(s + 2 * k, be);  // (B = G = R = v: BGR2GRAY gives v back)
    } else if (base >= FID_ENC_BGR16 && base <= FID_ENC_RGBA16) {
        const int ch = (base == FID_ENC_BGRA16 || base == FID_ENC_RGBA16) ? 4 : 3;
        const bool bfirst = base == FID_ENC_BGR16 || base == FID_ENC_BGRA16;
        const uint8_t *s = img + (long long)y * stride + 2 * ch * x0;
        for (int k = 0; k < n; k++) {
            const int c0 = k0_scale16(s + 2 * (ch * k), be), c1 = k0_scale16(s + 2 * (ch * k + 1), be), c2 = k0_scale16(s + 2 * (ch * k + 2), be);
            g[k] = k0_gray(bfirst ? c0 : c2, c1, bfirst ? c2 : c0);
        }
    } else {  // FID_ENC_YUV422 (UYVY; W is even, so a lane's pixels are whole pairs)
        const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SH = 20;
        const uint8_t *s = img + (long long)y * stride + 2 * x0;
        for (int k = 0; k < n; k += 2) {
            const int u = s[2 * k] - 128, y0 = s[2 * k + 1], v = s[2 * k + 2] - 128, y1 = s[2 * k + 3];
            const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
            const int p0 = (y0 - 16 > 0 ? y0 - 16 : 0) * CY, p1 = (y1 - 16 > 0 ? y1 - 16 : 0) * CY;
            g[k] = k0_gray(k0_sat8((p0 + buv) >> SH), k0_sat8((p0 + guv) >> SH), k0_sat8((p0 + ruv) >> SH));
            g[k + 1] = k0_gray(k0_sat8((p1 + buv) >> SH), k0_sat8((p1 + guv) >> SH), k0_sat8((p1 + ruv) >> SH));
        }
    }
    if (n == 4 && (((unsigned long long)o) & 3ull) == 0) {
        *reinterpret_cast<unsigned *>(o) = (unsigned)g[0] | ((unsigned)g[1] << 8) | ((unsigned)g[2] << 16) | ((unsigned)g[3] << 24);
    } else {
        for (int k = 0; k < n; k++) o[k] = (uint8_t)g[k];
    }
}

// ------------------------------------------------------------------------------------------------
// K1: multi-scale adaptive threshold.
//   adaptiveThreshold(MEAN_C, BINARY_INV, win, C): mean = round(boxsum / win^2) with BORDER_REPLICATE,
//   foreground iff src - mean <= -idelta, idelta = cvFloor(C) (THRESH_BINARY_INV)   <=>   2*boxsum >= (2*(src + idelta) - 1) * win^2
//   (exact: boxsum/win^2 never sits on a half for odd win).
// One workgroup = one TX x TY output tile: the tile plus a rmax halo is loaded once (clamped =
// replicate border), turned into a 2-D integral image in LDS, and all scales read their four corners
// from it.  A wave covers 64 consecutive x so one ballot yields two packed mask words per (row, scale).
template <int TX, int TY, int NT>
__global__ __launch_bounds__(NT) void k_threshold(const uint8_t *__restrict__ gray, long long gfstride,
                                                   uint32_t *__restrict__ masks, const DevParams P)
{
    extern __shared__ uint32_t I[];
    const int R = P.rmax;
    const int RW = TX + 2 * R, RH = TY + 2 * R;
    const int PT = (RW + 1) | 1;  // odd pitch: column walks are bank-conflict free
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NWAVES = NT / 64;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, f = blockIdx.z;
    const uint8_t *g = gray + (long long)f * gfstride;
    const int W = P.W, H = P.H, gs = P.gstride;

    for (int i = tid; i < PT; i += NT) I[i] = 0;
    for (int i = tid; i <= RH; i += NT) I[i * PT] = 0;
    // phase 1+2: load rows (clamped) and row-prefix them with a wave scan
    for (int ry = wid; ry < RH; ry += NWAVES) {
        int gy = y0 - R + ry;
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const uint8_t *grow = g + (long long)gy * gs;
        int carry = 0;
        for (int c = 0; c < RW; c += 64) {
            int rx = c + lane;
            int gx = x0 - R + rx;
            gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            int v = rx < RW ? (int)grow[gx] : 0;
            v = wave_iscan(v) + carry;
            if (rx < RW) I[(ry + 1) * PT + rx + 1] = (uint32_t)v;
            carry = __shfl(v, 63, WAVE);
        }
    }
    __syncthreads();
    // phase 3: column prefix, one thread per column
    for (int cx = 1 + tid; cx <= RW; cx += NT) {
        uint32_t acc = 0;
#pragma unroll 4
        for (int ry = 1; ry <= RH; ry++) {
            acc += I[ry * PT + cx];
            I[ry * PT + cx] = acc;
        }
    }
    __syncthreads();
    // phase 4: evaluate all scales
    constexpr int SEGS = TX / 64;
    const int S = P.nscales, TC = P.TC;
    const long long plane = (long long)P.TR * TC * MT_ROWS;
    for (int item = wid; item < TY * SEGS; item += NWAVES) {
        int ty = item / SEGS, seg = item % SEGS;
        int x = seg * 64 + lane;
        int gx = x0 + x, gy = y0 + ty;
        if (gy >= H) continue;  // wave-uniform
        int valid = gx < W;
        int gv = valid ? (int)g[(long long)gy * gs + gx] : 0;
        int t2 = 2 * (gv + P.idelta) - 1;
        int rc = ty + R, cc = x + R;
        uint32_t *mrow = masks + (long long)f * S * plane + mask_word(TC, gy + 1, MASK_PADW + ((x0 + seg * 64) >> 5));
        for (int s = 0; s < S; s++) {
            int win = P.win[s], r = win >> 1;
            const uint32_t *top = I + (rc - r) * PT, *bot = I + (rc + r + 1) * PT;
            int sum = (int)(bot[cc + r + 1] - top[cc + r + 1] - bot[cc - r] + top[cc - r]);
            int fg = valid && (2 * sum >= t2 * win * win);
            unsigned long long b = ballot64(fg);
            if (lane == 0) {
                mrow[(long long)s * plane] = (uint32_t)b;
                mrow[(long long)s * plane + MT_ROWS] = (uint32_t)(b >> 32);  // next tile column
            }
        }
    }
}

// v_writelane_b32: park a wave-uniform value in lane K of a VGPR (one VALU op, no exec juggling)
template <int K>
__device__ __forceinline__ uint32_t write_lane(uint32_t val, uint32_t old)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(K));
    return old;
}

template <int... Ks, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Ks...>, F &&f)
{
    (f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// ------------------------------------------------------------------------------------------------
// K1 (fast path): the same arithmetic for the window table WMIN, WMIN + WSTEP, ... (NS windows) fixed at
// compile time -- the node-default table 3, 7, ..., 51 (aruco_detect.cpp:690-693) -- so that every integral
// corner is one ds_read_b32 with an immediate offset.
//   tile          TX x TY = 128 x 120 output pixels per 1024-thread workgroup (16 waves, one workgroup per CU:
//                 the (TY + 2R + 1) x PT u32 integral takes 128.6 KB of the 160 KB LDS); halo re-read 1.97x
//   phase 1       wave per raw row: one dword (4 px) per lane, 4-px local prefix + wave scan -> row prefix in LDS
//   phase 2       column prefix in two levels (5 row chunks x 184 columns, 34 values in registers per thread)
//   phase 3       wave per (row, 64-px segment): per scale 4 LDS reads, 2*sum >= t2*win^2, v_cmp = the mask
//                 word pair; ballots are parked in lane k of the accumulators; each lane then stores its row's
//                 four words per scale into the tiled mask layout
template <int WMIN, int WSTEP, int NS>
struct ThrCfg {
    static constexpr int TX = 128, TY = 120, NT = 1024, NW = NT / 64;
    static constexpr int R = (WMIN + (NS - 1) * WSTEP) / 2;  // largest radius
    static constexpr int HL = (R + 3) & ~3;                  // left halo, dword aligned
    static constexpr int RAWW = (HL + TX + R + 3) & ~3;      // raw columns loaded per row
    static constexpr int NDW = RAWW / 4;                     // dwords per raw row (<= 64)
    static constexpr int RAWH = TY + 2 * R;                  // raw rows
    static constexpr int PT = RAWW + 4;                      // integral pitch (column 0 = zero column)
    static constexpr int NCH = 5;                            // row chunks of the column prefix
    static constexpr int CH = (RAWH + NCH - 1) / NCH;
    static constexpr int ROWS_PER_WAVE = (RAWH + NW - 1) / NW;
    static constexpr int ITEMS = (TY + NW - 1) / NW;         // output rows per wave
    static constexpr size_t LDS_BYTES = (size_t)((RAWH + 1) * PT + NCH * RAWW) * sizeof(uint32_t);
    static_assert(NDW <= 64, "one dword per lane");
    static_assert(NCH * RAWW <= NT, "column-prefix threads");
};

template <int WMIN, int WSTEP, int NS>
__global__ __launch_bounds__(1024) void k_threshold_fixed(const uint8_t *__restrict__ gray, long long gfstride,
                                                           uint32_t *__restrict__ masks, const DevParams P)
{
    using C = ThrCfg<WMIN, WSTEP, NS>;
    constexpr int TX = C::TX, TY = C::TY, NW = C::NW, R = C::R, HL = C::HL, RAWW = C::RAWW, NDW = C::NDW, RAWH = C::RAWH,
                  PT = C::PT, NCH = C::NCH, CH = C::CH;
    extern __shared__ uint32_t I[];       // (RAWH + 1) x PT, I[j][i] = sum of raw rows < j, raw cols < i
    uint32_t *tot = I + (RAWH + 1) * PT;  // NCH x RAWW chunk totals
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, f = blockIdx.z;
    const uint8_t *g = gray + (long long)f * gfstride;
    const int W = P.W, H = P.H, gs = P.gstride;

    // this wave's output rows: ty = wid + NW * k; their centre pixels are fetched now, used in phase 3
    uint32_t gpix[C::ITEMS][2];
#pragma unroll
    for (int k = 0; k < C::ITEMS; k++) {
        int gy = y0 + wid + NW * k;
        gy = gy < H ? gy : H - 1;
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            int gx = x0 + sg * 64 + lane;
            gx = gx < W ? gx : W - 1;
            gpix[k][sg] = g[(long long)gy * gs + gx];
        }
    }
    // zero row / zero column
    for (int i = tid; i < PT; i += C::NT) I[i] = 0;
    for (int i = tid; i <= RAWH; i += C::NT) I[i * PT] = 0;
    // ---- phase 1: row prefix
    {
        const bool fast = ((((uintptr_t)g) | (unsigned)gs) & 3) == 0 && x0 - HL >= 0 && x0 - HL + RAWW <= W;
        uint32_t raw[C::ROWS_PER_WAVE];
#pragma unroll
        for (int k = 0; k < C::ROWS_PER_WAVE; k++) {
            int ry = wid + NW * k;
            int gy = y0 - R + ry;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
            const uint8_t *grow = g + (long long)gy * gs;
            uint32_t v = 0;
            if (ry < RAWH && lane < NDW) {
                int gx = x0 - HL + 4 * lane;
                if (fast) {
                    v = *reinterpret_cast<const uint32_t *>(grow + gx);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        int xx = gx + b;
                        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                        v |= (uint32_t)grow[xx] << (8 * b);
                    }
                }
            }
            raw[k] = v;
        }
#pragma unroll
        for (int k = 0; k < C::ROWS_PER_WAVE; k++) {
            int ry = wid + NW * k;
            if (ry >= RAWH) break;  // wave-uniform
            uint32_t v = raw[k];
            int p0 = v & 0xff, p1 = p0 + ((v >> 8) & 0xff), p2 = p1 + ((v >> 16) & 0xff), p3 = p2 + (v >> 24);
            int e = wave_iscan(p3) - p3;
            if (lane < NDW) {
                uint32_t *dst = I + (ry + 1) * PT + 1 + 4 * lane;
                dst[0] = (uint32_t)(e + p0);
                dst[1] = (uint32_t)(e + p1);
                dst[2] = (uint32_t)(e + p2);
                dst[3] = (uint32_t)(e + p3);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: column prefix, chunk j of CH rows x column c per thread
    {
        const int c = tid % RAWW, j = tid / RAWW;
        const bool act = tid < NCH * RAWW;
        uint32_t v[CH];
        uint32_t *col = I + (1 + j * CH) * PT + 1 + c;
        if (act) {
#pragma unroll
            for (int k = 0; k < CH; k++) v[k] = (j * CH + k < RAWH) ? col[k * PT] : 0u;
#pragma unroll
            for (int k = 1; k < CH; k++) v[k] += v[k - 1];
            tot[j * RAWW + c] = v[CH - 1];
        }
        __syncthreads();
        if (act) {
            uint32_t off = 0;
#pragma unroll
            for (int q = 0; q < NCH - 1; q++) off += q < j ? tot[q * RAWW + c] : 0u;
#pragma unroll
            for (int k = 0; k < CH; k++)
                if (j * CH + k < RAWH) col[k * PT] = v[k] + off;
        }
    }
    __syncthreads();
    // ---- phase 3: all scales for this wave's rows
    uint4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) acc[s] = make_uint4(0u, 0u, 0u, 0u);
    static_for<C::ITEMS>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int ty = wid + NW * k;
        if (ty < TY) {  // wave-uniform
#pragma unroll
            for (int sg = 0; sg < 2; sg++) {
                const int x = sg * 64 + lane;
                const int t2 = 2 * ((int)gpix[k][sg] + P.idelta) - 1;
                const unsigned long long vmask = ballot64(x0 + x < W);
                // base = I[ry - R][cx - R] with ry = ty + R, cx = x + HL (raw coordinates of the pixel)
                const uint32_t *base = I + ty * PT + (x + HL - R);
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const int win = WMIN + s * WSTEP, r = win >> 1;
                    const int o_tl = (R - r) * PT + (R - r), o_tr = (R - r) * PT + (R + r + 1);
                    const int o_bl = (R + r + 1) * PT + (R - r), o_br = (R + r + 1) * PT + (R + r + 1);
                    int sum = (int)(base[o_br] - base[o_tr] - base[o_bl] + base[o_tl]);
                    unsigned long long b = ballot64(2 * sum >= __mul24(t2, win * win)) & vmask;
                    uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
                    if (sg == 0) {
                        acc[s].x = write_lane<k>(lo, acc[s].x);
                        acc[s].y = write_lane<k>(hi, acc[s].y);
                    } else {
                        acc[s].z = write_lane<k>(lo, acc[s].z);
                        acc[s].w = write_lane<k>(hi, acc[s].w);
                    }
                }
            }
        }
    });
    // lane k holds row wid + NW * k: one 16-byte store per (row, scale)
    {
        const int ty = wid + NW * lane;
        const int gy = y0 + ty;
        if (lane < C::ITEMS && ty < TY && gy < H) {
            const long long plane = (long long)P.TR * P.TC * MT_ROWS;
            uint32_t *mrow = masks + (long long)f * NS * plane + mask_word(P.TC, gy + 1, MASK_PADW + (x0 >> 5));
#pragma unroll
            for (int s = 0; s < NS; s++) {
                uint32_t *q = mrow + (long long)s * plane;  // the four words sit in four neighbouring tiles
                q[0] = acc[s].x;
                q[MT_ROWS] = acc[s].y;
                q[2 * MT_ROWS] = acc[s].z;
                q[3 * MT_ROWS] = acc[s].w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1 (stream path, the default for the node's window table): the same arithmetic with a small LDS footprint and a
// quarter of the LDS reads, so that it shares a CU with the latency-bound contour kernels of the other sub-batch.
//
//   workgroup      NW consumer waves + 1 producer wave walk down a strip of 64 * NW columns, 4 rows (a QUAD) per step
//   LDS ring       16 quads x RAWW columns x 8 bytes: for every raw column the INCLUSIVE ROW PREFIX of its 4 rows as
//                  4 x u16 (mod 2^16: only differences over <= 51 columns are ever taken, and those fit 14 bits).
//                  ring row = image row + 28, so ring quads are image-row quads; 47.6 KB for NW = 5 (3 workgroups per CU)
//   producer wave  per step one quad: byte loads (clamped = BORDER_REPLICATE), two rows per register (lo / hi 16 bits),
//                  one DPP wave scan per row pair and 64-column chunk (a chunk's fields cannot overflow: 64 * 255),
//                  chunk carry added with v_pk_add_u16, one ds_write_b64 per column; loads run one step ahead
//   consumer lane  = one column.  Per scale it keeps the running box sum V = sum over rows y-r..y+r of the horizontal
//                  window sum H(row) = RI[x + r] - RI[x - r - 1]:  V(y) = V(y - 1) + H(y + r) - H(y - r - 1).
//                  Per step and scale: 4 ds_read_b64 (left / right prefix of the quad that enters and of the quad that
//                  leaves), packed 16-bit subtractions for four rows at a time, v_alignbit to line the two quads up with
//                  the output quad (r is odd: exactly one of the two is off by an odd number of rows; the other is
//                  aligned or off by two rows = register renaming), four v_dot2c accumulations, four v_cmp whose SGPR
//                  result IS the mask word pair of 64 columns.  V is kept minus the per-scale constant of the test, so
//                  the test is V >= gray * win^2 (one v_mul_i32_i24 with a literal).
//   output         the word pairs are parked in lane (row & 63) of 26 registers (v_writelane) and stored every 64 rows,
//                  a row per lane, into the tiled mask layout
// One s_barrier per step orders ring slot reuse: in step k the producer fills quad k + 8 (slot (k + 8) & 15) while the
// consumers read quads k - 7 .. k + 7.
template <int WMIN, int WSTEP, int NS, int NW>
struct ThrStream {
    static constexpr int R = (WMIN + (NS - 1) * WSTEP) / 2;
    static constexpr int COLS = 64 * NW;
    static constexpr int PADL = R + 1;             // raw columns left of the strip (RI[x - r - 1] with r = R)
    static constexpr int RAWW = COLS + 2 * R + 1;  // raw columns: x - R - 1 .. x + R
    static constexpr int NCHUNK = (RAWW + 63) / 64;
    static constexpr int DEPTH = 16;               // ring depth in quads
    static constexpr int ROWPAD = 28;              // ring row = image row + ROWPAD
    static constexpr int LEAD = 7;                 // a step touches quads qo - LEAD .. qo + LEAD
    static constexpr int NT = 64 * (NW + 1);
    static constexpr size_t LDS_BYTES = (size_t)DEPTH * RAWW * 8;
    static_assert((WMIN & 1) == 1 && (WSTEP % 2) == 0, "odd windows");
    static_assert(R + 1 <= ROWPAD - 2 && (R + 3) / 4 <= LEAD && R <= 25, "ring geometry is for radii <= 25");
};

typedef unsigned short thr_u16x2 __attribute__((ext_vector_type(2)));
typedef short thr_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (thr_u16x2)(__builtin_bit_cast(thr_u16x2, a) - __builtin_bit_cast(thr_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (thr_u16x2)(__builtin_bit_cast(thr_u16x2, a) + __builtin_bit_cast(thr_u16x2, b)));
}
// inclusive scan over the 64 lanes, DPP only (no LDS crossbar)
__device__ __forceinline__ uint32_t wave_iscan_dpp(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// park a wave-uniform word pair in lane `sel` (wave-uniform, run-time) of two registers
__device__ __forceinline__ void park_pair(uint32_t &a0, uint32_t &a1, unsigned long long b, int sel)
{
    const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    // (M0 written by the SALU and read as a lane select by the next VALU instruction: no wait state is required for that pair)
    // not volatile: the statement is a pure function of its operands, so the scheduler may move LDS reads across it
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0"
                 : "+v"(a0), "+v"(a1)
                 : "s"(sel), "s"(lo), "s"(hi));
}

// The same parking for the four rows of a step through the EXEC mask instead of v_writelane: tools/valu_calib.hip measured
// v_writelane_b32 in the 4-cycle issue class (590 G wave-instructions/s on the chip, like most integer VALU operations) and a plain
// v_mov_b32 in the 2-cycle class (1000 G/s), and the lane select can ride on the scalar pipe, which this kernel leaves mostly
// idle: EXEC = 1 << sel, two v_mov per row (they write the one enabled lane), EXEC <<= 1 for the next row, EXEC restored at the
// end.  Eight 2-cycle VALU instead of eight 4-cycle ones per scale and step (104 of the 366 VALU instructions of a step).
// (SALU write of EXEC followed by a VALU needs no wait state on gfx9.)  -DFID_PARK_WRITELANE keeps the v_writelane form.
__device__ __forceinline__ void park_quad(uint32_t &a0, uint32_t &a1, unsigned long long b0, unsigned long long b1, unsigned long long b2,
                                          unsigned long long b3, int sel)
{
#ifdef FID_PARK_WRITELANE
    park_pair(a0, a1, b0, sel);
    park_pair(a0, a1, b1, sel + 1);
    park_pair(a0, a1, b2, sel + 2);
    park_pair(a0, a1, b3, sel + 3);
#else
    unsigned long long keep;
    asm("s_mov_b64 %2, exec\n\t"
        "s_lshl_b64 exec, 1, %3\n\t"
        "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b32 %0, %10\n\tv_mov_b32 %1, %11\n\t"
        "s_mov_b64 exec, %2"
        : "+v"(a0), "+v"(a1), "=&s"(keep)
        : "s"(sel), "s"((uint32_t)b0), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)b1), "s"((uint32_t)(b1 >> 32)), "s"((uint32_t)b2),
          "s"((uint32_t)(b2 >> 32)), "s"((uint32_t)b3), "s"((uint32_t)(b3 >> 32))
        : "scc");
#endif
}

// Round 5, the default: one v_mov_b64 per row (gfx90a+: a 64-bit move of an SGPR pair into a VGPR pair) instead of two v_mov_b32 --
// 52 instead of 104 parking instructions of a step's 366 VALU, the mask words of a scale in ONE register pair.  Measured on the
// bench batch (tools/gpu_sweep.sh, three rounds): the kernel alone 1.73 -> 1.63 ms per 128 frames, beside the other batch's kernels
// 3.9 - 4.0 -> 3.6 - 3.8 ms, whole pipeline +0.7 % (two contexts), +1 % (one); masks == in every parity test.  -DFID_PARK_MOV32
// keeps park_quad above.
__device__ __forceinline__ void park_quad64(unsigned long long &a, unsigned long long b0, unsigned long long b1, unsigned long long b2,
                                            unsigned long long b3, int sel)
{
    unsigned long long keep;
    asm("s_mov_b64 %1, exec\n\t"
        "s_lshl_b64 exec, 1, %2\n\t"
        "v_mov_b64 %0, %3\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b64 %0, %4\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b64 %0, %5\n\t"
        "s_lshl_b64 exec, exec, 1\n\t"
        "v_mov_b64 %0, %6\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(a), "=&s"(keep)
        : "s"(sel), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
        : "scc");
}

template <int WMIN, int WSTEP, int NS, int NW, bool SPLIT>
__global__ __launch_bounds__(64 * (NW + 1)) void k_threshold_stream(const uint8_t *__restrict__ gray, long long gfstride,
                                                                      uint32_t *__restrict__ masks, const DevParams P, int RS, int xcd_map)
{
    using C = ThrStream<WMIN, WSTEP, NS, NW>;
    constexpr int R = C::R, PADL = C::PADL, RAWW = C::RAWW, NCHUNK = C::NCHUNK, ROWPAD = C::ROWPAD, LEAD = C::LEAD;
    extern __shared__ __attribute__((aligned(16))) uint2 thr_ring[];  // [DEPTH][RAWW]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Which strip this workgroup takes.  The dispatcher deals workgroups out to the eight XCDs in turn (linear index mod 8), so
    // with the plain (x, y, frame) order the two neighbours of a strip -- whose 25 / 26 halo columns it reads, and with which it
    // shares every 128-byte line that straddles a strip border -- always sit on OTHER XCDs: each XCD's L2 fetched those lines
    // from HBM for itself (round 4: 4.28 MB fetched per 2.07 MB frame).  xcd_map: the k-th workgroup of XCD j takes strip
    // base(j) + k, so that an XCD works through a contiguous run of strips and the shared lines come out of its own L2.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_map) {
        const unsigned gx = gridDim.x, gy = gridDim.y, N = gx * gy * gridDim.z;
        const unsigned L = bx + gx * (by + gy * bz), j = L & 7u, k = L >> 3, q = N >> 3, r = N & 7u;
        const unsigned T = j * q + (j < r ? j : r) + k;
        bx = T % gx;
        const unsigned t2 = T / gx;
        by = t2 % gy;
        bz = t2 / gy;
    }
    const int xs = (int)bx * C::COLS, ys = (int)by * RS, f = (int)bz;
    const int W = P.W, H = P.H, gs = P.gstride;
    const uint8_t *g = gray + (long long)f * gfstride;
    const int yend = ys + RS < H ? ys + RS : H;
    const int nsteps = (yend - ys + 3) >> 2;
    const int Q0 = (ys >> 2) + ROWPAD / 4;  // ring quad of the first output quad (ys is a multiple of 4)

    // ---- producer side: one quad of raw rows -> packed row prefixes in the ring
    struct RawQuad {
        uint32_t v01[NCHUNK], v23[NCHUNK];
    };
    auto load_quad = [&](int Q, RawQuad &q) {
        const int yb = 4 * Q - ROWPAD;
        const uint8_t *rp[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int y = yb + i;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y);
            rp[i] = g + (long long)y * gs;
        }
#pragma unroll
        for (int c = 0; c < NCHUNK; c++) {
            int x = xs - PADL + 64 * c + lane;
            x = x < 0 ? 0 : (x >= W ? W - 1 : x);
            q.v01[c] = (uint32_t)rp[0][x] | ((uint32_t)rp[1][x] << 16);
            q.v23[c] = (uint32_t)rp[2][x] | ((uint32_t)rp[3][x] << 16);
        }
    };
    auto store_quad = [&](int Q, const RawQuad &q) {
        uint2 *dst = thr_ring + (Q & (C::DEPTH - 1)) * RAWW;
        uint32_t c01 = 0, c23 = 0;
#pragma unroll
        for (int c = 0; c < NCHUNK; c++) {
            const uint32_t s01 = wave_iscan_dpp(q.v01[c]), s23 = wave_iscan_dpp(q.v23[c]);
            const int j = 64 * c + lane;
            if (j < RAWW) dst[j] = make_uint2(pk_add16(s01, c01), pk_add16(s23, c23));
            c01 = pk_add16(c01, (uint32_t)__builtin_amdgcn_readlane((int)s01, 63));
            c23 = pk_add16(c23, (uint32_t)__builtin_amdgcn_readlane((int)s23, 63));
        }
    };
    // ---- prologue: every wave fills whole quads (a quad needs nothing from another quad)
    for (int q = Q0 - LEAD + wid; q <= Q0 + LEAD; q += NW + 1) {
        RawQuad rq;
        load_quad(q, rq);
        store_quad(q, rq);
    }
    __syncthreads();

    if (wid == NW) {
        // ---- the producer wave: quad Q0 + k + 8 during step k, its loads one step ahead
        const int last = Q0 + nsteps - 1 + LEAD;  // last quad any step reads
        RawQuad cur, nxt;
        if (Q0 + LEAD + 1 <= last) load_quad(Q0 + LEAD + 1, cur);
        for (int k = 0; k < nsteps; k++) {
            const int Q = Q0 + k + LEAD + 1;
            if (Q + 1 <= last) load_quad(Q + 1, nxt);
            if (Q <= last) store_quad(Q, cur);
            __syncthreads();
            cur = nxt;
        }
        return;
    }

    // ---- consumer waves
    const int colw = 64 * wid;                 // first column of this wave inside the strip
    const bool wave_on = xs + colw < W;        // a wave whose columns lie right of the image only keeps the barriers company
    const unsigned long long vmask = ballot64(xs + colw + lane < W);
    const uint32_t jb = (uint32_t)(colw + lane);  // raw column of RI[x - R - 1]: j = jb + (PADL - r - 1) / jb + PADL + r
    int zoff = 0;
    if (SPLIT) asm volatile("" : "+s"(zoff));
    auto slot_base = [&](int Q) { return thr_ring + (Q & (C::DEPTH - 1)) * RAWW + jb; };
    auto hquad = [&](int Q, int r) {  // horizontal window sums (window 2r + 1) of the four rows of aligned quad Q
        // the left prefix is addressed through a base the compiler cannot relate to the right one (opaque zero): two
        // ds_read_b64 (2 LDS cycles each) instead of one fused ds_read2_b64 (8 cycles, MI355X_MICROARCH.md LDS table)
        const uint2 *b = slot_base(Q);
        const uint2 hi = b[PADL + r], lo = (b + zoff)[PADL - r - 1];
        return make_uint2(pk_sub16(hi.x, lo.x), pk_sub16(hi.y, lo.y));
    };
    int V[NS];
    uint32_t ca[NS], cb[NS], cc[NS];  // carried halves of the previous aligned quads (see the step)
#ifndef FID_PARK_MOV32
    unsigned long long acc[NS];
#else
    uint32_t acc0[NS], acc1[NS];
#endif
    if (wave_on) {
        static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int r = (WMIN + s * WSTEP) / 2, w2 = (2 * r + 1) * (2 * r + 1);
            // box sum of row ys - 1: ring rows ys - 1 - r + ROWPAD .. ys - 1 + r + ROWPAD
            const uint16_t *ring16 = reinterpret_cast<const uint16_t *>(thr_ring);
            uint32_t sum = 0;
            for (int rho = ys - 1 - r + ROWPAD; rho <= ys - 1 + r + ROWPAD; rho++) {
                const int e = (((rho >> 2) & (C::DEPTH - 1)) * RAWW + (int)jb) * 4 + (rho & 3);
                sum += (uint16_t)(ring16[e + (PADL + r) * 4] - ring16[e + (PADL - r - 1) * 4]);
            }
            // the test  box >= (gray + idelta) * w2 - (w2 - 1) / 2  is kept as  V >= gray * w2
            V[s] = (int)sum - (P.idelta * w2 - (w2 - 1) / 2);
#ifndef FID_PARK_MOV32
            acc[s] = 0ull;
#else
            acc0[s] = acc1[s] = 0;
#endif
            if constexpr ((r & 3) == 1) {
                const uint2 e = hquad(Q0 + (r + 3) / 4 - 1, r);  // entering side, off by one row: previous quad, both halves
                ca[s] = e.x;
                cb[s] = e.y;
                cc[s] = hquad(Q0 - (r - 1) / 4 - 1, r).y;        // leaving side, off by two rows: previous quad, upper half
            } else {
                ca[s] = hquad(Q0 + (r + 1) / 4 - 1, r).y;        // entering side, off by three rows: previous quad, upper half
                cb[s] = cc[s] = 0;
            }
        });
    }
    const long long plane = (long long)P.TR * P.TC * MT_ROWS;
    uint32_t *mframe = masks + (long long)f * NS * plane;
    auto flush = [&](int yblk) {
        const int yrow = yblk + lane;
        if (yrow < yend) {
            const int wc = (xs + colw) >> 5;
            uint32_t *q = mframe + mask_word(P.TC, yrow + 1, MASK_PADW + wc);
            const bool two = xs + colw + 32 < W;
            const uint32_t m0w = (uint32_t)vmask, m1w = (uint32_t)(vmask >> 32);  // columns right of the image stay zero
#pragma unroll
            for (int s = 0; s < NS; s++) {
#ifndef FID_PARK_MOV32
                q[(long long)s * plane] = (uint32_t)acc[s] & m0w;
                if (two) q[(long long)s * plane + MT_ROWS] = (uint32_t)(acc[s] >> 32) & m1w;
#else
                q[(long long)s * plane] = acc0[s] & m0w;
                if (two) q[(long long)s * plane + MT_ROWS] = acc1[s] & m1w;
#endif
            }
        }
    };
    for (int k = 0; k < nsteps; k++) {
        if (wave_on) {
            const int qo = Q0 + k;
            const int sel = (4 * k) & 63;
            // gray of the four output rows of this column, from the prefix itself
            uint32_t g0, g1, g2, g3;
            {
                const uint2 *b = slot_base(qo);
                const uint2 hi = b[PADL], lo = b[PADL - 1];
                const uint32_t a = pk_sub16(hi.x, lo.x), c = pk_sub16(hi.y, lo.y);
                g0 = a & 0xffffu;
                g1 = a >> 16;
                g2 = c & 0xffffu;
                g3 = c >> 16;
            }
            // the four prefix quads of a scale are fetched one scale ahead of their use
            struct Rd {
                uint2 eh, el, lh, ll;  // raw right / left prefix quads of the entering and the leaving side
            };
            auto fetch = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int r = (WMIN + s * WSTEP) / 2;
                Rd q;
                constexpr int de = (r & 3) == 1 ? (r + 3) / 4 : (r + 1) / 4, dl = (r & 3) == 1 ? (r - 1) / 4 : (r + 1) / 4;
                const uint2 *be = slot_base(qo + de), *bl = slot_base(qo - dl);
                q.eh = be[PADL + r];
                q.el = (be + zoff)[PADL - r - 1];
                q.lh = bl[PADL + r];
                q.ll = (bl + zoff)[PADL - r - 1];
                return q;
            };
            Rd cur = fetch(std::integral_constant<int, 0>{});
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int r = (WMIN + s * WSTEP) / 2, w2 = (2 * r + 1) * (2 * r + 1);
                Rd nxt = cur;
                if constexpr (s + 1 < NS) nxt = fetch(std::integral_constant<int, s + 1>{});
                __builtin_amdgcn_sched_barrier(0);  // the reads stay up here: their latency runs under this scale's arithmetic
                const uint2 e = make_uint2(pk_sub16(cur.eh.x, cur.el.x), pk_sub16(cur.eh.y, cur.el.y));
                const uint2 l = make_uint2(pk_sub16(cur.lh.x, cur.ll.x), pk_sub16(cur.lh.y, cur.ll.y));
                uint32_t elo, ehi, llo, lhi;
                if constexpr ((r & 3) == 1) {
                    elo = __builtin_amdgcn_alignbit(cb[s], ca[s], 16);  // rows 1, 2 of the previous quad
                    ehi = __builtin_amdgcn_alignbit(e.x, cb[s], 16);    // row 3 of the previous, row 0 of the new one
                    llo = cc[s];                                        // rows 2, 3 of the previous leaving quad
                    lhi = l.x;                                          // rows 0, 1 of the new one
                    ca[s] = e.x;
                    cb[s] = e.y;
                    cc[s] = l.y;
                } else {
                    elo = __builtin_amdgcn_alignbit(e.x, ca[s], 16);    // row 3 of the previous quad, row 0 of the new one
                    ehi = __builtin_amdgcn_alignbit(e.y, e.x, 16);      // rows 1, 2 of the new one
                    llo = l.x;
                    lhi = l.y;
                    ca[s] = e.y;
                }
                const thr_s16x2 dlo = __builtin_bit_cast(thr_s16x2, pk_sub16(elo, llo));
                const thr_s16x2 dhi = __builtin_bit_cast(thr_s16x2, pk_sub16(ehi, lhi));
                const thr_s16x2 first = {1, 0}, second = {0, 1};
                int v = V[s];
                v = __builtin_amdgcn_sdot2(dlo, first, v, false);
                const unsigned long long b0 = ballot64(v >= __mul24((int)g0, w2));
                v = __builtin_amdgcn_sdot2(dlo, second, v, false);
                const unsigned long long b1 = ballot64(v >= __mul24((int)g1, w2));
                v = __builtin_amdgcn_sdot2(dhi, first, v, false);
                const unsigned long long b2 = ballot64(v >= __mul24((int)g2, w2));
                v = __builtin_amdgcn_sdot2(dhi, second, v, false);
                const unsigned long long b3 = ballot64(v >= __mul24((int)g3, w2));
                V[s] = v;
#ifndef FID_PARK_MOV32
                park_quad64(acc[s], b0, b1, b2, b3, sel);
#else
                park_quad(acc0[s], acc1[s], b0, b1, b2, b3, sel);
#endif
                cur = nxt;
            });
            if (sel == 60 || k == nsteps - 1) flush(ys + ((4 * k) & ~63));
        }
        __syncthreads();
    }
}

// ---- SeedHash: seed state -> seed index, one open-addressing table of 64-bit entries per frame.
// entry = generation (10 bits, 54..63) | state (x | y << 13 | d << 26 | scale << 29: 34 bits, 20..53) | seed index (20 bits).
// The generation is a per-call number (1..1023): entries of earlier calls count as empty, so the table is never cleared
// between calls (the host clears it when the number wraps).  Inserts (k_seed_index) and lookups (k_seg_link) are in
// different kernels; the table holds at most maxContours entries in >= 2 maxContours slots.
__device__ __forceinline__ unsigned long long seedhash_key(uint32_t state, int scale) { return (unsigned long long)state | ((unsigned long long)scale << 29); }
__device__ __forceinline__ unsigned seedhash_slot(unsigned long long key, int cap)
{
    return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & (unsigned)(cap - 1);
}
__device__ __forceinline__ void seedhash_insert(unsigned long long *__restrict__ tab, int cap, int gen, unsigned long long key, unsigned idx)
{
    const unsigned long long ent = ((unsigned long long)gen << 54) | (key << 20) | idx;
    unsigned h = seedhash_slot(key, cap);
    unsigned long long old = tab[h];
    for (;;) {
        if ((int)(old >> 54) != gen) {
            const unsigned long long prev = atomicCAS(&tab[h], old, ent);
            if (prev == old) return;
            old = prev;  // somebody else's entry of this call: move on (checked again at the top)
            continue;
        }
        h = (h + 1) & (unsigned)(cap - 1);
        old = tab[h];
    }
}
__device__ __forceinline__ unsigned seedhash_find(const unsigned long long *__restrict__ tab, int cap, int gen, unsigned long long key)
{
    unsigned h = seedhash_slot(key, cap);
    for (int probes = 0; probes < cap; probes++) {
        const unsigned long long e = tab[h];
        if ((int)(e >> 54) != gen) return SEG_INVALID;
        if (((e >> 20) & 0x3ffffffffull) == key) return (unsigned)(e & 0xfffffu);
        h = (h + 1) & (unsigned)(cap - 1);
    }
    return SEG_INVALID;
}

// ------------------------------------------------------------------------------------------------
// K2: start points of Suzuki-Abe border following, found without the sequential raster scan.
//   outer border start: the first pixel of a horizontal foreground run whose W neighbour is background
//     and none of whose pixels has a foreground N / NW / NE neighbour (a run that touches nothing above
//     it; necessary for holding the raster-first pixel of its 8-connected component)
//   hole border start: the pixel left of the first pixel of a horizontal background run whose every
//     pixel has foreground above it (necessary for the run to hold the raster-first pixel of a
//     4-connected hole); the run start itself has foreground N by the same test
// Runs are tested inside one 32-bit word (fill_toward_lsb below); a run that crosses a word border is
// kept as a candidate (K3 makes the exact decision by walking).  32 pixels per lane-op, one atomic per
// workgroup iteration on a per-frame counter.
// (In the bit-reversed word the spread is the carry chain of ONE addition: a seed bit plus the run's own bit carries through the
//  rest of the run and stops in the zero behind it -- two instructions and the reversals instead of a five-level Kogge-Stone
//  fill; v_bfrev_b32 is a full-rate instruction.)
__device__ __forceinline__ uint32_t fill_toward_msb_rev(uint32_t rseed, uint32_t rruns)
{
    // rseed must be a subset of rruns; both bit-reversed, and so is the result
    const uint32_t t = rruns + rseed;
    return (t & rseed) | (~t & rruns);  // the seeds themselves, and the bits of the run the carry went through
}
__device__ __forceinline__ uint32_t fill_toward_lsb(uint32_t seed, uint32_t runs)
{
    // spread every seed bit to all lower bits of its run of ones in `runs`
    const uint32_t m = __brev(runs);
    return __brev(fill_toward_msb_rev(__brev(seed) & m, m));
}
// bit 0 of fill_toward_lsb(seed, runs): does the run that holds bit 0 hold a seed
__device__ __forceinline__ uint32_t low_run_has_seed(uint32_t seed, uint32_t runs)
{
    const uint32_t low = runs & ~(runs + 1u);  // the run of ones that starts at bit 0 (empty if bit 0 is clear)
    return (low & seed) != 0u ? 1u : 0u;
}

// a / b and a % b for a < 2^22 (group numbers: scales x tile rows x column groups) with the reciprocal of b at hand: eight
// instructions.  (The compiler's expansion of a 64-bit division is ~ 90 VALU instructions, and this kernel had twelve of them in
// its loop body: 60 % of everything it executed.)
__device__ __forceinline__ void divmod_small(unsigned a, unsigned b, float rcp_b, unsigned &q, unsigned &r)
{
    q = (unsigned)((float)a * rcp_b);  // off by at most one
    int rr = (int)(a - q * b);
    if (rr < 0) {
        q--;
        rr += (int)b;
    } else if (rr >= (int)b) {
        q++;
        rr -= (int)b;
    }
    r = (unsigned)rr;
}

// One thread per mask word column x 4 rows (one 16-byte load per word column: its own, the left and the right
// one, plus single words for the row above and the row below the group); a wave covers one tile row of
// 16 word columns = 16 whole mask tiles (1 KB contiguous).
// Two kinds of starts are dropped on the spot because their border is shorter than any perimeter gate
// (when minPerimeterPixels allows it): isolated foreground pixels (a 1-point outer border) and isolated
// background pixels (a hole border of at most 8 points).
// HYB = true additionally emits the SEEDS of seed-accelerated tracing (fid_device.h): states (pixel, d) on a grid line -- all
// pixels of a grid row with the vertical-component directions, bit 0 of the words that start on a grid column with the
// horizontal-component ones -- whose neighbour in direction d is foreground and whose neighbour in direction
// seed_empty_dir(d) is background.  Seeds go to their own list (x | y << 13 | scale << 27, d); k_seed_index then builds the
// map state -> seed index (SeedHash above).  (A seed need not lie on a real border state: such a seed walks into a real
// border and nobody ever links to it.)
// Two groups per wave and iteration, in two phases: (A) every lane loads the four rows of its word column of both groups
// (one 16-byte load each) and drops the word columns that are all background -- such a word cannot hold a start or a seed
// whatever its neighbours are, and about half of them are; (B) the remaining (word column, four rows) ITEMS of the two
// groups are dealt out again, one per lane (compaction through a 128-byte LDS list), and only those load their
// neighbourhood and run the bit tests.  On the bench frames that halves the instructions of this kernel.
template <bool HYB>
__global__ __launch_bounds__(256) void k_find_starts(const uint32_t *__restrict__ masks, uint2 *__restrict__ starts,
                                                      DevCounts *__restrict__ counts, DevGlobal *__restrict__ G,
                                                      uint2 *__restrict__ seedq, const DevParams P)
{
    __shared__ int s_wsum[2][4];
    __shared__ unsigned s_base[2];
    __shared__ uint8_t s_items[4][128];
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = blockIdx.y;
    const int WW = P.WW, TC = P.TC, TR = P.TR, H = P.H, S = P.nscales;
    const int CG = (WW + 15) / 16;     // groups of 16 word columns
    const int ngroups = S * TR * CG;   // a group = 16 word columns x 16 rows (one tile row); < 2^22 for every supported size
    const float rcpTR = 1.0f / (float)TR, rcpCG = 1.0f / (float)CG;
    const long long plane = (long long)TR * TC * MT_ROWS;
    const unsigned cap = (unsigned)P.maxStarts, scap = (unsigned)P.maxContours;
    const bool drop1 = P.minPerim > 1, drop8 = P.minPerim > 8;
    uint2 *fst = starts + (long long)f * P.maxStarts;
    uint2 *fsq = HYB ? seedq + (long long)f * P.maxContours : nullptr;
    const int gm = (8 << P.seedShift) - 1;  // seed grid spacing - 1
    const uint32_t *fmasks = masks + (long long)f * S * plane;
    // XCD-aware order: a group also reads one row of the tile rows above and below it (whole 64-byte lines for one word).
    // Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md) and every XCD has its own L2, so the tile rows of one (scale,
    // column group) go to ONE XCD, neighbouring tile rows to waves that run at the same time: those lines then come from
    // that XCD's L2 instead of crossing the fabric three times.  (Placement only changes speed: any order is correct.)
    const int ncol = S * CG;                                   // (scale, column group) pairs
    const bool by_xcd = (gridDim.x & 7) == 0 && ncol >= 8;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    const int mycols = by_xcd ? (ncol - xcd + 7) / 8 : 0;  // columns xcd, xcd + 8, ...
    const int nitems = by_xcd ? mycols * TR : ngroups;     // groups this workgroup's XCD share holds
    const int nit8 = (nitems + 7) & ~7;
    // the it-th group of this share as (scale, tile row, column group); false: none
    struct GroupId {  // (plain values, returned by value: as reference parameters of the lambda they ended up in scratch memory)
        int s, tr, cg;
        bool ok;
    };
    auto group_of = [&](int it) -> GroupId {
        GroupId g = {0, 0, 0, false};
        if (it >= nitems) return g;
        unsigned a, b, c, d;
        if (by_xcd) {
            divmod_small((unsigned)it, (unsigned)TR, rcpTR, a, b);  // a-th column of this XCD, tile row b
            divmod_small((unsigned)xcd + 8u * a, (unsigned)CG, rcpCG, c, d);  // column = s * CG + cg
            g.s = (int)c;
            g.tr = (int)b;
            g.cg = (int)d;
        } else {
            divmod_small((unsigned)it, (unsigned)CG, rcpCG, a, b);  // it = (s * TR + tr) * CG + cg
            divmod_small(a, (unsigned)TR, rcpTR, c, d);
            g.s = (int)c;
            g.tr = (int)d;
            g.cg = (int)b;
        }
        g.ok = true;
        return g;
    };
    for (int i0 = by_xcd ? lb * 8 : (int)blockIdx.x * 8; i0 < nit8; i0 += (by_xcd ? nlb : (int)gridDim.x) * 8) {
        // ---- phase A: which word columns of the wave's two groups hold any foreground in their four rows
        const GroupId g0 = group_of(i0 + 2 * wid), g1 = group_of(i0 + 2 * wid + 1);
        unsigned long long msel[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            int nz = 0;
            const GroupId gt = t ? g1 : g0;
            if (gt.ok) {
                const int cg = gt.cg, tr = gt.tr, s = gt.s;
                const int w = cg * 16 + (lane >> 2), r4 = (lane & 3) * 4, yy0 = tr * MT_ROWS + r4;
                if (w < WW && yy0 <= H && yy0 + 3 >= 1) {
                    const uint4 c4 = *reinterpret_cast<const uint4 *>(fmasks + (long long)s * plane + ((long long)tr * TC + MASK_PADW + w) * MT_ROWS + r4);
                    nz = (c4.x | c4.y | c4.z | c4.w) != 0u;
                }
            }
            msel[t] = ballot64(nz);
            if (nz) s_items[wid][(t ? __popcll(msel[0]) : 0) + __popcll(msel[t] & ((1ull << lane) - 1ull))] = (uint8_t)(lane | (t << 6));
        }
        const int nsel = __popcll(msel[0]) + __popcll(msel[1]);  // wave-uniform
        // ---- phase B: at most two rounds of 64 items
        uint32_t outer[2][4], hole[2][4];
        uint32_t rowm[2][6], colb[2];  // seeds: the item's grid row (six directions x 32 pixels), bit 0 of its four rows x 8 directions
        int x_base[2], yy0v[2], sv[2], cntv[2], scntv[2], rowk[2];
        int cnt = 0, scnt = 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int k = 0; k < 4; k++) outer[r][k] = hole[r][k] = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) rowm[r][k] = 0;
            colb[r] = 0;
            x_base[r] = yy0v[r] = sv[r] = cntv[r] = scntv[r] = rowk[r] = 0;
            if (r * 64 >= nsel) continue;  // wave-uniform
            const int idx = r * 64 + lane;
            if (idx < nsel) {
                const int item = s_items[wid][idx];
                const bool second = (item >> 6) != 0;
                const int l = item & 63;
                const int cg = second ? g1.cg : g0.cg, tr = second ? g1.tr : g0.tr, s = second ? g1.s : g0.s;
                const int w = cg * 16 + (l >> 2), r4 = (l & 3) * 4;
                const int yy0 = tr * MT_ROWS + r4;  // padded row of this item's first row; image row = yy - 1
                const int xb = w * 32;
                const uint32_t *pl = fmasks + (long long)s * plane;
                const long long word0 = ((long long)tr * TC + MASK_PADW + w) * MT_ROWS + r4;
                const uint32_t *tile = pl + word0;
                const uint4 c4 = *reinterpret_cast<const uint4 *>(tile);
                const uint4 p4 = *reinterpret_cast<const uint4 *>(tile - MT_ROWS);
                const uint4 n4 = *reinterpret_cast<const uint4 *>(tile + MT_ROWS);
                uint32_t upc = 0, upp = 0, upn = 0, dnc = 0, dnp = 0, dnn = 0;
                if (yy0 > 0) {
                    const uint32_t *qq = pl + mask_word(TC, yy0 - 1, MASK_PADW + w);
                    upc = qq[0];
                    upp = qq[-MT_ROWS];
                    upn = qq[MT_ROWS];
                }
                {
                    const uint32_t *qq = pl + mask_word(TC, yy0 + 4, MASK_PADW + w);  // exists: TR has a spare tile row
                    dnc = qq[0];
                    dnp = qq[-MT_ROWS];
                    dnn = qq[MT_ROWS];
                }
                const uint32_t cc[6] = {upc, c4.x, c4.y, c4.z, c4.w, dnc};
                const uint32_t pp[6] = {upp, p4.x, p4.y, p4.z, p4.w, dnp};
                const uint32_t nn[6] = {upn, n4.x, n4.y, n4.z, n4.w, dnn};
                int c1 = 0, c2 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int y = yy0 + k - 1;
                    if (y < 0 || y >= H) continue;
                    const uint32_t cur = cc[k + 1], prevc = pp[k + 1], nextc = nn[k + 1];
                    const uint32_t u = cc[k], prevu = pp[k], nextu = nn[k];
                    const uint32_t d = cc[k + 2], prevd = pp[k + 2], nextd = nn[k + 2];
                    const uint32_t Wst = (cur << 1) | (prevc >> 31);
                    const uint32_t Est = (cur >> 1) | (nextc << 31);
                    const uint32_t NW = (u << 1) | (prevu >> 31);
                    const uint32_t NE = (u >> 1) | (nextu << 31);
                    const uint32_t below = d | (d << 1) | (prevd >> 31) | (d >> 1) | (nextd << 31);
                    // outer: starts of foreground runs that have no foreground above (N / NW / NE) anywhere
                    const uint32_t touch = cur & (NW | u | NE);
                    const uint32_t rcur = __brev(cur);
                    uint32_t o = cur & ~Wst & ~__brev(fill_toward_msb_rev(__brev(touch), rcur));
                    if (drop1) o &= Est | below;  // an isolated pixel is a complete 1-point contour
                    // hole: first pixel e of a background run (W neighbour foreground) that is closed above;
                    // the border-following start is the foreground pixel LEFT of e
                    const uint32_t bg = ~cur;
                    const uint32_t open = bg & ~u;  // background with background above: joins an earlier pixel
                    uint32_t e = bg & Wst & ~__brev(fill_toward_msb_rev(__brev(open), ~rcur));
                    const uint32_t bgn = ~nextc;
                    uint32_t en0 = bgn & (cur >> 31) & ~low_run_has_seed(bgn & ~nextu, bgn) & 1u;
                    if (drop8) {
                        // a background pixel whose four 4-neighbours are foreground is a whole hole of its own
                        e &= ~(u & d & Est);  // W is foreground by construction
                        en0 &= ~(nextu & nextd & (nextc >> 1));  // N, S, E of the next word's pixel 0
                    }
                    outer[r][k] = o;
                    hole[r][k] = (e >> 1) | (en0 << 31);
                    c1 += __popc(outer[r][k]) + __popc(hole[r][k]);
                    if (HYB) {
                        // seed states on a grid row (every pixel, vertical-component directions) and on a grid column (bit 0
                        // of a word that starts on one, horizontal-component directions; the diagonal ones of a pixel that is
                        // on both lines belong to the row)
                        const bool grow = (y & gm) == 0, gcol = (xb & gm) == 0;
                        if (gcol) {
                            // column seeds are states of pixel 0 of the word: bit 0 of the eight neighbour planes as one byte
                            // (bit dd = neighbour in direction dd: E NE N NW W SW S SE), the "must be empty" neighbours the
                            // same byte rotated by seed_empty_dir (odd directions + 1, even ones + 2)
                            uint32_t n8 = (cur >> 1) & 1u;            // E
                            n8 |= u & 2u;                             // NE = pixel 1 of the row above
                            n8 |= (u & 1u) << 2;                      // N
                            n8 |= (prevu >> 31) << 3;                 // NW
                            n8 |= (prevc >> 31) << 4;                 // W
                            n8 |= (prevd >> 31) << 5;                 // SW
                            n8 |= (d & 3u) << 6;                      // S, SE = pixels 0, 1 of the row below
                            const uint32_t x16 = n8 | (n8 << 8);
                            const uint32_t e8 = ((x16 >> 1) & 0xAAu) | ((x16 >> 2) & 0x55u);
                            // (the diagonal directions of a pixel that is on both lines belong to the row)
                            const uint32_t allowed = grow ? (SEED_DIRS_COL & ~SEED_DIRS_ROW) : SEED_DIRS_COL;
                            const uint32_t s8 = (cur & 1u) ? (n8 & ~e8 & allowed) : 0u;
                            colb[r] |= s8 << (k * 8);
                        }
                        if (grow) {
                            const uint32_t SEst = (d >> 1) | (nextd << 31), SWst = (d << 1) | (prevd >> 31);
                            const uint32_t nbp[8] = {Est, NE, u, NW, Wst, SWst, d, SEst};  // neighbour planes by direction
                            int ri = 0;
#pragma unroll
                            for (int dd = 0; dd < 8; dd++) {
                                if (!((SEED_DIRS_ROW >> dd) & 1u)) continue;
                                const uint32_t m = cur & nbp[dd] & ~nbp[seed_empty_dir(dd)];
                                rowm[r][ri++] = m;
                                c2 += __popc(m);
                            }
                            rowk[r] = k;
                        }
                    }
                }
                if (HYB) c2 += __popc(colb[r]);
                x_base[r] = xb;
                yy0v[r] = yy0;
                sv[r] = s;
                cntv[r] = c1;
                scntv[r] = c2;
                cnt += c1;
                scnt += c2;
            }
        }
        // ---- list slots: one atomic per workgroup iteration and list
        const int incl = wave_iscan(cnt);
        const int sincl = HYB ? wave_iscan(scnt) : 0;
        if (lane == 63) {
            s_wsum[0][wid] = incl;
            s_wsum[1][wid] = sincl;
        }
        __syncthreads();
        int wbase = 0, tot = 0, swbase = 0, stot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = s_wsum[0][k], sv2 = s_wsum[1][k];
            if (k < wid) {
                wbase += v;
                swbase += sv2;
            }
            tot += v;
            stot += sv2;
        }
        if (threadIdx.x == 0) {
#ifdef FS_NOATOMIC  // timing experiment only (wrong lists): what the returning atomics cost this kernel
            if (tot) s_base[0] = (unsigned)(blockIdx.x * 977u + (unsigned)i0) % (cap - 64u);
            if (HYB && stot) s_base[1] = (unsigned)(blockIdx.x * 331u + (unsigned)i0) % (scap - 64u);
#else
            if (tot) s_base[0] = atomicAdd((unsigned *)&counts[f].nstarts, (unsigned)tot);
            if (HYB && stot) s_base[1] = atomicAdd((unsigned *)&counts[f].nseeds, (unsigned)stot);
#endif
        }
        __syncthreads();
        if (tot) {
            unsigned off = s_base[0] + (unsigned)(wbase + incl - cnt);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (!cntv[r]) continue;
                const uint32_t meta = (uint32_t)f | ((uint32_t)sv[r] << 16);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t y = (uint32_t)(yy0v[r] + k - 1);
                    uint32_t o = outer[r][k], hh = hole[r][k];
                    while (o) {
                        int b = __ffs(o) - 1;
                        o &= o - 1;
                        if (off < cap) fst[off] = make_uint2((uint32_t)(x_base[r] + b) | (y << 16), meta);
                        off++;
                    }
                    while (hh) {
                        int b = __ffs(hh) - 1;
                        hh &= hh - 1;
                        if (off < cap) fst[off] = make_uint2((uint32_t)(x_base[r] + b) | (y << 16), meta | (1u << 24));
                        off++;
                    }
                }
            }
            if (threadIdx.x == 0 && s_base[0] + (unsigned)tot > cap) atomicOr(&G->overflow, 1u);
        }
        if (HYB && stot) {
            unsigned off = s_base[1] + (unsigned)(swbase + sincl - scnt);
            auto emit = [&](int x, int y, int sc, int dd) {
                if (off < scap) fsq[off] = make_uint2((uint32_t)x | ((uint32_t)y << 13) | ((uint32_t)sc << 27), (uint32_t)dd);
                off++;
            };
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (!scntv[r]) continue;
                {
                    const int y = yy0v[r] + rowk[r] - 1;
                    int ri = 0;
#pragma unroll
                    for (int dd = 0; dd < 8; dd++) {
                        if (!((SEED_DIRS_ROW >> dd) & 1u)) continue;
                        uint32_t m = rowm[r][ri++];
                        while (m) {
                            const int b = __ffs(m) - 1;
                            m &= m - 1;
                            emit(x_base[r] + b, y, sv[r], dd);
                        }
                    }
                }
                uint32_t cb = colb[r];
                while (cb) {
                    const int b = __ffs(cb) - 1;
                    cb &= cb - 1;
                    emit(x_base[r], yy0v[r] + (b >> 3) - 1, sv[r], b & 7);
                }
            }
            if (threadIdx.x == 0 && s_base[1] + (unsigned)stot > scap) atomicOr(&G->overflow, 2u);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Border following on the bit-packed padded mask.
// Directions (contours.cpp icvCodeDeltas): 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE.

// direction d -> step (packed 2-bit tables of dx + 1, dy + 1)
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x901Au >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }

struct MaskView {
    const uint32_t *base;  // (frame, scale) mask plane, tiled
    int TC;
};

// 8-neighbourhood occupancy of pixel (x, y): bit d = neighbour in direction d is foreground
__device__ __forceinline__ unsigned nb8(const MaskView &m, int x, int y)
{
    // bits x-1, x, x+1 of rows y-1, y, y+1 (padded rows y, y+1, y+2); the second word column is only
    // touched when the three bits straddle a word
    const int xb = x - 1 + MASK_PADW * 32;
    const int wi = xb >> 5, sh = xb & 31;
    const bool two = sh > 29;
    const uint32_t *p0 = m.base + mask_word(m.TC, y, wi);
    const uint32_t *p1 = m.base + mask_word(m.TC, y + 1, wi);
    const uint32_t *p2 = m.base + mask_word(m.TC, y + 2, wi);
    const uint32_t a0 = p0[0], a1 = p1[0], a2 = p2[0];
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    if (two) {
        b0 = p0[MT_ROWS];
        b1 = p1[MT_ROWS];
        b2 = p2[MT_ROWS];
    }
    unsigned tu = __builtin_amdgcn_alignbit(b0, a0, sh) & 7u;
    unsigned tm = __builtin_amdgcn_alignbit(b1, a1, sh) & 7u;
    unsigned td = __builtin_amdgcn_alignbit(b2, a2, sh) & 7u;
    return ((tm >> 2) & 1u) | (((tu >> 2) & 1u) << 1) | (((tu >> 1) & 1u) << 2) | ((tu & 1u) << 3) | ((tm & 1u) << 4) |
           ((td & 1u) << 5) | (((td >> 1) & 1u) << 6) | (((td >> 2) & 1u) << 7);
}

// padded raster index used to order discovery events like cvFindNextContour's scan
__device__ __forceinline__ int pidx(int x, int y, int W) { return (y + 1) * (W + 2) + (x + 1); }

// Contour points leave the walkers as CHAIN CODES (round 6; until then 4-byte points x | y << 16): 4 bits per point -- the
// direction 0..7 of the step from the point to its successor on the border -- eight to a 32-bit word (point k of a chunk in
// bits 4 (k & 7) of word k >> 3), in chunks of CK points = CKW words taken from a per-launch pool while the border is followed;
// chunk_tab[slot][k] names the chunk that holds points [CK*k, CK*k + CK) of a contour.  Codes do not depend on where a piece
// of a border ends up in a contour: k_seg_copy moves them (one BYTE per point in the dense array) and the consumers
// (k_approx, k_refine_contour) turn them back into points with a prefix sum from the contour's first point, which the contour
// record carries anyway.  0.5 + 1 bytes per point through HBM instead of 4 + 4 twice.
#define CK 64
#define CKW 8
#define REC_NO_CHUNK 0xffffffffu  // copy record: the row's chunks are all in chunk_tab (probe survivors, trace mode 1)
// packed step of a chain code: dx + 65536 dy as a 32-bit integer (x + 65536 y is linear: sums of these are sums of steps)
__device__ __forceinline__ uint32_t code_delta(unsigned c)
{
    const unsigned c2 = (c & 7u) * 2u;
    return (((0x901Au >> c2) & 3u) | (((0xA901u >> c2) & 3u) << 16)) - 0x10001u;
}
// eight codes of a word (one per nibble) -> one per byte, low four in .x
__device__ __forceinline__ uint2 codes_nibbles_to_bytes(uint32_t w)
{
    uint32_t a = w & 0xffffu, b = w >> 16;
    a = (a | (a << 8)) & 0x00ff00ffu;
    b = (b | (b << 8)) & 0x00ff00ffu;
    return make_uint2((a | (a << 4)) & 0x0f0f0f0fu, (b | (b << 4)) & 0x0f0f0f0fu);
}
// A wave turns 512 consecutive chain codes back into points: lane l holds the codes of points k .. k + 7 (k = k0 + 8 l, one
// per byte of w), `base` = point k0 (wave-uniform).  Point k + i = base + the steps in front of it: a serial sum over the
// lane's own eight, a DPP scan over the lanes' totals.  The lane writes its eight points (two 16-byte LDS stores: k is a
// multiple of 8; the last lane of a contour may write up to seven words past the contour's length, never read) and the
// function returns point k0 + 512.  Bytes past the contour's end may hold anything: they only ever reach those words.
__device__ __forceinline__ uint32_t codes8_to_points(uint32_t *pts, int k, int count, uint32_t base, uint2 w)
{
    uint32_t p[8], s = 0;
    const uint32_t w2[2] = {w.x << 1, w.y << 1};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned c2 = (w2[i >> 2] >> (8 * (i & 3))) & 14u;
        p[i] = s - (uint32_t)i * 0x10001u;  // (the steps are summed with their bias dx + 1, dy + 1)
        s += ((0x901Au >> c2) & 3u) | (((0xA901u >> c2) & 3u) << 16);
    }
    s -= 8u * 0x10001u;
    const uint32_t incl = wave_iscan_dpp(s);
    const uint32_t mine = base + incl - s;
    if (k < count) {
        *reinterpret_cast<uint4 *>(pts + k) = make_uint4(mine + p[0], mine + p[1], mine + p[2], mine + p[3]);
        *reinterpret_cast<uint4 *>(pts + k + 4) = make_uint4(mine + p[4], mine + p[5], mine + p[6], mine + p[7]);
    }
    return base + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}
// entries per contour in chunk_tab: the windowed walk may run WALK_RUN (< 64) points past maxPerimeterPixels before it notices
__device__ __host__ inline int chunk_tab_pitch(const DevParams &P) { return P.maxPerim / CK + 3; }
// A frame's table is laid out BY ENTRY: entry k of row r (rows 0 .. maxContours - 1: the seeds' segments, maxContours .. 2 maxContours
// - 1: the probe survivors) at [k * 2 maxContours + r].  Nearly every walker only ever has entries 0 and 1 (128 points): with a row
// per walker those were 8 bytes in a 64-byte line of their own, written once and fetched once; by entry they are two dense arrays.
__device__ __forceinline__ long long chunk_tab_at(const DevParams &P, int f, unsigned row, unsigned k)
{
    return ((long long)f * chunk_tab_pitch(P) + k) * (2ll * P.maxContours) + row;
}

// K3 probe passes: one lane per start.  Walks the border exactly as icvFetchContour does (contours.cpp).
// A start is kept only if it can be the canonical one of its border (the pixel where cvFindNextContour's
// raster scan would have started it): outer borders start at their raster-first pixel, hole borders left of
// the raster-first background pixel of the hole.  A walker gives up as soon as it meets an earlier pixel;
// to make that happen fast on staircase edges a second cursor walks the border BACKWARDS.
// ~98 % of the starts die within a few steps while ~2 % run for hundreds to thousands, and a wave lasts as
// long as its longest lane, so the starts are sieved twice before the full walk:
//   LEVEL 0   every start, at most PROBE0_STEPS steps in each direction -> surv1   (kills ~90 %)
//   LEVEL 1   surv1, at most PROBE1_STEPS steps                         -> surv
// What is still undecided (or closed with a length that passes the perimeter gate) is appended to the next
// list with one atomic per wave.  Contours shorter than a probe are rejected here for good when they fail
// minMarkerPerimeterRate.
#ifndef PROBE0_STEPS
#define PROBE0_STEPS 6
#endif
#ifndef PROBE1_STEPS
#define PROBE1_STEPS 32
#endif
// STOPSEED (trace mode 2): a start whose walk meets a seed state is dropped -- its border is a seed cycle, k_seg_cycles
// finds it without a start.
template <int STEPS, int LEVEL, bool STOPSEED = false>
__global__ __launch_bounds__(256) void k_probe(const uint32_t *__restrict__ masks, const uint2 *__restrict__ in_list,
                                                uint2 *__restrict__ out_list, DevCounts *__restrict__ counts,
                                                DevGlobal *__restrict__ G, const DevParams P)
{
    const int f = blockIdx.y;
    const int lane = lane_id();
    // LEVEL 0: starts -> surv1, LEVEL 1: surv1 -> surv, LEVEL 2: starts -> surv (single sieve)
    unsigned n = (unsigned)(LEVEL != 1 ? counts[f].nstarts : counts[f].nsurv1);
    n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
    int *out_count = LEVEL == 0 ? &counts[f].nsurv1 : &counts[f].nsurv;
    const int W = P.W, S = P.nscales;
    const long long plane = (long long)P.TR * P.TC * MT_ROWS;
    const uint2 *fin = in_list + (long long)f * P.maxStarts;
    uint2 *fout = out_list + (long long)f * P.maxStarts;
    for (unsigned i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + lane;
        const bool active = i < n;
        uint2 st = active ? fin[i] : make_uint2(0u, 0u);
        int x0 = st.x & 0xffff, y0 = st.x >> 16;
        int s = (st.y >> 16) & 0xff, hole = (st.y >> 24) & 1;
        MaskView m;
        m.base = masks + ((long long)f * S + s) * plane;
        m.TC = P.TC;
        // canonical key: outer = own index, hole = index of the background pixel to the right
        const int key = hole ? pidx(x0 + 1, y0, W) : pidx(x0, y0, W);
        const int s_end = hole ? 0 : 4;
        const int sgm = (8 << P.seedShift) - 1;
        auto on_seed = [&](int x, int y, int d, unsigned nbh) { return seed_state(x, y, d, sgm) && !((nbh >> seed_empty_dir(d)) & 1u); };
        int count = 0, ok = active, closed = 0;
        unsigned nb = ok ? nb8(m, x0, y0) : 0u;
        if (!ok) {
        } else if (nb == 0) {
            count = 1;  // single pixel domain
            closed = 1;
        } else {
            // do { s = (s - 1) & 7; } while (*i1 == 0 && s != s_end)  == first foreground clockwise from s_end - 1
            int sdir;
            {
                unsigned nb2 = nb | (nb << 8);
                int c0 = (s_end - 1) & 7;
                unsigned win = (nb2 >> (c0 + 1)) & 0xffu;
                int t = 7 - (31 - __clz((int)win));
                sdir = (c0 - t) & 7;
            }
            const int i1x = x0 + dir_dx(sdir), i1y = y0 + dir_dy(sdir);
            // backward cursor starts on i1 with forward direction pointing at the start pixel
            int bx = i1x, by = i1y, bf = (sdir + 4) & 7;
            if (!hole && pidx(bx, by, W) < key) ok = 0;
            if (STOPSEED && on_seed(x0, y0, sdir, nb)) ok = 0;  // the start state itself
            int cx = x0, cy = y0;
            while (ok) {
                // ---- forward step: first foreground counter-clockwise from sdir + 1
                unsigned nb2 = nb | (nb << 8);
                int start = (sdir + 1) & 7;
                unsigned rot = (nb2 >> start) & 0xffu;
                int t = __ffs(rot) - 1;  // rot != 0: the neighbour we came from is foreground
                if (hole) {
                    // background pixels examined in the 4-directions belong to this border's hole region
                    for (int q = 0; q < t; q++) {
                        int d = (start + q) & 7;
                        if (!(d & 1) && pidx(cx + dir_dx(d), cy + dir_dy(d), W) < key) ok = 0;
                    }
                }
                int sn = (start + t) & 7;
                count++;
                int nx = cx + dir_dx(sn), ny = cy + dir_dy(sn);
                if (!ok || count > P.maxPerim) {
                    ok = 0;
                    break;
                }
                if (nx == x0 && ny == y0 && cx == i1x && cy == i1y) {
                    closed = 1;
                    break;
                }
                if (count >= STEPS) break;
                cx = nx;
                cy = ny;
                if (!hole && pidx(cx, cy, W) < key) {
                    ok = 0;
                    break;
                }
                sdir = (sn + 4) & 7;
                nb = nb8(m, cx, cy);
                if (STOPSEED && on_seed(cx, cy, sdir, nb)) {
                    ok = 0;
                    break;
                }
                // ---- backward step: predecessor = first foreground clockwise from bf - 1
                {
                    unsigned bn = nb8(m, bx, by);
                    unsigned bn2 = bn | (bn << 8);
                    int c0 = (bf - 1) & 7;
                    unsigned win = (bn2 >> (c0 + 1)) & 0xffu;
                    int tz = 7 - (31 - __clz((int)win));
                    if (hole) {
                        for (int q = 0; q < tz; q++) {
                            int d = (c0 - q) & 7;
                            if (!(d & 1) && pidx(bx + dir_dx(d), by + dir_dy(d), W) < key) ok = 0;
                        }
                    }
                    int bd = (c0 - tz) & 7;
                    if (STOPSEED && on_seed(bx, by, bd, bn)) ok = 0;  // the state (pixel, back direction) the cursor stood in
                    bx += dir_dx(bd);
                    by += dir_dy(bd);
                    bf = (bd + 4) & 7;
                    if (!hole && pidx(bx, by, W) < key) ok = 0;
                }
            }
        }
        // still undecided, or closed with a length that passes the perimeter gate
        const int keep = ok && (!closed || (count >= P.minPerim && count <= P.maxPerim));
        const unsigned long long mk = ballot64(keep);
        if (mk) {
            const int leader = __ffsll((long long)mk) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd((unsigned *)out_count, (unsigned)__popcll(mk));
            base = __shfl(base, leader, WAVE);
            const unsigned idx = base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
            if (keep) {
                if (idx < (unsigned)P.maxStarts) fout[idx] = st;
                else atomicOr(&G->overflow, 1u);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3 full pass, windowed.  A border-following step is one dependent memory round trip (the 3x3
// neighbourhood of the new pixel); with 64 independent walkers per wave the round trip of the slowest lane
// is paid on every step.  Here each lane keeps a private WINDOW of the mask in LDS -- 64 px x 16 rows = 128 bytes
// (2 word columns x four 4-row quarters of mask tiles), placed ahead of the direction of travel -- and steps inside it at LDS
// latency.  A lane that leaves its window parks; every WALK_CKPT iterations (or as soon as nobody can step)
// the wave reaches a CHECKPOINT where everything that touches memory is batched and asynchronous:
//   * s_waitcnt vmcnt(0): window refills (LDS-DMA, global_load_lds_dwordx4, 8 per lane, no VGPRs in
//     flight) and the survivor prefetch issued at the PREVIOUS checkpoint have landed -> those lanes resume
//   * finished walkers retire (contour slot written), idle lanes take the next survivors of the wave's range
//   * every walker without a spare pool chunk gets one from the wave's ARENA (a run of WALK_ARENA chunks
//     taken from the frame pool with one atomic, then handed out by ballot + popcount), parked lanes issue
//     their refill
// so the memory latency of one lane overlaps the steps of the others and a checkpoint waits on a round trip
// of its own only when the arena runs dry.  Points go straight to the contour's pool chunk (fire-and-forget
// stores).
// One step = 6 LDS words -> 8 neighbour bits -> one LDS table look-up that yields the next direction and
// which background 4-neighbours the step examined (the hole-border canonical test).
// The backward cursor of the probe pass is not needed here: it only ever rejects, and the forward cursor
// visits every pixel of the border.
#ifndef WALK_CKPT
#define WALK_CKPT 8
#endif
#define WALK_RUN 32     // most steps between two checkpoints
#define WALK_GRAB 64    // survivors a wave takes from the frame's work queue per atomic
#define WALK_ARENA 256  // pool chunks a wave takes per atomic
#ifndef WALK_STEAL_MIN
#define WALK_STEAL_MIN 32  // idle lanes a wave must have before it looks for another frame's queue
#endif

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// raw neighbourhood byte (bits 0-2 row above x-1..x+1, bit 3 W, bit 4 E, bits 5-7 row below) -> direction order
__device__ __forceinline__ unsigned raw_to_nb(unsigned raw)
{
    return ((raw >> 4) & 1u) | (((raw >> 2) & 1u) << 1) | (((raw >> 1) & 1u) << 2) | ((raw & 1u) << 3) | (((raw >> 3) & 1u) << 4) |
           (raw & 0xe0u);
}

__device__ __forceinline__ int first_dir(unsigned nb, int s_end)
{
    // do { s = (s - 1) & 7; } while (*i1 == 0 && s != s_end)  == first foreground clockwise from s_end - 1
    const unsigned nb2 = nb | (nb << 8);
    const int c0 = (s_end - 1) & 7;
    const unsigned win = (nb2 >> (c0 + 1)) & 0xffu;
    const int t = 7 - (31 - __clz((int)win));
    return (c0 - t) & 7;
}

// step table: index raw | backdir << 8 -> next direction | code << 3 | seed-state flag << 6; code = the smallest-offset
// background 4-neighbour the search passed over (0 none, else 4 | positive << 1 | whole-row)
__device__ __forceinline__ void build_step_lut(uint8_t *lut, int tid, int nthreads)
{
    for (int e = tid; e < 2048; e += nthreads) {
        const unsigned raw = (unsigned)e & 0xffu;
        const unsigned nb = raw_to_nb(raw);
        const int sd = e >> 8;
        const int start = (sd + 1) & 7;
        const unsigned rot = ((nb | (nb << 8)) >> start) & 0xffu;
        const int t = rot ? __ffs(rot) - 1 : 0;
        unsigned seen = 0;
        for (int q = 0; q < t; q++) seen |= 1u << ((start + q) & 7);
        const int code = (seen & 4u) ? 5 : (seen & 16u) ? 4 : (seen & 1u) ? 6 : (seen & 64u) ? 7 : 0;
        // seed-state flag (to be combined with the grid-line test seed_state(x, y, sd) by the walker)
        const int seed = ((nb >> sd) & 1u) && !((nb >> seed_empty_dir(sd)) & 1u);
        lut[e] = (uint8_t)(((start + t) & 7) | (code << 3) | (seed << 6));
    }
}

// raw 3x3 neighbourhood byte of pixel (cx, cy) read from lane `lane`'s window (origin: pixel column wx0 of
// bit 0, padded row wy0 of window row 0)
__device__ __forceinline__ unsigned win_raw(const uint32_t *s_winw, int lane4, int cx, int cy, int wx0, int wy0)
{
    const int xr = cx + (MASK_PADW * 32 - 1) - wx0;  // bit of x-1 in the 64-bit window row
    const int rr = cy - wy0;                         // window row of image row y-1
    unsigned t3[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const int r = rr + d;
        const int idx = ((r & ~3) << 6) + (r & 3) + lane4;  // chunk (r >> 2), lane, element (r & 3)
        const uint32_t w0 = s_winw[idx], w1 = s_winw[idx + 4 * 64 * 4];
        const unsigned long long v = ((unsigned long long)w1 << 32) | w0;
        t3[d] = (unsigned)(v >> xr);
    }
    return (t3[0] & 7u) | ((t3[1] & 1u) << 3) | ((t3[1] & 4u) << 2) | ((t3[2] & 7u) << 5);
}

// MODE 0: a walker follows a whole border from a probe survivor and applies the canonical-start test as it goes.
// MODE 1: a walker follows one SEGMENT, from its seed state to the next seed state, and records length and minima.
// MODE 2: as MODE 0, but the walker stops in front of the first seed state it meets (k_seg_chain takes over from there).
template <int MODE>
// Four independent walker waves per workgroup (one per SIMD: one-wave workgroups were all placed on the same SIMD of a
// CU, which capped a CU at one SIMD's issue rate); they share nothing but the step table.
#define WALK_WAVES 4
__global__ __launch_bounds__(64 * WALK_WAVES) void k_walk_full(const uint32_t *__restrict__ masks, const uint2 *__restrict__ surv,
                                                   uint4 *__restrict__ contours, uint32_t *__restrict__ chunk_tab,
                                                   uint32_t *__restrict__ pool, DevSeg *__restrict__ segs, DevPend *__restrict__ pend,
                                                   DevCounts *__restrict__ counts, DevGlobal *__restrict__ G, const DevParams P)
{
    constexpr bool SEG = MODE == 1;
    // window: 64 px x 16 rows per lane = 2 word columns x 4 row quarters; chunk j = c * 4 + q (16 bytes = rows
    // 4q..4q+3 of word column c) of lane l lives at s_win[j * 64 + l].  8 KB per wave -> 16 waves per CU.
    __shared__ uint4 s_win_all[WALK_WAVES][8 * 64];
    uint4 *s_win = s_win_all[threadIdx.x >> 6];
    // step table: index raw | backdir << 8 -> next direction | code << 3, code = the smallest-offset background
    // 4-neighbour the search passed over (0 none, else 4 | positive << 1 | whole-row)
    __shared__ uint8_t s_lut[2048];
    const uint32_t *s_winw = reinterpret_cast<const uint32_t *>(s_win);
    int f = blockIdx.y;  // the frame whose queue this wave is handing out; when that queue runs dry the wave moves on to the
                         // next frame that still has walkers waiting WITHOUT waiting for its own lanes: a lane keeps the
                         // frame of its walker (lf), point chunks come from one pool for the whole launch
#ifdef FID_DEBUG_STATS
    const unsigned long long d_k0 = __builtin_readcyclecounter();
    unsigned long long d_kexh = 0;
#endif
    const int lane = lane_id();
    const int lane4 = lane * 4;
    build_step_lut(s_lut, threadIdx.x, 64 * WALK_WAVES);
    __syncthreads();
    const unsigned ccap = (unsigned)P.maxContours, pcap = (unsigned)P.maxChunks * (unsigned)P.nframes;
    const int W = P.W, S = P.nscales, TC = P.TC, TR = P.TR, F = P.nframes;
    const int W2 = W + 2;
    const int sgm = (8 << P.seedShift) - 1;  // seed grid spacing - 1
    const long long plane = (long long)TR * TC * MT_ROWS;
    enum { ST_IDLE = 0, ST_ACTIVE, ST_NEED, ST_LOADING, ST_FINAL };

    {
        // the queue being handed out (wave-uniform)
        unsigned n = 0;
        const uint2 *fin = surv;
        unsigned *qhead = nullptr;
        auto set_queue = [&](int fr) {
            n = (unsigned)(SEG ? counts[fr].nseeds : counts[fr].nsurv);
            n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
            if (n > ccap) {  // more walkers than contour rows: reported, never silently dropped
                if (lane == 0) atomicOr(&G->overflow, 2u);
                n = ccap;
            }
            fin = surv + (long long)fr * (SEG ? P.maxContours : P.maxStarts);
            qhead = (unsigned *)(SEG ? &counts[fr].nwalk2 : &counts[fr].nwalk);
        };
        set_queue(f);
        int all_done = 0;  // no frame of the launch has walkers waiting any more (wave-uniform)
        // per-lane views of the walker's own frame lf: contour rows, chunk rows (seeds 0 .. maxContours-1, survivors
        // maxContours .. 2 maxContours-1); point chunks are numbered across the whole launch
        int lf = f;
        auto fco_of = [&](int fr) { return contours + (long long)fr * P.maxContours; };
        auto tab_at = [&](int fr, unsigned sl, unsigned k) -> uint32_t & { return chunk_tab[chunk_tab_at(P, fr, (SEG ? 0u : ccap) + sl, k)]; };
        uint32_t *const fpool = pool;
        // the wave's current batch of the frame's survivor queue: [next, rend), records of batch base .. base + 63 in `pre`
        unsigned next = 0, rend = 0, pre_base = 0;  // wave-uniform
        int exhausted = 0, pre_ready = 0;           // wave-uniform
        uint2 pre = make_uint2(0u, 0u);
        // per-lane walker
        int state = ST_IDLE;
        uint2 st = make_uint2(0u, 0u);
        const uint32_t *pl = masks;  // mask plane of the walker's scale
        int x0 = 0, y0 = 0, hole = 0, key = 0;
        unsigned slot = 0;
        int cx = 0, cy = 0, pc = 0, sdir = 0, i1x = 0, i1y = 0, first = 0;
        int count = 0, ok = 0, closed = 0;
        int wx0 = 0, wy0 = 0;  // window origin: pixel column of bit 0 / padded row of window row 0
        int ndx = 1, ndy = 1;  // direction of travel used to place the next window
        unsigned chunkA = 0, chunkB = 0;  // pool chunks of the even / odd 64-point blocks around `count`
        int kreg = 0;                     // highest block index that has a chunk
        unsigned ovf = 0;
        unsigned mout = 0xffffffffu, mhole = 0xffffffffu;  // MODE 1: running minima of the segment
        int too_long = 0, stopped = 0;
        int brx = -1, bry = -1, brd = -1;  // MODE 1: state remembered for the cycle test
        uint32_t acc = 0;  // the chain codes of the last (up to) eight points, oldest in the low nibble; every eighth step they leave as ONE word
        unsigned arena_next = 0, arena_end = 0;  // wave-uniform
#ifdef FID_DEBUG_STATS
        unsigned long long d_iters = 0, d_ckpts = 0, d_active = 0, d_ckcyc = 0, d_waitcyc = 0, d_forced = 0;
        unsigned long long d_seg[5] = {0, 0, 0, 0, 0}, d_lanes[4] = {0, 0, 0, 0}, d_mark = 0;
#define FID_DSEG(K)                                            \
    {                                                          \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        d_seg[K] += t_ - d_mark;                               \
        d_mark = t_;                                           \
    }
        const unsigned long long d_t0 = __builtin_readcyclecounter();
#endif
        for (;;) {
            // ================= checkpoint =================
#ifdef FID_DEBUG_STATS
            const unsigned long long d_c0 = __builtin_readcyclecounter();
            d_ckpts++;
#endif
            wait_vmcnt0();
#ifdef FID_DEBUG_STATS
            d_waitcyc += __builtin_readcyclecounter() - d_c0;
            d_mark = __builtin_readcyclecounter();
            d_lanes[0] += __popcll(ballot64(state == ST_NEED));
            d_lanes[1] += __popcll(ballot64(state == ST_LOADING));
            d_lanes[2] += __popcll(ballot64(state == ST_FINAL));
            d_lanes[3] += __popcll(ballot64(state == ST_IDLE));
#endif
            if (next < rend) pre_ready = 1;  // the batch's records were requested at the previous checkpoint
            if (state == ST_LOADING) {
                state = ST_ACTIVE;
                if (first) {
                    // the walker's first look at its start pixel: single-pixel domain, initial direction
                    //   do { s = (s - 1) & 7; } while (*i1 == 0 && s != s_end)  == first foreground clockwise from s_end - 1
                    first = 0;
                    const unsigned raw = win_raw(s_winw, lane4, cx, cy, wx0, wy0);
                    if (raw == 0) {
                        count = 1;  // (the contour's only point is its start: no step, no code)
                        closed = 1;
                        mout = (unsigned)pc;
                        state = ST_FINAL;
                    } else {
                        // a seed starts in its seed state; a survivor as icvFetchContour starts a border
                        sdir = SEG ? (int)(st.y & 7u) : first_dir(raw_to_nb(raw), hole ? 0 : 4);
                        i1x = x0 + dir_dx(sdir);
                        i1y = y0 + dir_dy(sdir);
                        if (!SEG && !hole && pidx(i1x, i1y, W) < key) {
                            ok = 0;
                            state = ST_FINAL;
                        }
                    }
                }
            }
            if ((state == ST_ACTIVE || state == ST_NEED) && count > P.maxPerim) {
                // checked here, not per step: a walk overshoots by at most WALK_RUN points
                ok = 0;
                too_long = 1;
                state = ST_FINAL;
            }
#ifdef FID_DEBUG_STATS
            FID_DSEG(0)  // loading -> active, first look, perimeter cap
#endif
            // ---- retire finished walkers
            if (state == ST_FINAL) {
                {
                    // the codes of the last count % 8 points are still in the shift register (newest in the top nibble)
                    const int rem = count & 7, b0 = count - rem;
                    if (rem) fpool[((b0 & CK ? chunkB : chunkA) * CKW) + (unsigned)((b0 & (CK - 1)) >> 3)] = acc >> (4 * (8 - rem));
                }
                if (SEG) {
                    DevSeg *r = segs + (long long)lf * P.maxContours + slot;
                    r->next_key = seed_key(cx, cy, sdir);  // the seed state the walk stopped in front of
                    r->n = too_long || !ok ? SEG_INVALID : (unsigned)count;
                    r->mout = mout;
                    r->mhole = mhole;
                } else {
                    // stopped in front of a seed state (MODE 2): k_seg_chain decides; else decided here
                    const int accept = ok && closed && !stopped && count >= P.minPerim && count <= P.maxPerim;
                    fco_of(lf)[slot] = make_uint4(st.x, st.y, accept ? (unsigned)count : 0u, (unsigned)key);
                    if (MODE == 2) {
                        DevPend *pd = pend + (long long)lf * P.maxContours + slot;
                        pd->p = ok && stopped ? (unsigned)count : 0u;
                        pd->next_key = seed_key(cx, cy, sdir);
                    }
                }
                state = ST_IDLE;
#ifdef FID_DEBUG_STATS
                if (FID_DEBUG_STATS + 0 < 0 || MODE == FID_DEBUG_STATS + 0) {
                    atomicMax(&G->dbg[13], (unsigned long long)count);
                    atomicAdd(&G->dbg[14], (unsigned long long)count);
                }
#endif
            }
#ifdef FID_DEBUG_STATS
            FID_DSEG(1)  // retire
#endif
            // ---- hand out new work to idle lanes
            int fresh = 0;
            const unsigned long long idle = ballot64(state == ST_IDLE);
            if (exhausted && next == rend && !all_done && __popcll(idle) >= WALK_STEAL_MIN) {
                // this frame's queue is empty and enough lanes are idle to make it worth a look: hand out the next frame
                // that still has walkers waiting (cyclic order from a wave-specific offset, so that the waves that run dry
                // together do not all descend on the same queue); 64 frames are examined per load
                int nextf = -1;
                const int hop = (int)((blockIdx.x * WALK_WAVES + (threadIdx.x >> 6)) * 37u % (unsigned)F);
                for (int k0 = 1; k0 < F && nextf < 0; k0 += 64) {
                    const int k = k0 + lane;
                    int has = 0;
                    int fr = (f + hop + k) % F;
                    if (fr == f) fr = -1;
                    if (k < F && fr >= 0) {
                        const unsigned done = __hip_atomic_load((unsigned *)(SEG ? &counts[fr].nwalk2 : &counts[fr].nwalk), __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                        unsigned m = (unsigned)(SEG ? counts[fr].nseeds : counts[fr].nsurv);
                        m = m < (unsigned)P.maxStarts ? m : (unsigned)P.maxStarts;
                        m = m < ccap ? m : ccap;
                        has = done < m;
                    }
                    const unsigned long long hb = ballot64(has);
                    if (hb) nextf = __shfl(fr, __ffsll((long long)hb) - 1, WAVE);
                }
                if (nextf < 0) {
                    all_done = 1;
                } else {
                    f = nextf;
                    set_queue(f);
                    exhausted = 0;
                    next = rend = 0;
                    pre_ready = 0;
                }
            }
            if (idle) {
                if (next == rend && !exhausted) {
                    // take the next batch of the frame's survivors; its records arrive by the next checkpoint
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(qhead, (unsigned)WALK_GRAB);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n) {
                        exhausted = 1;
#ifdef FID_DEBUG_STATS
                        if (!d_kexh) d_kexh = __builtin_readcyclecounter() - d_k0;
#endif
                    } else {
                        next = pre_base = base;
                        rend = base + WALK_GRAB < n ? base + WALK_GRAB : n;
                        pre_ready = 0;
                        // the survivor list is roughly ordered by threshold scale, and the largest windows hold the
                        // longest borders: hand the list out back to front so that those do not end up in the tail
                        if (base + lane < rend) pre = fin[n - 1 - (base + lane)];
                    }
                } else if (next < rend && pre_ready) {
                    const int rank = __popcll(idle & ((1ull << lane) - 1ull));
                    const int src = (int)(next - pre_base) + rank;
                    const unsigned gx = __shfl(pre.x, src & 63, WAVE), gy = __shfl(pre.y, src & 63, WAVE);
                    if (state == ST_IDLE && next + (unsigned)rank < rend) {
                        slot = n - 1 - (next + (unsigned)rank);  // the walker's index in its list (handed out back to front)
                        if (slot < ccap) {
                            lf = f;
                            st = make_uint2(gx, gy);
                            int s;
                            if (SEG) {  // x | y << 13 | scale << 27, back direction of the seed state
                                x0 = gx & 0x1fff;
                                y0 = (gx >> 13) & 0x1fff;
                                hole = 0;
                                s = gx >> 27;
                                mout = mhole = 0xffffffffu;
                                too_long = 0;
                                brx = bry = brd = -1;
                            } else {
                                x0 = st.x & 0xffff;
                                y0 = st.x >> 16;
                                s = (st.y >> 16) & 0xff;
                                hole = (st.y >> 24) & 1;
                            }
                            pl = masks + ((long long)f * S + s) * plane;
                            key = hole ? pidx(x0 + 1, y0, W) : pidx(x0, y0, W);
                            count = 0;
                            closed = 0;
                            stopped = 0;
                            ok = 1;
                            first = 1;
                            cx = x0;
                            cy = y0;
                            pc = pidx(x0, y0, W);
                            ndx = 1;
                            ndy = 1;
                            kreg = 1;
                            fresh = 1;
                            state = ST_NEED;
                        } else {
                            ovf |= 2u;
                        }
                    }
                    const unsigned nidle = (unsigned)__popcll(idle);
                    next = next + nidle < rend ? next + nidle : rend;
                }
            }
#ifdef FID_DEBUG_STATS
            FID_DSEG(2)  // queue switch, grab, hand-out
#endif
            // ---- pool chunks from the wave's arena: two for a fresh walker (blocks 0 and 1), one for every walker
            //      that has entered its last chunked block (count >> 6 == kreg)
            {
                const int want1 = !fresh && (state == ST_ACTIVE || state == ST_NEED) && (count >> 6) == kreg;
                const unsigned long long b1 = ballot64(want1), b2 = ballot64(fresh);
                const unsigned total = (unsigned)__popcll(b1) + 2u * (unsigned)__popcll(b2);
                if (total) {
                    if (arena_next + total > arena_end) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd((unsigned *)&counts[0].npool, (unsigned)WALK_ARENA);  // one pool per launch
                        arena_next = __builtin_amdgcn_readfirstlane(base);
                        arena_end = arena_next + WALK_ARENA;
                    }
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    const unsigned mine = arena_next + (unsigned)__popcll(b1 & lt) + 2u * (unsigned)__popcll(b2 & lt);
                    arena_next += total;
                    if (fresh || want1) {
                        if (mine + 1 >= pcap) {
                            ovf |= 8u;
                            ok = 0;
                            if (SEG) {
                                segs[(long long)lf * P.maxContours + slot].n = SEG_INVALID;
                            } else {
                                fco_of(lf)[slot] = make_uint4(st.x, st.y, 0u, (unsigned)key);
                                if (MODE == 2) pend[(long long)lf * P.maxContours + slot].p = 0u;
                            }
                            state = ST_IDLE;
                        } else if (fresh) {
                            chunkA = mine;
                            chunkB = mine + 1;
                            tab_at(lf, slot, 0) = chunkA;
                            tab_at(lf, slot, 1) = chunkB;
                        } else {
                            kreg++;
                            if (kreg & 1) chunkB = mine;
                            else chunkA = mine;
                            tab_at(lf, slot, (unsigned)kreg) = mine;
                        }
                    }
                }
            }
#ifdef FID_DEBUG_STATS
            FID_DSEG(3)  // chunks
#endif
            // ---- window refills
            if (state == ST_NEED) {
                // padded coordinates of the 3x3 neighbourhood: bits xb .. xb+2, rows cy .. cy+2; the window starts on a
                // multiple of 4 rows (a 16-byte quarter of a mask tile) and lies ahead of the direction of travel
                const int xb = cx - 1 + MASK_PADW * 32;
                int tx = ndx >= 0 ? (xb >> 5) : ((xb + 2) >> 5) - 1;
                int wy = ndy >= 0 ? (cy & ~3) : ((cy + 2) & ~3) - 12;
                tx = tx < 0 ? 0 : (tx > TC - 2 ? TC - 2 : tx);
                wy = wy < 0 ? 0 : (wy > TR * MT_ROWS - 16 ? TR * MT_ROWS - 16 : wy);
                wx0 = tx * 32;
                wy0 = wy;
                static_for<4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int row = wy + 4 * q;
                    const uint32_t *g = pl + ((long long)(row >> 4) * TC + tx) * MT_ROWS + (row & 15);
                    // (the instruction offset moves the LDS destination as well as the source: compensate in the base)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)((char *)s_win + q * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)((char *)s_win + (4 + q) * 1024 - MT_ROWS * 4),
                                                     16, MT_ROWS * 4, 0);
                });
                state = ST_LOADING;
            }
#ifdef FID_DEBUG_STATS
            FID_DSEG(4)  // refill issue
            d_ckcyc += __builtin_readcyclecounter() - d_c0;
#endif
            if (all_done && ballot64(state != ST_IDLE) == 0) break;  // every queue empty, everybody retired
            // ================= up to WALK_CKPT border-following steps inside the windows =================
            // (the next checkpoint comes after WALK_CKPT steps if some lane is waiting for one -- parked, loading, finished,
            //  or idle with survivors still queued -- and after WALK_RUN steps at the latest: chunk hand-out and the
            //  perimeter cap rely on that bound)
            const int work_left = !all_done;
            for (int it = 0; it < WALK_RUN; it++) {
                const unsigned long long act = ballot64(state == ST_ACTIVE);
                if (it >= WALK_CKPT && ballot64(state != ST_ACTIVE && (state != ST_IDLE || work_left))) break;
                if (act == 0) {
#ifdef FID_DEBUG_STATS
                    d_forced += it == 0;
#endif
                    break;
                }
#ifdef FID_DEBUG_STATS
                d_iters++;
                d_active += __popcll(act);
#endif
                if (state == ST_ACTIVE) {
                    const unsigned raw = win_raw(s_winw, lane4, cx, cy, wx0, wy0);
                    const unsigned e = s_lut[raw | ((unsigned)sdir << 8)];
                    const int sn = e & 7, code = (e >> 3) & 7;
                    const int hmag = (code & 1) ? W2 : 1;
                    const int hoff = (code & 2) ? hmag : -hmag;
                    if (SEG) {
                        if (count > 0 && (e & 0x40u) && seed_state(cx, cy, sdir, sgm)) {
                            closed = 1;  // the next seed state: the segment ends in front of it
                            state = ST_FINAL;
                        } else if (count > 0 && cx == brx && cy == bry && sdir == brd) {
                            // Brent's cycle test: a seed that is not a real border state has walked into a border without
                            // seeds and is going round it; nobody links to such a seed
                            ok = 0;
                            state = ST_FINAL;
                        } else {
                            if ((count & (count - 1)) == 0) {  // remember the state at every power of two
                                brx = cx;
                                bry = cy;
                                brd = sdir;
                            }
                            const unsigned hv = code ? (unsigned)(pc + hoff) : 0xffffffffu;
                            mhole = hv < mhole ? hv : mhole;
                            mout = (unsigned)pc < mout ? (unsigned)pc : mout;
                            acc = __builtin_amdgcn_alignbit((uint32_t)sn, acc, 4);
                            if ((count & 7) == 7) fpool[((count & CK ? chunkB : chunkA) * CKW) + (unsigned)((count & (CK - 1)) >> 3)] = acc;
                            count++;
                            const int dx = dir_dx(sn), dy = dir_dy(sn);
                            cx += dx;
                            cy += dy;
                            pc += __mul24(dy, W2) + dx;
                            sdir = (sn + 4) & 7;
                            ndx = dx;
                            ndy = dy;
                            const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - wx0), rr = (unsigned)(cy - wy0);
                            state = (xr > 61u || rr > 13u) ? ST_NEED : ST_ACTIVE;
                        }
                    } else if (MODE == 2 && count > 0 && (e & 0x40u) && seed_state(cx, cy, sdir, sgm)) {
                        stopped = 1;  // the first seed state on this border: the segment chain continues from here
                        state = ST_FINAL;
                    } else {
                        // background pixels examined in the 4-directions belong to this border's hole region
                        int bad = hole && code && (pc + hoff < key);
                        acc = __builtin_amdgcn_alignbit((uint32_t)sn, acc, 4);
                        if ((count & 7) == 7) fpool[((count & CK ? chunkB : chunkA) * CKW) + (unsigned)((count & (CK - 1)) >> 3)] = acc;
                        count++;
                        const int dx = dir_dx(sn), dy = dir_dy(sn);
                        const int nx = cx + dx, ny = cy + dy;
                        const int cl = !bad && nx == x0 && ny == y0 && cx == i1x && cy == i1y;
                        cx = nx;
                        cy = ny;
                        pc += __mul24(dy, W2) + dx;
                        bad |= !cl && !hole && pc < key;
                        sdir = (sn + 4) & 7;
                        ndx = dx;
                        ndy = dy;
                        // still inside the window?  bits of x-1..x+1 in [0, 64), padded rows cy..cy+2 in [0, 16)
                        const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - wx0), rr = (unsigned)(cy - wy0);
                        const int outside = xr > 61u || rr > 13u;
                        closed = cl;
                        ok = !bad;
                        state = (bad || cl) ? ST_FINAL : outside ? ST_NEED : ST_ACTIVE;
                    }
                }
            }
        }
        if (ovf) atomicOr(&G->overflow, ovf);
#ifdef FID_DEBUG_STATS
        if (lane == 0 && (FID_DEBUG_STATS + 0 < 0 || MODE == FID_DEBUG_STATS + 0)) {  // -DFID_DEBUG_STATS=<mode>: that walk only; -1: all
            atomicAdd(&G->dbg[0], d_iters);
            atomicAdd(&G->dbg[1], d_ckpts);
            atomicAdd(&G->dbg[2], d_active);
            atomicAdd(&G->dbg[3], d_ckcyc);
            atomicAdd(&G->dbg[4], d_waitcyc);
            atomicAdd(&G->dbg[5], d_forced);
            atomicAdd(&G->dbg[6], (unsigned long long)(__builtin_readcyclecounter() - d_t0));
            atomicAdd(&G->dbg[7], 1ull);
            for (int k = 0; k < 5; k++) atomicAdd(&G->dbg[16 + k], d_seg[k]);
            for (int k = 0; k < 4; k++) atomicAdd(&G->dbg[21 + k], d_lanes[k]);
        }
#endif
    }
#ifdef FID_DEBUG_STATS
    if (lane == 0 && (FID_DEBUG_STATS + 0 < 0 || MODE == FID_DEBUG_STATS + 0)) {
        const unsigned long long d_kt = __builtin_readcyclecounter() - d_k0;
        atomicAdd(&G->dbg[8], d_kt);
        atomicMax(&G->dbg[9], d_kt);
        atomicAdd(&G->dbg[10], d_kexh);
        atomicMax(&G->dbg[11], d_kexh);
        atomicAdd(&G->dbg[12], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Seed-accelerated tracing (k_find_starts<true> + k_walk_full<1>, <2> + the three passes below).
// A border-following state is (pixel, direction back to the previous pixel); one step maps a state to the next one and does
// not depend on how the walk was started, so a border is a cycle of states.  SEED states cut every cycle that contains some
// into segments: seed i owns the states from its start state up to, not including, the next seed state.  The whole-border
// walk of a probe survivor then only has to reach the first seed state on its border (it keeps applying the canonical
// test on the way and closes by itself on a border without seeds); the rest comes from the segment records:
//   k_seg_link     next seed state -> seed index (SeedHash) for every segment and every stopped survivor
//   k_seg_chain    every stopped survivor: once around the seed cycle, summing lengths and taking the two running minima;
//                  the acceptance test is the one the whole-border walk applies (no pixel -- outer -- or examined
//                  background 4-neighbour -- hole -- with a raster index below the start's key; perimeter gate)
//   k_seg_copy     the pieces of the accepted contours (the survivor's own points, then the segments in cycle order, the
//                  last one cut where the survivor started; listed by k_seg_chain) into a dense array for k_approx
// The longest sequential piece is the longest seed-free stretch of a border, not the longest border.

// the frame's seed list -> SeedHash, one thread per seed (the inserts wait for their compare-and-swap: a kernel of its own
// hides that behind a few thousand of them in flight; inside k_find_starts one lane inserting the seeds of a grid row one
// after the other held its whole wave up)
__global__ __launch_bounds__(256) void k_seed_index(const uint2 *__restrict__ seedq, unsigned long long *__restrict__ seedhash,
                                                     const DevCounts *__restrict__ counts, const DevParams P)
{
    const int f = blockIdx.y;
    unsigned ns = (unsigned)counts[f].nseeds;
    ns = ns < (unsigned)P.maxContours ? ns : (unsigned)P.maxContours;
    const uint2 *fsq = seedq + (long long)f * P.maxContours;
    unsigned long long *fsh = seedhash + (long long)f * P.seedHashCap;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const uint2 r = fsq[i];
        seedhash_insert(fsh, P.seedHashCap, P.seedGen, seedhash_key((r.x & 0x3ffffffu) | (r.y << 26), (int)(r.x >> 27)), i);
    }
}

__global__ __launch_bounds__(256) void k_seg_link(const uint2 *__restrict__ seedq, DevSeg *__restrict__ segs,
                                                   const uint2 *__restrict__ surv, DevPend *__restrict__ pend,
                                                   const unsigned long long *__restrict__ seedhash, DevCounts *__restrict__ counts,
                                                   DevGlobal *__restrict__ G, const DevParams P)
{
    const int f = blockIdx.y;
    unsigned ns = (unsigned)counts[f].nseeds, nv = (unsigned)counts[f].nsurv;
    ns = ns < (unsigned)P.maxContours ? ns : (unsigned)P.maxContours;
    nv = nv < (unsigned)P.maxContours ? nv : (unsigned)P.maxContours;
    const unsigned long long *fsh = seedhash + (long long)f * P.seedHashCap;
    const uint2 *fsq = seedq + (long long)f * P.maxContours;
    DevSeg *fsg = segs + (long long)f * P.maxContours;
    const uint2 *fsv = surv + (long long)f * P.maxStarts;
    DevPend *fpd = pend + (long long)f * P.maxContours;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < ns + nv; i += gridDim.x * blockDim.x) {
        if (i < ns) {
            const unsigned sc = fsq[i].x >> 27;
            // (a segment that was abandoned -- too long, pool exhausted -- has no next seed)
            const unsigned nx = fsg[i].n == SEG_INVALID ? SEG_INVALID
                                                        : seedhash_find(fsh, P.seedHashCap, P.seedGen, seedhash_key(fsg[i].next_key, (int)sc));
            fsg[i].next_idx = nx < ns ? nx : SEG_INVALID;  // (a seed beyond the table's capacity was reported by k_find_starts)
        } else {
            const unsigned j = i - ns;
            if (fpd[j].p) {
                const unsigned sc = (fsv[j].y >> 16) & 0xffu;
                const unsigned nx = seedhash_find(fsh, P.seedHashCap, P.seedGen, seedhash_key(fpd[j].next_key, (int)sc));
                fpd[j].next_idx = nx < ns ? nx : SEG_INVALID;
            }
        }
    }
}

// stopped survivors: once around the seed cycle.  Accepted contours (also the ones k_walk_full<2> closed by itself, which
// it appended already) end up in `contours` (start, meta, length, key) with {survivor index, own points, first seed}.
// An accepted contour gets its place in the dense point array here (cbase) and goes round its cycle a second time to leave
// one COPY RECORD per piece {chunk row, place in the dense array, points}: the survivor's own points, then the segments in
// cycle order, the last one cut where the survivor started.  k_seg_copy moves the pieces, all of them in parallel (walking
// the chain inside the copy kernel made one wave wait for two dependent loads per segment).
__global__ __launch_bounds__(64) void k_seg_chain(const uint2 *__restrict__ surv, const DevPend *__restrict__ pend,
                                                   const uint4 *__restrict__ wres, const DevSeg *__restrict__ segs,
                                                   uint4 *__restrict__ contours,
                                                   uint4 *__restrict__ cinfo, uint32_t *__restrict__ cbase, uint4 *__restrict__ recs,
                                                   DevCounts *__restrict__ counts, DevGlobal *__restrict__ G, const DevParams P)
{
    const int f = blockIdx.y;
    const int lane = lane_id();
    unsigned nv = (unsigned)counts[f].nsurv;
    nv = nv < (unsigned)P.maxContours ? nv : (unsigned)P.maxContours;
    const uint2 *fsv = surv + (long long)f * P.maxStarts;
    const DevPend *fpd = pend + (long long)f * P.maxContours;
    const DevSeg *fsg = segs + (long long)f * P.maxContours;
    uint4 *fco = contours + (long long)f * P.maxContours;
    uint4 *fci = cinfo + (long long)f * P.maxContours;
    uint32_t *fcb = cbase + (long long)f * P.maxContours;
    uint4 *frc = recs + (long long)f * 2 * P.maxContours;
    const unsigned dcap = (unsigned)P.maxChunks * CK, rcap = 2u * (unsigned)P.maxContours;
    const int W = P.W;
    for (unsigned i0 = blockIdx.x * 64; i0 < nv; i0 += gridDim.x * 64) {
        const unsigned i = i0 + lane;
        int accept = 0;
        unsigned L = 0, key = 0, first = SEG_INVALID, own = 0, hops = 0;
        uint2 st = make_uint2(0u, 0u);
        if (i < nv) {
            const DevPend pd = fpd[i];
            if (!pd.p) {
                // decided by the walker itself (a border without seeds): accepted iff it left a length
                const uint4 w = wres[(long long)f * P.maxContours + i];
                if (w.z) {
                    st = make_uint2(w.x, w.y);
                    L = own = w.z;
                    key = w.w;
                    accept = 1;
                }
            } else {
                st = fsv[i];
                const int x0 = st.x & 0xffff, y0 = st.x >> 16;
                const int hole = (st.y >> 24) & 1;
                key = (unsigned)(hole ? pidx(x0 + 1, y0, W) : pidx(x0, y0, W));
                first = pd.next_idx;
                own = pd.p;
                unsigned cur = first;
                for (int h = 0; h <= P.maxPerim && cur != SEG_INVALID; h++) {  // (every segment has >= 1 state)
                    const DevSeg r = fsg[cur];
                    if (r.n == SEG_INVALID || r.n == 0u) break;
                    if ((hole ? r.mhole : r.mout) < key) break;  // a smaller key on the border: not the canonical start
                    L += r.n;
                    hops++;
                    if (L > (unsigned)P.maxPerim) break;
                    if (r.next_idx == first) {
                        accept = L >= (unsigned)P.minPerim;
                        break;
                    }
                    cur = r.next_idx;
                }
                if (first == SEG_INVALID) atomicOr(&G->overflow, 16u);  // a seed state without a seed: must not happen
            }
        }
        const unsigned long long mk = ballot64(accept);
        if (mk) {
            // contour slots, dense points and copy records of the wave's accepted contours: one atomic each
            // (a contour's codes start on a multiple of 8 bytes: k_approx reads them eight to a lane)
            const unsigned myL = accept ? (L + 7u) & ~7u : 0u, myR = accept ? hops + 1u : 0u;
            const unsigned sL = wave_iscan_dpp(myL), sR = wave_iscan_dpp(myR);
            const unsigned totL = (unsigned)__builtin_amdgcn_readlane((int)sL, 63), totR = (unsigned)__builtin_amdgcn_readlane((int)sR, 63);
            unsigned bslot = 0, bdense = 0, brec = 0;
            if (lane == 63) {
                bslot = atomicAdd((unsigned *)&counts[f].ncontours, (unsigned)__popcll(mk));
                bdense = atomicAdd((unsigned *)&counts[f].ndense, totL);
                brec = atomicAdd((unsigned *)&counts[f].nrec, totR);
            }
            bslot = (unsigned)__builtin_amdgcn_readlane((int)bslot, 63);
            bdense = (unsigned)__builtin_amdgcn_readlane((int)bdense, 63);
            brec = (unsigned)__builtin_amdgcn_readlane((int)brec, 63);
            const unsigned idx = bslot + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
            if (accept) {
                if (idx < (unsigned)P.maxContours) {
                    fco[idx] = make_uint4(st.x, st.y, L, key);
                    fci[idx] = make_uint4(i, own, first, 0u);
                    const unsigned dst0 = bdense + sL - myL;
                    unsigned rec = brec + sR - myR;
                    if (dst0 + L > dcap || rec + myR > rcap) {
                        atomicOr(&G->overflow, 8u);
                        fcb[idx] = SEG_INVALID;
                    } else {
                        fcb[idx] = dst0;
                        frc[rec++] = make_uint4((unsigned)P.maxContours + i, dst0, own < L ? own : L, REC_NO_CHUNK);  // the survivor's own points
                        unsigned off = own, cur = first;
                        while (off < L && cur != SEG_INVALID) {
                            const DevSeg r = fsg[cur];
                            if (r.n == 0u || r.n == SEG_INVALID) break;  // (never for a segment of an accepted cycle)
                            const unsigned take = r.n < L - off ? r.n : L - off;
                            frc[rec++] = make_uint4(cur, dst0 + off, take, REC_NO_CHUNK);
                            off += take;
                            cur = r.next_idx;
                        }
                        for (; rec < brec + sR; rec++) frc[rec] = make_uint4(0u, 0u, 0u, 0u);  // (records reserved, not needed)
                    }
                } else {
                    atomicOr(&G->overflow, 2u);
                }
            }
        }
    }
}

// the pieces of the accepted contours, from their pool chunks into the dense array: one wave per copy record
// {chunk row, place in the dense array, points | first point of the piece inside its row << 16, the row's first chunk or REC_NO_CHUNK}
__global__ __launch_bounds__(256) void k_seg_copy(const uint4 *__restrict__ recs, const uint32_t *__restrict__ chunk_tab,
                                                   const uint32_t *__restrict__ pool, uint32_t *__restrict__ dense,
                                                   const DevCounts *__restrict__ counts, const DevParams P, int part)
{
    // part -1: the frame's whole record array (trace mode 1); 0 / 1: its first / second half (trace mode 2, lists A / B)
    const int f = blockIdx.y;
    const int lane = lane_id();
    unsigned nr = (unsigned)(part == 1 ? counts[f].nrec2 : counts[f].nrec);
    const unsigned rcap = (part < 0 ? 2u : 1u) * (unsigned)P.maxContours;
    nr = nr < rcap ? nr : rcap;
    const uint4 *frc = recs + (long long)f * 2 * P.maxContours + (part == 1 ? (unsigned)P.maxContours : 0u);
    const uint32_t *fpool = pool;  // chunks are numbered across the whole launch
    uint8_t *fd = reinterpret_cast<uint8_t *>(dense) + (long long)f * P.maxChunks * CK;  // (one byte per point: its chain code)
    // (the record is the wave's, not the lane's: its index through readfirstlane, so that the record and everything derived from it
    //  -- length, first point, chunk, table base -- live in scalar registers and the scalar unit does that arithmetic)
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long tstride = 2ll * P.maxContours;
    for (unsigned ri = blockIdx.x * 4 + wv; ri < nr; ri += gridDim.x * 4) {
        const uint4 r = frc[ri];
        const unsigned np = r.z & 0xffffu, p0 = r.z >> 16;
        uint8_t *out = fd + r.y;
        if (r.w != REC_NO_CHUNK && p0 + np <= 2 * CK) {
            // a segment's record names its first two chunks (consecutive in the pool): no trip to the table for its first 128 points
            const uint32_t *src = fpool + (long long)r.w * CKW;
            for (unsigned k = lane; k < np; k += 64) {
                const unsigned k2 = k + p0;
                out[k] = (uint8_t)((src[k2 >> 3] >> ((k2 & 7) * 4)) & 7u);
            }
        } else {
            // (a segment longer than that: its table row holds the chunks from the third on; survivors' rows hold all of theirs)
            const uint32_t *trow = chunk_tab + chunk_tab_at(P, f, r.x, 0);
            const bool direct = r.w != REC_NO_CHUNK;
            for (unsigned k = lane; k < np; k += 64) {
                const unsigned k2 = k + p0;
                const uint32_t id = direct && k2 < 2 * CK ? r.w + (k2 >> 6) : trow[(long long)(k2 >> 6) * tstride];
                out[k] = (uint8_t)((fpool[(long long)id * CKW + ((k2 & 63) >> 3)] >> ((k2 & 7) * 4)) & 7u);
            }
        }
    }
}

// K3 probe passes, table-driven (trace mode 2).  Same decisions as k_probe<…, STOPSEED = true> -- a start survives only while it
// can still be the canonical start of a border that has NO seed state and can pass the perimeter gate -- but a step is what a
// walker's step is: the raw 3 x 3 neighbourhood byte (three mask rows, one alignbit each) and ONE table look-up per cursor
// (forward: next direction, the smallest examined background 4-neighbour, seed-state flag; backward: previous direction, ditto)
// instead of nb8()'s bit shuffles, a find-first-set on the rotated neighbourhood and a loop over the examined directions.
// (SQ counters, round 3: the two arithmetic probe passes were 3.1 M of the pipeline's 17.9 M VALU wave-instructions per frame.)
__device__ __forceinline__ unsigned raw8(const MaskView &m, int x, int y)
{
    const int xb = x - 1 + MASK_PADW * 32;
    const int wi = xb >> 5, sh = xb & 31;
    const uint32_t *p0 = m.base + mask_word(m.TC, y, wi);
    const uint32_t *p1 = m.base + mask_word(m.TC, y + 1, wi);
    const uint32_t *p2 = m.base + mask_word(m.TC, y + 2, wi);
    const uint32_t a0 = p0[0], a1 = p1[0], a2 = p2[0];
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    if (sh > 29) {
        b0 = p0[MT_ROWS];
        b1 = p1[MT_ROWS];
        b2 = p2[MT_ROWS];
    }
    const unsigned tu = __builtin_amdgcn_alignbit(b0, a0, sh), tm = __builtin_amdgcn_alignbit(b1, a1, sh), td = __builtin_amdgcn_alignbit(b2, a2, sh);
    return (tu & 7u) | ((tm & 1u) << 3) | ((tm & 4u) << 2) | ((td & 7u) << 5);
}

// backward step table: index raw | bf << 8 (bf = direction from the cursor's pixel to its successor) -> direction to the
// predecessor (first foreground clockwise from bf - 1) | code << 3 (the smallest-offset background 4-neighbour that search
// passed over: 0 none, else 4 | positive << 1 | whole-row, as in the forward table) | seed-state flag of the state (pixel, that
// direction) << 6 (to be combined with the grid-line test)
__device__ __forceinline__ void build_back_lut(uint8_t *lut, int tid, int nthreads)
{
    for (int e = tid; e < 2048; e += nthreads) {
        const unsigned raw = (unsigned)e & 0xffu;
        const unsigned nb = raw_to_nb(raw);
        const int bf = e >> 8;
        const int c0 = (bf - 1) & 7;
        const unsigned win = (((nb | (nb << 8)) >> (c0 + 1)) & 0xffu);
        const int tz = win ? 7 - (31 - __clz((int)win)) : 0;
        unsigned seen = 0;
        for (int q = 0; q < tz; q++) seen |= 1u << ((c0 - q) & 7);
        const int code = (seen & 4u) ? 5 : (seen & 16u) ? 4 : (seen & 1u) ? 6 : (seen & 64u) ? 7 : 0;
        const int bd = (c0 - tz) & 7;
        const int seed = nb && !((nb >> seed_empty_dir(bd)) & 1u);
        lut[e] = (uint8_t)(bd | (code << 3) | (seed << 6));
    }
}

// the two step tables of k_probe_lut, built ONCE per context into global memory (forward table, then backward table: 4 KB).
// Until round 6 every workgroup of k_probe_lut built them itself: 16 entries a thread at 60 - 80 instructions each (the "seen"
// loop diverges), i.e. ~1 100 wave-instructions per wave in front of a main loop that handles two starts per thread in ~400 --
// the table build was the larger part of both probe kernels (measured: profiles/sq_cycles.json before / after).
__global__ __launch_bounds__(256) void k_probe_tables(uint8_t *__restrict__ tables)
{
    build_step_lut(tables, threadIdx.x, 256);
    build_back_lut(tables + 2048, threadIdx.x, 256);
}

template <int STEPS, int LEVEL>
__global__ __launch_bounds__(256) void k_probe_lut(const uint32_t *__restrict__ masks, const uint2 *__restrict__ in_list,
                                                    uint2 *__restrict__ out_list, DevCounts *__restrict__ counts,
                                                    DevGlobal *__restrict__ G, const DevParams P, const uint4 *__restrict__ tables)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_fwd[2048], s_bwd[2048];
    // (one 16-byte load and store per thread: the tables come out of L2)
    {
        const uint4 v = tables[threadIdx.x];
        if (threadIdx.x < 128) reinterpret_cast<uint4 *>(s_fwd)[threadIdx.x] = v;
        else reinterpret_cast<uint4 *>(s_bwd)[threadIdx.x - 128] = v;
    }
    __syncthreads();
    const int f = blockIdx.y;
    const int lane = lane_id();
    unsigned n = (unsigned)(LEVEL != 1 ? counts[f].nstarts : counts[f].nsurv1);
    n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
    int *out_count = LEVEL == 0 ? &counts[f].nsurv1 : &counts[f].nsurv;
    const int W = P.W, S = P.nscales, W2 = P.W + 2;
    const int sgm = (8 << P.seedShift) - 1;
    const long long plane = (long long)P.TR * P.TC * MT_ROWS;
    const uint2 *fin = in_list + (long long)f * P.maxStarts;
    uint2 *fout = out_list + (long long)f * P.maxStarts;
    for (unsigned i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + lane;
        const bool active = i < n;
        const uint2 st = active ? fin[i] : make_uint2(0u, 0u);
        const int x0 = st.x & 0xffff, y0 = st.x >> 16;
        const int s = (st.y >> 16) & 0xff, hole = (st.y >> 24) & 1;
        MaskView m;
        m.base = masks + ((long long)f * S + s) * plane;
        m.TC = P.TC;
        const int key = hole ? pidx(x0 + 1, y0, W) : pidx(x0, y0, W);
        int count = 0, ok = active, closed = 0;
        unsigned raw = ok ? raw8(m, x0, y0) : 0u;
        if (ok && raw == 0) {
            count = 1;  // single pixel domain
            closed = 1;
        } else if (ok) {
            const unsigned nb0 = raw_to_nb(raw);
            int sdir = first_dir(nb0, hole ? 0 : 4);
            const int i1x = x0 + dir_dx(sdir), i1y = y0 + dir_dy(sdir);
            int bx = i1x, by = i1y, bf = (sdir + 4) & 7, pb = pidx(bx, by, W);
            if (!hole && pb < key) ok = 0;
            if (seed_state(x0, y0, sdir, sgm) && !((nb0 >> seed_empty_dir(sdir)) & 1u)) ok = 0;  // the start state itself is a seed state
            int cx = x0, cy = y0, pc = pidx(x0, y0, W);
            while (ok) {
                // ---- forward step
                const unsigned e = s_fwd[raw | ((unsigned)sdir << 8)];
                const int sn = e & 7, code = (e >> 3) & 7;
                if (hole && code) {
                    const int hmag = (code & 1) ? W2 : 1;
                    if (pc + ((code & 2) ? hmag : -hmag) < key) ok = 0;  // an examined background 4-neighbour in front of the key
                }
                count++;
                const int dx = dir_dx(sn), dy = dir_dy(sn);
                const int nx = cx + dx, ny = cy + dy;
                if (!ok || count > P.maxPerim) {
                    ok = 0;
                    break;
                }
                if (nx == x0 && ny == y0 && cx == i1x && cy == i1y) {
                    closed = 1;
                    break;
                }
                if (count >= STEPS) break;
                cx = nx;
                cy = ny;
                pc += __mul24(dy, W2) + dx;
                if (!hole && pc < key) {
                    ok = 0;
                    break;
                }
                sdir = sn ^ 4;
                raw = raw8(m, cx, cy);
                if ((s_fwd[raw | ((unsigned)sdir << 8)] & 0x40u) && seed_state(cx, cy, sdir, sgm)) {  // a seed state: the border is a seed cycle
                    ok = 0;
                    break;
                }
                // ---- backward step
                {
                    const unsigned braw = raw8(m, bx, by);
                    const unsigned be = s_bwd[braw | ((unsigned)bf << 8)];
                    const int bd = be & 7, bcode = (be >> 3) & 7;
                    if (hole && bcode) {
                        const int hmag = (bcode & 1) ? W2 : 1;
                        if (pb + ((bcode & 2) ? hmag : -hmag) < key) ok = 0;
                    }
                    if ((be & 0x40u) && seed_state(bx, by, bd, sgm)) ok = 0;  // the state (pixel, back direction) the cursor stood in
                    const int bdx = dir_dx(bd), bdy = dir_dy(bd);
                    bx += bdx;
                    by += bdy;
                    pb += __mul24(bdy, W2) + bdx;
                    bf = bd ^ 4;
                    if (!hole && pb < key) ok = 0;
                }
            }
        }
        const int keep = ok && (!closed || (count >= P.minPerim && count <= P.maxPerim));
        const unsigned long long mk = ballot64(keep);
        if (mk) {
            const int leader = __ffsll((long long)mk) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd((unsigned *)out_count, (unsigned)__popcll(mk));
            base = __shfl(base, leader, WAVE);
            const unsigned idx = base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
            if (keep) {
                if (idx < (unsigned)P.maxStarts) fout[idx] = st;
                else atomicOr(&G->overflow, 1u);
            }
        }
    }
}

// The second probe pass with REFILLED lanes (round 6).  k_probe_lut<32, 1> gives every lane one start and the wave then steps
// until its last lane is done: the starts die after anything between 7 and 32 steps, and the wave issued all 32 (lane utilisation
// 0.45, 0.81 M VALU wave-instructions per frame).  Here a wave owns a contiguous share of the list and keeps its lanes busy: when
// PROBE_REFILL_MIN lanes are idle (or nobody is stepping) the finished ones hand in their verdict -- survivors collect in an LDS
// list that leaves with one atomic per 64 -- and take the next starts of the share.  Start-up and one step are the statements of
// k_probe_lut's loop, one step per trip; the decisions per start are the same, the order of the survivor list was never defined.
#ifndef PROBE_REFILL_MIN
#define PROBE_REFILL_MIN 16
#endif
template <int STEPS, int LEVEL>
__global__ __launch_bounds__(256) void k_probe_refill(const uint32_t *__restrict__ masks, const uint2 *__restrict__ in_list,
                                                       uint2 *__restrict__ out_list, DevCounts *__restrict__ counts,
                                                       DevGlobal *__restrict__ G, const DevParams P, const uint4 *__restrict__ tables)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_fwd[2048], s_bwd[2048];
    __shared__ uint2 s_out[4][128];
    {
        const uint4 v = tables[threadIdx.x];
        if (threadIdx.x < 128) reinterpret_cast<uint4 *>(s_fwd)[threadIdx.x] = v;
        else reinterpret_cast<uint4 *>(s_bwd)[threadIdx.x - 128] = v;
    }
    __syncthreads();
    const int f = blockIdx.y;
    const int lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned n = (unsigned)(LEVEL != 1 ? counts[f].nstarts : counts[f].nsurv1);
    n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
    int *out_count = LEVEL == 0 ? &counts[f].nsurv1 : &counts[f].nsurv;
    const int W = P.W, S = P.nscales, W2 = P.W + 2;
    const int sgm = (8 << P.seedShift) - 1;
    const long long plane = (long long)P.TR * P.TC * MT_ROWS;
    const uint2 *fin = in_list + (long long)f * P.maxStarts;
    uint2 *fout = out_list + (long long)f * P.maxStarts;
    uint2 *obuf = s_out[wv];
    // the wave's share of the list
    const unsigned nwv = gridDim.x * 4u, me = blockIdx.x * 4u + (unsigned)wv;
    unsigned next = (unsigned)(((unsigned long long)n * me) / nwv);
    const unsigned hi = (unsigned)(((unsigned long long)n * (me + 1u)) / nwv);
    int outn = 0;  // survivors waiting in obuf (wave-uniform)
    auto flush = [&](int cnt) {  // the first cnt entries of obuf leave
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned *)out_count, (unsigned)cnt);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < cnt) {
            if (base + (unsigned)lane < (unsigned)P.maxStarts) fout[base + lane] = obuf[lane];
            else atomicOr(&G->overflow, 1u);
        }
    };
    enum { PH_IDLE = 0, PH_RUN, PH_DONE };
    int phase = PH_IDLE;
    uint2 st = make_uint2(0u, 0u);
    MaskView m;
    m.base = masks;
    m.TC = P.TC;
    int x0 = 0, y0 = 0, hole = 0, key = 0, count = 0, ok = 0, closed = 0;
    unsigned raw = 0;
    int sdir = 0, i1x = 0, i1y = 0, bx = 0, by = 0, bf = 0, pb = 0, cx = 0, cy = 0, pc = 0;
    for (;;) {
        const unsigned long long running = ballot64(phase == PH_RUN);
        if (running == 0 || 64 - __popcll(running) >= PROBE_REFILL_MIN) {
            // ---- verdicts of the finished lanes
            const int keep = phase == PH_DONE && ok && (!closed || (count >= P.minPerim && count <= P.maxPerim));
            const unsigned long long mk = ballot64(keep);
            if (mk) {
                if (keep) obuf[outn + __popcll(mk & ((1ull << lane) - 1ull))] = st;
                outn += __popcll(mk);
                if (outn >= 64) {
                    flush(64);
                    outn -= 64;
                    if (lane < outn) {
                        const uint2 t = obuf[64 + lane];
                        obuf[lane] = t;
                    }
                }
            }
            if (phase == PH_DONE) phase = PH_IDLE;
            // ---- the next starts of the share
            const unsigned long long idle = ballot64(phase == PH_IDLE);
            if (next < hi) {
                const unsigned mine = next + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                if (phase == PH_IDLE && mine < hi) {
                    st = fin[mine];
                    x0 = st.x & 0xffff;
                    y0 = st.x >> 16;
                    const int s = (st.y >> 16) & 0xff;
                    hole = (st.y >> 24) & 1;
                    m.base = masks + ((long long)f * S + s) * plane;
                    key = hole ? pidx(x0 + 1, y0, W) : pidx(x0, y0, W);
                    count = 0;
                    ok = 1;
                    closed = 0;
                    raw = raw8(m, x0, y0);
                    phase = PH_RUN;
                    if (raw == 0) {
                        count = 1;  // single pixel domain
                        closed = 1;
                        phase = PH_DONE;
                    } else {
                        const unsigned nb0 = raw_to_nb(raw);
                        sdir = first_dir(nb0, hole ? 0 : 4);
                        i1x = x0 + dir_dx(sdir);
                        i1y = y0 + dir_dy(sdir);
                        bx = i1x;
                        by = i1y;
                        bf = (sdir + 4) & 7;
                        pb = pidx(bx, by, W);
                        if (!hole && pb < key) ok = 0;
                        if (seed_state(x0, y0, sdir, sgm) && !((nb0 >> seed_empty_dir(sdir)) & 1u)) ok = 0;  // the start state itself is a seed state
                        cx = x0;
                        cy = y0;
                        pc = pidx(x0, y0, W);
                        if (!ok) phase = PH_DONE;
                    }
                }
                const unsigned nid = (unsigned)__popcll(idle);
                next = next + nid < hi ? next + nid : hi;
            } else if (running == 0 && ballot64(phase != PH_IDLE) == 0) {
                break;  // the share is handed out, nobody steps, every verdict is in
            }
        }
        if (phase == PH_RUN) {
            // ---- one step of k_probe_lut's loop; whatever leaves that loop ends the start here
            bool done = false;
            const unsigned e = s_fwd[raw | ((unsigned)sdir << 8)];
            const int sn = e & 7, code = (e >> 3) & 7;
            if (hole && code) {
                const int hmag = (code & 1) ? W2 : 1;
                if (pc + ((code & 2) ? hmag : -hmag) < key) ok = 0;  // an examined background 4-neighbour in front of the key
            }
            count++;
            const int dx = dir_dx(sn), dy = dir_dy(sn);
            const int nx = cx + dx, ny = cy + dy;
            if (!ok || count > P.maxPerim) {
                ok = 0;
                done = true;
            } else if (nx == x0 && ny == y0 && cx == i1x && cy == i1y) {
                closed = 1;
                done = true;
            } else if (count >= STEPS) {
                done = true;
            } else {
                cx = nx;
                cy = ny;
                pc += __mul24(dy, W2) + dx;
                if (!hole && pc < key) {
                    ok = 0;
                    done = true;
                } else {
                    sdir = sn ^ 4;
                    raw = raw8(m, cx, cy);
                    if ((s_fwd[raw | ((unsigned)sdir << 8)] & 0x40u) && seed_state(cx, cy, sdir, sgm)) {  // a seed state: the border is a seed cycle
                        ok = 0;
                        done = true;
                    } else {
                        // ---- backward step
                        const unsigned braw = raw8(m, bx, by);
                        const unsigned be = s_bwd[braw | ((unsigned)bf << 8)];
                        const int bd = be & 7, bcode = (be >> 3) & 7;
                        if (hole && bcode) {
                            const int hmag = (bcode & 1) ? W2 : 1;
                            if (pb + ((bcode & 2) ? hmag : -hmag) < key) ok = 0;
                        }
                        if ((be & 0x40u) && seed_state(bx, by, bd, sgm)) ok = 0;  // the state (pixel, back direction) the cursor stood in
                        const int bdx = dir_dx(bd), bdy = dir_dy(bd);
                        bx += bdx;
                        by += bdy;
                        pb += __mul24(bdy, W2) + bdx;
                        bf = bd ^ 4;
                        if (!hole && pb < key) ok = 0;
                        if (!ok) done = true;
                    }
                }
            }
            if (done) phase = PH_DONE;
        }
    }
    if (outn) flush(outn);
}

// ================================================================================================
// Trace mode 2: CYCLE TRACING.  Every border that has a seed state is a cycle of segments; its length, its kind (outer /
// hole), its canonical start and therefore the exact point order of cvFindContours all follow from the segment records
// (DevSegC, fid_device.h), so no probe survivor has to find it: the probe passes and the survivor walk only keep the borders
// WITHOUT a seed state (the ones that curl up inside one grid cell) and drop a start as soon as its walk meets a seed state.
//   k_seed_walk    every seed follows its segment (below)
//   k_seg_link2    next seed state -> seed index; marks the seeds some segment runs into
//   k_seg_cycles   one lane per segment: once around its cycle while one of its two start candidates can still be the
//                  cycle's smallest; the lane that holds the canonical start lists the pieces (copy records) -- plus the
//                  borders the survivor walk closed by itself
//   k_seg_copy     as before (records now carry a source offset: the first segment is cut at the start state)

// step table of the seed walker: index raw | backdir << 8 -> next direction | starts-an-outer-border << 3 |
// starts-a-hole-border << 4 | seed state on a grid column << 5 | seed state on a grid row << 6
__device__ __forceinline__ void build_step_lut2(uint8_t *lut, int tid, int nthreads)
{
    for (int e = tid; e < 2048; e += nthreads) {
        const unsigned raw = (unsigned)e & 0xffu;
        const unsigned nb = raw_to_nb(raw);
        const int sd = e >> 8;
        const int start = (sd + 1) & 7;
        const unsigned rot = ((nb | (nb << 8)) >> start) & 0xffu;
        const int t = rot ? __ffs(rot) - 1 : 0;
        // the state icvFetchContour begins a border in: (start pixel, direction of the first foreground neighbour found
        // clockwise from s_end - 1), s_end = 4 (W, background) for an outer border, 0 (E, background) for a hole border
        const int fo = nb && !((nb >> 4) & 1u) && sd == first_dir(nb, 4);
        const int fh = nb && !(nb & 1u) && sd == first_dir(nb, 0);
        const int seed = ((nb >> sd) & 1u) && !((nb >> seed_empty_dir(sd)) & 1u);
        const int scol = seed && ((SEED_DIRS_COL >> sd) & 1u), srow = seed && ((SEED_DIRS_ROW >> sd) & 1u);
        lut[e] = (uint8_t)(((start + t) & 7) | (fo << 3) | (fh << 4) | (scol << 5) | (srow << 6));
    }
}

// The seed walker.  As k_walk_full<1> (persistent waves, per-frame queues, LDS windows filled by LDS-DMA at batched
// checkpoints, points into pool chunks), with what the round-2 counters asked for:
//   * window = 2 x 2 mask tiles (64 px x 32 rows, 256 bytes per lane), TOROIDAL: tile (tr, tc) lives in slot (tr & 1, tc & 1), so
//     moving the window by one tile replaces two tiles and keeps two.  A walker that travels (net displacement since the last
//     checkpoint) gets the tiles ahead of it fetched while it keeps stepping in the tiles it has: a lane waits for memory only
//     at its first window and after a turn the prefetch did not foresee (round 2: 30 of 64 lanes sat in a refill at any time)
//   * start states of borders are recognised by the step table (two more bits), minima and their positions kept per segment
//   * the cycle test compares one packed state key; the segment record leaves as two 16-byte stores
#define SW_WAVES 2
#ifndef SW_VGPR_ATTR
#define SW_VGPR_ATTR
#endif
#ifndef SW_CKPT
#define SW_CKPT 8
#endif
#ifndef SW_RUN
#define SW_RUN 32
#endif
// SURV (round 5): the same machinery for the PROBE SURVIVORS -- what k_walk_full<2> does (a survivor starts a border the way
// icvFetchContour would, applies the canonical-start test as it goes, stops in front of the first seed state or closes the
// border by itself; results into the survivors' contour rows and DevPend) on the seed walker's toroidal prefetched windows
// instead of k_walk_full's reload-when-left windows (23 % of the lanes stepping there).  seedq = the survivor list, segs unused;
// wres / pend unused for seeds.
template <bool SURV>
__global__ __launch_bounds__(64 * SW_WAVES) SW_VGPR_ATTR void k_seed_walk(const uint32_t *__restrict__ masks, const uint2 *__restrict__ seedq,
                                                              uint32_t *__restrict__ chunk_tab, uint32_t *__restrict__ pool,
                                                              DevSegC *__restrict__ segs, DevCounts *__restrict__ counts,
                                                              DevGlobal *__restrict__ G, const DevParams P, uint4 *__restrict__ wres,
                                                              DevPend *__restrict__ pend_out)
{
    // chunk j = pcx * 8 + (row >> 2 & 7) of lane l (16 bytes: rows 4q..4q+3 of one word column) at s_win[j * 64 + l]
    __shared__ uint4 s_win_all[SW_WAVES][16 * 64];
    __shared__ uint8_t s_lut[2048];
    uint4 *s_win = s_win_all[threadIdx.x >> 6];
    const uint32_t *s_winw = reinterpret_cast<const uint32_t *>(s_win);
    const int lane = lane_id();
    const int lane4 = lane * 4;
    if (SURV) build_step_lut(s_lut, threadIdx.x, 64 * SW_WAVES);  // (next direction | hole-canonical code << 3 | seed flag << 6)
    else build_step_lut2(s_lut, threadIdx.x, 64 * SW_WAVES);
    __syncthreads();
    int f = blockIdx.y;
    const unsigned ccap = (unsigned)P.maxContours, pcap = (unsigned)P.maxChunks * (unsigned)P.nframes;
    const int S = P.nscales, TC = P.TC, TR = P.TR, F = P.nframes;
    const int W = P.W, W2 = P.W + 2;
    const int sgm = (8 << P.seedShift) - 1;
    const long long plane = (long long)TR * TC * MT_ROWS;
    enum { ST_IDLE = 0, ST_ACTIVE, ST_NEED, ST_LOADING, ST_FINAL };
#ifdef FID_DEBUG_STATS
    unsigned long long d_iters = 0, d_ckpts = 0, d_active = 0, d_lanes[4] = {0, 0, 0, 0}, d_ckcyc = 0;
    const unsigned long long d_t0 = __builtin_readcyclecounter();
#endif

    unsigned n = 0;
    const uint2 *fin = seedq;
    unsigned *qhead = nullptr;
    auto set_queue = [&](int fr) {
        n = (unsigned)(SURV ? counts[fr].nsurv : counts[fr].nseeds);
        if (SURV) n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
        if (n > ccap) {
            if (lane == 0) atomicOr(&G->overflow, 2u);
            n = ccap;
        }
        fin = seedq + (long long)fr * (SURV ? P.maxStarts : P.maxContours);
        qhead = (unsigned *)(SURV ? &counts[fr].nwalk : &counts[fr].nwalk2);
    };
    set_queue(f);
    int all_done = 0;
    unsigned next = 0, rend = 0, pre_base = 0;
    int exhausted = 0, pre_ready = 0;
    uint2 pre = make_uint2(0u, 0u);
    // per-lane walker
    int state = ST_IDLE;
    int lf = f;
    const uint32_t *pl = masks;
    unsigned slot = 0;
    int cx = 0, cy = 0, pc = 0, sdir = 0, count = 0, ok = 0, too_long = 0;
    int cxp = 0, cyp = 0;            // position at the last checkpoint (net travel decides what is fetched ahead)
    int br = 0, bc = 0;              // window base: mask tiles (br..br+1) x (bc..bc+1)
    int pbr = 0, pbc = 0, pend = 0;  // the base the loads in flight will give
    int vxb = 0, vyb = 0;            // usable part of the window: padded bit / row of its origin ...
    unsigned vxs = 0, vys = 0;       // ... and extent minus the 3 x 3 footprint
    unsigned chunkA = 0, chunkB = 0, chunk0 = 0;
    int kreg = 0;
    unsigned ovf = 0;
    unsigned ko = 0xffffffffu, kh = 0xffffffffu, po = 0, ph = 0;
    unsigned brkey = 0xffffffffu;
    uint32_t acc = 0;  // chain codes of the last (up to) eight points
    unsigned arena_next = 0, arena_end = 0;
    // SURV: the survivor record, its start pixel / kind / key, the pixel after the start, what the walk found
    uint2 sst = make_uint2(0u, 0u);
    int x0 = 0, y0 = 0, hole = 0, key = 0, i1x = 0, i1y = 0, first = 0, closed = 0, stopped = 0;
    // raw 3 x 3 neighbourhood byte of the walker's pixel out of its toroidal window
    auto raw_here = [&]() {
        const int xb = cx + (MASK_PADW * 32 - 1);
        const int sh = xb & 31;
        const bool odd = (xb >> 5) & 1;
        unsigned t3[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int r = cy + d;
            const int idx = ((r << 6) & 0x700) | (r & 3) | lane4;
            const uint32_t wa = s_winw[idx], wb = s_winw[idx + 2048];
            t3[d] = __builtin_amdgcn_alignbit(odd ? wa : wb, odd ? wb : wa, sh);
        }
        return (t3[0] & 7u) | ((t3[1] & 1u) << 3) | ((t3[1] & 4u) << 2) | ((t3[2] & 7u) << 5);
    };
    for (;;) {
        // ================= checkpoint =================
#ifdef FID_DEBUG_STATS
        const unsigned long long d_c0 = __builtin_readcyclecounter();
        d_ckpts++;
#endif
        wait_vmcnt0();
#ifdef FID_DEBUG_STATS
        d_lanes[0] += __popcll(ballot64(state == ST_NEED));
        d_lanes[1] += __popcll(ballot64(state == ST_LOADING));
        d_lanes[2] += __popcll(ballot64(state == ST_FINAL));
        d_lanes[3] += __popcll(ballot64(state == ST_IDLE));
#endif
        if (next < rend) pre_ready = 1;
        if (pend) {  // the tiles requested at the last checkpoint have landed: the whole 2 x 2 window is usable
            pend = 0;
            br = pbr;
            bc = pbc;
            vxb = bc * 32;
            vyb = br * MT_ROWS;
            vxs = 64 - 3;
            vys = 2 * MT_ROWS - 3;
            if (state == ST_LOADING || state == ST_NEED) {  // (NEED: it stepped off the shared tiles while the new ones were in flight)
                const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - vxb), rr = (unsigned)(cy - vyb);
                state = (xr > vxs || rr > vys) ? ST_NEED : ST_ACTIVE;
            }
        }
        if (SURV && first && state == ST_ACTIVE) {
            // the survivor's first look at its start pixel: single-pixel domain, initial direction
            //   do { s = (s - 1) & 7; } while (*i1 == 0 && s != s_end)  == first foreground clockwise from s_end - 1
            first = 0;
            const unsigned raw = raw_here();
            if (raw == 0) {
                count = 1;  // (the contour's only point is its start: no step, no code)
                closed = 1;
                state = ST_FINAL;
            } else {
                sdir = first_dir(raw_to_nb(raw), hole ? 0 : 4);
                i1x = x0 + dir_dx(sdir);
                i1y = y0 + dir_dy(sdir);
                if (!hole && pidx(i1x, i1y, W) < key) {
                    ok = 0;
                    state = ST_FINAL;
                }
            }
        }
        if ((state == ST_ACTIVE || state == ST_NEED) && count > P.maxPerim) {
            ok = 0;
            too_long = 1;
            state = ST_FINAL;
        }
        // ---- retire
        if (state == ST_FINAL) {
            const int rem = count & 7, b0 = count - rem;
            if (rem) pool[((b0 & CK ? chunkB : chunkA) * CKW) + (unsigned)((b0 & (CK - 1)) >> 3)] = acc >> (4 * (8 - rem));
            if (SURV) {
                // stopped in front of a seed state: k_seg_cycles decides; else decided here
                const int accept = ok && closed && !stopped && count >= P.minPerim && count <= P.maxPerim;
                wres[(long long)lf * P.maxContours + slot] = make_uint4(sst.x, sst.y, accept ? (unsigned)count : 0u, (unsigned)key);
                DevPend *pd = pend_out + (long long)lf * P.maxContours + slot;
                pd->p = ok && stopped ? (unsigned)count : 0u;
                pd->next_key = seed_key(cx, cy, sdir);
            } else {
                uint4 *r = reinterpret_cast<uint4 *>(segs + (long long)lf * P.maxContours + slot);
                r[0] = make_uint4(SEG_INVALID, too_long || !ok ? SEG_INVALID : (unsigned)count, ko, kh);  // next_idx (k_seg_link2 fills it in), n, ko, kh
                r[1] = make_uint4(seed_key(cx, cy, sdir), po | (ph << 16), 0u, chunk0);                  // next_key, pos, linked, chunk0
            }
            state = ST_IDLE;
        }
        // ---- hand out new work
        int fresh = 0;
        const unsigned long long idle = ballot64(state == ST_IDLE);
        if (exhausted && next == rend && !all_done && __popcll(idle) >= WALK_STEAL_MIN) {
            int nextf = -1;
            const int hop = (int)((blockIdx.x * SW_WAVES + (threadIdx.x >> 6)) * 37u % (unsigned)F);
            for (int k0 = 1; k0 < F && nextf < 0; k0 += 64) {
                const int k = k0 + lane;
                int has = 0;
                int fr = (f + hop + k) % F;
                if (fr == f) fr = -1;
                if (k < F && fr >= 0) {
                    const unsigned done = __hip_atomic_load((unsigned *)(SURV ? &counts[fr].nwalk : &counts[fr].nwalk2), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
                    unsigned m = (unsigned)(SURV ? counts[fr].nsurv : counts[fr].nseeds);
                    if (SURV) m = m < (unsigned)P.maxStarts ? m : (unsigned)P.maxStarts;
                    m = m < ccap ? m : ccap;
                    has = done < m;
                }
                const unsigned long long hb = ballot64(has);
                if (hb) nextf = __shfl(fr, __ffsll((long long)hb) - 1, WAVE);
            }
            if (nextf < 0) {
                all_done = 1;
            } else {
                f = nextf;
                set_queue(f);
                exhausted = 0;
                next = rend = 0;
                pre_ready = 0;
            }
        }
        if (idle) {
            if (next == rend && !exhausted) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(qhead, (unsigned)WALK_GRAB);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n) {
                    exhausted = 1;
                } else {
                    next = pre_base = base;
                    rend = base + WALK_GRAB < n ? base + WALK_GRAB : n;
                    pre_ready = 0;
                    if (base + lane < rend) pre = fin[n - 1 - (base + lane)];
                }
            } else if (next < rend && pre_ready) {
                const int rank = __popcll(idle & ((1ull << lane) - 1ull));
                const int src = (int)(next - pre_base) + rank;
                const unsigned gx = __shfl(pre.x, src & 63, WAVE), gy = __shfl(pre.y, src & 63, WAVE);
                if (state == ST_IDLE && next + (unsigned)rank < rend) {
                    slot = n - 1 - (next + (unsigned)rank);
                    lf = f;
                    if (SURV) {  // x | y << 16, scale << 16 | hole << 24
                        sst = make_uint2(gx, gy);
                        x0 = cx = gx & 0xffff;
                        y0 = cy = gx >> 16;
                        hole = (gy >> 24) & 1;
                        pl = masks + ((long long)f * S + (int)((gy >> 16) & 0xff)) * plane;
                        pc = pidx(cx, cy, P.W);
                        key = hole ? pidx(x0 + 1, y0, P.W) : pc;
                        cxp = cx;  // (no direction yet: the first window is centred on the start)
                        cyp = cy;
                        sdir = 0;
                        first = 1;
                        closed = stopped = 0;
                    } else {
                    cx = gx & 0x1fff;
                    cy = (gx >> 13) & 0x1fff;
                    sdir = (int)(gy & 7u);
                    pl = masks + ((long long)f * S + (int)(gx >> 27)) * plane;
                    pc = pidx(cx, cy, P.W);
                    // the state was entered from the neighbour in direction sdir: travelling the other way
                    cxp = cx + 2 * dir_dx(sdir);
                    cyp = cy + 2 * dir_dy(sdir);
                    }
                    count = 0;
                    ok = 1;
                    too_long = 0;
                    ko = kh = 0xffffffffu;
                    po = ph = 0;
                    brkey = 0xffffffffu;
                    kreg = 1;
                    fresh = 1;
                    state = ST_NEED;
                }
                const unsigned nidle = (unsigned)__popcll(idle);
                next = next + nidle < rend ? next + nidle : rend;
            }
        }
        // ---- pool chunks
        {
            const int want1 = !fresh && (state == ST_ACTIVE || state == ST_NEED || state == ST_LOADING) && (count >> 6) == kreg;
            const unsigned long long b1 = ballot64(want1), b2 = ballot64(fresh);
            const unsigned total = (unsigned)__popcll(b1) + 2u * (unsigned)__popcll(b2);
            if (total) {
                if (arena_next + total > arena_end) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd((unsigned *)&counts[0].npool, (unsigned)WALK_ARENA);
                    arena_next = __builtin_amdgcn_readfirstlane(base);
                    arena_end = arena_next + WALK_ARENA;
                }
                const unsigned long long lt = (1ull << lane) - 1ull;
                const unsigned mine = arena_next + (unsigned)__popcll(b1 & lt) + 2u * (unsigned)__popcll(b2 & lt);
                arena_next += total;
                if (fresh || want1) {
                    const unsigned trow = (SURV ? ccap : 0u) + slot;  // (survivors: the upper rows)
                    if (mine + 1 >= pcap) {
                        ovf |= 8u;
                        if (SURV) {
                            wres[(long long)lf * P.maxContours + slot] = make_uint4(sst.x, sst.y, 0u, (unsigned)key);
                            pend_out[(long long)lf * P.maxContours + slot].p = 0u;
                        } else {
                            segs[(long long)lf * P.maxContours + slot].n = SEG_INVALID;
                        }
                        state = ST_IDLE;
                        pend = 0;
                    } else if (fresh) {
                        chunkA = chunk0 = mine;
                        chunkB = mine + 1;
                        if (SURV) {  // (a segment's first two chunks travel in its record: the table only holds the ones beyond)
                            chunk_tab[chunk_tab_at(P, lf, trow, 0)] = chunkA;
                            chunk_tab[chunk_tab_at(P, lf, trow, 1)] = chunkB;
                        }
                    } else {
                        kreg++;
                        if (kreg & 1) chunkB = mine;
                        else chunkA = mine;
                        chunk_tab[chunk_tab_at(P, lf, trow, (unsigned)kreg)] = mine;
                    }
                }
            }
        }
        // ---- windows: walkers outside theirs get a new one around them, travelling walkers get the tiles ahead
        {
            const bool had = state == ST_ACTIVE;
            const bool mv = had || state == ST_NEED;
            int nbr = br, nbc = bc;
            if (mv) {
                const int xb = cx + (MASK_PADW * 32 - 1), yy = cy;  // padded bit of x-1, padded row of y-1
                const int sdx = cx - cxp, sdy = cy - cyp;
                if (had) {
                    nbr = sdy >= 2 ? (yy >> 4) : sdy <= -2 ? ((yy + 2) >> 4) - 1 : br;
                    nbc = sdx >= 2 ? (xb >> 5) : sdx <= -2 ? ((xb + 2) >> 5) - 1 : bc;
                } else {
                    nbr = sdy > 0 ? (yy >> 4) : sdy < 0 ? ((yy + 2) >> 4) - 1 : ((yy & 15) >= 8 ? (yy >> 4) : (yy >> 4) - 1);
                    nbc = sdx > 0 ? (xb >> 5) : sdx < 0 ? ((xb + 2) >> 5) - 1 : ((xb & 31) >= 16 ? (xb >> 5) : (xb >> 5) - 1);
                }
                nbr = nbr < 0 ? 0 : (nbr > TR - 2 ? TR - 2 : nbr);
                nbc = nbc < 0 ? 0 : (nbc > TC - 2 ? TC - 2 : nbc);
                cxp = cx;
                cyp = cy;
            }
            const bool moved = mv && (!had || nbr != br || nbc != bc);
            if (ballot64(moved)) {
                static_for<4>([&](auto sc) {
                    constexpr int s = decltype(sc)::value, pr = s >> 1, pcx = s & 1;
                    const int ntr = nbr + ((pr ^ nbr) & 1), ntc = nbc + ((pcx ^ nbc) & 1);
                    const int otr = br + ((pr ^ br) & 1), otc = bc + ((pcx ^ bc) & 1);
                    if (moved && (!had || ntr != otr || ntc != otc)) {
                        const uint32_t *g = pl + ((long long)ntr * TC + ntc) * MT_ROWS;
                        static_for<4>([&](auto qc) {
                            constexpr int q = decltype(qc)::value;
                            // (the instruction offset moves the LDS destination as well as the source: compensate in the base)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                             (__attribute__((address_space(3))) void *)((char *)s_win + (pcx * 8 + pr * 4 + q) * 1024 - q * 16),
                                                             16, q * 16, 0);
                        });
                    }
                });
                if (moved) {
                    pend = 1;
                    pbr = nbr;
                    pbc = nbc;
                    if (had) {
                        // until they land the walker keeps the tiles both windows share (it stands on them: a window only ever
                        // moves one tile, towards the side the walker is on)
                        const int r0 = nbr > br ? nbr : br, c0 = nbc > bc ? nbc : bc;
                        vyb = r0 * MT_ROWS;
                        vys = (unsigned)((nbr == br ? 2 : 1) * MT_ROWS - 3);
                        vxb = c0 * 32;
                        vxs = (unsigned)((nbc == bc ? 2 : 1) * 32 - 3);
                        const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - vxb), rr = (unsigned)(cy - vyb);
                        if (xr > vxs || rr > vys) state = ST_LOADING;
                    } else {
                        state = ST_LOADING;
                    }
                }
            }
        }
#ifdef FID_DEBUG_STATS
        d_ckcyc += __builtin_readcyclecounter() - d_c0;
#endif
        if (all_done && ballot64(state != ST_IDLE) == 0) break;
        // ================= steps =================
        const int work_left = !all_done;
        for (int it = 0; it < SW_RUN; it++) {
            const unsigned long long act = ballot64(state == ST_ACTIVE);
            if (it >= SW_CKPT && ballot64(state != ST_ACTIVE && (state != ST_IDLE || work_left))) break;
            if (act == 0) break;
#ifdef FID_DEBUG_STATS
            d_iters++;
            d_active += __popcll(act);
#endif
            if (SURV && state == ST_ACTIVE) {
                const unsigned raw = raw_here();
                const unsigned e = s_lut[raw | ((unsigned)sdir << 8)];
                const int sn = e & 7, code = (e >> 3) & 7;
                const int hmag = (code & 1) ? W2 : 1;
                const int hoff = (code & 2) ? hmag : -hmag;
                if (count > 0 && (e & 0x40u) && seed_state(cx, cy, sdir, sgm)) {
                    stopped = 1;  // the first seed state on this border: its seed cycle has the border already
                    state = ST_FINAL;
                } else {
                    // background pixels examined in the 4-directions belong to this border's hole region
                    int bad = hole && code && (pc + hoff < key);
                    acc = __builtin_amdgcn_alignbit((uint32_t)sn, acc, 4);
                    if ((count & 7) == 7) pool[((count & CK ? chunkB : chunkA) * CKW) + (unsigned)((count & (CK - 1)) >> 3)] = acc;
                    count++;
                    const int dx = dir_dx(sn), dy = dir_dy(sn);
                    const int nx = cx + dx, ny = cy + dy;
                    const int cl = !bad && nx == x0 && ny == y0 && cx == i1x && cy == i1y;
                    cx = nx;
                    cy = ny;
                    pc += __mul24(dy, W2) + dx;
                    bad |= !cl && !hole && pc < key;
                    sdir = sn ^ 4;
                    const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - vxb), rr = (unsigned)(cy - vyb);
                    closed = cl;
                    ok = !bad;
                    state = (bad || cl) ? ST_FINAL : (xr > vxs || rr > vys) ? ST_NEED : ST_ACTIVE;
                }
            } else if (!SURV && state == ST_ACTIVE) {
                const unsigned raw = raw_here();
                const unsigned e = s_lut[raw | ((unsigned)sdir << 8)];
                const unsigned skey = seed_key(cx, cy, sdir);
                const bool onseed = ((e & 0x20u) && (cx & sgm) == 0) || ((e & 0x40u) && (cy & sgm) == 0);
                if (count > 0 && onseed) {
                    state = ST_FINAL;  // the next seed state: the segment ends in front of it
                } else if (count > 0 && skey == brkey) {
                    ok = 0;  // Brent's cycle test: a seed that is no border state went round a border without seeds
                    state = ST_FINAL;
                } else {
                    if ((count & (count - 1)) == 0) brkey = skey;
                    {
                        const bool lo = (e & 8u) && (unsigned)pc < ko, lh = (e & 16u) && (unsigned)pc + 1u < kh;
                        ko = lo ? (unsigned)pc : ko;
                        po = lo ? (unsigned)count : po;
                        kh = lh ? (unsigned)pc + 1u : kh;
                        ph = lh ? (unsigned)count : ph;
                    }
                    const int sn = e & 7;
                    acc = __builtin_amdgcn_alignbit((uint32_t)sn, acc, 4);
                    if ((count & 7) == 7) pool[((count & CK ? chunkB : chunkA) * CKW) + (unsigned)((count & (CK - 1)) >> 3)] = acc;
                    count++;
                    const int dx = dir_dx(sn), dy = dir_dy(sn);
                    cx += dx;
                    cy += dy;
                    pc += __mul24(dy, W2) + dx;
                    sdir = sn ^ 4;
                    const unsigned xr = (unsigned)(cx + (MASK_PADW * 32 - 1) - vxb), rr = (unsigned)(cy - vyb);
                    state = (xr > vxs || rr > vys) ? ST_NEED : ST_ACTIVE;
                }
            }
        }
    }
    if (ovf) atomicOr(&G->overflow, ovf);
#ifdef FID_DEBUG_STATS
    if (lane == 0) {
        atomicAdd(&G->dbg[0], d_iters);
        atomicAdd(&G->dbg[1], d_ckpts);
        atomicAdd(&G->dbg[2], d_active);
        atomicAdd(&G->dbg[3], d_ckcyc);
        atomicAdd(&G->dbg[6], (unsigned long long)(__builtin_readcyclecounter() - d_t0));
        atomicAdd(&G->dbg[7], 1ull);
        for (int k = 0; k < 4; k++) atomicAdd(&G->dbg[21 + k], d_lanes[k]);
    }
#endif
}

__global__ __launch_bounds__(256) void k_seg_link2(const uint2 *__restrict__ seedq, DevSegC *__restrict__ segs,
                                                    const unsigned long long *__restrict__ seedhash, const DevCounts *__restrict__ counts,
                                                    const DevParams P)
{
    const int f = blockIdx.y;
    unsigned ns = (unsigned)counts[f].nseeds;
    ns = ns < (unsigned)P.maxContours ? ns : (unsigned)P.maxContours;
    const unsigned long long *fsh = seedhash + (long long)f * P.seedHashCap;
    const uint2 *fsq = seedq + (long long)f * P.maxContours;
    DevSegC *fsg = segs + (long long)f * P.maxContours;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        if (fsg[i].n == SEG_INVALID) continue;  // (abandoned: too long, pool exhausted -- its border cannot be accepted)
        const unsigned nx = seedhash_find(fsh, P.seedHashCap, P.seedGen, seedhash_key(fsg[i].next_key, (int)(fsq[i].x >> 27)));
        if (nx < ns) {
            fsg[i].next_idx = nx;
            fsg[nx].linked = 1u;
        }
    }
}

// One lane per segment (then one per probe survivor that closed a seedless border by itself).  A segment knows the smallest
// outer-start key ko and hole-start key kh among its own states; it goes once around its cycle while one of the two is still
// the smallest seen.  The cycle is an outer border iff min ko < min kh, and then starts at the state with min ko; else a hole
// border that starts at the state with min kh -- so exactly one lane per cycle ends up accepting, the one whose segment holds
// the canonical start.  It then lists the pieces: its own segment from the start state on, the other segments in cycle order,
// its own segment up to the start state.
// SC_HOPS: segments of a walk noted in LDS (8 bytes x 64 lanes each).  48 for calls of a few frames; 0 for batches -- there the
// 24 KB per workgroup crowd the CUs' LDS while another batch's k_resolve (70 KB) waits for room: the two-context rate fell from
// 32.7 to 31.5 k frames/s with the list on, and a batch does not wait for one cycle's walk anyway.
template <unsigned SC_HOPS>
__global__ __launch_bounds__(64) void k_seg_cycles(const uint2 *__restrict__ seedq, const DevSegC *__restrict__ segs,
                                                    const DevPend *__restrict__ pend, const uint4 *__restrict__ wres,
                                                    uint4 *__restrict__ contours, uint4 *__restrict__ cinfo, uint32_t *__restrict__ cbase,
                                                    uint4 *__restrict__ recs, DevCounts *__restrict__ counts, DevGlobal *__restrict__ G,
                                                    const DevParams P, int part, int warm)
{
    // part 0: the segments (contour list A: slots [0, maxContours / 2), records [0, maxContours)); part 1: the survivors (list B:
    // slots [maxContours / 2, maxContours), records [maxContours, 2 maxContours)) -- the two parts run on different streams
    // The lane that holds a cycle's start goes round it hop by hop, every hop a load that depends on the one before; for a single
    // frame that chain IS the kernel's duration (fifty hops at ~0.65 us, twice: 67 us).  Two things shorten it:
    //   * the segments it passes are noted in LDS (SC_HOPS per lane), so that the second time round -- the copy records -- reads
    //     them from there;
    //   * warm != 0 (calls of a few frames): the workgroups of an XCD first read the frame's segment table between them, one
    //     access per 128-byte line, so that the hops hit that XCD's L2 instead of going to memory for lines another XCD wrote.
    __shared__ uint2 s_hops[SC_HOPS ? SC_HOPS : 1][SC_HOPS ? 64 : 1];
    const int f = blockIdx.y;
    const int lane = lane_id();
    unsigned ns = (unsigned)counts[f].nseeds, nv = (unsigned)counts[f].nsurv;
    ns = ns < (unsigned)P.maxContours ? ns : (unsigned)P.maxContours;
    nv = nv < (unsigned)P.maxContours ? nv : (unsigned)P.maxContours;
    const unsigned lcap = (unsigned)P.maxContours / 2u, rcap = (unsigned)P.maxContours;
    const uint2 *fsq = seedq + (long long)f * P.maxContours;
    const DevSegC *fsg = segs + (long long)f * P.maxContours;
    const DevPend *fpd = pend + (long long)f * P.maxContours;
    uint4 *fco = contours + (long long)f * P.maxContours + (part ? lcap : 0u);
    uint4 *fci = cinfo + (long long)f * P.maxContours + (part ? lcap : 0u);
    uint32_t *fcb = cbase + (long long)f * P.maxContours + (part ? lcap : 0u);
    uint4 *frc = recs + (long long)f * 2 * P.maxContours + (part ? rcap : 0u);
    unsigned *cnt_slots = (unsigned *)(part ? &counts[f].ncontours2 : &counts[f].ncontours);
    unsigned *cnt_recs = (unsigned *)(part ? &counts[f].nrec2 : &counts[f].nrec);
    const unsigned dcap = (unsigned)P.maxChunks * CK;
    const unsigned W2 = (unsigned)P.W + 2u;
    const unsigned ibeg = part ? ns : 0u, iend = part ? ns + nv : ns;
    if (warm && !part) {
        // (workgroups go to the XCDs in turn: blockIdx.x & 7 names this one's, blockIdx.x >> 3 its place among that XCD's)
        const unsigned nlines = (ns * (unsigned)sizeof(DevSegC) + 127u) >> 7, nx = (gridDim.x + 7u) >> 3;
        unsigned acc = 0;
        for (unsigned k = (blockIdx.x >> 3) * 64u + (unsigned)lane; k < nlines; k += nx * 64u)
            acc += *reinterpret_cast<const volatile uint32_t *>(reinterpret_cast<const char *>(fsg) + (size_t)k * 128u);
        asm volatile("" ::"v"(acc));
    }
    for (unsigned i0 = ibeg + blockIdx.x * 64; i0 < iend; i0 += gridDim.x * 64) {
        const unsigned i = i0 + lane;
        int accept = 0, hole = 0;
        unsigned L = 0, key = 0, hops = 0, pos = 0, n0 = 0, nx0 = SEG_INVALID;
        uint2 st = make_uint2(0u, 0u);
        if (i < ns) {
            const DevSegC s = fsg[i];
            if (s.n != SEG_INVALID && s.n != 0u && s.linked && (s.ko & s.kh) != 0xffffffffu) {
                bool co = s.ko != 0xffffffffu, ch = s.kh != 0xffffffffu, closed = false;
                unsigned KO = s.ko, KH = s.kh, cur = s.next_idx;
                L = s.n;
                hops = 1;
                while (cur != SEG_INVALID) {
                    if (cur == i) {
                        closed = true;
                        break;
                    }
                    DevSegC r;
                    {
                        const uint4 q = reinterpret_cast<const uint4 *>(fsg + cur)[0];  // next_idx, n, ko, kh: one 16-byte load per hop
                        r.next_idx = q.x;
                        r.n = q.y;
                        r.ko = q.z;
                        r.kh = q.w;
                    }
                    if (r.n == SEG_INVALID || r.n == 0u) break;
                    co = co && !(r.ko < s.ko);
                    ch = ch && !(r.kh < s.kh);
                    if (!co && !ch) break;  // both candidates beaten: another segment holds the start
                    KO = r.ko < KO ? r.ko : KO;
                    KH = r.kh < KH ? r.kh : KH;
                    L += r.n;
                    if (SC_HOPS && hops - 1u < SC_HOPS) s_hops[hops - 1u][lane] = make_uint2(cur, r.n);  // the hops-th segment of the walk
                    hops++;
                    if (L > (unsigned)P.maxPerim) break;
                    cur = r.next_idx;
                }
                if (closed && L >= (unsigned)P.minPerim && L <= (unsigned)P.maxPerim) {
                    if (co && KO < KH) {
                        accept = 1;
                        hole = 0;
                        key = s.ko;
                        pos = s.pos & 0xffffu;
                    } else if (ch && KH < KO) {
                        accept = 1;
                        hole = 1;
                        key = s.kh;
                        pos = s.pos >> 16;
                    }
                }
                if (accept) {
                    n0 = s.n;
                    nx0 = s.next_idx;
                    const unsigned ky = key / W2, kx = key - ky * W2;  // padded raster index -> pixel (hole: the pixel left of it)
                    st = make_uint2((kx - 1u - (unsigned)hole) | ((ky - 1u) << 16), (uint32_t)f | ((fsq[i].x >> 27) << 16) | ((uint32_t)hole << 24));
                }
            }
        } else if (i < iend) {
            const unsigned j = i - ns;
            if (!fpd[j].p) {  // decided by the survivor walk itself (a border without seeds): accepted iff it left a length
                const uint4 w = wres[(long long)f * P.maxContours + j];
                if (w.z) {
                    st = make_uint2(w.x, w.y);
                    L = w.z;
                    key = w.w;
                    accept = 2;
                }
            }
        }
        const unsigned long long mk = ballot64(accept);
        if (mk) {
            // (a contour's codes start on a multiple of 8 bytes: k_approx reads them eight to a lane)
            const unsigned myL = accept ? (L + 7u) & ~7u : 0u, myR = accept == 1 ? hops + 1u : (accept == 2 ? 1u : 0u);
            const unsigned sL = wave_iscan_dpp(myL), sR = wave_iscan_dpp(myR);
            const unsigned totL = (unsigned)__builtin_amdgcn_readlane((int)sL, 63), totR = (unsigned)__builtin_amdgcn_readlane((int)sR, 63);
            unsigned bslot = 0, bdense = 0, brec = 0;
            if (lane == 63) {
                bslot = atomicAdd(cnt_slots, (unsigned)__popcll(mk));
                bdense = atomicAdd((unsigned *)&counts[f].ndense, totL);
                brec = atomicAdd(cnt_recs, totR);
            }
            bslot = (unsigned)__builtin_amdgcn_readlane((int)bslot, 63);
            bdense = (unsigned)__builtin_amdgcn_readlane((int)bdense, 63);
            brec = (unsigned)__builtin_amdgcn_readlane((int)brec, 63);
            const unsigned idx = bslot + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
            if (accept) {
                if (idx < lcap) {
                    fco[idx] = make_uint4(st.x, st.y, L, key);
                    fci[idx] = make_uint4(i, pos, nx0, 0u);
                    const unsigned dst0 = bdense + sL - myL;
                    unsigned rec = brec + sR - myR;
                    if (dst0 + L > dcap || rec + myR > rcap) {
                        atomicOr(&G->overflow, 8u);
                        fcb[idx] = SEG_INVALID;
                    } else if (accept == 2) {
                        fcb[idx] = dst0;
                        frc[rec] = make_uint4((unsigned)P.maxContours + (i - ns), dst0, L, REC_NO_CHUNK);
                    } else {
                        fcb[idx] = dst0;
                        const unsigned ck0 = fsg[i].chunk0;
                        frc[rec++] = make_uint4(i, dst0, (n0 - pos) | (pos << 16), ck0);  // from the start state to the end of its segment
                        unsigned off = n0 - pos, cur = nx0;
                        {
                            // (an accepted cycle was walked to its end: hops - 1 segments behind the first, the first SC_HOPS of
                            //  them noted; the rest, if any, hop by hop as the first time)
                            const unsigned noted = !SC_HOPS ? 0u : (hops - 1u < SC_HOPS ? hops - 1u : SC_HOPS);
                            for (unsigned h = 0; h < noted; h++) {
                                const uint2 e = s_hops[h][lane];
                                frc[rec++] = make_uint4(e.x, dst0 + off, e.y, fsg[e.x].chunk0);
                                off += e.y;
                                cur = e.x;
                            }
                            if (noted) cur = hops - 1u > noted ? fsg[cur].next_idx : i;
                        }
                        while (cur != i && cur != SEG_INVALID && off < L) {
                            const uint2 q = reinterpret_cast<const uint2 *>(fsg + cur)[0];  // next_idx, n
                            frc[rec++] = make_uint4(cur, dst0 + off, q.y, fsg[cur].chunk0);  // (the same 32-byte record)
                            off += q.y;
                            cur = q.x;
                        }
                        frc[rec++] = make_uint4(i, dst0 + off, pos, ck0);  // ... and the states in front of it
                        for (; rec < brec + sR; rec++) frc[rec] = make_uint4(0u, 0u, 0u, 0u);
                    }
                } else {
                    atomicOr(&G->overflow, 2u);
                }
            }
        }
    }
}

// points per contour the first (short-LDS) launch of k_approx accepts
#define K4_SHORT_PTS 2048
#define K4_SHORT_STACK 256
#define K4_LONG_STACK 1024
__device__ __host__ inline int pts_cap_first(const DevParams &P) { return P.maxPerim < K4_SHORT_PTS ? P.maxPerim : K4_SHORT_PTS; }

// ------------------------------------------------------------------------------------------------
// K4: one wave per accepted contour: gather its points from the chunk pool into LDS (one coalesced 256-byte
// load per chunk), then approxPolyDP(closed, eps = size * polygonalApproxAccuracyRate) exactly as approx.cpp
// approxPolyDP_<int> orders its work (the slice stack is sequential, each slice's farthest-point search is a
// wave reduction with first-maximum tie-break), then _findMarkerContours' gates (aruco.cpp): 4 points,
// convex, min side, distance to the image border.
// LDS (points + slice stack) is what limits residency, so the launch is bucketed by contour length: the
// first launch takes contours of at most pts_cap points with a short stack and flags the rare contour whose
// slice stack overflows (bit 25 of the slot's meta word); the second launch, sized for maxPerimeterPixels,
// takes the longer and the flagged ones.
#define K4_RETRY_BIT (1u << 25)
__global__ __launch_bounds__(64) void k_approx(uint4 *__restrict__ contours, const uint32_t *__restrict__ chunk_tab,
                                                const uint32_t *__restrict__ pool, DevCand *__restrict__ cands,
                                                DevCounts *__restrict__ counts, DevGlobal *__restrict__ G, const DevParams P,
                                                int pts_cap, int stack_cap, int second_pass, const uint32_t *__restrict__ dense,
                                                const uint32_t *__restrict__ cbase, int part = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t pts[];  // pts_cap points, then stack_cap slices
    int2 *stack = reinterpret_cast<int2 *>(pts + pts_cap);
    __shared__ int dst[2 * 16];
    const int lane = lane_id();
    const int f = blockIdx.y;
    // legacy path: contour slot = survivor index; segment tracing (dense != nullptr): the accepted contours, compact
    // (part: trace mode 2 keeps two contour lists per frame -- 1: the seed cycles, first half of the slots; 2: the seedless
    //  borders, second half -- worked on by two launches on different streams; 0: one list)
    unsigned n = (unsigned)(dense ? (part == 2 ? counts[f].ncontours2 : counts[f].ncontours) : counts[f].nsurv);
    n = n < (unsigned)P.maxStarts ? n : (unsigned)P.maxStarts;
    n = n < (unsigned)P.maxContours ? n : (unsigned)P.maxContours;
    const unsigned loff = part == 2 ? (unsigned)P.maxContours / 2u : 0u;
    if (part) n = n < (unsigned)P.maxContours / 2u ? n : (unsigned)P.maxContours / 2u;
    const int W = P.W, H = P.H;
    uint4 *fco = contours + (long long)f * P.maxContours + loff;
    const uint32_t *fpool = pool;  // chunks are numbered across the whole launch
    // this workgroup's slots are blockIdx.x, blockIdx.x + gridDim.x, ...: 64 of them are looked at with one
    // load (a lane each); only the accepted ones of the right length class are then processed in turn
    for (unsigned cb = blockIdx.x; cb < n; cb += gridDim.x * 64) {
      const unsigned myci = cb + (unsigned)lane * gridDim.x;
      uint4 mine = make_uint4(0u, 0u, 0u, 0u);
      if (myci < n) mine = fco[myci];
      int take = mine.z != 0;  // count 0 = slot of a walk that was dropped
      if (second_pass) take = take && ((int)mine.z > pts_cap_first(P) || (mine.y & K4_RETRY_BIT));
      else take = take && (int)mine.z <= pts_cap;
      unsigned long long todo = ballot64(take);
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const unsigned ci = cb + (unsigned)src * gridDim.x;
        uint4 c;
        c.x = (unsigned)__builtin_amdgcn_readlane((int)mine.x, src);  // (src is wave-uniform: v_readlane, not a trip through the LDS crossbar)
        c.y = (unsigned)__builtin_amdgcn_readlane((int)mine.y, src);
        c.z = (unsigned)__builtin_amdgcn_readlane((int)mine.z, src);
        c.w = (unsigned)__builtin_amdgcn_readlane((int)mine.w, src);
        const int count = (int)c.z;
        const int x0 = c.x & 0xffff, y0 = c.x >> 16;
        const int s = (c.y >> 16) & 0xff, hole = (c.y >> 24) & 1;
        __syncthreads();
        // ---- gather the border
        uint32_t pref = 0x80000000u | ci;  // where the points stay for CORNER_REFINE_CONTOUR: the table row of the whole-border walk ...
        if (dense) {
            const uint32_t cb0 = cbase[(long long)f * P.maxContours + loff + ci];
            if (cb0 == SEG_INVALID) continue;
            pref = cb0;  // ... or the contour's place in the frame's dense point array
            // (one code byte per point, the contour's on a multiple of 8: eight to a lane, 512 points a trip)
            const uint8_t *src = reinterpret_cast<const uint8_t *>(dense) + (long long)f * P.maxChunks * CK + cb0;
            uint32_t base = c.x;
            for (int k0 = 0; k0 < count; k0 += 512) {
                const int k = k0 + 8 * lane;
                uint2 w = make_uint2(0u, 0u);
                if (k < count) w = *reinterpret_cast<const uint2 *>(src + k);
                base = codes8_to_points(pts, k, count, base, w);
            }
        } else {
            // (the walker's own chunks: a lane per code word)
            uint32_t base = c.x;
            for (int k0 = 0; k0 < count; k0 += 512) {
                const int k = k0 + 8 * lane;
                uint32_t w = 0u;
                if (k < count)  // (the survivors' rows)
                    w = fpool[(long long)chunk_tab[chunk_tab_at(P, f, (unsigned)P.maxContours + ci, (unsigned)k >> 6)] * CKW + ((k & (CK - 1)) >> 3)];
                base = codes8_to_points(pts, k, count, base, codes_nibbles_to_bytes(w));
            }
        }
        __syncthreads();
        // ---- approxPolyDP
        double eps = (double)count * P.polyAcc;
        eps *= eps;
        int new_count = 0, reject = 0, top = 0;
        int rs_start = 0, pos = 0, le_eps = 0;
        int sx = 0, sy = 0;
        // 1. find approximately two farthest points
        for (int it = 0; it < 3; it++) {
            pos = (pos + rs_start) % count;
            uint32_t sp = pts[pos];
            sx = sp & 0xffff;
            sy = sp >> 16;
            // points j = 1 .. count-1 at index (pos + j) % count; READ_PT leaves pos back at its start
            // (per lane the largest distance and the FIRST index that has it -- j only grows inside a lane, so a strict compare
            //  keeps it; the first-maximum tie-break across lanes is wave_argmax_first's, once per pass)
            unsigned bd = 0, bj = 0;
            for (int j = 1 + lane; j < count; j += 64) {
                int idx = pos + j;
                idx = idx >= count ? idx - count : idx;
                uint32_t p = pts[idx];
                int dx = (int)(p & 0xffff) - sx, dy = (int)(p >> 16) - sy;
                unsigned d = (unsigned)(dx * dx + dy * dy);
                const bool gt = d > bd;
                bd = gt ? d : bd;
                bj = gt ? (unsigned)j : bj;
            }
            unsigned md, mj;
            wave_argmax_first(bd, bd ? bj : 0xffffffffu, md, mj);
            if (md > 0) rs_start = (int)mj;
            le_eps = (double)md <= eps;
            // after the loop READ_PT has advanced pos by count (mod count): pos unchanged
        }
        if (!le_eps) {
            int2 rs, sl;
            rs.y = sl.x = pos % count;
            sl.y = rs.x = (rs_start + sl.x) % count;
            if (lane == 0) {
                stack[0] = rs;
                stack[1] = sl;
            }
            top = 2;
        } else {
            if (lane == 0) {
                dst[0] = sx;
                dst[1] = sy;
            }
            new_count = 1;
        }
        __syncthreads();
        // 3. recursive process
        while (top > 0 && !reject) {
            int2 sl = stack[--top];
            uint32_t ep = pts[sl.y];
            int ex = ep & 0xffff, ey = ep >> 16;
            uint32_t sp = pts[sl.x];
            sx = sp & 0xffff;
            sy = sp >> 16;
            int mcount = sl.y - sl.x;
            if (mcount < 0) mcount += count;
            mcount -= 1;  // interior points
            int split = 0;
            if (mcount > 0) {
                int dx = ex - sx, dy = ey - sy;
                unsigned lbd = 0, lbt = 0;
                bool any = false;
                for (int t = lane; t < mcount; t += 64) {
                    int idx = sl.x + 1 + t;
                    idx = idx >= count ? idx - count : idx;
                    uint32_t p = pts[idx];
                    int px = p & 0xffff, py = p >> 16;
                    int cr = (py - sy) * dx - (px - sx) * dy;
                    unsigned d = (unsigned)(cr < 0 ? -cr : cr);
                    const bool gt = d > lbd || !any;  // (the first point of the lane counts even at distance 0: its index is the tie-break)
                    lbd = gt ? d : lbd;
                    lbt = gt ? (unsigned)t : lbt;
                    any = true;
                }
                unsigned mdist, mt;
                wave_argmax_first(any ? lbd : 0u, any ? lbt : 0xffffffffu, mdist, mt);
                double max_dist = (double)mdist;
                int bt = (int)mt;
                le_eps = max_dist * max_dist <= eps * ((double)dx * dx + (double)dy * dy);
                if (!le_eps) {
                    split = sl.x + 1 + bt;
                    split = split >= count ? split - count : split;
                }
            } else {
                le_eps = 1;
            }
            __syncthreads();
            if (le_eps) {
                if (new_count >= 9) {
                    reject = 1;  // the clean-up pass removes at most half: more than 8 can never end as 4
                } else {
                    if (lane == 0) {
                        dst[2 * new_count] = sx;
                        dst[2 * new_count + 1] = sy;
                    }
                    new_count++;
                }
            } else {
                if (top + 2 > stack_cap) {
                    reject = 1;
                    if (lane == 0) {
                        if (second_pass) atomicOr(&G->overflow, 4u);
                        else fco[ci].y = c.y | K4_RETRY_BIT;  // the second launch has the long stack
                    }
                } else {
                    if (lane == 0) {
                        stack[top] = make_int2(split, sl.y);   // right_slice
                        stack[top + 1] = make_int2(sl.x, split);  // slice
                    }
                    top += 2;
                }
            }
            __syncthreads();
        }
        if (reject || new_count < 4) continue;
        // last stage: remove extra points on the [almost] straight lines (wave-uniform scalar code)
        {
            const int cnt = new_count;
            int posd = cnt - 1, wpos, i;
            int spx, spy, ptx, pty, epx, epy;
#define RD(X, Y)                  \
    do {                          \
        X = dst[2 * posd];        \
        Y = dst[2 * posd + 1];    \
        if (++posd >= cnt) posd = 0; \
    } while (0)
            RD(spx, spy);
            wpos = posd;
            RD(ptx, pty);
            for (i = 0; i < cnt && new_count > 2; i++) {
                RD(epx, epy);
                double dx = epx - spx, dy = epy - spy;
                double dist = fabs((double)(ptx - spx) * dy - (double)(pty - spy) * dx);
                double sip = (double)(ptx - spx) * (epx - ptx) + (double)(pty - spy) * (epy - pty);
                __syncthreads();
                if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
                    new_count--;
                    spx = epx;
                    spy = epy;
                    if (lane == 0) {
                        dst[2 * wpos] = spx;
                        dst[2 * wpos + 1] = spy;
                    }
                    if (++wpos >= cnt) wpos = 0;
                    __syncthreads();
                    RD(ptx, pty);
                    i++;
                    continue;
                }
                spx = ptx;
                spy = pty;
                if (lane == 0) {
                    dst[2 * wpos] = spx;
                    dst[2 * wpos + 1] = spy;
                }
                if (++wpos >= cnt) wpos = 0;
                ptx = epx;
                pty = epy;
                __syncthreads();
            }
#undef RD
        }
        __syncthreads();
        if (new_count != 4) continue;
        int ax[4], ay[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ax[k] = dst[2 * k];
            ay[k] = dst[2 * k + 1];
        }
        // isContourConvex_<int> (convhull.cpp)
        {
            int prevx = ax[2], prevy = ay[2], curx = ax[3], cury = ay[3];
            int dx0 = curx - prevx, dy0 = cury - prevy, orientation = 0, convex = 1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                prevx = curx;
                prevy = cury;
                curx = ax[k];
                cury = ay[k];
                int dx = curx - prevx, dy = cury - prevy;
                int dxdy0 = dx * dy0, dydx0 = dy * dx0;
                orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
                if (orientation == 3) convex = 0;
                dx0 = dx;
                dy0 = dy;
            }
            if (!convex) continue;
        }
        {
            int maxdim = W > H ? W : H;
            double minDistSq = (double)maxdim * maxdim;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int k1 = (k + 1) & 3;
                double d = (double)(ax[k] - ax[k1]) * (double)(ax[k] - ax[k1]) + (double)(ay[k] - ay[k1]) * (double)(ay[k] - ay[k1]);
                minDistSq = minDistSq < d ? minDistSq : d;
            }
            double mcd = (double)count * P.minCornerDistRate;
            if (minDistSq < mcd * mcd) continue;
            int tooNear = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (ax[k] < P.minDistToBorder || ay[k] < P.minDistToBorder || ax[k] > W - 1 - P.minDistToBorder ||
                    ay[k] > H - 1 - P.minDistToBorder)
                    tooNear = 1;
            if (tooNear) continue;
        }
        if (lane == 0) {
            int o = atomicAdd(&counts[f].ncand, 1);
            if (o < P.maxCands) {
                DevCand cd;
                cd.scale = s;
                cd.size = count;
                cd.sx = x0;
                cd.sy = y0;
                cd.hole = hole;
                cd.key = c.w;
                cd.pref = pref;
                cd.pad = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    cd.c[2 * k] = (float)ax[k];
                    cd.c[2 * k + 1] = (float)ay[k];
                }
                cands[(long long)f * P.maxCands + o] = cd;
            } else {
                atomicOr(&counts[f].overflow, 1);
            }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// K5a: restore OpenCV's candidate order (scale ascending; inside a scale cv::findContours returns the
// RETR_LIST contours newest-first = discovery position descending) by rank sort, and apply
// _reorderCandidatesCorners (aruco.cpp).
// gridDim.y > 1: workgroup = 64 candidates x 4 quarters of the key list (partial ranks meet in LDS), gridDim.y workgroups share a
// frame's candidates -- a call of a few frames spreads the n * n comparisons over several CUs (one workgroup: 15 us for a single frame's 600).
__global__ __launch_bounds__(256) void k_sort_cands(const DevCand *__restrict__ cands, DevCand *__restrict__ sorted,
                                                    float4 *__restrict__ cmeta, DevCounts *__restrict__ counts, const DevParams P)
{
    extern __shared__ unsigned long long keys[];
    __shared__ int s_rank[256];
    const int f = blockIdx.x;
    int n = counts[f].ncand;
    n = n < P.maxCands ? n : P.maxCands;
    if ((int)blockIdx.y * (gridDim.y > 1 ? 64 : 256) >= n) return;
    const DevCand *src = cands + (long long)f * P.maxCands;
    DevCand *dstc = sorted + (long long)f * P.maxCands;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        keys[i] = ((unsigned long long)(unsigned)src[i].scale << 32) | (0xffffffffu - src[i].key);
    __syncthreads();
    // (a frame with a workgroup to itself -- a batch -- keeps every thread on a candidate of its own: no partial ranks to add up)
    const int parts = gridDim.y > 1 ? 4 : 1, ipb = 256 / parts;
    const int il = threadIdx.x & (ipb - 1), part = threadIdx.x / ipb;
    const int j0 = (int)((long long)n * part / parts), j1 = (int)((long long)n * (part + 1) / parts);
    for (int i0 = (int)blockIdx.y * ipb; i0 < n; i0 += (int)gridDim.y * ipb) {
        const int i = i0 + il;
        if (part == 0) s_rank[il] = 0;
        __syncthreads();
        DevCand c;
        if (i < n) {
            if (part == 0) c = src[i];  // (in flight under the comparisons)
            const unsigned long long k = keys[i];
            int r = 0;
            for (int j = j0; j < j1; j++) r += keys[j] < k;
            atomicAdd(&s_rank[il], r);
        }
        __syncthreads();
        if (i < n && part == 0) {
            const int rank = s_rank[il];
            double dx1 = c.c[2] - c.c[0], dy1 = c.c[3] - c.c[1];
            double dx2 = c.c[4] - c.c[0], dy2 = c.c[5] - c.c[1];
            double cross = (dx1 * dy2) - (dy1 * dx2);
            if (cross < 0.0) {
                float tx = c.c[2], ty = c.c[3];
                c.c[2] = c.c[6];
                c.c[3] = c.c[7];
                c.c[6] = tx;
                c.c[7] = ty;
            }
            dstc[rank] = c;
            // corner sums (exact: integer coordinates) and contour size for k_near's centroid test
            cmeta[(long long)f * P.maxCands + rank] =
                make_float4(c.c[0] + c.c[2] + c.c[4] + c.c[6], c.c[1] + c.c[3] + c.c[5] + c.c[7], (float)c.size, 0.f);
        }
        __syncthreads();
    }
}

// The near matrix of a frame (bit j of row i set: candidates i < j are too close) is stored as its upper triangle, row after
// row: row i keeps its words (i >> 5) .. nw - 1, nw = ceil(n / 32).  k_resolve copies the whole thing into LDS in one go.
__device__ __forceinline__ int near_row_off(int i, int nw)
{
    const int q = i >> 5;  // row i keeps words q .. nw - 1
    return i * nw - 16 * q * (q - 1) - (i - 32 * q) * q;
}
// K5b: _filterTooCloseCandidates pair test (aruco.cpp): bit j of near[f][i][j>>5] for j > i.
// For any cyclic shift the mean squared corner distance is at least the squared distance of the corner means
// (Jensen), so a pair whose centroids are far enough apart cannot be near: that test runs on a compact
// {corner sums, size} record and skips the candidate loads for almost every pair.  The margin covers the
// float rounding of the reference's ax * ax + ay * ay.
// A wave takes (row i, 64 columns): lane = column j, so the 64 records of a block arrive in ONE coalesced load (a thread that
// walked the 32 columns of a word waited for 32 loads one after the other: 27 us for a single frame's 600 candidates), the few
// lanes whose centroid test fails to rule the pair out fetch their candidate and run the four shifts side by side, and a
// ballot makes the two words.  Every word of the row's part of the triangle is written (zeros included).
__global__ __launch_bounds__(256) void k_near(const DevCand *__restrict__ sorted, const float4 *__restrict__ cmeta,
                                               uint32_t *__restrict__ nearb, const DevCounts *__restrict__ counts,
                                               const DevParams P)
{
    const int f = blockIdx.y;
    int n = counts[f].ncand;
    n = n < P.maxCands ? n : P.maxCands;
    const int NW = P.maxCands >> 5;
    const int nw = (n + 31) >> 5, nw2 = (n + 63) >> 6;
    const DevCand *cs = sorted + (long long)f * P.maxCands;
    const float4 *cm = cmeta + (long long)f * P.maxCands;
    uint32_t *nb = nearb + (long long)f * P.maxCands * NW;
    const int lane = lane_id();
    const float rate_f = (float)P.minMarkerDistRate * 1.0001f;
    const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    // One (row i, 64-column block w2) item: lane = column j.
    auto item = [&](int i, int w2) {
        const int j = w2 * 64 + lane;
        const bool valid = j > i && j < n;
        const float4 ma = cm[i];
        bool close = false;
        if (valid) {
            // (a PRE-filter: it only has to keep every pair the exact test below can accept, so single precision with a margin
            //  that dwarfs its rounding does what the double-precision form did with a fifth of the issue slots -- round 5)
            const float4 mo = cm[j];
            const float sz = ma.z < mo.z ? ma.z : mo.z;
            const float lim0 = sz * rate_f;
            const float lim = 16.f * (lim0 * lim0 * 1.001f + 1.f) + 1.f;
            const float dx = ma.x - mo.x, dy = ma.y - mo.y;
            close = dx * dx + dy * dy < lim;  // (otherwise: centroids too far apart for any shift)
        }
        bool near = false;
        if (close) {
            const DevCand a = cs[i];
            const DevCand o = cs[j];
            const int minimumPerimeter = a.size < o.size ? a.size : o.size;
            double mmd = (double)minimumPerimeter * P.minMarkerDistRate;
            mmd = mmd * mmd;
#pragma unroll
            for (int fc = 0; fc < 4; fc++) {
                double distSq = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int modC = (c + fc) & 3;
                    const float ax = a.c[2 * modC] - o.c[2 * c];
                    const float ay = a.c[2 * modC + 1] - o.c[2 * c + 1];
                    distSq += ax * ax + ay * ay;
                }
                distSq /= 4.;
                near = near || distSq < mmd;
            }
        }
        const unsigned long long bits = ballot64(near);
        if (lane < 2) {
            const int w = 2 * w2 + lane;
            if (w >= (i >> 5) && w < nw) nb[near_row_off(i, nw) + w - (i >> 5)] = (uint32_t)(bits >> (32 * lane));
        }
    };
    // (round 6) Batches: a wave takes the ROWS wave, wave + nwaves, ... and in a row only the blocks from the one that holds i itself
    // on (its words belong to the row) -- until now every (row, block) pair was an item found by a division, and the half below
    // the diagonal was skipped one `continue` at a time: a third of this kernel's instructions.  A call of a few frames has more
    // waves than rows (the grid is sized for the latency of ONE frame): there the items stay spread over all waves.
    if (nwaves <= n) {
        // (round 6, second step) ... and the exact test is taken off the 64-column sweep: a marker's outline comes back as a
        // candidate at most scales, so ~5 % of a row's pairs pass the centroid test -- three or four lanes of every block then ran
        // the four-shift test with the other sixty idle (lane utilisation 0.23, the lowest of the pipeline).  The sweep now only
        // QUEUES the columns that pass (ballot + popcount into an LDS list), the exact test runs on 64 queued pairs at a time and
        // sets its bits in the row's words in LDS, and the row's part of the triangle leaves from there.  Same pairs, same
        // arithmetic per pair; only the grouping changed.
        __shared__ uint32_t s_row[4][128];  // (max_candidates_per_frame <= 4096: 128 words a row)
        __shared__ uint16_t s_q[4][128];
        const int wv = (int)(threadIdx.x >> 6);
        uint32_t *row = s_row[wv];
        uint16_t *q = s_q[wv];
        row[lane] = 0u;
        row[lane + 64] = 0u;
        for (int i = __builtin_amdgcn_readfirstlane(wave); i < n; i += nwaves) {  // (the row is the wave's: scalar registers)
            const float4 ma = cm[i];
            const DevCand a = cs[i];
            int qn = 0;  // wave-uniform
            auto exact = [&](int cnt) {  // the first min(cnt, 64) queued columns
                bool near = false;
                int j = 0;
                if (lane < cnt) {
                    j = q[lane];
                    const DevCand o = cs[j];
                    const int minimumPerimeter = a.size < o.size ? a.size : o.size;
                    double mmd = (double)minimumPerimeter * P.minMarkerDistRate;
                    mmd = mmd * mmd;
#pragma unroll
                    for (int fc = 0; fc < 4; fc++) {
                        double distSq = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int modC = (c + fc) & 3;
                            const float ax = a.c[2 * modC] - o.c[2 * c];
                            const float ay = a.c[2 * modC + 1] - o.c[2 * c + 1];
                            distSq += ax * ax + ay * ay;
                        }
                        distSq /= 4.;
                        near = near || distSq < mmd;
                    }
                }
                if (near) atomicOr(&row[j >> 5], 1u << (j & 31));
            };
            for (int w2 = i >> 6; w2 < nw2; w2++) {
                const int j = w2 * 64 + lane;
                bool close = false;
                if (j > i && j < n) {
                    const float4 mo = cm[j];
                    const float sz = ma.z < mo.z ? ma.z : mo.z;
                    const float lim0 = sz * rate_f;
                    const float lim = 16.f * (lim0 * lim0 * 1.001f + 1.f) + 1.f;
                    const float dx = ma.x - mo.x, dy = ma.y - mo.y;
                    close = dx * dx + dy * dy < lim;
                }
                const unsigned long long cb = ballot64(close);
                if (cb) {
                    if (close) q[qn + __popcll(cb & ((1ull << lane) - 1ull))] = (uint16_t)j;
                    qn += __popcll(cb);
                    if (qn >= 64) {
                        exact(64);
                        qn -= 64;
                        if (lane < qn) {
                            const uint16_t t = q[64 + lane];
                            q[lane] = t;
                        }
                    }
                }
            }
            if (qn) exact(qn);
            // the row's words (i >> 5) .. nw - 1, then clean for the next row
            for (int w = (i >> 5) + lane; w < nw; w += 64) {
                nb[near_row_off(i, nw) + w - (i >> 5)] = row[w];
                row[w] = 0u;
            }
        }
    } else {
        for (int it = wave; it < n * nw2; it += nwaves) {
            const int i = it / nw2, w2 = it - i * nw2;
            if (w2 * 64 + 63 < i) continue;  // (the block that holds i itself is kept)
            item(i, w2);
        }
    }
}

// K5c: the sequential part of _filterTooCloseCandidates: near pairs are visited in (i, j) order, a pair
// whose members are both still alive removes the one with the smaller contour (ties: the first).
// For a live i this means: scan its live near j > i in order; every j with size_j < size_i dies, the
// first j with size_j >= size_i kills i and ends the row.
// What row i does depends only on rows of its own connected component of the near graph (a marker seen at 13 threshold
// scales, inside and outside border: a clique of up to 26), so the components are resolved side by side, each one by one
// wave in row order with lanes owning 32-bit words of a row: the workgroup copies the triangle into LDS, labels the components
// (minimum index, propagated along the edges with pointer jumping) and deals them out to its waves.  One workgroup per frame.
// (One wave taking the 600 rows of a bench frame one after the other spent 180 us on LDS latencies; one THREAD per component
// 130 us: a clique's rows are long.)
// A triangle that does not fit the LDS budget goes the old way: one wave, rows in order, straight from global memory.
__global__ __launch_bounds__(1024) void k_resolve(const DevCand *__restrict__ sorted, const uint32_t *__restrict__ nearb,
                                                 DevCand *__restrict__ filtered, DevCounts *__restrict__ counts,
                                                 unsigned *__restrict__ worklist, unsigned *__restrict__ nwork,
                                                 const DevParams P, int lds_words, DevGlobal *__restrict__ G, int reg_max)
{
    // reg_max: components of up to this many candidates (at most 64) are resolved in registers; 0: every one by the LDS-row loop
#ifdef FID_DEBUG_STATS
    unsigned long long d_t[8];
    int d_k = 0, d_iters = 0;
#define RES_MARK() d_t[d_k++] = __builtin_readcyclecounter();
#else
#define RES_MARK()
#endif
    RES_MARK()
    extern __shared__ int sizes[];     // maxCands sizes | labels | component sizes, then lds_words words for the near matrix
    __shared__ uint32_t s_rem[128];    // per 32 candidates: who has been removed
    __shared__ uint32_t s_alive[128];  // ... who is left / where the first of them goes
    __shared__ int s_off[128];
    __shared__ int s_mem[16][64];    // per wave: the members of the component it is resolving
    const int f = blockIdx.x, lane = lane_id(), tid = threadIdx.x, nt = blockDim.x;
    int n = counts[f].ncand;
    n = n < P.maxCands ? n : P.maxCands;
    const int NW = P.maxCands >> 5;
    const DevCand *cs = sorted + (long long)f * P.maxCands;
    const uint32_t *nb = nearb + (long long)f * P.maxCands * NW;
    const int nw = (n + 31) >> 5;
    int *label = sizes + P.maxCands, *csize = label + P.maxCands;
    uint32_t *s_near = reinterpret_cast<uint32_t *>(csize + P.maxCands);
    const int tri = n > 0 ? near_row_off(n - 1, nw) + nw - ((n - 1) >> 5) : 0;  // words of the triangle
    const bool in_lds = tri <= lds_words;
    for (int i = tid; i < n; i += nt) {
        sizes[i] = cs[i].size;
        label[i] = i;
        csize[i] = 0;
    }
    if (tid < 128) s_rem[tid] = 0u;
    if (in_lds) {
        // (the frame's part of nearb starts on a 16-byte boundary: maxCands is a multiple of 32)
        const uint4 *src = reinterpret_cast<const uint4 *>(nb);
        uint4 *dst = reinterpret_cast<uint4 *>(s_near);
        const int nq = (tri + 3) >> 2;
        for (int t0 = 0; t0 < nq; t0 += 4 * nt) {  // four loads in flight per thread
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + u * nt + tid;
                v[u] = t < nq ? src[t] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + u * nt + tid;
                if (t < nq) dst[t] = v[u];
            }
        }
    }
    __syncthreads();
    RES_MARK()
    if (in_lds) {
        // ---- components: label = smallest index
        for (;;) {
#ifdef FID_DEBUG_STATS
            d_iters++;
#endif
            int changed = 0;
            // (four threads share a row, each takes every fourth word: a thread walking all the words of its row waited for
            //  some twenty LDS reads one after the other, 13 000 cycles a round)
            for (int i = tid >> 2; i < n; i += nt >> 2) {
                const int l0 = label[i];
                int li = l0;
                const uint32_t *row = s_near + near_row_off(i, nw) - (i >> 5);
                for (int w = (i >> 5) + (tid & 3); w < nw; w += 4) {
                    uint32_t bits = row[w];
                    while (bits) {
                        const int j = w * 32 + __ffs(bits) - 1;
                        bits &= bits - 1;
                        const int lj = label[j];
                        if (lj < li) {
                            li = lj;
                        } else if (lj > li) {
                            atomicMin(&label[j], li);
                            changed = 1;
                        }
                    }
                }
                const int ll = label[li];  // pointer jumping
                li = ll < li ? ll : li;
                if (li < l0) {
                    atomicMin(&label[i], li);
                    changed = 1;
                }
            }
            if (!__syncthreads_or(changed)) break;
        }
        RES_MARK()
        for (int i = tid; i < n; i += nt) atomicAdd(&csize[label[i]], 1);
        __syncthreads();
        RES_MARK()
        // ---- a component of TWO candidates (the inside and the outside border of a quad seen at one scale) needs no wave: its
        //      root's thread finds the other one in its row and removes the smaller (ties: the root, the first of the pair)
        for (int i = tid; i < n; i += nt) {
            if (label[i] != i || csize[i] != 2) continue;
            const uint32_t *row = s_near + near_row_off(i, nw) - (i >> 5);
            int j = -1;
            for (int w = i >> 5; w < nw && j < 0; w++) {
                const uint32_t bits = row[w];
                if (bits) j = w * 32 + __ffs(bits) - 1;
            }
            if (j >= 0) {
                const int dead = sizes[j] >= sizes[i] ? i : j;
                atomicOr(&s_rem[dead >> 5], 1u << (dead & 31));
            }
        }
        // ---- the larger components are dealt out to the waves; a wave takes the rows of a component in order, lanes own 32-bit
        //      words of the row (other waves set other bits of the same removed-set words: LDS atomics)
        const int nwaves = (int)blockDim.x >> 6, wv = tid >> 6;
        int rootrank = 0;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int ci = c0 + lane;
            unsigned long long rb = ballot64(ci < n && label[ci] == ci && csize[ci] > 2);
            while (rb) {
                const int r = c0 + __ffsll((long long)rb) - 1;
                rb &= rb - 1;
                if ((rootrank++ % nwaves) != wv) continue;
                const int cs = csize[r];
                if (cs <= reg_max) {
                    // A component of at most 64 candidates (a marker seen at 13 scales, inside and outside border: 26) is resolved in
                    // REGISTERS: lane m holds member m's row as a 64-bit mask over the component's members (in index order) and its
                    // size, and the rows are taken in order with v_readlane and scalar bit operations -- no LDS round trip per
                    // row (the LDS-row loop below: ~1 us per row, 24 us of a single frame's 38 us)
                    int *mem = s_mem[wv];
                    int cnt = 0;
                    for (int m0 = r & ~63; cnt < cs; m0 += 256) {  // (four blocks of labels in flight per step)
                        int lab[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int mi = m0 + 64 * u + lane;
                            lab[u] = mi < n ? label[mi] : -1;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int mi = m0 + 64 * u + lane;
                            const bool is = mi >= r && lab[u] == r;
                            const unsigned long long mb = ballot64(is);
                            if (is) mem[cnt + __popcll(mb & ((1ull << lane) - 1ull))] = mi;
                            cnt += __popcll(mb);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the wave's own LDS writes, read by its other lanes)
                    const int g = lane < cs ? mem[lane] : 0;
                    const int sz = lane < cs ? sizes[g] : 0;
                    // (every near j of a member is a member: its row, cut down to the members.  One independent LDS read per
                    //  member and lane -- scanning the row's words and looking every set bit up was a chain of ~60 dependent reads)
                    unsigned long long mask = 0ull;
                    {
                        const uint32_t *row = s_near + near_row_off(g, nw) - (g >> 5);
#pragma unroll 4
                        for (int k = 1; k < cs; k++) {  // (member k from the list in LDS, not v_readlane: a convergent operation keeps the loop from being unrolled)
                            const int gk = mem[k];
                            if (lane < k) mask |= (unsigned long long)((row[gk >> 5] >> (gk & 31)) & 1u) << k;
                        }
                    }
                    const int mlo = (int)(unsigned)mask, mhi = (int)(unsigned)(mask >> 32);
                    unsigned long long alive = cs == 64 ? ~0ull : (1ull << cs) - 1ull;
                    for (int i = 0; i < cs; i++) {  // wave-uniform
                        if (!((alive >> i) & 1ull)) continue;
                        const unsigned long long rowm =
                            (((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, i) << 32) | (unsigned)__builtin_amdgcn_readlane(mlo, i)) & alive;
                        if (!rowm) continue;
                        const int szi = __builtin_amdgcn_readlane(sz, i);
                        const unsigned long long ge = ballot64(sz >= szi) & rowm;
                        if (ge) {  // the first live near j with size_j >= size_i removes i; the live near j in front of it are removed
                            const int kj = __ffsll((long long)ge) - 1;
                            alive &= ~(rowm & ((1ull << kj) - 1ull));
                            alive &= ~(1ull << i);
                        } else {
                            alive &= ~rowm;
                        }
                    }
                    if (lane < cs && !((alive >> lane) & 1ull)) atomicOr(&s_rem[g >> 5], 1u << (g & 31));
                    continue;
                }
                int left = cs;
                for (int m0 = r & ~63; left > 0; m0 += 64) {
                    const int mi = m0 + lane;
                    unsigned long long mb = ballot64(mi < n && mi >= r && label[mi] == r);
                    left -= __popcll(mb);
                    while (mb) {
                        const int i = m0 + __ffsll((long long)mb) - 1;  // wave-uniform
                        mb &= mb - 1;
                        const int wi = i >> 5;
                        if ((s_rem[wi] >> (i & 31)) & 1u) continue;
                        const int szi = sizes[i];
                        const uint32_t *row = s_near + near_row_off(i, nw) - wi;
                        int firstKill = INT_MAX;
                        uint32_t live[2] = {0u, 0u};
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int w = lane + 64 * k;
                            if (w < nw && w >= wi) {
                                const uint32_t bits = row[w] & ~s_rem[w];
                                live[k] = bits;
                                uint32_t t = bits;
                                while (t) {
                                    const int j = w * 32 + __ffs(t) - 1;
                                    t &= t - 1;
                                    if (sizes[j] >= szi) {
                                        firstKill = firstKill < j ? firstKill : j;
                                        break;
                                    }
                                }
                            }
                        }
                        const int jk = wave_min_i32_dpp(firstKill);
                        // every live near j < jk has size_j < size_i and is removed; jk, if any, removes i
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int w = lane + 64 * k;
                            if (live[k]) {
                                uint32_t m;
                                if (jk == INT_MAX || (jk >> 5) > w) m = 0xffffffffu;
                                else if ((jk >> 5) < w) m = 0;
                                else m = (1u << (jk & 31)) - 1u;
                                if (live[k] & m) atomicOr(&s_rem[w], live[k] & m);
                            }
                        }
                        if (jk != INT_MAX && lane == 0) atomicOr(&s_rem[wi], 1u << (i & 31));
                    }
                }
            }
        }
    } else if (tid < 64) {
        // removed bits: lane l owns words l, l+64, ... (maxCands <= 4096 -> at most 2 words per lane)
        uint32_t rem0 = 0, rem1 = 0;
        // Row i of the near matrix does not depend on what has been removed so far: the rows of the next RB candidates are
        // fetched together (their loads overlap) and then resolved one after the other.
        constexpr int RB = 8;
        for (int i0 = 0; i0 < n; i0 += RB) {
            uint32_t row[RB][2];
#pragma unroll
            for (int u = 0; u < RB; u++) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int w = lane + 64 * k, i = i0 + u;
                    row[u][k] = (i < n && w < nw && w >= (i >> 5)) ? nb[near_row_off(i, nw) + w - (i >> 5)] : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int i = i0 + u;
                if (i >= n) break;  // wave-uniform
                if (ballot64((row[u][0] | row[u][1]) != 0u) == 0ull) continue;  // nothing near i
                const int wi = i >> 5;  // wave-uniform: the word of the removed set that holds candidate i sits in lane wi & 63
                const uint32_t rw = (uint32_t)__builtin_amdgcn_readlane((int)(wi < 64 ? rem0 : rem1), wi & 63);
                if ((rw >> (i & 31)) & 1u) continue;  // wave-uniform
                int szi = sizes[i];
                int firstKill = INT_MAX;
                uint32_t live0 = 0, live1 = 0;
                // each lane scans its words
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int w = lane + 64 * k;
                    if (w < nw && w >= wi) {
                        uint32_t bits = row[u][k] & ~(k == 0 ? rem0 : rem1);
                        if (k == 0) live0 = bits; else live1 = bits;
                        uint32_t t = bits;
                        while (t) {
                            int b = __ffs(t) - 1;
                            t &= t - 1;
                            int j = w * 32 + b;
                            if (sizes[j] >= szi) {
                                firstKill = firstKill < j ? firstKill : j;
                                break;
                            }
                        }
                    }
                }
                int jk = wave_min_i32_dpp(firstKill);
                // every live near j < jk has size_j < size_i and is removed
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int w = lane + 64 * k;
                    uint32_t bits = k == 0 ? live0 : live1;
                    if (bits) {
                        uint32_t m;
                        if (jk == INT_MAX || (jk >> 5) > w) m = 0xffffffffu;
                        else if ((jk >> 5) < w) m = 0;
                        else m = (1u << (jk & 31)) - 1u;
                        if (k == 0) rem0 |= bits & m; else rem1 |= bits & m;
                    }
                }
                if (jk != INT_MAX) {
                    if ((wi & 63) == lane) {
                        if (wi < 64) rem0 |= 1u << (i & 31); else rem1 |= 1u << (i & 31);
                    }
                }
            }
        }
        s_rem[lane] = rem0;
        s_rem[lane + 64] = rem1;
    }
    __syncthreads();
    RES_MARK()
    if (tid < 64) {
        // places of the survivors, in order
        int base = 0;
        for (int w0 = 0; w0 < nw; w0 += 64) {
            int w = w0 + lane;
            uint32_t alive = 0;
            if (w < nw) {
                alive = ~s_rem[w];
                int hi = n - w * 32;
                if (hi < 32) alive &= (1u << hi) - 1u;
            }
            int cnt = __popc(alive);
            int incl = wave_iscan(cnt);
            if (w < nw) {
                s_alive[w] = alive;
                s_off[w] = base + incl - cnt;
            }
            base += __shfl(incl, 63, WAVE);
        }
        unsigned o = 0;
        if (lane == 0) {
            counts[f].nfilt = base;
            o = atomicAdd(nwork, (unsigned)base);
        }
        o = (unsigned)__builtin_amdgcn_readfirstlane((int)o);
        for (int k = lane; k < base; k += 64) worklist[o + k] = ((unsigned)f << 16) | (unsigned)k;
    }
    __syncthreads();
    RES_MARK()
    // the whole workgroup moves them
    for (int i = tid; i < n; i += nt) {
        const uint32_t alive = s_alive[i >> 5], bit = 1u << (i & 31);
        if (alive & bit) filtered[(long long)f * P.maxCands + s_off[i >> 5] + __popc(alive & (bit - 1u))] = cs[i];
    }
#ifdef FID_DEBUG_STATS
    __syncthreads();
    RES_MARK()
    if (tid == 0 && f == 0) {
        for (int q = 1; q < d_k; q++) G->dbg[24 + q] = d_t[q] - d_t[q - 1];
        G->dbg[31] = (unsigned long long)d_iters | ((unsigned long long)n << 32);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// K6: _identifyOneCandidate (aruco.cpp): one wave per candidate.
//   getPerspectiveTransform = 8x8 LU with partial pivoting (hal LU64f order), one matrix element per lane;
//   warpPerspective(INTER_NEAREST) into an LDS patch; meanStdDev; Otsu (sequential, as
//   getThreshVal_Otsu_8u); cell majority; border test; Dictionary::identify by XOR+popcount.
__device__ __forceinline__ int sat_round_int(double v)
{
    // saturate_cast<int>(double) after the std::max/min clamp of WarpPerspectiveInvoker
    v = fmax((double)INT_MIN, fmin((double)INT_MAX, v));
    return (int)rint(v);
}

__global__ __launch_bounds__(64) void k_identify(const uint8_t *__restrict__ gray, long long gfstride,
                                                  const DevCand *__restrict__ filtered, const unsigned *__restrict__ worklist,
                                                  const unsigned *__restrict__ nwork, const uint8_t *__restrict__ dict,
                                                  DevIdent *__restrict__ ident, const DevParams P)
{
    extern __shared__ uint8_t patch[];  // S*S bytes
    __shared__ int hist[256];
    __shared__ uint8_t cellbits[FID_MAX_CELLS * FID_MAX_CELLS];
    __shared__ int s_thr;
    __shared__ double2 s_ny[256];             // Otsu, per bin: (i * p_i, refined reciprocal of q1) ...
    __shared__ double s_q1[256], s_pm[256];   // ... q1; p_i, later mu1 (no loop reads an array it writes)
    const int lane = lane_id();
    const unsigned n = *nwork;
    const int ms = P.markerSize, bb = P.borderBits, msb = ms + 2 * bb, cellSize = P.cellSize;
    const int SZ = msb * cellSize;
    const int W = P.W, H = P.H;
    for (unsigned wi = blockIdx.x; wi < n; wi += gridDim.x) {
        const unsigned item = worklist[wi];
        const int f = item >> 16, k = item & 0xffff;
        const DevCand *cdp = filtered + (long long)f * P.maxCands + k;  // (read in place: a copy indexed by lane went through scratch memory)
        const uint8_t *g = gray + (long long)f * gfstride;
        DevIdent *out = ident + (long long)f * P.maxCands + k;
        __syncthreads();
#ifdef ID_TIMING
        unsigned long long ti[6];
        ti[0] = __builtin_readcyclecounter();
#define ID_T(k) ti[k] = __builtin_readcyclecounter();
#else
#define ID_T(k)
#endif
        for (int i = lane; i < 256; i += 64) hist[i] = 0;
        // ---- getPerspectiveTransform(src = candidate corners, dst = patch corners), lane = row*8 + col
        const int row = lane >> 3, col = lane & 7;
        double a, b;
        {
            const float fs1 = (float)SZ - 1.f;
            const int pi = row & 3;
            float sxp = cdp->c[2 * pi], syp = cdp->c[2 * pi + 1];
            float dxp = (pi == 1 || pi == 2) ? fs1 : 0.f;
            float dyp = (pi >= 2) ? fs1 : 0.f;
            float dsel = row < 4 ? dxp : dyp;
            // rows 0-3: [x y 1 0 0 0 -x*X -y*X], rows 4-7: [0 0 0 x y 1 -x*Y -y*Y]
            double v = 0.;
            int c3 = row < 4 ? col : col - 3;
            if (col < 6) {
                if (c3 == 0) v = sxp;
                else if (c3 == 1) v = syp;
                else if (c3 == 2) v = 1.;
                else v = 0.;
                if (row >= 4 && col < 3) v = 0.;
                if (row < 4 && col >= 3) v = 0.;
            } else if (col == 6) {
                v = (double)(-sxp * dsel);
            } else {
                v = (double)(-syp * dsel);
            }
            a = v;
            b = dsel;
        }
        int singular = 0;
        for (int i = 0; i < 8; i++) {
            // pivot search down column i
            int kp = i;
            double best = fabs(bcast_f64(a, i * 8 + i));
            for (int j = i + 1; j < 8; j++) {
                double v = fabs(bcast_f64(a, j * 8 + i));
                if (v > best) {
                    best = v;
                    kp = j;
                }
            }
            if (best < DBL_EPSILON * 100) {
                singular = 1;
                break;
            }
            if (kp != i) {
                double ai = shfl_f64(a, i * 8 + col), ak = shfl_f64(a, kp * 8 + col);
                double bi = bcast_f64(b, i * 8), bk = bcast_f64(b, kp * 8);
                if (row == i) { a = ak; b = bk; }
                else if (row == kp) { a = ai; b = bi; }
            }
            double d = -1 / bcast_f64(a, i * 8 + i);
            double alpha = shfl_f64(a, row * 8 + i) * d;
            double piv = shfl_f64(a, i * 8 + col);
            double pb = bcast_f64(b, i * 8);
            if (row > i) {
                if (col > i) a += alpha * piv;
                b += alpha * pb;
            }
        }
        double M[9];
        if (!singular) {
            double x[8];
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                double s = bcast_f64(b, i * 8);
#pragma unroll
                for (int kk = i + 1; kk < 8; kk++) s -= bcast_f64(a, i * 8 + kk) * x[kk];
                x[i] = s / bcast_f64(a, i * 8 + i);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) M[i] = x[i];
            M[8] = 1.;
        }
        // ---- invert (cv::invert 3x3 fast path) and warp
        double Mi[9];
        int okinv = 0;
        if (!singular) {
#define Sd(r, c) M[(r)*3 + (c)]
            double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) - Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
                       Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
            if (d != 0.) {
                okinv = 1;
                d = 1. / d;
                Mi[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
                Mi[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
                Mi[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
                Mi[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
                Mi[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
                Mi[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
                Mi[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
                Mi[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
                Mi[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
            }
#undef Sd
        }
        __syncthreads();
        ID_T(1)
        long long ssum = 0, ssq = 0;
        const int in0 = cellSize / 2, in1 = SZ - cellSize / 2;
        // (round 4: seven patch pixels of a lane at a time -- their source pixels are fetched together, then stored and counted: one
        //  global-memory latency per seven pixels instead of one per pixel (the loads were 35 of the 55 k cycles this loop took
        //  for a 56 x 56 patch); the row / column of a pixel is carried along instead of divided out.  Same arithmetic per pixel.)
        {
            constexpr int WU = 7;
            int py = lane / SZ, px1 = lane - py * SZ;  // (one division per candidate)
            for (int p0 = lane; p0 < SZ * SZ; p0 += 64 * WU) {
                uint8_t vv[WU];
                int yy[WU], xx[WU];
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    const int y = py, x1 = px1;
                    yy[u] = y;
                    xx[u] = x1;
                    uint8_t v = 0;
                    if (okinv && p0 + 64 * u < SZ * SZ) {
                        double X0 = Mi[0] * 0 + Mi[1] * y + Mi[2];
                        double Y0 = Mi[3] * 0 + Mi[4] * y + Mi[5];
                        double W0 = Mi[6] * 0 + Mi[7] * y + Mi[8];
                        double Wd = W0 + Mi[6] * x1;
                        Wd = Wd ? 1. / Wd : 0;
                        int X = sat_round_int((X0 + Mi[0] * x1) * Wd);
                        int Y = sat_round_int((Y0 + Mi[3] * x1) * Wd);
                        X = X < SHRT_MIN ? SHRT_MIN : (X > SHRT_MAX ? SHRT_MAX : X);
                        Y = Y < SHRT_MIN ? SHRT_MIN : (Y > SHRT_MAX ? SHRT_MAX : Y);
                        if ((unsigned)X < (unsigned)W && (unsigned)Y < (unsigned)H) v = g[(long long)Y * P.gstride + X];
                    }
                    vv[u] = v;
                    px1 += 64;  // the next pixel of this lane: 64 further along the row-major patch
                    while (px1 >= SZ) {
                        px1 -= SZ;
                        py++;
                    }
                }
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    const int p = p0 + 64 * u;
                    if (p >= SZ * SZ) break;
                    const uint8_t v = vv[u];
                    patch[p] = v;
                    atomicAdd(&hist[v], 1);
                    if (yy[u] >= in0 && yy[u] < in1 && xx[u] >= in0 && xx[u] < in1) {
                        ssum += v;
                        ssq += (int)v * (int)v;
                    }
                }
            }
        }
        ssum = wave_sum_i64(ssum);
        ssq = wave_sum_i64(ssq);
        __syncthreads();
        ID_T(2)
        // ---- meanStdDev on the inner region
        const int nin = (in1 - in0) * (in1 - in0);
        double scale = nin ? 1. / nin : 0.;
        double mean = ssum * scale;
        double var = ssq * scale - mean * mean;
        double stddev = sqrt(var > 0. ? var : 0.);
        int uniform_bits = -1;
        if (stddev < P.minOtsuStdDev) uniform_bits = mean > 127 ? 1 : 0;
        if (uniform_bits < 0) {
            // getThreshVal_Otsu_8u (the histogram in registers, four bins per lane)
            const int hreg[4] = {hist[lane], hist[lane + 64], hist[lane + 128], hist[lane + 192]};
            {
                const double sc = 1. / (SZ * SZ);
                // mu = sum of i * h_i, then * sc: every term and every partial sum is an integer below 2^53, so the double sum of the
                // loop is exact whatever its order -- an integer wave reduction gives the same number
                double mu = (double)wave_sum_i32(lane * hreg[0] + (64 + lane) * hreg[1] + (128 + lane) * hreg[2] + (192 + lane) * hreg[3]);
                mu *= sc;
                // getThreshVal_Otsu_8u's loop carries (q1, mu1) from bin to bin:  p_i = h_i * sc;  mu1 *= q1;  q1 += p_i;
                // skip the bin if q1 or 1 - q1 is within FLT_EPSILON of 0 / 1;  mu1 = (mu1 + i * p_i) / q1;  then mu2, sigma, the maximum.
                // As written that is 256 dependent steps of ~260 cycles (an IEEE f64 division each): 29 us of this kernel's 62.
                // Only two things are truly sequential, and both are short:
                //   1. q1: 256 dependent additions (their rounding depends on the order);
                //   2. mu1: per bin  a = mu1 * q1_prev + i * p_i  and the LAST three operations of the division a / q1_i --
                //      q0 = a * y,  r = fma(-q1_i, q0, a),  mu1 = fma(r, y, q0)  with y the twice-refined v_rcp_f64(q1_i).
                // Everything else hangs off q1 alone and is done for all bins side by side, four per lane: p_i, i * p_i, the skip
                // test (q1 rises, so the skipped bins are a stretch at each end and the loop of 2 runs over the bins between them
                // without a branch), the reciprocal y and its two refinements, and afterwards mu2, sigma and the arg-max.
                // The division is the compiler's own sequence (v_rcp_f64, two fma refinements, q0, r, fma) without v_div_scale /
                // v_div_fixup, which only act on operands near the ends of the exponent range: here 0 <= a <= 255 (a == 0 gives 0
                // through the same three operations) and FLT_EPSILON <= q1 <= 1.  Same operations on the same operands as the loop.
                // (the loops read arrays they do not write: distinct LDS objects, so that their loads can run ahead of the stores)
                double pk[4], qk[4];
                bool skipk[4];
                unsigned long long skipm[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    pk[k4] = hreg[k4] * sc;
                    s_pm[k4 * 64 + lane] = pk[k4];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (lane == 0) {  // (one lane, one branch around the whole loop: a test inside it keeps its loads from running ahead)
                    double q1 = 0;
#pragma unroll 16
                    for (int i = 0; i < 256; i++) {
                        q1 += s_pm[i];
                        s_q1[i] = q1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int i = k4 * 64 + lane;
                    const double q1i = s_q1[i], q2 = 1. - q1i;
                    qk[k4] = q1i;
                    skipk[k4] = fmin(q1i, q2) < FLT_EPSILON || fmax(q1i, q2) > 1. - FLT_EPSILON;
                    double y = __builtin_amdgcn_rcp(q1i);
                    double e = __builtin_fma(-q1i, y, 1.);
                    y = __builtin_fma(y, e, y);
                    e = __builtin_fma(-q1i, y, 1.);
                    y = __builtin_fma(y, e, y);
                    s_ny[i] = make_double2(i * pk[k4], y);
                    skipm[k4] = ballot64(skipk[k4]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                int i_lo = 256, i_hi = 256;  // the bins the loop does not skip: [i_lo, i_hi)
#pragma unroll
                for (int k4 = 3; k4 >= 0; k4--)
                    if (~skipm[k4]) i_lo = k4 * 64 + __ffsll((long long)~skipm[k4]) - 1;
#pragma unroll
                for (int k4 = 3; k4 >= 0; k4--) {
                    const int lo = i_lo - k4 * 64;  // skipped bins of this word at or behind i_lo
                    const unsigned long long m = lo >= 64 ? 0ull : (lo <= 0 ? skipm[k4] : skipm[k4] & ~((1ull << lo) - 1ull));
                    if (m) i_hi = k4 * 64 + __ffsll((long long)m) - 1;
                }
                if (lane == 0) {
                    double mu1 = 0;
#pragma unroll 8
                    for (int i = i_lo; i < i_hi; i++) {
                        const double qprev = i ? s_q1[i - 1] : 0., q1i = s_q1[i];
                        const double2 ny = s_ny[i];
                        const double t = mu1 * qprev;
                        const double a2 = t + ny.x;
                        const double q0 = a2 * ny.y;
                        const double r0 = __builtin_fma(-q1i, q0, a2);
                        mu1 = __builtin_fma(r0, ny.y, q0);
                        s_pm[i] = mu1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                double pm[4], pq[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const int i = k4 * 64 + lane;
                    pm[k4] = s_pm[i];
                    pq[k4] = (skipk[k4] || i >= i_hi) ? -1. : qk[k4];  // (q1 < 0: the loop skipped this bin)
                }
                double best = 0.;
                int bi = 0;
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const double q1t = pq[k4], mu1t = pm[k4];
                    const double q2 = 1. - q1t;
                    const double mu2 = (mu - q1t * mu1t) / q2;
                    const double sigma = q1t * q2 * (mu1t - mu2) * (mu1t - mu2);
                    if (q1t >= 0. && sigma > best) {  // (bins of a lane in rising order: the first one that holds the lane's maximum)
                        best = sigma;
                        bi = k4 * 64 + lane;
                    }
                }
                const double gmax = wave_max_nonneg_f64(best);
                const int first = wave_min_i32(best == gmax && gmax > 0. ? bi : INT_MAX);
                const double max_val = gmax > 0. ? (double)first : 0.;
                if (lane == 0) s_thr = (int)floor(max_val);
            }
            __syncthreads();
            ID_T(3)
            const int thr = s_thr;
            const int cs = cellSize - 2 * P.cellMargin;
            for (int c = lane; c < msb * msb; c += 64) {
                int cy = c / msb, cx = c - cy * msb;
                int Xs = cx * cellSize + P.cellMargin, Ys = cy * cellSize + P.cellMargin;
                int nz = 0;
                for (int yy = 0; yy < cs; yy++)
                    for (int xx = 0; xx < cs; xx++) nz += patch[(Ys + yy) * SZ + Xs + xx] > thr;
                cellbits[c] = (unsigned)nz > (unsigned)(cs * cs) / 2 ? 1 : 0;
            }
        } else {
            for (int c = lane; c < msb * msb; c += 64) cellbits[c] = (uint8_t)uniform_bits;
        }
        __syncthreads();
        ID_T(4)
        for (int c = lane; c < msb * msb; c += 64) out->bits[c] = cellbits[c];
        // ---- _getBorderErrors (every lane computes the same scalar answer from LDS)
        int borderErrors = 0;
        for (int y = 0; y < msb; y++)
            for (int kk = 0; kk < bb; kk++) {
                borderErrors += cellbits[y * msb + kk] != 0;
                borderErrors += cellbits[y * msb + msb - 1 - kk] != 0;
            }
        for (int x = bb; x < msb - bb; x++)
            for (int kk = 0; kk < bb; kk++) {
                borderErrors += cellbits[kk * msb + x] != 0;
                borderErrors += cellbits[(msb - 1 - kk) * msb + x] != 0;
            }
        int id = -1, rot = -1;
        if (borderErrors <= P.maxBorderErr) {
            // candidate bytes (Dictionary::getByteListFromBits, rotation 0), nbytes <= 8
            unsigned long long cw = 0;
            {
                const int nb2 = ms * ms;
                for (int r = 0, t = 0; r < ms; r++)  // (row and column carried along: a division per bit otherwise)
                    for (int c = 0; c < ms; c++, t++) {
                        int byte = t >> 3, q = t & 7;
                        int inbyte = nb2 - 8 * byte;
                        inbyte = inbyte > 8 ? 8 : inbyte;  // the last partial byte is right-aligned
                        unsigned long long bit = cellbits[(r + bb) * msb + c + bb];
                        cw |= bit << (8 * byte + (inbyte - 1 - q));
                    }
            }
            const int nbytes = P.nbytes;
            int bestm = INT_MAX, bestr = 0;
            // 64 markers a round, a lane each; four rounds' tables are fetched together -- with four bytes per rotation (5 x 5
            // markers) a marker's four rotations are one aligned 16-byte load -- and then compared round by round, the first
            // round with a hit ends the search (a round used to wait for sixteen byte loads of its own: id 238 of DICT_5X5_250
            // cost 23 k cycles, id 47 11 k)
            for (int m0 = 0; m0 < P.nMarkers && bestm == INT_MAX; m0 += 256) {
                uint4 tw4[4];
                if (nbytes == 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int mi = m0 + 64 * u + lane;
                        tw4[u] = mi < P.nMarkers ? reinterpret_cast<const uint4 *>(dict)[mi] : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int mi = m0 + 64 * u + lane;
                    int myr = -1;
                    if (mi < P.nMarkers) {
                        int cmin = ms * ms + 1;
                        for (int r = 0; r < 4; r++) {
                            unsigned long long tw = 0;
                            if (nbytes == 4) {
                                tw = r == 0 ? tw4[u].x : (r == 1 ? tw4[u].y : (r == 2 ? tw4[u].z : tw4[u].w));
                            } else {
                                const uint8_t *t = dict + ((long long)mi * 4 + r) * nbytes;
                                for (int q = 0; q < nbytes; q++) tw |= (unsigned long long)t[q] << (8 * q);
                            }
                            int ham = __popcll(tw ^ cw);
                            if (ham < cmin) {
                                cmin = ham;
                                myr = r;
                            }
                        }
                        if (cmin > P.maxCorr) myr = -1;
                    }
                    unsigned long long hit = ballot64(myr >= 0);
                    if (hit) {
                        int src = __ffsll((long long)hit) - 1;
                        bestm = m0 + 64 * u + src;
                        bestr = __shfl(myr, src, WAVE);
                        break;
                    }
                }
            }
            if (bestm != INT_MAX) {
                id = bestm;
                rot = bestr;
            }
        }
        if (lane == 0) {
            out->id = id;
            out->rot = rot;
        }
#ifdef ID_TIMING
        ID_T(5)
        if (lane == 0 && (wi & 63u) == 0) printf("identify cand %u: cycles homography %llu warp %llu otsu %llu cells %llu border+dictionary %llu (id %d)\n", wi, ti[1] - ti[0], ti[2] - ti[1], ti[3] - ti[2], ti[4] - ti[3], ti[5] - ti[4], id);
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// K7a: collect identified candidates in order, rotate corners (std::rotate by 4 - rotation) and apply
// _filterDetectedMarkers (aruco.cpp; pointPolygonTest from geometry.cpp).  One wave per frame.
__device__ __forceinline__ double point_polygon_test4(const float *cnt, float ptx, float pty)
{
    int counter = 0;
    float vx = cnt[6], vy = cnt[7], v0x, v0y;
    for (int i = 0; i < 4; i++) {
        v0x = vx;
        v0y = vy;
        vx = cnt[2 * i];
        vy = cnt[2 * i + 1];
        if ((v0y <= pty && vy <= pty) || (v0y > pty && vy > pty) || (v0x < ptx && vx < ptx)) {
            if (pty == vy && (ptx == vx || (pty == v0y && ((v0x <= ptx && ptx <= vx) || (vx <= ptx && ptx <= v0x))))) return 0;
            continue;
        }
        double dist = (double)(pty - v0y) * (vx - v0x) - (double)(ptx - v0x) * (vy - v0y);
        if (dist == 0) return 0;
        if (vy < v0y) dist = -dist;
        counter += dist > 0;
    }
    return counter % 2 == 0 ? -1 : 1;
}

__global__ __launch_bounds__(64) void k_filter_markers(const DevCand *__restrict__ filtered, const DevIdent *__restrict__ ident,
                                                        fid_marker *__restrict__ pre, DevCounts *__restrict__ counts,
                                                        const DevParams P, fid_marker *__restrict__ gscratch, int lds_cap,
                                                        int *__restrict__ accsrc, int *__restrict__ mksrc)
{
    // accsrc [maxCands] per frame (scratch) / mksrc [maxMarkers] per frame: the filtered candidate a marker came from -- its
    // contour is what CORNER_REFINE_CONTOUR fits (aruco.cpp carries `contours` beside `candidates` through both filters)
    // the identified markers of the frame: in LDS when they fit lds_cap (a few dozen do; the kernel used to ask for maxCands
    // entries = 73 KB per frame and waited for CUs with that much LDS free: 0.4 ms for 5 us of work), else in the frame's
    // slice of a global scratch array
    extern __shared__ fid_marker acc_lds[];
    const int f = blockIdx.x, lane = lane_id();
    int nf = counts[f].nfilt;
    const DevCand *cs = filtered + (long long)f * P.maxCands;
    const DevIdent *idn = ident + (long long)f * P.maxCands;
    int nhit = 0;
    for (int k0 = 0; k0 < nf; k0 += 64) nhit += __popcll(ballot64(k0 + lane < nf && idn[k0 + lane].id >= 0));
    fid_marker *acc = nhit <= lds_cap ? acc_lds : gscratch + (long long)f * P.maxCands;
    int base = 0;
    for (int k0 = 0; k0 < nf; k0 += 64) {
        int k = k0 + lane;
        int id = -1, rot = 0;
        if (k < nf) {
            id = idn[k].id;
            rot = idn[k].rot;
        }
        unsigned long long hit = ballot64(id >= 0);
        int off = base + __popcll(hit & ((1ull << lane) - 1ull));
        if (id >= 0) {
            fid_marker m;
            m.id = id;
            for (int c = 0; c < 4; c++) {
                int sc = (c + 4 - rot) & 3;
                m.corners[2 * c] = cs[k].c[2 * sc];
                m.corners[2 * c + 1] = cs[k].c[2 * sc + 1];
            }
            acc[off] = m;
            accsrc[(long long)f * P.maxCands + off] = k;
        }
        base += __popcll(hit);
    }
    const int nacc = base;
    __syncthreads();
    // toRemove flags: pure function of the pairs (the loops never test toRemove before comparing)
    int outbase = 0;
    for (int j0 = 0; j0 < nacc; j0 += 64) {
        int j = j0 + lane;
        int removed = 0;
        if (j < nacc) {
            for (int i = 0; i < nacc; i++) {
                if (i == j || acc[i].id != acc[j].id) continue;
                int a = i < j ? i : j, bq = i < j ? j : i;  // pair (a, b), a < b
                // first: is b inside a?  then: is a inside b?
                int b_in_a = 1;
                for (int q = 0; q < 4; q++)
                    if (point_polygon_test4(acc[a].corners, acc[bq].corners[2 * q], acc[bq].corners[2 * q + 1]) < 0) {
                        b_in_a = 0;
                        break;
                    }
                if (b_in_a) {
                    if (j == bq) removed = 1;
                    continue;
                }
                int a_in_b = 1;
                for (int q = 0; q < 4; q++)
                    if (point_polygon_test4(acc[bq].corners, acc[a].corners[2 * q], acc[a].corners[2 * q + 1]) < 0) {
                        a_in_b = 0;
                        break;
                    }
                if (a_in_b && j == a) removed = 1;
            }
        }
        int keep = j < nacc && !removed;
        unsigned long long hit = ballot64(keep);
        int off = outbase + __popcll(hit & ((1ull << lane) - 1ull));
        if (keep) {
            if (off < P.maxMarkers) {
                pre[(long long)f * P.maxMarkers + off] = acc[j];
                mksrc[(long long)f * P.maxMarkers + off] = accsrc[(long long)f * P.maxCands + j];
            }
        }
        outbase += __popcll(hit);
    }
    if (lane == 0) {
        counts[f].nacc = nacc;
        if (outbase > P.maxMarkers) {
            atomicOr(&counts[f].overflow, 2);
            outbase = P.maxMarkers;
        }
        counts[f].nmark = outbase;
    }
}

// K7b: cornerSubPix (cornersubpix.cpp) with getRectSubPix 8u->32f (samplers.cpp), one wave per corner.
// The (2w+3)^2 patch and the per-tap products are computed in parallel; each of the five accumulators is
// summed by its own lane in the reference's (i, j) order so the float corner equals the sequential result
// bit for bit.
#define SP_MAXWIN 7
__global__ __launch_bounds__(64) void k_subpix(const uint8_t *__restrict__ gray, long long gfstride,
                                                const fid_marker *__restrict__ pre, fid_marker *__restrict__ out,
                                                const DevCounts *__restrict__ counts, const float *__restrict__ maskw,
                                                const DevParams P)
{
    __shared__ float sp[(2 * SP_MAXWIN + 3) * (2 * SP_MAXWIN + 3)];
    // (rows padded to a multiple of 16 taps; the pad holds -0.0, the one value x + pad == x holds for bit for bit, for every x)
    constexpr int SP_NTP = ((2 * SP_MAXWIN + 1) * (2 * SP_MAXWIN + 1) + 15) & ~15;
    __shared__ __attribute__((aligned(16))) double prod[5][SP_NTP];
    // the gray pixels every iteration's patch can touch while the corner stays within win + 1 pixels of where it started
    // (further away the result is discarded anyway): fetched once, so that an iteration does not wait for global memory
    constexpr int SP_CMAX = (2 * SP_MAXWIN + 3) + 1 + 2 * (SP_MAXWIN + 1);
    __shared__ uint8_t s_img[SP_CMAX * SP_CMAX];
    __shared__ float s_c[2];
    __shared__ int s_flag;
    const int lane = lane_id();
    const int win = P.subpixWin, ww = 2 * win + 1, pw = ww + 2;
    const int W = P.W, H = P.H, gs = P.gstride;
    const int total = P.nframes * P.maxMarkers * 4;
    for (int t = ww * ww + lane; t < SP_NTP; t += 64)
#pragma unroll
        for (int q = 0; q < 5; q++) prod[q][t] = -0.0;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int f = item / (P.maxMarkers * 4), r = item % (P.maxMarkers * 4);
        int mk = r >> 2, cn = r & 3;
        if (mk >= counts[f].nmark) continue;
        const fid_marker *src = pre + (long long)f * P.maxMarkers + mk;
        fid_marker *dstm = out + (long long)f * P.maxMarkers + mk;
        const uint8_t *g = gray + (long long)f * gfstride;
        if (cn == 0 && lane == 0) dstm->id = src->id;
        const float cTx = src->corners[2 * cn], cTy = src->corners[2 * cn + 1];
        float cIx = cTx, cIy = cTy;
        if (P.refine != 1) {  // CORNER_REFINE_NONE; CORNER_REFINE_CONTOUR: k_refine_contour writes the corners afterwards
            if (lane == 0) {
                dstm->corners[2 * cn] = cTx;
                dstm->corners[2 * cn + 1] = cTy;
            }
            continue;
        }
        const int cm = win + 1, csz = pw + 1 + 2 * cm;
        const int X0 = (int)floorf(cTx - (pw - 1) * 0.5f) - cm, Y0 = (int)floorf(cTy - (pw - 1) * 0.5f) - cm;
        const bool cached = X0 >= 0 && Y0 >= 0 && X0 + csz <= W && Y0 + csz <= H;
        __syncthreads();  // (the previous item's readers of s_img are done)
        if (cached)
            for (int p = lane; p < csz * csz; p += 64) {
                const int i = p / csz, j = p - i * csz;
                s_img[p] = g[(long long)(Y0 + i) * gs + X0 + j];
            }
        // (the window weights of this lane's taps: read once, not once per iteration -- a global load in front of every
        //  iteration's products sat on the critical path)
        float wgt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wgt[k] = lane + 64 * k < ww * ww ? maskw[lane + 64 * k] : 0.f;
        constexpr int SP_PR = ((2 * SP_MAXWIN + 3) * (2 * SP_MAXWIN + 3) + 63) / 64;
        int poff[SP_PR], pjj[SP_PR];
#pragma unroll
        for (int r = 0; r < SP_PR; r++) {
            const int pp = lane + 64 * r;
            const int i = pp / pw, j = pp - i * pw;
            poff[r] = i * csz + j;
            pjj[r] = pp < pw * pw ? j : -1;
        }
        int iter = 0;
#ifdef SP_TIMING
        unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tq;
#define SP_T(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - tq; tq = now_; }
        tq = __builtin_readcyclecounter();
#else
#define SP_T(acc)
#endif
        for (;;) {
            // getRectSubPix(src, (pw, pw), cI) -> sp
            float cx = cIx - (pw - 1) * 0.5f, cy = cIy - (pw - 1) * 0.5f;
            int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
            __syncthreads();
            if (cached && ipx >= X0 && ipy >= Y0 && ipx + pw + 1 <= X0 + csz && ipy + pw + 1 <= Y0 + csz) {
                // the fast path below, pixels from the LDS copy (same arithmetic)
                float a = cx - ipx, b = cy - ipy;
                a = a > 0.0001f ? a : 0.0001f;
                float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
                double s = (1. - a) / a;
                const uint8_t *s0 = s_img + (ipy - Y0) * csz + (ipx - X0);
                // (the patch positions of this lane -- row, column, offset into the cached pixels -- do not change from iteration to
                //  iteration: computed once in front of the loop, poff / pjj; the same reads and the same arithmetic)
#pragma unroll
                for (int r = 0; r < SP_PR; r++) {
                    const int j = pjj[r];
                    if (j < 0) continue;
                    const uint8_t *q = s0 + poff[r];  // = sr + j
                    float t = a12 * q[1] + a22 * q[1 + csz];
                    float prev;
                    if (j == 0) {
                        prev = (1 - a) * (b1 * q[0] + b2 * q[csz]);
                    } else {
                        float tp = a12 * q[0] + a22 * q[csz];
                        prev = (float)(tp * s);
                    }
                    sp[lane + 64 * r] = prev + t;
                }
            } else if (0 <= ipx && ipx + pw < W && 0 <= ipy && ipy + pw < H) {
                // getRectSubPix_8u32f fast path: dst[j] = prev_j + t_j, prev_0 = (1-a)(b1 s[0] + b2 s[step]),
                // prev_j = (float)(t_{j-1} * s), t_j = a12 s[j+1] + a22 s[j+1+step]
                float a = cx - ipx, b = cy - ipy;
                a = a > 0.0001f ? a : 0.0001f;
                float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
                double s = (1. - a) / a;
                const uint8_t *s0 = g + (long long)ipy * gs + ipx;
                for (int p = lane; p < pw * pw; p += 64) {
                    int i = p / pw, j = p - i * pw;
                    const uint8_t *sr = s0 + (long long)i * gs;
                    float t = a12 * sr[j + 1] + a22 * sr[j + 1 + gs];
                    float prev;
                    if (j == 0) {
                        prev = (1 - a) * (b1 * sr[0] + b2 * sr[gs]);
                    } else {
                        float tp = a12 * sr[j] + a22 * sr[j + gs];
                        prev = (float)(tp * s);
                    }
                    sp[p] = prev + t;
                }
            } else {
                // getRectSubPix_Cn_ with replicated border (adjustRect semantics == clamped coordinates)
                float a = cx - ipx, b = cy - ipy;
                float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
                float b1 = 1.f - b, b2 = b;
                int inside = 0 <= ipx && ipx < W - pw && 0 <= ipy && ipy < H - pw;
                // emulate adjustRect: r = [rx, rw) x [ry, rh) are the columns/rows that interpolate normally
                int rx, ry, rw, rh;
                long long basey, basex;
                {
                    int x = ipx, y = ipy;
                    basey = 0; basex = 0;
                    if (y >= 0) { basey += y; ry = 0; } else ry = -y < pw ? -y : pw;
                    if (y + pw < H) rh = pw; else { rh = H - y - 1; if (rh < 0) { basey += rh; rh = 0; } }
                    if (x >= 0) { basex += x; rx = 0; } else rx = -x < pw ? -x : pw;
                    if (x + pw < W) rw = pw; else { rw = W - x - 1; if (rw < 0) { basex += rw; rw = 0; } }
                    basex -= rx;
                }
                for (int p = lane; p < pw * pw; p += 64) {
                    int i = p / pw, j = p - i * pw;
                    float v;
                    if (inside) {
                        const uint8_t *sr = g + (long long)(ipy + i) * gs + ipx;
                        v = sr[j] * a11 + sr[j + 1] * a12 + sr[j + gs] * a21 + sr[j + gs + 1] * a22;
                    } else {
                        // row pointer after i iterations of the reference loop: src advances only while
                        // ry <= row < rh
                        int adv = 0;
                        {
                            int lo = ry, hi = rh < i ? rh : i;  // rows r in [0, i) with r >= ry && r < rh advance
                            adv = hi > lo ? hi - lo : 0;
                        }
                        long long y0r = basey + adv;
                        long long y1r = (i < ry || i >= rh) ? y0r : y0r + 1;
                        const uint8_t *sr = g + y0r * gs + basex;
                        const uint8_t *sr2 = g + y1r * gs + basex;
                        if (j < rx) v = sr[rx] * b1 + sr2[rx] * b2;
                        else if (j < rw) v = sr[j] * a11 + sr[j + 1] * a12 + sr2[j] * a21 + sr2[j + 1] * a22;
                        else v = sr[rw] * b1 + sr2[rw] * b2;
                    }
                    sp[p] = v;
                }
            }
            __syncthreads();
            SP_T(tp0)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = lane + 64 * k;
                if (t >= ww * ww) break;
                int i = t / ww, j = t - i * ww;
                const float *q = sp + (i + 1) * pw + (j + 1);
                double m = wgt[k];
                double tgx = q[1] - q[-1];
                double tgy = q[pw] - q[-pw];
                double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                double px = j - win, py = i - win;
                prod[0][t] = gxx;
                prod[1][t] = gxy;
                prod[2][t] = gyy;
                prod[3][t] = gxx * px + gxy * py;
                prod[4][t] = gxy * px + gyy * py;
            }
            __syncthreads();
            SP_T(tp1)
            // the five accumulators are summed in the reference's tap order, one accumulator per lane (0..4)
            double accv = 0;
            if (lane < 5) {
                // in tap order, as the reference accumulates; sixteen LDS reads are issued together, then added one after
                // the other (one read per add left every add waiting for LDS: 6 of the 7 us an iteration took)
                // (round 4: whole groups only -- the row's pad is -0.0 -- so that an addition is ONE dependent instruction; the
                //  predicated form `if (t < nt) acc += v` was a compare and two selects on top of every add: 5 700 of an
                //  iteration's 9 000 cycles went into these sums)
                const double *pr = prod[lane];
                const int nt = ww * ww;
                for (int t0 = 0; t0 < nt; t0 += 16) {
                    double v[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) v[k] = pr[t0 + k];
#pragma unroll
                    for (int k = 0; k < 16; k++) accv += v[k];
                }
            }
            const double a = bcast_f64(accv, 0), b = bcast_f64(accv, 1), c = bcast_f64(accv, 2);
            const double bb1 = bcast_f64(accv, 3), bb2 = bcast_f64(accv, 4);
            SP_T(tp2)
            if (lane == 0) {
                int flag = 0;  // 0 continue, 1 stop
                double det = a * c - b * b;
                if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) {
                    flag = 1;
                } else {
                    double scale = 1.0 / det;
                    float nx = (float)(cIx + c * scale * bb1 - b * scale * bb2);
                    float ny = (float)(cIy - b * scale * bb1 + a * scale * bb2);
                    double err = (nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy);
                    s_c[0] = nx;
                    s_c[1] = ny;
                    if (nx < 0 || nx >= W || ny < 0 || ny >= H) flag = 1;
                    else if (!(iter + 1 < P.subpixMaxIter && err > P.subpixEps)) flag = 1;
                    flag |= 2;  // moved
                }
                s_flag = flag;
            }
            __syncthreads();
            int flag = s_flag;
            if (flag & 2) {
                cIx = s_c[0];
                cIy = s_c[1];
            }
            iter++;
            SP_T(tp3)
            if (flag & 1) break;
        }
#ifdef SP_TIMING
        if (lane == 0 && iter >= 20) printf("subpix corner: %d iterations; cycles per iteration: patch %llu products %llu sums %llu solve+sync %llu\n", iter, tp0 / iter, tp1 / iter, tp2 / iter, tp3 / iter);
#endif
        if (fabsf(cIx - cTx) > win || fabsf(cIy - cTy) > win) {
            cIx = cTx;
            cIy = cTy;
        }
        if (lane == 0) {
            dstm->corners[2 * cn] = cIx;
            dstm->corners[2 * cn + 1] = cIy;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7c: CORNER_REFINE_CONTOUR -- aruco.cpp (4.2.0) _refineCandidateLines / _interpolate2Dline / _getCrossPoint, what the node selects
// with doCornerRefinement = true, cornerRefinementSubPix = false (aruco_detect.cpp:274-283, :700-711).  One wave per marker.
// The contour of the marker's candidate is still in HBM where the tracing left it (the dense point array of the traced modes, the
// chunk pool of the whole-border walk): the wave reads it 64 points at a time and sorts every point into the group of the corner
// that precedes it on the contour -- a corner is a contour point, so the match is a compare of packed words and the group of a
// lane is the corner of the highest matching lane at or below it (four ballots), carried from chunk to chunk; the points in front
// of the first corner go to the last one's group, as the reference appends them.  A side's least-squares line needs n, sum t,
// sum t^2, sum v, sum t v over its points (t the coordinate with the larger extent): sums of products of pixel coordinates,
// EXACT in integers whatever the order -- per-lane partial sums, DPP reductions -- and the reference's float arithmetic starts
// where its own does: cv::solve(DECOMP_NORMAL) rounds those sums (accumulated in double) to float once each, hal::LU32f solves
// the 2 x 2 system (lanes 0..3, one side each), Matx22f::solve crosses adjacent lines (a lane per corner).  A side of exactly
// two points skips the normal equations in cv::solve (m == n): LU32f then pivots the row with the larger t to the top, so the
// system is rebuilt in that order from the sums (t_hi, t_lo = the extent; v from sum v and sum t v) and the result is the
// reference's whichever order the points came in.  A side of one point makes cv::solve throw: the frame is flagged
// (DevCounts::overflow bit 2) and the call reports FID_E_CV_EXCEPTION for it, as the node publishes nothing (:391-393).
struct RefineSide {  // exact sums of one side, wave-uniform after the reductions
    int n, sx, sy, minx, maxx, miny, maxy;
    long long sxx, syy, sxy;
};

__device__ __forceinline__ int wave_max_i32(int v)
{
#define S_(C, M)                                  \
    {                                             \
        const int o = FID_DPP(INT_MIN, v, C, M);  \
        v = o > v ? o : v;                        \
    }
    FID_DPP_SCAN_STEPS(S_)
#undef S_
    return __builtin_amdgcn_readlane(v, 63);
}

// hal::LU32f (LUImpl<float>, eps = FLT_EPSILON * 10) on a 2 x 2 system with one right-hand side; false: singular
__device__ __forceinline__ bool lu32f_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float &x0, float &x1)
{
    const float eps = FLT_EPSILON * 10;
    if (fabsf(a10) > fabsf(a00)) {
        float t = a00; a00 = a10; a10 = t;
        t = a01; a01 = a11; a11 = t;
        t = b0; b0 = b1; b1 = t;
    }
    if (fabsf(a00) < eps) return false;
    const float d = -1 / a00;
    const float alpha = a10 * d;
    a11 += alpha * a01;
    b1 += alpha * b0;
    if (fabsf(a11) < eps) return false;
    b1 = b1 / a11;
    float s = b0;
    s -= a01 * b1;
    x0 = s / a00;
    x1 = b1;
    return true;
}

// PTS(k, in): packed point k (x | y << 16) of the contour, k in [0, count); asked for k = base + lane, base = 0, 64, 128, ... in turn
template <typename PTS>
__device__ __forceinline__ bool refine_candidate_lines_wave(PTS pts, int count, const float cin[8], float &ox, float &oy)
{
    const int lane = lane_id();
    uint32_t ck[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ck[j] = (uint32_t)(int)cin[2 * j] | ((uint32_t)(int)cin[2 * j + 1] << 16);
    // per-lane partial sums of the five groups (4: in front of the first corner)
    int pn[5], psx[5], psy[5], pminx[5], pmaxx[5], pminy[5], pmaxy[5];
    long long pxx[5], pyy[5], pxy[5];
#pragma unroll
    for (int g = 0; g < 5; g++) {
        pn[g] = psx[g] = psy[g] = 0;
        pminx[g] = pminy[g] = INT_MAX;
        pmaxx[g] = pmaxy[g] = INT_MIN;
        pxx[g] = pyy[g] = pxy[g] = 0;
    }
    int carry = 4;                        // group of the last corner met so far (wave-uniform)
    int ci0 = -1, ci1 = 0, ci2 = 0, ci3 = 0;  // int cornerIndex[4] = {-1}: as the reference writes it
    for (int base = 0; base < count; base += 64) {
        const int k = base + lane;
        const bool in = k < count;
        const uint32_t pk = pts(k, in);  // (every lane asks: the chain-code sources sum their steps across the wave)
        const uint32_t p = in ? pk : 0xffffffffu;
        const unsigned long long m0 = ballot64(in && p == ck[0]), m1 = ballot64(in && p == ck[1]), m2 = ballot64(in && p == ck[2]),
                                 m3 = ballot64(in && p == ck[3]);
        const unsigned long long mall = m0 | m1 | m2 | m3;
        if (m0) ci0 = base + 63 - __clzll((long long)m0);
        if (m1) ci1 = base + 63 - __clzll((long long)m1);
        if (m2) ci2 = base + 63 - __clzll((long long)m2);
        if (m3) ci3 = base + 63 - __clzll((long long)m3);
        const unsigned long long below = mall & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        int grp = carry;
        if (below) {
            const int top = 63 - __clzll((long long)below);
            // (a pixel that equals two corners cannot happen: the four corners are distinct points -- the last j wins as in the reference)
            grp = ((m3 >> top) & 1ull) ? 3 : (((m2 >> top) & 1ull) ? 2 : (((m1 >> top) & 1ull) ? 1 : 0));
        }
        if (mall) {
            const int top = 63 - __clzll((long long)mall);
            carry = ((m3 >> top) & 1ull) ? 3 : (((m2 >> top) & 1ull) ? 2 : (((m1 >> top) & 1ull) ? 1 : 0));
        }
        if (in) {
            const int x = (int)(p & 0xffffu), y = (int)(p >> 16);
            const long long xx = (long long)x * x, yy = (long long)y * y, xy = (long long)x * y;
#pragma unroll
            for (int g = 0; g < 5; g++) {
                const bool h = grp == g;
                pn[g] += h ? 1 : 0;
                psx[g] += h ? x : 0;
                psy[g] += h ? y : 0;
                pxx[g] += h ? xx : 0;
                pyy[g] += h ? yy : 0;
                pxy[g] += h ? xy : 0;
                pminx[g] = h && x < pminx[g] ? x : pminx[g];
                pmaxx[g] = h && x > pmaxx[g] ? x : pmaxx[g];
                pminy[g] = h && y < pminy[g] ? y : pminy[g];
                pmaxy[g] = h && y > pmaxy[g] ? y : pmaxy[g];
            }
        }
    }
    if (carry == 4) return false;  // no corner on the contour (cannot happen: approxPolyDP picks contour points)
    // reductions; lane g (0..3) keeps side g
    RefineSide mine = {0, 0, 0, INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0, 0, 0}, extra = mine;
#pragma unroll
    for (int g = 0; g < 5; g++) {
        RefineSide t;
        t.n = wave_sum_i32(pn[g]);
        t.sx = wave_sum_i32(psx[g]);
        t.sy = wave_sum_i32(psy[g]);
        t.sxx = wave_sum_i64(pxx[g]);
        t.syy = wave_sum_i64(pyy[g]);
        t.sxy = wave_sum_i64(pxy[g]);
        t.minx = wave_min_i32(pminx[g]);
        t.maxx = wave_max_i32(pmaxx[g]);
        t.miny = wave_min_i32(pminy[g]);
        t.maxy = wave_max_i32(pmaxy[g]);
        if (g == 4) extra = t;
        else if (lane == g) mine = t;
    }
    if (lane == carry && extra.n) {  // "saves extra group into corresponding"
        mine.n += extra.n;
        mine.sx += extra.sx;
        mine.sy += extra.sy;
        mine.sxx += extra.sxx;
        mine.syy += extra.syy;
        mine.sxy += extra.sxy;
        mine.minx = extra.minx < mine.minx ? extra.minx : mine.minx;
        mine.maxx = extra.maxx > mine.maxx ? extra.maxx : mine.maxx;
        mine.miny = extra.miny < mine.miny ? extra.miny : mine.miny;
        mine.maxy = extra.maxy > mine.maxy ? extra.maxy : mine.maxy;
    }
    // _interpolate2Dline, lanes 0..3
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    bool bad = false;
    if (lane < 4) {
        if (mine.n < 2) {
            bad = true;
        } else {
            const bool xm = (float)mine.maxx - (float)mine.minx > (float)mine.maxy - (float)mine.miny;
            float a00, a01, a10, a11, b0, b1;
            if (mine.n == 2) {
                // m == n: no normal equations.  LU32f puts the row with the larger |t| first (strict compare; equal t: singular)
                const int thi = xm ? mine.maxx : mine.maxy, tlo = xm ? mine.minx : mine.miny;
                const long long sv = xm ? mine.sy : mine.sx;
                long long vhi, vlo;
                if (thi != tlo) {
                    vhi = (mine.sxy - (long long)tlo * sv) / (long long)(thi - tlo);
                    vlo = sv - vhi;
                } else {
                    vhi = xm ? mine.miny : mine.minx;  // (both points share t: the system is singular whatever v is)
                    vlo = sv - vhi;
                }
                a00 = (float)thi; a01 = 1.f; a10 = (float)tlo; a11 = 1.f;
                b0 = (float)vhi; b1 = (float)vlo;
            } else {
                const long long st = xm ? mine.sx : mine.sy, stt = xm ? mine.sxx : mine.syy, sv = xm ? mine.sy : mine.sx;
                a00 = (float)(double)stt;
                a01 = a10 = (float)(double)st;
                a11 = (float)(double)mine.n;
                b0 = (float)(double)mine.sxy;
                b1 = (float)(double)sv;
            }
            float c0 = 0.f, c1 = 0.f;
            if (!lu32f_2x2(a00, a01, a10, a11, b0, b1, c0, c1)) c0 = c1 = 0.f;  // if( !result ) dst = Scalar(0)
            if (xm) { l0 = c0; l1 = -1.f; l2 = c1; }
            else    { l0 = -1.f; l1 = c0; l2 = c1; }
        }
    }
    if (ballot64(bad)) return false;
    int inc = 1;
    inc = ((ci0 > ci1) && (ci3 > ci0)) ? -1 : inc;
    inc = ((ci2 > ci3) && (ci1 > ci2)) ? -1 : inc;
    // _getCrossPoint(lines[i], lines[(i + 1) % 4]) (inc < 0) or (lines[i], lines[(i + 3) % 4]): lane i = corner i
    const int other = (lane + (inc < 0 ? 1 : 3)) & 3;
    const float m0 = __shfl(l0, other, WAVE), m1 = __shfl(l1, other, WAVE), m2 = __shfl(l2, other, WAVE);
    {
        const float a00 = l0, a01 = l1, a10 = m0, a11 = m1;
        const float b0 = -l2, b1 = -m2;
        float d = a00 * a11 - a01 * a10;
        if (d == 0) {
            ox = oy = 0.f;
        } else {
            d = 1 / d;
            ox = (b0 * a11 - b1 * a01) * d;
            oy = (b1 * a00 - b0 * a10) * d;
        }
    }
    return true;
}

struct RefinePtsDense {  // caller-supplied points (fid_refine_contour_corners)
    const uint32_t *p;
    __device__ __forceinline__ uint32_t operator()(int k, bool in) const { return in ? p[k] : 0u; }
};
// the tracing leaves chain codes: point k = the contour's first point + the steps in front of it (a wave scan per 64 points,
// the running point carried from trip to trip)
struct RefinePtsCodes {  // the traced modes' dense array: one code byte per point
    const uint8_t *p;
    uint32_t carry;
    __device__ __forceinline__ uint32_t step(uint32_t d)
    {
        const uint32_t incl = wave_iscan_dpp(d);
        const uint32_t pt = carry + incl - d;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        return pt;
    }
    __device__ __forceinline__ uint32_t operator()(int k, bool in) { return step(in ? code_delta(p[k]) : 0u); }
};
struct RefinePtsChunks : RefinePtsCodes {  // the whole-border walk (FID_TRACE=legacy): code words in the pool chunks of the contour's table row
    const uint32_t *row, *pool;  // row: entry 0 of the contour's row; entry k is `pitch` words on
    long long pitch;
    __device__ __forceinline__ uint32_t operator()(int k, bool in)
    {
        return step(in ? code_delta(pool[(long long)row[(k / CK) * pitch] * CKW + ((k & (CK - 1)) >> 3)] >> ((k & 7) * 4)) : 0u);
    }
};

#define REFINE_LEGACY_BIT 0x80000000u
__global__ __launch_bounds__(64) void k_refine_contour(const fid_marker *__restrict__ pre, fid_marker *__restrict__ out,
                                                        const int *__restrict__ mksrc, const DevCand *__restrict__ filtered,
                                                        const uint32_t *__restrict__ dense, const uint32_t *__restrict__ chunk_tab,
                                                        const uint32_t *__restrict__ pool, DevCounts *__restrict__ counts, const DevParams P)
{
    const int lane = lane_id();
    const int f = blockIdx.y;
    int nm = counts[f].nmark;
    nm = nm < P.maxMarkers ? nm : P.maxMarkers;
    for (int mk = blockIdx.x; mk < nm; mk += gridDim.x) {
        const fid_marker *src = pre + (long long)f * P.maxMarkers + mk;
        fid_marker *dstm = out + (long long)f * P.maxMarkers + mk;
        const DevCand *cd = filtered + (long long)f * P.maxCands + mksrc[(long long)f * P.maxMarkers + mk];
        const int count = cd->size;
        const uint32_t pref = cd->pref;
        const uint32_t first = (uint32_t)cd->sx | ((uint32_t)cd->sy << 16);  // the contour's first point: where its chain codes start
        float cin[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cin[k] = src->corners[k];
        float ox = 0.f, oy = 0.f;
        bool ok;
        if (pref & REFINE_LEGACY_BIT) {
            RefinePtsChunks pts;
            pts.p = nullptr;
            pts.carry = first;
            pts.row = chunk_tab + chunk_tab_at(P, f, (unsigned)P.maxContours + (pref & ~REFINE_LEGACY_BIT), 0u);
            pts.pitch = 2ll * P.maxContours;
            pts.pool = pool;
            ok = refine_candidate_lines_wave(pts, count, cin, ox, oy);
        } else {
            RefinePtsCodes pts{reinterpret_cast<const uint8_t *>(dense) + (long long)f * P.maxChunks * CK + pref, first};
            ok = refine_candidate_lines_wave(pts, count, cin, ox, oy);
        }
        if (!ok) {
            if (lane == 0) atomicOr(&counts[f].overflow, 4);
            ox = cin[2 * (lane & 3)];
            oy = cin[2 * (lane & 3) + 1];
        }
        if (lane == 0) dstm->id = src->id;
        if (lane < 4) {
            dstm->corners[2 * lane] = ox;
            dstm->corners[2 * lane + 1] = oy;
        }
    }
}

// the same on caller-supplied contours (fid_refine_contour_corners: one marker per workgroup, item i = points [off[i], off[i + 1]))
__global__ __launch_bounds__(64) void k_refine_contour_pts(const uint32_t *__restrict__ pts, const int *__restrict__ off,
                                                            float *__restrict__ corners, int *__restrict__ status)
{
    const int i = blockIdx.x, lane = lane_id();
    float cin[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cin[k] = corners[8 * i + k];
    float ox = 0.f, oy = 0.f;
    RefinePtsDense src{pts + off[i]};
    const bool ok = refine_candidate_lines_wave(src, off[i + 1] - off[i], cin, ox, oy);
    if (lane == 0) status[i] = ok ? 0 : 1;
    if (ok && lane < 4) {
        corners[8 * i + 2 * lane] = ox;
        corners[8 * i + 2 * lane + 1] = oy;
    }
}

// ------------------------------------------------------------------------------------------------
// K8: per-marker pose = cv::solvePnP(SOLVEPNP_ITERATIVE) for the 4 coplanar marker corners as
// calibration.cpp cvFindExtrinsicCameraParams2 does it: undistort (5 fixed-point iterations) -> homography
// between the marker square and the normalised image points -> R,t -> Levenberg-Marquardt on the distorted
// reprojection error with CvLevMarq's state machine (lambda 1e-3, x10 / /10, 20 iterations, eps FLT_EPSILON,
// analytic Jacobians of cvProjectPoints2 / cvRodrigues2), followed by getReprojectionError /
// calcFiducialArea / object_error of aruco_detect.cpp:203-221,179-200,493-495.
// Eight lanes per marker: lane g owns residual g (corner g>>1, x or y); J^T J is reduced with xor-shuffles.
// What is only an initial guess or a damped linear solve is computed in closed form (square-to-quad
// homography instead of the 9x9 DLT eigenproblem, LDL^T instead of SVD back-substitution): the converged
// minimum is what is compared (tolerance in tests/, measured ~1e-12).
__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = __shfl_xor((unsigned)u, mask, WAVE);
    unsigned hi = __shfl_xor((unsigned)(u >> 32), mask, WAVE);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// a double from the lane a DPP control names (two v_mov_b32 with a DPP operand: no trip through the LDS crossbar)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// sum over an aligned group of eight lanes, in every lane of the group.  The same three additions with the same operands as the
// xor-shuffle form (lane ^ 1: quad_perm [1,0,3,2]; lane ^ 2: quad_perm [2,3,0,1]; the other quad of the group: row_half_mirror
// -- after the second step every lane of a quad holds the same value, so lane 7 - i serves as well as lane i ^ 4), without the
// six ds_bpermute round trips: Levenberg-Marquardt reduces 28 such sums per iteration, 84 dependent LDS latencies that were
// most of k_pose's time.
__device__ __forceinline__ double grp_sum8(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    return v;
}

struct PoseCam {
    double K[9];
    double D[5];
    double fiducial_len;
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(K); v(D); v(fiducial_len);
    }
};
// CvLevMarq's damping factor exp(lambdaLg10 * log(10.)) for lambdaLg10 = -16 .. 16 as the HOST's libm gives it (glibc's exp / log,
// what the reference runs on; generated with Python's math.exp(k * math.log(10.0)), hexadecimal literals = the exact doubles):
// a table look-up instead of a device exp() in every Levenberg-Marquardt step -- and the reference's values, not the device
// library's.
__device__ __forceinline__ double lm_lambda(int lg10)
{
    static const double t[33] = {0x1.cd2b297d889a0p-54, 0x1.203af9ee755f8p-50, 0x1.6849b86a12b93p-47, 0x1.c25c268497664p-44, 0x1.19799812dea04p-40, 0x1.5fd7fe179648cp-37, 0x1.b7cdfd9d7bd9cp-34, 0x1.12e0be826d687p-30, 0x1.5798ee2308c2fp-27, 0x1.ad7f29abcaf44p-24, 0x1.0c6f7a0b5ed87p-20, 0x1.4f8b588e368e5p-17, 0x1.a36e2eb1c4326p-14, 0x1.0624dd2f1a9f9p-10, 0x1.47ae147ae1478p-7, 0x1.9999999999998p-4, 0x1.0000000000000p+0, 0x1.4000000000001p+3, 0x1.9000000000003p+6, 0x1.f400000000006p+9, 0x1.3880000000005p+13, 0x1.86a000000000ep+16, 0x1.e84800000000bp+19, 0x1.312d000000003p+23, 0x1.7d7840000000cp+26, 0x1.dcd6500000018p+29, 0x1.2a05f20000015p+33, 0x1.74876e800000ap+36, 0x1.d1a94a2000015p+39, 0x1.2309ce5400013p+43, 0x1.6bcc41e900008p+46, 0x1.c6bf52634002fp+49, 0x1.1c37937e08011p+53};
    lg10 = lg10 < -16 ? -16 : (lg10 > 16 ? 16 : lg10);
    return t[lg10 + 16];
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi, fully unrolled (static register indexing)
__device__ __forceinline__ void jacobi3(double A[3][3], double V[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1. : 0.;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-60 * dg || off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2. * apq);
                double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                double c = 1. / sqrt(t * t + 1.), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double vpk = V[p][k], vqk = V[q][k];
                    V[p][k] = c * vpk - s * vqk;
                    V[q][k] = s * vpk + c * vqk;
                }
            }
    }
}

// R <- U * Vt of its SVD  ( = R * (RtR)^(-1/2) ), as cvRodrigues2 does before reading the axis
__device__ __forceinline__ void orthonormalize3(double R[9])
{
    double A[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[i][j] = R[i] * R[j] + R[3 + i] * R[3 + j] + R[6 + i] * R[6 + j];
    jacobi3(A, V);
    double Pm[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Pm[i][j] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double w = A[k][k];
        double is = w > 1e-300 ? 1. / sqrt(w) : 0.;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Pm[i][j] += V[k][i] * V[k][j] * is;
    }
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i * 3 + j] = R[i * 3] * Pm[0][j] + R[i * 3 + 1] * Pm[1][j] + R[i * 3 + 2] * Pm[2][j];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = T[i];
}

__device__ __forceinline__ void rodrigues_m2v(const double Rin[9], double r[3])
{
    double R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = Rin[i];
    orthonormalize3(R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

// cvRodrigues2 vector -> matrix with dR/dr (J[i*9+k] = dR_k / dr_i)
__device__ __forceinline__ void rodrigues_v2m(const double r_in[3], double R[9], double J[27], bool wantJ)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
        if (wantJ) {
#pragma unroll
            for (int i = 0; i < 27; i++) J[i] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c, s;
    sincos(theta, &s, &c);  // (one argument reduction for the pair)
    const double c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k] + s * r_x[k];
    if (wantJ) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double drrt[27] = {rx + rx, ry, rz, ry, 0,       0,  rz, 0,  0,       0, rx, 0, rx, ry + ry,
                                 rz,      0,  rz, 0,  0,       0,  rx, 0,  0,       ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
#pragma unroll
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

// cvProjectPoints2Internal for ONE object point and ONE image coordinate (sel = 0: x, 1: y);
// Jrow[0..2] = d/d rvec, Jrow[3..5] = d/d tvec
__device__ __forceinline__ double project_one(const double M[3], const double param[6], const double K[9], const double k[5],
                                               int sel, double Jrow[6], bool wantJ)
{
    double R[9], dRdr[27];
    rodrigues_v2m(param, R, dRdr, wantJ);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double X = M[0], Y = M[1], Z = M[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + param[3];
    double y = R[3] * X + R[4] * Y + R[5] * Z + param[4];
    double z = R[6] * X + R[7] * Y + R[8] * Z + param[5];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    const double icdist2 = 1.;
    double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
    double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
    double out = sel == 0 ? xd * fx + cx : yd * fy + cy;
    if (wantJ) {
        const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double dr2dt = 2 * x * dxdt[j] + 2 * y * dydt[j];
            double dcdist_dt = k[0] * dr2dt + 2 * k[1] * r2 * dr2dt + 3 * k[4] * r4 * dr2dt;
            double da1dt = 2 * (x * dydt[j] + y * dxdt[j]);
            double dmxdt = (dxdt[j] * cdist * icdist2 + x * dcdist_dt * icdist2 + k[2] * da1dt + k[3] * (dr2dt + 4 * x * dxdt[j]));
            double dmydt = (dydt[j] * cdist * icdist2 + y * dcdist_dt * icdist2 + k[2] * (dr2dt + 4 * y * dydt[j]) + k[3] * da1dt);
            Jrow[3 + j] = sel == 0 ? fx * dmxdt : fy * dmydt;
        }
        const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                                 X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
        const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                                 X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
        const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                                 X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
            double dydr = z * (dy0dr[j] - y * dz0dr[j]);
            double dr2dr = 2 * x * dxdr + 2 * y * dydr;
            double dcdist_dr = (k[0] + 2 * k[1] * r2 + 3 * k[4] * r4) * dr2dr;
            double da1dr = 2 * (x * dydr + y * dxdr);
            double dmxdr = (dxdr * cdist * icdist2 + x * dcdist_dr * icdist2 + k[2] * da1dr + k[3] * (dr2dr + 4 * x * dxdr));
            double dmydr = (dydr * cdist * icdist2 + y * dcdist_dr * icdist2 + k[2] * (dr2dr + 4 * y * dydr) + k[3] * da1dr);
            Jrow[j] = sel == 0 ? fx * dmxdr : fy * dmydr;
        }
    }
    return out;
}

// solve (JtJ with its diagonal scaled by 1 + lambda) x = JtErr, JtJ symmetric positive definite (packed upper
// triangle, row-major: index of (a, b), a <= b, is a*6 - a*(a-1)/2 + (b - a)); LDL^T, unrolled
__device__ __forceinline__ void solve6_spd(const double S[21], const double g[6], double lambda, double x[6])
{
    double A[6][6];
    {
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) {
                A[a][b] = S[idx];
                A[b][a] = S[idx];
                idx++;
            }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) A[i][i] *= 1. + lambda;
    double L[6][6], Dg[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * Dg[k];
        Dg[j] = d;
        double id = d != 0. ? 1. / d : 0.;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * Dg[k];
            L[i][j] = v * id;
        }
    }
    double yv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = g[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i][k] * yv[k];
        yv[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = Dg[i] != 0. ? yv[i] / Dg[i] : 0.;
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k];
        x[i] = v;
    }
}

__device__ double dist2f_d(float x1f, float y1f, float x2f, float y2f)
{
    double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
    double dx = x1 - x2, dy = y1 - y2;
    return sqrt(dx * dx + dy * dy);
}

__global__ __launch_bounds__(64) void k_pose(const fid_marker *__restrict__ markers, const int *__restrict__ nmark_per_frame,
                                              int nmark_stride_ints, const double *__restrict__ lens, int nframes,
                                              int per_frame, PoseCam cam, fid_pose_out *__restrict__ out)
{
    const int total = nframes * per_frame;
    const int g = threadIdx.x & 7;       // residual index inside the marker's lane group
    const int pi = g >> 1, sel = g & 1;  // corner, coordinate
    const int groups_per_block = blockDim.x >> 3;
    for (int item0 = blockIdx.x * groups_per_block; item0 < total; item0 += gridDim.x * groups_per_block) {
        const int item = item0 + (threadIdx.x >> 3);
        bool live = item < total;
        int f = 0, kk = 0;
        if (live) {
            f = item / per_frame;
            kk = item - f * per_frame;
            live = kk < nmark_per_frame[(long long)f * nmark_stride_ints];
        }
        if (!live) continue;  // group-uniform
        const fid_marker mk = markers[item];
        const double len = lens ? lens[item] : cam.fiducial_len;
        const double *K = cam.K, *kd = cam.D;
        const float ml = (float)len;
        const float hx = ml / 2.f;
        // object point of this lane: (-h, h), (h, h), (h, -h), (-h, -h)   aruco_detect.cpp:151-161
        const double M[3] = {(double)((pi == 1 || pi == 2) ? hx : -hx), (double)((pi < 2) ? hx : -hx), 0.};
        const double mobs = (double)mk.corners[g];
        // ---- cvUndistortPoints (5 iterations) on every corner (each lane needs all four for the homography)
        double mnx[4], mny[4];
        {
            const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double x = mk.corners[2 * i], y = mk.corners[2 * i + 1], u = x, v = y;
                x = (x - cx) * ifx;
                y = (y - cy) * ify;
                const double x0 = x, y0 = y;
                for (int j = 0; j < 5; j++) {
                    double r2 = x * x + y * y;
                    double icdist = (1) / (1 + ((kd[4] * r2 + kd[1]) * r2 + kd[0]) * r2);
                    if (icdist < 0) {
                        x = (u - cx) * ifx;
                        y = (v - cy) * ify;
                        break;
                    }
                    double deltaX = 2 * kd[2] * x * y + kd[3] * (r2 + 2 * x * x);
                    double deltaY = kd[2] * (r2 + 2 * y * y) + 2 * kd[3] * x * y;
                    x = (x0 - deltaX) * icdist;
                    y = (y0 - deltaY) * icdist;
                }
                // findHomography converts its inputs to float
                mnx[i] = (double)(float)x;
                mny[i] = (double)(float)y;
            }
        }
        double param[6];
        {
            // homography marker plane -> normalised image: unit square (0,0),(1,0),(1,1),(0,1) -> quad (Heckbert),
            // composed with (X, Y) -> ((X + h) / 2h, (h - Y) / 2h)
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
                const double hq = (double)hx;
                const double s = 1. / (2. * hq);
                // H = Hunit * [[s, 0, .5], [0, -s, .5], [0, 0, 1]]
                h[0] = a * s;  h[1] = -b * s;  h[2] = 0.5 * a + 0.5 * b + c;
                h[3] = d * s;  h[4] = -e * s;  h[5] = 0.5 * d + 0.5 * e + ff;
                h[6] = gg * s; h[7] = -hh * s; h[8] = 0.5 * gg + 0.5 * hh + 1.;
                okh = h[8] != 0.;
                if (okh) {
                    const double sc = 1. / h[8];
#pragma unroll
                    for (int i = 0; i < 9; i++) h[i] *= sc;
                }
            }
            double R[9];
            param[3] = param[4] = param[5] = 0.;
            if (okh) {
                const double h1_norm = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
                const double h2_norm = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
                const double s1 = 1. / fmax(h1_norm, DBL_EPSILON), s2 = 1. / fmax(h2_norm, DBL_EPSILON);
                const double stt = 2. / fmax(h1_norm + h2_norm, DBL_EPSILON);
                param[3] = h[2] * stt;
                param[4] = h[5] * stt;
                param[5] = h[8] * stt;
                h[0] *= s1; h[3] *= s1; h[6] *= s1;
                h[1] *= s2; h[4] *= s2; h[7] *= s2;
                h[2] = h[3] * h[7] - h[6] * h[4];
                h[5] = h[6] * h[1] - h[0] * h[7];
                h[8] = h[0] * h[4] - h[3] * h[1];
                double rtmp[3], dummy[27];
                rodrigues_m2v(h, rtmp);
                rodrigues_v2m(rtmp, R, dummy, false);
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            }
            rodrigues_m2v(R, param);
        }
        // ---- CvLevMarq
        double prevParam[6], S[21], gJ[6], Jrow[6];
        double err = 0, prevErrNorm = 0, errNorm = 0;
        int lambdaLg10 = -3, iters = 0, state = 1;
        // (CvLevMarq: lambda = exp(lambdaLg10 * log(10.)): lm_lambda)
#pragma unroll
        for (int i = 0; i < 6; i++) prevParam[i] = param[i];
        for (;;) {
            bool needJ = false, needErr = false;
            if (state == 1) {
                needJ = needErr = true;
                state = 2;
            } else if (state == 2) {
                {
                    int idx = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int b = a; b < 6; b++) S[idx++] = grp_sum8(Jrow[a] * Jrow[b]);
                        gJ[a] = grp_sum8(Jrow[a] * err);
                    }
                }
#pragma unroll
                for (int i = 0; i < 6; i++) prevParam[i] = param[i];
                double xs[6];
                solve6_spd(S, gJ, lm_lambda(lambdaLg10), xs);
#pragma unroll
                for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
                if (iters == 0) prevErrNorm = sqrt(grp_sum8(err * err));
                needErr = true;
                state = 3;
            } else {
                errNorm = sqrt(grp_sum8(err * err));
                bool retry = false;
                if (errNorm > prevErrNorm) {
                    if (++lambdaLg10 <= 16) {
                        double xs[6];
                        solve6_spd(S, gJ, lm_lambda(lambdaLg10), xs);
#pragma unroll
                        for (int i = 0; i < 6; i++) param[i] = prevParam[i] - xs[i];
                        needErr = true;
                        state = 3;
                        retry = true;
                    }
                }
                if (!retry) {
                    lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
                    double dn = 0, pn = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
                        pn += prevParam[i] * prevParam[i];
                    }
                    double rel = sqrt(dn) / (sqrt(pn) + DBL_EPSILON);
                    if (++iters >= 20 || rel < FLT_EPSILON) break;
                    prevErrNorm = errNorm;
                    needJ = needErr = true;
                    state = 2;
                }
            }
            if (!needErr) break;
            double pr = project_one(M, param, K, kd, sel, Jrow, needJ);
            err = pr - mobs;
        }
        // ---- getReprojectionError: projections rounded to float (vector<Point2f>), error = sum |d|^2 / 4
        double prj = project_one(M, param, K, kd, sel, Jrow, false);
        double dcoord = mobs - (double)(float)prj;
        double d2 = dcoord * dcoord;
        double pt2 = d2 + dpp_f64<0xB1>(d2);  // dx^2 + dy^2 of this corner (the lane beside this one: quad_perm [1,0,3,2])
        double e = sqrt(pt2);
        double contrib = sel == 0 ? e * e : 0.;
        double totalErr = grp_sum8(contrib);
        if (g == 0) {
            fid_pose_out o;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                o.rvec[i] = param[i];
                o.tvec[i] = param[3 + i];
            }
            const double rerr = totalErr / 4.0;
            o.image_error = rerr;
            const float *c = mk.corners;
            // calcFiducialArea (Heron on two triangles)
            double a1 = dist2f_d(c[0], c[1], c[2], c[3]);
            double b1 = dist2f_d(c[0], c[1], c[6], c[7]);
            double c1 = dist2f_d(c[2], c[3], c[6], c[7]);
            double a2 = dist2f_d(c[2], c[3], c[4], c[5]);
            double b2 = dist2f_d(c[4], c[5], c[6], c[7]);
            double c2 = c1;
            double s1 = (a1 + b1 + c1) / 2.0, s2 = (a2 + b2 + c2) / 2.0;
            a1 = sqrt(s1 * (s1 - a1) * (s1 - b1) * (s1 - c1));
            a2 = sqrt(s2 * (s2 - a2) * (s2 - b2) * (s2 - c2));
            o.fiducial_area = a1 + a2;
            double nt = sqrt(o.tvec[0] * o.tvec[0] + o.tvec[1] * o.tvec[1] + o.tvec[2] * o.tvec[2]);
            o.object_error = (rerr / dist2f_d(c[0], c[1], c[4], c[5])) * (nt / cam.fiducial_len);
            out[item] = o;
        }
    }
}
