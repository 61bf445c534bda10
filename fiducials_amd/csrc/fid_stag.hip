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

// ------------------------------------------------------------------------------------------------ C-ABI
struct fid_stag_ctx {
    int device = 0, maxW = 0, maxH = 0, libraryHD = 0, errorCorrection = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_src = nullptr, *d_smooth = nullptr, *d_dir = nullptr, *d_edge = nullptr;
    int16_t *d_grad = nullptr;
    unsigned *d_rowhist = nullptr, *d_bandhist = nullptr, *d_tot = nullptr, *d_bstart = nullptr, *d_n = nullptr;
    int32_t *d_sorted = nullptr;
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
    void *dev[] = {c->d_src, c->d_smooth, c->d_dir, c->d_edge, c->d_grad, c->d_sorted, c->d_rowhist, c->d_bandhist, c->d_tot, c->d_bstart, c->d_n};
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
    }
    if (!src) return FID_E_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return FID_E_HIP;
    return hipMemcpy(dst, src, (size_t)need, hipMemcpyDeviceToHost) == hipSuccess ? FID_OK : FID_E_HIP;
}

}  // extern "C"
