// fid_stag_lines.hip -- STag rows s5, s6: segment validation, EDLines line fitting / joining, line validation.
// Part of the fid_stag.hip translation unit (included there; not compiled on its own).
// ------------------------------------------------------------------------------------------------ K11: segment validation
// ValidateEdgeSegments (ValidateEdgeSegments.cpp:365-413) after the second smoothing of DetectEdgesByEDPF
// (ED.cpp:176-178: SmoothImage(sigma = 1 / 2.5) = cv::GaussianBlur(Size(0, 0), 0.4): OpenCV picks ksize 3 and the 8.8
// fixed-point kernel [10 236 10] / 256, one rounding at the end -- restated, "parity unpinned").
//   k_stag_smooth3_prewitt   the 3x3 blur fused with ComputePrewitt3x3 (:63-115): gradient map + histogram
//   k_stag_valid_prob        H[g] = P(gradient >= g) (:107-111), np = sum len (len - 1) / 2 (:381-385)
//   k_stag_test_segments     TestSegment (:134-199), one wave per segment, the recursion on an explicit stack that lives
//                            in the scratch slots of the segment's own pixels
//   k_stag_extract           ExtractNewSegments (:319-360): runs of still-marked pixels of >= 10 (count pass, scan, write pass)
__device__ __forceinline__ void k_stag_smooth3_prewitt_impl(const uint8_t *__restrict__ src, int stride, int W, int H,
                                                              uint8_t *__restrict__ smooth, int16_t *__restrict__ grad,
                                                              unsigned *__restrict__ hist)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SY + 4][SX + 4 + 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[SY + 4][SX + 4];
    __shared__ __attribute__((aligned(16))) uint8_t s_sm[SY + 2][SX + 2 + 2];
    __shared__ unsigned s_hist[STAG_BINS];
    const int x0 = blockIdx.x * SX, y0 = blockIdx.y * SY;
    const int tid = threadIdx.x;
    for (int i = tid; i < STAG_BINS; i += 256) s_hist[i] = 0;
    // (round 6) the frame's histogram is kept in STAG_VHIST_SLICES copies, a tile adds to the one its place picks and
    // k_stag_valid_prob sums them: 2 040 tiles of a 1080p frame sent their ~30 non-empty bins each to the same ~300 words -- 26 of the
    // kernel's 35 us for a single frame were those atomics queueing up (measured: a build without the flush)
    unsigned *hslice = hist + (size_t)((blockIdx.x + 5u * blockIdx.y) & (STAG_VHIST_SLICES - 1)) * STAG_BINS;
    // (round 6) interior tiles, four pixels a thread: as k_stag_smooth_grad's (fid_stag.hip) -- the 3-tap [10 236 10] passes as one
    // v_dot4_u32_u8 per output (horizontal) and three 32-bit multiply-adds (vertical: 8.8 x 8.8 fixed point needs the 32 bits),
    // Prewitt from column sums, the histogram as below.  The same integers as the byte-a-thread form.
    if (x0 >= 4 && x0 + SX + 4 <= W && y0 >= 2 && y0 + SY + 2 <= H && ((stride | W) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        uint32_t(*w_src)[18] = reinterpret_cast<uint32_t(*)[18]>(&s_src[0][0]);  // bytes x0 - 4 .. x0 + 67 of rows y0 - 2 .. y0 + 17 (20 x 72 bytes)
        uint32_t(*w_h)[34] = reinterpret_cast<uint32_t(*)[34]>(&s_h[0][0]);       // 20 x 34 words: smoothed columns x0 - 2 .. x0 + 65, two a word
        uint32_t(*w_sm)[17] = reinterpret_cast<uint32_t(*)[17]>(&s_sm[0][0]);     // 18 x 17 words
        for (int i = tid; i < 20 * 18; i += 256) {
            const int r = i / 18, c = i - r * 18;
            w_src[r][c] = *reinterpret_cast<const uint32_t *>(src + (long long)(y0 - 2 + r) * stride + (x0 - 4 + 4 * c));
        }
        __syncthreads();
        // horizontal: smoothed column x0 - 2 + 4 j + k from bytes k + 1 .. k + 3 of words j, j + 1 (weights 10, 236, 10)
        for (int i = tid; i < 20 * 17; i += 256) {
            const int r = i / 17, j = i - r * 17;
            const uint32_t d0 = w_src[r][j], d1 = w_src[r][j + 1];
            const uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), 0x000AEC0Au, 0u, false);
            const uint32_t h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), 0x000AEC0Au, 0u, false);
            const uint32_t h2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), 0x000AEC0Au, 0u, false);
            const uint32_t h3 = __builtin_amdgcn_udot4(d1, 0x000AEC0Au, 0u, false);
            w_h[r][2 * j] = h0 | (h1 << 16);
            w_h[r][2 * j + 1] = h2 | (h3 << 16);
        }
        __syncthreads();
        // vertical: smoothed row y0 - 1 + r from rows r .. r + 2; (acc + 32768) >> 16
        for (int i = tid; i < 18 * 17; i += 256) {
            const int r = i / 17, j = i - r * 17;
            uint32_t px[4];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t a = w_h[r][2 * j + q], b = w_h[r + 1][2 * j + q], c = w_h[r + 2][2 * j + q];
                px[2 * q] = (10u * (a & 0xffffu) + 236u * (b & 0xffffu) + 10u * (c & 0xffffu) + 32768u) >> 16;
                px[2 * q + 1] = (10u * (a >> 16) + 236u * (b >> 16) + 10u * (c >> 16) + 32768u) >> 16;
            }
            w_sm[r][j] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        }
        __syncthreads();
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
            uint32_t gw[2] = {0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int gxv = S[3 + k] - S[1 + k], gyv = D[1 + k] + D[2 + k] + D[3 + k];
                gxv = gxv < 0 ? -gxv : gxv;
                gyv = gyv < 0 ? -gyv : gyv;
                const int g = gxv + gyv;
                gw[k >> 1] |= (uint32_t)g << (16 * (k & 1));
                // the tile's histogram, as below: the lanes that hold the first lane's value are counted by one atomic
                const int g0 = __builtin_amdgcn_readfirstlane(g);
                const unsigned long long same = __ballot(g == g0);
                if (g == g0) {
                    if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(same)) atomicAdd(&s_hist[g0], (unsigned)__builtin_popcountll(same));
                } else {
                    atomicAdd(&s_hist[g], 1u);
                }
            }
            const long long idx = (long long)(y0 + r) * W + (x0 + 4 * j);
            *reinterpret_cast<uint32_t *>(smooth + idx) = __builtin_amdgcn_alignbyte(m1, m0, 2);
            *reinterpret_cast<uint2 *>(grad + idx) = make_uint2(gw[0], gw[1]);
        }
        __syncthreads();
        for (int i = tid; i < STAG_BINS; i += 256)
            if (s_hist[i]) atomicAdd(&hslice[i], s_hist[i]);
        return;
    }
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
            // the tile's histogram: most of a row's 64 pixels share ONE value (flat image: 0 - 3), and 64 LDS atomics on one word
            // are 64 serial operations -- the lanes that hold the first lane's value are counted by one of them, the rest go singly
            const int g0 = __builtin_amdgcn_readfirstlane(g);
            const unsigned long long same = __ballot(g == g0);
            if (g == g0) {
                if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(same)) atomicAdd(&s_hist[g0], (unsigned)__builtin_popcountll(same));
            } else {
                atomicAdd(&s_hist[g], 1u);
            }
        }
        grad[idx] = (int16_t)g;
    }
    __syncthreads();
    for (int i = tid; i < STAG_BINS; i += 256)
        if (s_hist[i]) atomicAdd(&hslice[i], s_hist[i]);
}
__global__ __launch_bounds__(256) void k_stag_smooth3_prewitt(const uint8_t *__restrict__ src, int stride, int W, int H, uint8_t *__restrict__ smooth, int16_t *__restrict__ grad, unsigned *__restrict__ hist)
{
    k_stag_smooth3_prewitt_impl(src, stride, W, H, smooth, grad, hist);
}
struct k_stag_smooth3_prewitt_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const uint8_t *__restrict__ src, int stride, int W, int H, uint8_t *__restrict__ smooth, int16_t *__restrict__ grad, unsigned *__restrict__ hist) const { k_stag_smooth3_prewitt_impl(src, stride, W, H, smooth, grad, hist); }
};

// one workgroup: cumulative histogram from the top -> H[g]; np over the segments (32-bit int arithmetic as in the reference)
__device__ __forceinline__ void k_stag_valid_prob_impl(const unsigned *__restrict__ hist, int W, int H, const int2 *__restrict__ segs,
                                                         const int *__restrict__ counters, double *__restrict__ prob, int *__restrict__ np_out)
{
    __shared__ unsigned s_part[512];
    const int tid = threadIdx.x;
    constexpr int PER = STAG_BINS / 512;
    // suffix sums: thread t owns bins [t * PER, t * PER + PER)
    unsigned loc[PER];
    unsigned acc = 0;
    for (int k = PER - 1; k >= 0; k--) {
        for (int sl = 0; sl < STAG_VHIST_SLICES; sl++) acc += hist[(size_t)sl * STAG_BINS + tid * PER + k];  // (the slices of k_stag_smooth3_prewitt)
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
__global__ __launch_bounds__(512) void k_stag_valid_prob(const unsigned *__restrict__ hist, int W, int H, const int2 *__restrict__ segs, const int *__restrict__ counters, double *__restrict__ prob, int *__restrict__ np_out)
{
    k_stag_valid_prob_impl(hist, W, H, segs, counters, prob, np_out);
}
struct k_stag_valid_prob_fn {
    static constexpr int kBounds = 512;
    __device__ __forceinline__ void operator()(const unsigned *__restrict__ hist, int W, int H, const int2 *__restrict__ segs, const int *__restrict__ counters, double *__restrict__ prob, int *__restrict__ np_out) const { k_stag_valid_prob_impl(hist, W, H, segs, counters, prob, np_out); }
};

__device__ __forceinline__ void k_stag_test_segments_impl(const int2 *__restrict__ segs, const int *__restrict__ counters,
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
__global__ __launch_bounds__(256) void k_stag_test_segments(const int2 *__restrict__ segs, const int *__restrict__ counters, const int2 *__restrict__ pix, const int16_t *__restrict__ vgrad, int W, const double *__restrict__ prob, const int *__restrict__ np_in, double div, int2 *__restrict__ stackmem, uint8_t *__restrict__ edge)
{
    k_stag_test_segments_impl(segs, counters, pix, vgrad, W, prob, np_in, div, stackmem, edge);
}
struct k_stag_test_segments_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int2 *__restrict__ segs, const int *__restrict__ counters, const int2 *__restrict__ pix, const int16_t *__restrict__ vgrad, int W, const double *__restrict__ prob, const int *__restrict__ np_in, double div, int2 *__restrict__ stackmem, uint8_t *__restrict__ edge) const { k_stag_test_segments_impl(segs, counters, pix, vgrad, W, prob, np_in, div, stackmem, edge); }
};

// ExtractNewSegments: one wave per segment.  write = 0: counts[seg] = number of runs of >= 10 marked pixels; write = 1:
// the runs go to out[] from counts[seg] (exclusive prefix sums by then) on.
__device__ __forceinline__ void k_stag_extract_impl(const int2 *__restrict__ segs, const int *__restrict__ counters,
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
__global__ __launch_bounds__(256) void k_stag_extract(const int2 *__restrict__ segs, const int *__restrict__ counters, const int2 *__restrict__ pix, const uint8_t *__restrict__ edge, int W, int *__restrict__ counts, int2 *__restrict__ out, int write)
{
    k_stag_extract_impl(segs, counters, pix, edge, W, counts, out, write);
}
struct k_stag_extract_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int2 *__restrict__ segs, const int *__restrict__ counters, const int2 *__restrict__ pix, const uint8_t *__restrict__ edge, int W, int *__restrict__ counts, int2 *__restrict__ out, int write) const { k_stag_extract_impl(segs, counters, pix, edge, W, counts, out, write); }
};

// exclusive prefix sums over per-item counts: one workgroup per array (blockIdx.x picks it), 8 consecutive items per thread,
// wave scans on DPP, one barrier pair per 32 768 items (1 024 threads).  (The first version, a Hillis-Steele scan of 1024 items at a time with
// 20 barriers each, took 68 us for the ~40 k anchors of a frame.)
struct StagScanJobs {
    int *counts[2];
    int *total[2];
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(counts); v(total);
    }
};
__device__ __forceinline__ void k_stag_scan_counts_n_impl(StagScanJobs J, const int *__restrict__ counters)
{
    __shared__ int s_w[16];
    int *__restrict__ counts = J.counts[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = counters[0];
    // (round 6: 32 items a thread, read and written as 16-byte vectors -- a trip of this loop is a memory round trip each way and two
    //  barriers, ~6 us whatever it carries, and the ~40 k anchors of a frame took five trips with 8 items a thread and 1 024 threads
    //  -- 38 us -- and twenty with the batch's 256 threads)
    constexpr int PER = 32;
    int carry = 0;
    // (any block size up to 1 024: round 6 launches these with 256 threads -- a 1 024-thread workgroup waits for sixteen free wave
    //  slots on ONE CU, and beside the other groups' kernels that wait was ten times the scan: 9 us alone, 90 us in the batch)
    const int NT = (int)blockDim.x, NWV = NT >> 6;
    for (int base = 0; base < n; base += NT * PER) {
        const int i0 = base + tid * PER;
        int v[PER], sum = 0;
        const bool whole = i0 + PER <= n;
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) {
                const int4 q = *reinterpret_cast<const int4 *>(counts + i0 + k);
                v[k] = q.x;
                v[k + 1] = q.y;
                v[k + 2] = q.z;
                v[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) v[k] = i0 + k < n ? counts[i0 + k] : 0;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) sum += v[k];
        const int incl = wave_iscan(sum);
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = k < NWV ? s_w[k] : 0;
            wbase += k < wv ? t : 0;
            tot += t;
        }
        int run = carry + wbase + incl - sum;
#pragma unroll
        for (int k = 0; k < PER; k++) {  // v[k]: the item -> its exclusive prefix
            const int t = v[k];
            v[k] = run;
            run += t;
        }
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) *reinterpret_cast<int4 *>(counts + i0 + k) = make_int4(v[k], v[k + 1], v[k + 2], v[k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++)
                if (i0 + k < n) counts[i0 + k] = v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *J.total[blockIdx.x] = carry;
}
__global__ __launch_bounds__(1024) void k_stag_scan_counts_n(StagScanJobs J, const int *__restrict__ counters)
{
    k_stag_scan_counts_n_impl(J, counters);
}
struct k_stag_scan_counts_n_fn {
    static constexpr int kBounds = 1024;
    __device__ __forceinline__ void operator()(StagScanJobs J, const int *__restrict__ counters) const { k_stag_scan_counts_n_impl(J, counters); }
};
__device__ __forceinline__ void k_stag_scan_counts_impl(int *__restrict__ counts, const int *__restrict__ counters, int *__restrict__ total)
{
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = counters[0];
    // (round 6: 32 items a thread, read and written as 16-byte vectors -- a trip of this loop is a memory round trip each way and two
    //  barriers, ~6 us whatever it carries, and the ~40 k anchors of a frame took five trips with 8 items a thread and 1 024 threads
    //  -- 38 us -- and twenty with the batch's 256 threads)
    constexpr int PER = 32;
    int carry = 0;
    // (any block size up to 1 024: round 6 launches these with 256 threads -- a 1 024-thread workgroup waits for sixteen free wave
    //  slots on ONE CU, and beside the other groups' kernels that wait was ten times the scan: 9 us alone, 90 us in the batch)
    const int NT = (int)blockDim.x, NWV = NT >> 6;
    for (int base = 0; base < n; base += NT * PER) {
        const int i0 = base + tid * PER;
        int v[PER], sum = 0;
        const bool whole = i0 + PER <= n;
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) {
                const int4 q = *reinterpret_cast<const int4 *>(counts + i0 + k);
                v[k] = q.x;
                v[k + 1] = q.y;
                v[k + 2] = q.z;
                v[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) v[k] = i0 + k < n ? counts[i0 + k] : 0;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) sum += v[k];
        const int incl = wave_iscan(sum);
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = k < NWV ? s_w[k] : 0;
            wbase += k < wv ? t : 0;
            tot += t;
        }
        int run = carry + wbase + incl - sum;
#pragma unroll
        for (int k = 0; k < PER; k++) {  // v[k]: the item -> its exclusive prefix
            const int t = v[k];
            v[k] = run;
            run += t;
        }
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) *reinterpret_cast<int4 *>(counts + i0 + k) = make_int4(v[k], v[k + 1], v[k + 2], v[k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++)
                if (i0 + k < n) counts[i0 + k] = v[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}
__global__ __launch_bounds__(1024) void k_stag_scan_counts(int *__restrict__ counts, const int *__restrict__ counters, int *__restrict__ total)
{
    k_stag_scan_counts_impl(counts, counters, total);
}
struct k_stag_scan_counts_fn {
    static constexpr int kBounds = 1024;
    __device__ __forceinline__ void operator()(int *__restrict__ counts, const int *__restrict__ counters, int *__restrict__ total) const { k_stag_scan_counts_impl(counts, counters, total); }
};

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
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(x); v(y); v(xx); v(yy); v(xy);
    }
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

// Where k_stag_split_lines reads a segment's pixels and prefix sums (round 5): every batch of ten good pixels fetched its next
// 64 pixels and every refit its ten prefix values from GLOBAL memory, two round trips a batch and some eighty batches on a
// marker's outline -- two thirds of the kernel's time waiting.  A segment of <= lds_pix pixels (k_stag_split_lines) lives in the wave's share of
// the LDS instead (pixels packed x | y << 16; x / y sums fit 32 bits there), longer ones stay where they were.
struct SlLds {
    const int *px;                      // x | y << 16
    const int *sx, *sy;                 // [n + 1]
    const long long *sxx, *syy, *sxy;   // [n + 1]
    __device__ __forceinline__ double X(int k) const { return (double)(px[k] & 0xffff); }
    __device__ __forceinline__ double Y(int k) const { return (double)(px[k] >> 16); }
    __device__ __forceinline__ long long dX(int b, int n) const { return (long long)(sx[b + n] - sx[b]); }
    __device__ __forceinline__ long long dY(int b, int n) const { return (long long)(sy[b + n] - sy[b]); }
    __device__ __forceinline__ long long dXX(int b, int n) const { return sxx[b + n] - sxx[b]; }
    __device__ __forceinline__ long long dYY(int b, int n) const { return syy[b + n] - syy[b]; }
    __device__ __forceinline__ long long dXY(int b, int n) const { return sxy[b + n] - sxy[b]; }
};
struct SlGlobal {
    const int2 *px;  // (row, column)
    StagPrefix P;
    __device__ __forceinline__ double X(int k) const { return (double)px[k].y; }
    __device__ __forceinline__ double Y(int k) const { return (double)px[k].x; }
    __device__ __forceinline__ long long dX(int b, int n) const { return P.x[b + n] - P.x[b]; }
    __device__ __forceinline__ long long dY(int b, int n) const { return P.y[b + n] - P.y[b]; }
    __device__ __forceinline__ long long dXX(int b, int n) const { return P.xx[b + n] - P.xx[b]; }
    __device__ __forceinline__ long long dYY(int b, int n) const { return P.yy[b + n] - P.yy[b]; }
    __device__ __forceinline__ long long dXY(int b, int n) const { return P.xy[b + n] - P.xy[b]; }
};

// LineFit with a known orientation (LineSegment.cpp:703-733) over pixels [base, base + count)
template <class Src>
__device__ void sl_fit_known(const Src &P, int base, int count, int invert, double *a, double *b)
{
    if (count < 2) return;
    const double S = count;
    double Sx = (double)P.dX(base, count), Sy = (double)P.dY(base, count);
    double Sxx, Sxy = (double)P.dXY(base, count);
    if (invert) {
        const double t = Sx;
        Sx = Sy;
        Sy = t;
        Sxx = (double)P.dYY(base, count);
    } else {
        Sxx = (double)P.dXX(base, count);
    }
    const double D = S * Sxx - Sx * Sx;
    *a = (Sxx * Sy - Sx * Sxy) / D;
    *b = (S * Sxy - Sx * Sy) / D;
}

// LineFit with orientation choice and fitting error (LineSegment.cpp:628-697) over pixels [base, base + count)
template <class Src>
__device__ void sl_fit_first(const Src &P, int base, int count, double *a, double *b, double *e, int *invert)
{
    if (count < 2) return;
    const double Sx0 = (double)P.dX(base, count), Sy0 = (double)P.dY(base, count);
    const double mx = Sx0 / count, my = Sy0 / count;
    double dx = 0.0, dy = 0.0;
    for (int i = 0; i < count; i++) {
        const double xi = P.X(base + i), yi = P.Y(base + i);
        dx += (xi - mx) * (xi - mx);
        dy += (yi - my) * (yi - my);
    }
    const int inv = dx < dy ? 1 : 0;
    *invert = inv;
    sl_fit_known(P, base, count, inv, a, b);
    double error = 0.0;
    if (*b == 0.0) {
        for (int i = 0; i < count; i++) {
            const double yi = inv ? P.X(base + i) : P.Y(base + i);
            error += fabs((*a) - yi);
        }
        *e = error / count;
    } else {
        for (int i = 0; i < count; i++) {
            const double xi = inv ? P.Y(base + i) : P.X(base + i), yi = inv ? P.X(base + i) : P.Y(base + i);
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
// SplitSegment2Lines + JoinCollinearLines of one segment by one wave, on either source
template <class Src>
__device__ __forceinline__ void sl_split_body(const Src &P, int n, int seg, int first, int lane, int min_line_len, double line_error,
                                              fid_stag_line *__restrict__ slots, int *__restrict__ counts)
{
    fid_stag_line *L = slots + first / 9;
    int nl = 0;
    const int MLL = min_line_len;
    int base = 0, noPixels = n, firstPixelIndex = 0;
    while (noPixels >= MLL) {
        bool valid = false;
        double lastA = 0, lastB = 0;
        int lastInvert = 0;
        // first window (sliding by one pixel) whose MLL-pixel fit has error <= 0.5: 64 positions per round
        while (noPixels >= MLL) {
            const int avail = noPixels - MLL + 1;  // window starts base .. base + avail - 1
            double a = 0, bq = 0, e = 1e300;
            int inv = 0;
            if (lane < avail) sl_fit_first(P, base + lane, MLL, &a, &bq, &e, &inv);
            const unsigned long long okm = __ballot(lane < avail && e <= 0.5);
            if (okm) {
                const int j = __builtin_ctzll(okm);
                lastA = __shfl(a, j, 64); lastB = __shfl(bq, j, 64); lastInvert = __shfl(inv, j, 64);
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
                if (k < noPixels) ok = sl_min_dist(P.X(base + k), P.Y(base + k), lastA, lastB, lastInvert) <= line_error;
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
                while (idx < noPixels - 1 && sl_min_dist(P.X(base + idx), P.Y(base + idx), lastA, lastB, lastInvert) > line_error) idx++;
                sl_min_dist(P.X(base + idx), P.Y(base + idx), lastA, lastB, lastInvert, &sx, &sy);
                const int skipped = idx;
                idx = lastGoodIndex;
                while (idx > 0 && sl_min_dist(P.X(base + idx), P.Y(base + idx), lastA, lastB, lastInvert) > line_error) idx--;
                sl_min_dist(P.X(base + idx), P.Y(base + idx), lastA, lastB, lastInvert, &ex, &ey);
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

// LDS of k_stag_split_lines: four waves, each `lds_pix` pixels + the prefix sums over them (36 bytes a pixel); the launch passes
// lds_pix and asks for SL_LDS_BYTES(lds_pix) -- 1 024 pixels for a frame on its own (148 KB, one workgroup per CU: 200 of them),
// 256 for a group of frames (37 KB: four workgroups per CU stay resident), 0: every segment on the global road
__host__ __device__ constexpr int sl_lds_wave_bytes(int lds_pix) { return ((lds_pix + 1) * (3 * 4 + 3 * 8) + 63) & ~63; }
#define SL_LDS_BYTES(lds_pix) ((lds_pix) > 0 ? 4 * sl_lds_wave_bytes(lds_pix) : 0)
template <bool FM = false>
__device__ __forceinline__ void k_stag_split_lines_impl(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int2 *__restrict__ pix,
                                                          StagPrefix PF, int min_line_len, double line_error, fid_stag_line *__restrict__ slots,
                                                          int *__restrict__ counts, int lds_pix, int min_n, int max_n)
{
    extern __shared__ long long s_sl[];
    // (a wave per segment, any number of waves per workgroup: a frame on its own runs four with 1 024 pixels of LDS each; a group of
    //  frames takes the segments in two launches of ONE-wave workgroups -- the long ones (min_n < n) with 1 024 pixels of LDS, which
    //  until round 6 took the global-memory road in a group, the short ones with 256 -- so that a workgroup asks a CU for 9 or 37 KB
    //  and one wave slot instead of 37 KB and four)
    const int wv = threadIdx.x >> 6;
    const int seg = (int)STAG_BX<FM>() * (int)(blockDim.x >> 6) + wv, lane = threadIdx.x & 63;
    if (seg >= *nsegs) return;
    const int first = segs[seg].x, n = segs[seg].y;
    if (n <= min_n || n > max_n) return;  // (the other launch's segment)
    const int2 *px = pix + first;
    if (n <= lds_pix) {
        // the wave's share: three 64-bit arrays, then three 32-bit ones
        long long *sxx = s_sl + (size_t)wv * (sl_lds_wave_bytes(lds_pix) / 8), *syy = sxx + (lds_pix + 1), *sxy = syy + (lds_pix + 1);
        int *sx = (int *)(sxy + (lds_pix + 1)), *sy = sx + (lds_pix + 1), *spx = sy + (lds_pix + 1);
        long long cxx = 0, cyy = 0, cxy = 0;
        int cx = 0, cy = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            long long x = 0, y = 0;
            if (k < n) {
                x = px[k].y;
                y = px[k].x;
                spx[k] = (int)x | ((int)y << 16);
            }
            const long long ix = wave_iscan_ll(x, lane), iy = wave_iscan_ll(y, lane), ixx = wave_iscan_ll(x * x, lane),
                            iyy = wave_iscan_ll(y * y, lane), ixy = wave_iscan_ll(x * y, lane);
            if (k < n) {  // exclusive value at k
                sx[k] = cx + (int)(ix - x); sy[k] = cy + (int)(iy - y); sxx[k] = cxx + ixx - x * x; syy[k] = cyy + iyy - y * y; sxy[k] = cxy + ixy - x * y;
            }
            cx += (int)__shfl(ix, 63, 64); cy += (int)__shfl(iy, 63, 64); cxx += __shfl(ixx, 63, 64); cyy += __shfl(iyy, 63, 64); cxy += __shfl(ixy, 63, 64);
        }
        if (lane == 0) {
            sx[n] = cx; sy[n] = cy; sxx[n] = cxx; syy[n] = cyy; sxy[n] = cxy;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        SlLds P;
        P.px = spx; P.sx = sx; P.sy = sy; P.sxx = sxx; P.syy = syy; P.sxy = sxy;
        sl_split_body(P, n, seg, first, lane, min_line_len, line_error, slots, counts);
        return;
    }
    // the prefix arrays of this segment live at [first + seg, first + seg + n]: one extra slot per segment
    SlGlobal G;
    G.px = px;
    const int pb = first + seg;
    long long *gx = PF.x + pb, *gy = PF.y + pb, *gxx = PF.xx + pb, *gyy = PF.yy + pb, *gxy = PF.xy + pb;
    G.P.x = gx; G.P.y = gy; G.P.xx = gxx; G.P.yy = gyy; G.P.xy = gxy;
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
                gx[k] = cx + ix - x; gy[k] = cy + iy - y; gxx[k] = cxx + ixx - x * x; gyy[k] = cyy + iyy - y * y; gxy[k] = cxy + ixy - x * y;
            }
            cx += __shfl(ix, 63, 64); cy += __shfl(iy, 63, 64); cxx += __shfl(ixx, 63, 64); cyy += __shfl(iyy, 63, 64); cxy += __shfl(ixy, 63, 64);
        }
        if (lane == 0) {
            gx[n] = cx; gy[n] = cy; gxx[n] = cxx; gyy[n] = cyy; gxy[n] = cxy;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    sl_split_body(G, n, seg, first, lane, min_line_len, line_error, slots, counts);
}
__global__ __launch_bounds__(256) void k_stag_split_lines(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int2 *__restrict__ pix, StagPrefix PF, int min_line_len, double line_error, fid_stag_line *__restrict__ slots, int *__restrict__ counts, int lds_pix, int min_n, int max_n)
{
    k_stag_split_lines_impl(segs, nsegs, pix, PF, min_line_len, line_error, slots, counts, lds_pix, min_n, max_n);
}
struct k_stag_split_lines_fn {
    static constexpr int kBounds = 256;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int2 *__restrict__ pix, StagPrefix PF, int min_line_len, double line_error, fid_stag_line *__restrict__ slots, int *__restrict__ counts, int lds_pix, int min_n, int max_n) const { k_stag_split_lines_impl<true>(segs, nsegs, pix, PF, min_line_len, line_error, slots, counts, lds_pix, min_n, max_n); }
};

// the lines of every segment, one after the other in segment order (counts hold exclusive prefix sums by now)
__device__ __forceinline__ void k_stag_gather_lines_impl(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int *__restrict__ counts,
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
__global__ __launch_bounds__(64) void k_stag_gather_lines(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int *__restrict__ counts, const int *__restrict__ total, const fid_stag_line *__restrict__ slots, fid_stag_line *__restrict__ out)
{
    k_stag_gather_lines_impl(segs, nsegs, counts, total, slots, out);
}
struct k_stag_gather_lines_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(const int2 *__restrict__ segs, const int *__restrict__ nsegs, const int *__restrict__ counts, const int *__restrict__ total, const fid_stag_line *__restrict__ slots, fid_stag_line *__restrict__ out) const { k_stag_gather_lines_impl(segs, nsegs, counts, total, slots, out); }
};

// ------------------------------------------------------------------------------------------------ K13: line validation
// ValidateLineSegments (EDLines.cpp:274-409): a line is kept if enough of its pixels have a gradient direction within
// 22.5 degrees of the line (Helmholtz principle, number of false alarms from a table).  Lines of >= 80 pixels pass untested,
// lines of <= 25 pixels are tested on all pixels of a 2-pixel-wide rectangle around them (EnumerateRectPoints, :417-600, the
// LSD rectangle iterator), the others on their own pixels first and on the rectangle if that fails.  One WAVE per line (round 6; one lane until then).
// Host-made tables (functions of the image size only, evaluated with the host's libm exactly as the reference does):
//   atan_lut[i] = atan(i / 1024)  (myAtan2, MyMath.cpp:12-72);  kmin[n] = the NFALUT entry (NFA.cpp:13-44).
struct StagLineTables {
    const double *atan_lut;  // 1025 entries
    const int *kmin;         // kmin[n]: smallest number of aligned pixels out of n that validates; n <= kmin_n
    int kmin_n;
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(atan_lut); v(kmin); v(kmin_n);
    }
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

// ValidateLineSegmentRect (EDLines.cpp:612-690) with the rectangle iterator of EnumerateRectPoints (:417-600): the reference runs
//     y++; while (y > ye && x <= vx[2]) { x++; if (x > vx[2]) break; ys = ...(x); ye = ...(x); y = ceil(ys); }  if (x > vx[2]) break;  point (x, y)
// until maxNoOfPoints points are out, i.e. the columns x = ceil(vx[0]) .. vx[2] in turn and in every column the rows from
// ceil(ys(x)) while !(y > ye(x)).
// By a WAVE (round 6): ys and ye are functions of x alone (the branches below are the reference's, term for term), so a lane takes a column, a wave scan over the columns' point counts applies the iterator's cap (the first
// maxNoOfPoints points in its order), and the two tallies are integer sums.  One lane per line walked up to ~250 pixels one
// dependent load after the other: 111 us for a frame's lines.
__device__ bool sl_validate_rect_wave(const uint8_t *__restrict__ src, int W, int H, const fid_stag_line &ls, double lineAngle, const StagLineTables &T,
                                      int lane)
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
    const int xfirst = (int)ceil(vx[0]);
    const int maxNoOfPoints = (int)(fabs(ls.sx - ls.ex) + fabs(ls.sy - ls.ey)) * 4;
    int count = 0, aligned = 0, taken = 0;  // taken: points of the columns in front of this trip's (wave-uniform)
    for (int c0 = 0; taken < maxNoOfPoints; c0 += 64) {
        const int x = xfirst + c0 + lane;
        const bool col = !(x > vx[2]);
        if (!__ballot(col)) break;
        int y0 = 0, nx = 0;
        if (col) {
            double ys, ye;
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
            y0 = (int)ceil(ys);
            // the rows y0, y0 + 1, ... while !(y > ye)
            if (!((double)y0 > ye)) nx = (int)floor(ye) - y0 + 1;
        }
        const int incl = wave_iscan(nx);
        const int before = taken + incl - nx;
        int room = maxNoOfPoints - before;
        room = room < 0 ? 0 : room;
        const int mine = nx < room ? nx : room;
        for (int k = 0; k < mine; k++) {
            const int al = sl_aligned(src, W, H, y0 + k, x, lineAngle, T.atan_lut);
            if (al >= 0) {
                count++;
                aligned += al;
            }
        }
        taken += __builtin_amdgcn_readlane(incl, 63);
    }
    count = wave_sum_i32(count);
    aligned = wave_sum_i32(aligned);
    return count <= T.kmin_n ? aligned >= T.kmin[count] : false;
}

__device__ __forceinline__ void k_stag_validate_lines_impl(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines,
                                                            const uint8_t *__restrict__ src, int W, int H, const int2 *__restrict__ vsegs,
                                                            const int2 *__restrict__ pix, StagLineTables T, int *__restrict__ flags)
{
    // (round 6) a WAVE per line: the pixels of the line across the lanes, the rectangle's columns across the lanes
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (i >= *nlines) return;
    const double PI = 3.14159265358979323846;
    const fid_stag_line ls = lines[i];
    double lineAngle = ls.invert == 0 ? atan(ls.b) : atan(1.0 / ls.b);
    if (lineAngle < 0) lineAngle += PI;
    bool valid;
    if (ls.len >= 80) {
        valid = true;
    } else if (ls.len <= 25) {
        valid = sl_validate_rect_wave(src, W, H, ls, lineAngle, T, lane);
    } else {
        const int2 *p = pix + vsegs[ls.segmentNo].x + ls.firstPixelIndex;
        int count = 0, aligned = 0;
        for (int j = lane; j < ls.len; j += 64) {
            const int al = sl_aligned(src, W, H, p[j].x, p[j].y, lineAngle, T.atan_lut);
            if (al >= 0) {
                count++;
                aligned += al;
            }
        }
        count = wave_sum_i32(count);
        aligned = wave_sum_i32(aligned);
        valid = count <= T.kmin_n ? aligned >= T.kmin[count] : false;
        if (!valid) valid = sl_validate_rect_wave(src, W, H, ls, lineAngle, T, lane);
    }
    if (lane == 0) flags[i] = valid ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_stag_validate_lines(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, const uint8_t *__restrict__ src, int W, int H, const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, StagLineTables T, int *__restrict__ flags)
{
    k_stag_validate_lines_impl(lines, nlines, src, W, H, vsegs, pix, T, flags);
}
struct k_stag_validate_lines_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, const uint8_t *__restrict__ src, int W, int H, const int2 *__restrict__ vsegs, const int2 *__restrict__ pix, StagLineTables T, int *__restrict__ flags) const { k_stag_validate_lines_impl(lines, nlines, src, W, H, vsegs, pix, T, flags); }
};

// keep the valid lines, in order (flags hold exclusive prefix sums by now)
__device__ __forceinline__ void k_stag_compact_lines_impl(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines,
                                                            const int *__restrict__ pos, const int *__restrict__ total,
                                                            fid_stag_line *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = *nlines;
    if (i >= n) return;
    const int next = i + 1 < n ? pos[i + 1] : *total;
    if (next != pos[i]) out[pos[i]] = lines[i];
}
__global__ __launch_bounds__(256) void k_stag_compact_lines(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, const int *__restrict__ pos, const int *__restrict__ total, fid_stag_line *__restrict__ out)
{
    k_stag_compact_lines_impl(lines, nlines, pos, total, out);
}
struct k_stag_compact_lines_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const fid_stag_line *__restrict__ lines, const int *__restrict__ nlines, const int *__restrict__ pos, const int *__restrict__ total, fid_stag_line *__restrict__ out) const { k_stag_compact_lines_impl(lines, nlines, pos, total, out); }
};
