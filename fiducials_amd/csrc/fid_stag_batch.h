// fid_stag_batch.h -- STag frames as a grid dimension (BASELINE cfg 5, round 3).
//
// The frame pipeline of fid_stag.hip is a host state machine of SEGMENTS with ~60 launch sites.  Round 2 carried several frames
// at once by giving each its own context AND stream (16 - 22 of them, pinned against the 24 hardware queues).  Here a GROUP of
// frames goes through the same state machine in lockstep on ONE stream: while a segment's host code runs for every frame of the
// group, its launches are RECORDED instead of issued -- launch site by launch site, the arguments of every frame into a table --
// and then each site is launched ONCE for the whole group: grid (largest per-frame grid, frames in blockIdx.z), a trampoline
// kernel that picks its frame's argument tuple out of the table (kernel-argument memory: scalar loads) and runs the unchanged
// kernel body (k_stag_X_impl, produced from the frame-at-a-time kernels by tools/stag_kernels_to_impl.py).  A workgroup beyond
// its own frame's grid returns at once.  Host waits, launches and small copies per frame drop by the group size; the whole-image
// passes of a group fill the chip together instead of one 2-Mpixel frame at a time.
// Frames that take another road (sequential routing fallback, empty frame) simply record other sites: sites are merged by
// their id (source order), so a frame's launches keep their order and frames never depend on one another.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#ifndef STAG_MAXF
// frames per merged launch (more are launched in pieces): what 4 KB of kernel-argument memory hold of the routing kernels' ~250-byte
// tuples.  Round 5 tried the routing kernels' context behind a pointer (a tuple of ~80 bytes, groups of 32): groups of 32 measured
// 7 % SLOWER than groups of 16 on the cfg 5 batch (128 slots: 4 260 - 4 290 against 4 580 frames/s) -- the walk / extraction /
// refinement kernels are bound by workgroup slots, twice the frames take twice the rounds -- and pointers LOADED from memory are
// generic pointers to the compiler (flat_load / flat_store, which also count as LDS operations), where pointers that arrive as
// kernel arguments are promoted to global ones: k_stag_route_walk 1.40 -> 2.15 ms per group.  Reverted: by value, 16.
#define STAG_MAXF 16
#endif

template <typename... A>
struct StagTup;
template <>
struct StagTup<> {
};
template <typename H, typename... T>
struct StagTup<H, T...> {
    H head;
    StagTup<T...> tail;
};
template <int I, typename H, typename... T>
__host__ __device__ __forceinline__ const auto &stag_get(const StagTup<H, T...> &t)
{
    if constexpr (I == 0) return t.head;
    else return stag_get<I - 1>(t.tail);
}
template <typename H, typename... T, typename V, typename... R>
__host__ inline void stag_fill(StagTup<H, T...> &t, const V &v, const R &...r)
{
    t.head = (H)v;
    if constexpr (sizeof...(T) > 0) stag_fill(t.tail, r...);
}

// the parameter list of a kernel functor's operator()
template <typename F>
struct StagSig;
template <typename C, typename... A>
struct StagSig<void (C::*)(A...) const> {
    using tup = StagTup<std::remove_cv_t<A>...>;
    template <typename Fn, size_t... I>
    static __device__ __forceinline__ void call(const tup &t, std::index_sequence<I...>)
    {
        Fn{}(stag_get<(int)I>(t)...);
    }
    static constexpr size_t n = sizeof...(A);
};

template <typename Fn>
struct StagTab {
    using sig = StagSig<decltype(&Fn::operator())>;
    using tup = typename sig::tup;
    // kernel-argument memory is 4 KB: as many frames per launch as fit
    static constexpr int kFit = 4032 / (int)(sizeof(tup) + 8);
    static constexpr int kMax = kFit < 1 ? 1 : (kFit > STAG_MAXF ? STAG_MAXF : kFit);
    unsigned gx[kMax], gy[kMax];
    tup a[kMax];
};

template <typename Fn>
__global__ __launch_bounds__(Fn::kBounds) void k_stag_batch(const StagTab<Fn> tab)
{
    const int f = blockIdx.z;
    if (blockIdx.x >= tab.gx[f] || blockIdx.y >= tab.gy[f]) return;
    StagTab<Fn>::sig::template call<Fn>(tab.a[f], std::make_index_sequence<StagTab<Fn>::sig::n>{});
}

// ---- host side: the recorder of one group
struct StagRecorder {
    struct Op {
        int site;
        void (*flush)(StagRecorder &, void *);
        void *state;
    };
    struct Copy {  // a memset / memcpy of one frame, issued at its site in order
        int site, kind;  // kind 0: memset (val), 1..: hipMemcpyKind + 1
        void *dst;
        const void *src;
        int val;
        size_t bytes;
    };
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<Op> ops;
    std::vector<Copy> copies;
    bool failed = false;
    long long merged_launches = 0, recorded_launches = 0, order_flushes = 0;
    int frame_last_site = -1;  // the last site the frame now recording has used in this round (stag_order_guard)
};
extern thread_local StagRecorder *g_stag_rec;
static inline bool stag_flush(StagRecorder &R);
// Merged launches go out in SITE order (= source position), which is a frame's execution order only while the frame passes its
// sites in increasing order within a round.  That holds for the state machine as it stands; a site inside a loop, or a helper
// defined above its caller, would silently reorder a frame's dependent kernels.  So every record checks it: a frame that comes
// back to a site at or below its last one gets everything recorded so far issued first (correct, just less merged).
static inline void stag_order_guard(StagRecorder &R, int site)
{
    if (site <= R.frame_last_site) {
        (void)stag_flush(R);
        R.order_flushes++;
    }
    R.frame_last_site = site;
}

template <typename Fn, int SITE>
struct StagSite {
    static thread_local StagTab<Fn> tab;
    static thread_local int n;
    static thread_local dim3 block;
    static thread_local size_t lds;
    static thread_local unsigned mx, my;
    static void launch(StagRecorder &R)
    {
        if (n == 0) return;
        hipLaunchKernelGGL(k_stag_batch<Fn>, dim3(mx, my, (unsigned)n), block, lds, R.stream, tab);
        if (hipGetLastError() != hipSuccess) R.failed = true;
        R.merged_launches++;
        n = 0;
        mx = my = 0;
        lds = 0;
    }
    static void flush(StagRecorder &R, void *) { launch(R); }
    template <typename... V>
    static void record(StagRecorder &R, dim3 grid, dim3 blk, size_t l, const V &...v)
    {
        // the table is full: everything recorded so far goes out (in site order, this site included), then recording goes on -- a
        // frame's launches stay in order, because what it recorded before this point is issued before what it records after it
        stag_order_guard(R, SITE);
        if (n == StagTab<Fn>::kMax) (void)stag_flush(R);
        if (n == 0) R.ops.push_back({SITE, &StagSite::flush, nullptr});
        stag_fill(tab.a[n], v...);
        tab.gx[n] = grid.x;
        tab.gy[n] = grid.y;
        mx = grid.x > mx ? grid.x : mx;
        my = grid.y > my ? grid.y : my;
        block = blk;
        lds = l > lds ? l : lds;
        n++;
        R.recorded_launches++;
    }
};
template <typename Fn, int SITE>
thread_local StagTab<Fn> StagSite<Fn, SITE>::tab;
template <typename Fn, int SITE>
thread_local int StagSite<Fn, SITE>::n = 0;
template <typename Fn, int SITE>
thread_local dim3 StagSite<Fn, SITE>::block;
template <typename Fn, int SITE>
thread_local size_t StagSite<Fn, SITE>::lds = 0;
template <typename Fn, int SITE>
thread_local unsigned StagSite<Fn, SITE>::mx = 0;
template <typename Fn, int SITE>
thread_local unsigned StagSite<Fn, SITE>::my = 0;

// a launch site of the state machine: issued at once (frame-at-a-time entry points) or recorded (group mode)
#define STAG_LAUNCH(K, grid, block, lds, st, ...)                                                          \
    do {                                                                                                   \
        if (g_stag_rec && g_stag_rec->on) StagSite<K##_fn, __COUNTER__>::record(*g_stag_rec, grid, block, lds, __VA_ARGS__); \
        else hipLaunchKernelGGL(K, grid, block, lds, st, __VA_ARGS__);                                     \
    } while (0)

// ---- the small device operations of the state machine as kernels, so that a group issues each of them ONCE: fills, device
// copies, and the few bytes of counters that go back to the host after every segment (written straight into the context's
// pinned host block through its device alias: no copy engine round per frame)
__device__ __forceinline__ void k_stag_memset_impl(uint8_t *dst, int val, unsigned long long bytes)
{
    const unsigned long long head = ((16ull - ((unsigned long long)dst & 15ull)) & 15ull) < bytes ? ((16ull - ((unsigned long long)dst & 15ull)) & 15ull) : bytes;
    const unsigned long long n16 = (bytes - head) >> 4, tail = bytes - head - (n16 << 4);
    const unsigned v = (unsigned)(val & 0xff) * 0x01010101u;
    uint4 *mid = reinterpret_cast<uint4 *>(dst + head);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) mid[i] = make_uint4(v, v, v, v);
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) dst[threadIdx.x] = (uint8_t)val;
        if (threadIdx.x < tail) dst[head + (n16 << 4) + threadIdx.x] = (uint8_t)val;
    }
}
__global__ __launch_bounds__(256) void k_stag_memset(uint8_t *dst, int val, unsigned long long bytes) { k_stag_memset_impl(dst, val, bytes); }
struct k_stag_memset_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(uint8_t *dst, int val, unsigned long long bytes) const { k_stag_memset_impl(dst, val, bytes); }
};
__device__ __forceinline__ void k_stag_memcpy_impl(uint8_t *dst, const uint8_t *src, unsigned long long bytes)
{
    if ((((unsigned long long)dst | (unsigned long long)src) & 15ull) == 0) {
        const unsigned long long n16 = bytes >> 4;
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        if (blockIdx.x == 0 && threadIdx.x < (bytes & 15ull)) dst[(n16 << 4) + threadIdx.x] = src[(n16 << 4) + threadIdx.x];
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < bytes; i += (unsigned long long)gridDim.x * 256) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void k_stag_memcpy(uint8_t *dst, const uint8_t *src, unsigned long long bytes) { k_stag_memcpy_impl(dst, src, bytes); }
struct k_stag_memcpy_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(uint8_t *dst, const uint8_t *src, unsigned long long bytes) const { k_stag_memcpy_impl(dst, src, bytes); }
};

// pinned host blocks the kernels may write into: host address range -> device alias (registered by fid_stag_create)
struct StagHostAlias {
    const char *host;
    size_t bytes;
    char *dev;
};
extern std::vector<StagHostAlias> g_stag_aliases;
extern std::mutex g_stag_alias_mutex;
static inline void *stag_device_alias(const void *host_ptr, size_t bytes)
{
    std::lock_guard<std::mutex> g(g_stag_alias_mutex);
    for (const auto &a : g_stag_aliases)
        if ((const char *)host_ptr >= a.host && (const char *)host_ptr + bytes <= a.host + a.bytes) return a.dev + ((const char *)host_ptr - a.host);
    return nullptr;
}

template <int SITE>
static inline hipError_t stag_memset_site(void *dst, int val, size_t bytes, hipStream_t st)
{
    if (g_stag_rec && g_stag_rec->on) {
        const unsigned long long n16 = bytes >> 4;
        const unsigned gx = (unsigned)(n16 / 256 + 1 > 1024 ? 1024 : n16 / 256 + 1);
        StagSite<k_stag_memset_fn, SITE>::record(*g_stag_rec, dim3(gx), dim3(256), 0, (uint8_t *)dst, val, (unsigned long long)bytes);
        return hipSuccess;
    }
    return hipMemsetAsync(dst, val, bytes, st);
}
template <int SITE>
static inline hipError_t stag_memcpy_site(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st)
{
    if (g_stag_rec && g_stag_rec->on) {
        void *d = dst;
        bool as_kernel = kind == hipMemcpyDeviceToDevice;
        if (kind == hipMemcpyDeviceToHost && bytes <= 4096) {
            d = stag_device_alias(dst, bytes);
            as_kernel = d != nullptr;
        }
        if (as_kernel) {
            const unsigned long long n16 = bytes >> 4;
            const unsigned gx = (unsigned)(n16 / 256 + 1 > 1024 ? 1024 : n16 / 256 + 1);
            StagSite<k_stag_memcpy_fn, SITE>::record(*g_stag_rec, dim3(gx), dim3(256), 0, (uint8_t *)d, (const uint8_t *)src, (unsigned long long)bytes);
            return hipSuccess;
        }
        stag_order_guard(*g_stag_rec, SITE);
        g_stag_rec->copies.push_back({SITE, 1 + (int)kind, dst, src, 0, bytes});
        g_stag_rec->ops.push_back({SITE, nullptr, nullptr});
        return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, bytes, kind, st);
}
#define STAG_MEMSET(dst, val, bytes, st) stag_memset_site<__COUNTER__>(dst, val, bytes, st)
#define STAG_MEMCPY(dst, src, bytes, kind, st) stag_memcpy_site<__COUNTER__>(dst, src, bytes, kind, st)

// issue everything that was recorded, site by site in source order
static inline bool stag_flush(StagRecorder &R)
{
    // stable order by site; the copies of a site go out in recording order
    std::vector<int> sites;
    for (const auto &o : R.ops) sites.push_back(o.site);
    std::sort(sites.begin(), sites.end());
    sites.erase(std::unique(sites.begin(), sites.end()), sites.end());
    for (int s : sites) {
        // (a copy site can hold BOTH kinds in one round: a frame whose few bytes go through the alias kernel and a frame whose
        //  larger block takes the copy engine -- e.g. the marker hand-over of a frame with 12 and of one with 40 markers.  Until
        //  round 5 the kernel kind made the loop skip the site's copies: the second frame handed over stale markers.)
        for (const auto &o : R.ops)
            if (o.site == s && o.flush) {
                o.flush(R, o.state);
                break;
            }
        for (const auto &c : R.copies) {
            if (c.site != s) continue;
            const hipError_t e = c.kind == 0 ? hipMemsetAsync(c.dst, c.val, c.bytes, R.stream)
                                             : hipMemcpyAsync(c.dst, c.src, c.bytes, (hipMemcpyKind)(c.kind - 1), R.stream);
            if (e != hipSuccess) R.failed = true;
        }
    }
    R.ops.clear();
    R.copies.clear();
    return !R.failed;
}
