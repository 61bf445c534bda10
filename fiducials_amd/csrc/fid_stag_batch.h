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

#include "fid_device.h"

#include <algorithm>
#include <type_traits>
#include <mutex>
#include <utility>
#include <vector>

#ifndef STAG_MAXF
// Frames per merged launch.  Rounds 3 - 5 kept every frame's argument tuple BY VALUE in the launch's 4 KB of kernel-argument memory:
// ~250 bytes per frame for the routing kernels, i.e. groups of 16.  Round 5 tried the routing context behind a pointer (32 frames):
// pointers LOADED from memory are generic pointers to the compiler (flat_load / flat_store), k_stag_route_walk 1.40 -> 2.15 ms.
// Round 6: the table holds frame 0's tuple once and, per frame, two 64-bit DELTAS and one optional scalar.  Every context carves all
// its device buffers out of ONE slab at the same offsets (fid_stag_create), so frame f's pointer = frame 0's + (slab_f - slab_0);
// the few pointers into a context's pinned host block (the alias kernels that carry counters back) move by a second delta.  The
// trampoline rebuilds frame f's tuple from frame 0's with pointer arithmetic on kernel-argument pointers -- which stay GLOBAL
// pointers -- in a handful of scalar instructions.  32 bytes a frame: a launch carries up to ~110 frames, groups of 64 by default.
#define STAG_MAXF 64
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

// ---- walking over everything a tuple holds, in a fixed order: pointers and scalars, through arrays and through the argument
// structs (each has a visit() that names its members).  The host flattens a tuple into 64-bit words with it (to find out how a
// frame's arguments differ from frame 0's), the device rebuilds frame f's tuple with it.
template <typename T, typename = void>
struct stag_has_visit : std::false_type {
};
template <typename T>
struct stag_has_visit<T, std::void_t<decltype(std::declval<T &>().visit(std::declval<int (*)(int)>()))>> : std::true_type {
};
template <typename T, typename V>
__host__ __device__ __forceinline__ void stag_visit(T &x, V &v)
{
    if constexpr (std::is_pointer<T>::value) {
        v.ptr(x);
    } else if constexpr (std::is_array<T>::value) {
#pragma unroll
        for (size_t i = 0; i < std::extent<T>::value; i++) stag_visit(x[i], v);
    } else if constexpr (std::is_arithmetic<T>::value || std::is_enum<T>::value) {
        v.scalar(x);
    } else {
        x.visit([&v](auto &m) { stag_visit(m, v); });
    }
}
template <typename V>
__host__ __device__ __forceinline__ void stag_visit_tup(StagTup<> &, V &)
{
}
template <typename H, typename... T, typename V>
__host__ __device__ __forceinline__ void stag_visit_tup(StagTup<H, T...> &t, V &v)
{
    stag_visit(t.head, v);
    stag_visit_tup(t.tail, v);
}
template <typename S>
__host__ __device__ __forceinline__ unsigned long long stag_scalar_bits(const S &x)
{
    static_assert(sizeof(S) <= 8, "a kernel argument scalar of more than 8 bytes");
    unsigned long long u = 0;
    __builtin_memcpy(&u, &x, sizeof(S));
    return u;
}
#define STAG_MAX_WORDS 96  // pointers + scalars of one argument tuple (the routing kernels: ~45)
struct StagFlat {  // host: a tuple as 64-bit words
    unsigned long long v[STAG_MAX_WORDS];
    unsigned char is_ptr[STAG_MAX_WORDS];
    int n = 0;
    bool too_long = false;
    template <typename P>
    void ptr(P &p)
    {
        if (n < STAG_MAX_WORDS) { v[n] = (unsigned long long)(uintptr_t)p; is_ptr[n++] = 1; } else too_long = true;
    }
    template <typename S>
    void scalar(S &x)
    {
        if (n < STAG_MAX_WORDS) { v[n] = stag_scalar_bits(x); is_ptr[n++] = 0; } else too_long = true;
    }
};
struct StagRebase {  // device: frame 0's tuple -> frame f's
    long long dS, dP;
    unsigned long long cls[3];  // two bits per word: 0 = as it is, 1 = + dS (the context's slab), 2 = + dP (its pinned block)
    int var_slot;               // the one scalar word that differs between frames (-1: none)
    unsigned long long var;
    int i;
    template <typename P>
    __device__ __forceinline__ void ptr(P &p)
    {
        const unsigned c = (unsigned)(cls[i >> 5] >> ((i & 31) * 2)) & 3u;
        const long long d = c == 1 ? dS : (c == 2 ? dP : 0);
        if (p) p = (P)((const char *)p + d);
        i++;
    }
    template <typename S>
    __device__ __forceinline__ void scalar(S &x)
    {
        if (i == var_slot) __builtin_memcpy(&x, &var, sizeof(S));
        i++;
    }
};

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
    // kernel-argument memory is 4 KB: frame 0's tuple, the class bits, 32 bytes per frame
    static constexpr int kFit = (4032 - (int)sizeof(tup) - 48) / 32;
    static constexpr int kMax = kFit < 1 ? 1 : (kFit > STAG_MAXF ? STAG_MAXF : kFit);
    tup a0;
    unsigned long long cls[3];
    int var_slot, pad_;
    long long dS[kMax], dP[kMax];
    unsigned long long var[kMax];
    unsigned gx[kMax], gy[kMax];
};

// Dispatch order.  The hardware hands out the workgroups of a grid x-fastest, so with frames in blockIdx.z a group's launch works
// through frame 0's items, then frame 1's ...: the LONG items of the last frames (a marker's outline to walk, its longest segment to
// fit) start when most of the launch has drained -- at 64 frames the launch ends with them.  A functor that declares kFrameMinor is
// launched with the frames in blockIdx.x and its own x index in blockIdx.z (its body reads STAG_BX<true>()): item 0 of every frame,
// then item 1 of every frame ...; with the items in longest-first order (k_stag_comp_tilemax's list) the long ones all start at once.
template <typename F, typename = void>
struct stag_frame_minor : std::false_type {
};
template <typename F>
struct stag_frame_minor<F, std::void_t<decltype(F::kFrameMinor)>> : std::integral_constant<bool, F::kFrameMinor> {
};
template <typename Fn>
__global__ __launch_bounds__(Fn::kBounds) void k_stag_batch(const StagTab<Fn> tab)
{
    constexpr bool FM = stag_frame_minor<Fn>::value;
    const int f = FM ? blockIdx.x : blockIdx.z;
    if ((FM ? blockIdx.z : blockIdx.x) >= tab.gx[f] || blockIdx.y >= tab.gy[f]) return;
    typename StagTab<Fn>::tup a = tab.a0;
    StagRebase rb = {tab.dS[f], tab.dP[f], {tab.cls[0], tab.cls[1], tab.cls[2]}, tab.var_slot, tab.var[f], 0};
    stag_visit_tup(a, rb);
    StagTab<Fn>::sig::template call<Fn>(a, std::make_index_sequence<StagTab<Fn>::sig::n>{});
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
    long long merged_launches = 0, recorded_launches = 0, order_flushes = 0, unmergeable = 0;
    int frame_last_site = -1;  // the last site the frame now recording has used in this round (stag_order_guard)
    // the frame now recording: where its context's slab and its pinned block (device alias) lie
    const char *slab = nullptr, *pin = nullptr;
    size_t slab_bytes = 0, pin_bytes = 0;
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
    static thread_local StagFlat flat0;
    static thread_local const char *slab0, *pin0;
    static thread_local int n;
    static thread_local dim3 block;
    static thread_local size_t lds;
    static thread_local unsigned mx, my;
    static thread_local const char *name;
    static void launch(StagRecorder &R)
    {
        if (n == 0) return;
        if (lds > 0 && name) fid_launch_log(name, block.x * block.y * block.z, lds);
        const bool fm = stag_frame_minor<Fn>::value && mx <= 65535u;  // (blockIdx.z is 16 bits; larger grids keep the frames in z)
        if (stag_frame_minor<Fn>::value && !fm) R.failed = true;  // (cannot happen for the functors that ask for it: their grids are components / segments of a frame)
        hipLaunchKernelGGL(k_stag_batch<Fn>, fm ? dim3((unsigned)n, my, mx) : dim3(mx, my, (unsigned)n), block, lds, R.stream, tab);
        if (hipGetLastError() != hipSuccess) R.failed = true;
        R.merged_launches++;
        n = 0;
        mx = my = 0;
        lds = 0;
    }
    static void flush(StagRecorder &R, void *) { launch(R); }
    // can the frame whose tuple flattens to `fl` ride in the launch that frame 0 opened?  Every pointer must be frame 0's moved by
    // this frame's slab / pinned-block delta (or the very same pointer), every scalar equal -- but for ONE scalar slot per site,
    // which travels per frame (a byte count, an LDS size)
    static bool fits(const StagFlat &fl, long long dS, long long dP, int &var_here)
    {
        if (fl.n != flat0.n || fl.too_long) return false;
        var_here = -1;
        for (int i = 0; i < fl.n; i++) {
            if (fl.is_ptr[i] != flat0.is_ptr[i]) return false;
            if (fl.is_ptr[i]) {
                const unsigned c = (unsigned)(tab.cls[i >> 5] >> ((i & 31) * 2)) & 3u;
                const unsigned long long want = flat0.v[i] ? flat0.v[i] + (unsigned long long)(c == 1 ? dS : (c == 2 ? dP : 0)) : 0ull;
                if (fl.v[i] != want) return false;
            } else if (fl.v[i] != flat0.v[i]) {
                if (var_here >= 0 || (tab.var_slot >= 0 && tab.var_slot != i)) return false;
                var_here = i;
            }
        }
        return true;
    }
    template <typename... V>
    static void record(StagRecorder &R, dim3 grid, dim3 blk, size_t l, const V &...v)
    {
        // a frame's launches stay in order: what it recorded before a flush is issued before what it records after it
        stag_order_guard(R, SITE);
        typename StagTab<Fn>::tup cur;
        stag_fill(cur, v...);
        StagFlat fl;
        stag_visit_tup(cur, fl);
        const long long dS = R.slab - slab0, dP = R.pin - pin0;
        int var_here = -1;
        if (n > 0 && (n == StagTab<Fn>::kMax || !fits(fl, dS, dP, var_here))) {
            // the table is full, or this frame's arguments are not frame 0's moved by its deltas (a context of another size, a
            // buffer outside its slab): everything recorded so far goes out (in site order, this site included), then recording
            // goes on with this frame as a launch's first
            if (n < StagTab<Fn>::kMax) R.unmergeable++;
            (void)stag_flush(R);
        }
        if (n == 0) {
            R.ops.push_back({SITE, &StagSite::flush, nullptr});
            tab.a0 = cur;
            flat0 = fl;
            slab0 = R.slab;
            pin0 = R.pin;
            tab.cls[0] = tab.cls[1] = tab.cls[2] = 0;
            tab.var_slot = -1;
            for (int i = 0; i < fl.n && !fl.too_long; i++) {
                if (!fl.is_ptr[i] || !fl.v[i]) continue;
                const char *p = (const char *)(uintptr_t)fl.v[i];
                const unsigned long long c = (R.slab && p >= R.slab && p < R.slab + R.slab_bytes) ? 1ull : ((R.pin && p >= R.pin && p < R.pin + R.pin_bytes) ? 2ull : 0ull);
                tab.cls[i >> 5] |= c << ((i & 31) * 2);
            }
            if (fl.too_long) tab.var_slot = -2;  // (never matches a slot: such a site launches frame by frame)
            tab.dS[0] = tab.dP[0] = 0;
            tab.var[0] = 0;
        } else {
            if (var_here >= 0 && tab.var_slot < 0) {  // the first frame that differs in a scalar: the frames before it carry frame 0's value
                tab.var_slot = var_here;
                for (int k = 0; k < n; k++) tab.var[k] = flat0.v[var_here];
            }
            tab.dS[n] = dS;
            tab.dP[n] = dP;
            tab.var[n] = tab.var_slot >= 0 ? fl.v[tab.var_slot] : 0;
        }
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
thread_local StagFlat StagSite<Fn, SITE>::flat0;
template <typename Fn, int SITE>
thread_local const char *StagSite<Fn, SITE>::slab0 = nullptr;
template <typename Fn, int SITE>
thread_local const char *StagSite<Fn, SITE>::pin0 = nullptr;
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
template <typename Fn, int SITE>
thread_local const char *StagSite<Fn, SITE>::name = nullptr;

// a launch site of the state machine: issued at once (frame-at-a-time entry points) or recorded (group mode)
#define STAG_LAUNCH(K, grid, block, lds, st, ...)                                                          \
    do {                                                                                                   \
        if (g_stag_rec && g_stag_rec->on) {                                                                \
            using Site_ = StagSite<K##_fn, __COUNTER__>;                                                   \
            Site_::name = #K "[g]";                                                                        \
            Site_::record(*g_stag_rec, grid, block, lds, __VA_ARGS__);                                     \
        } else {                                                                                           \
            if ((size_t)(lds) > 0) fid_launch_log(#K, dim3(block).x * dim3(block).y * dim3(block).z, (size_t)(lds)); \
            hipLaunchKernelGGL(K, grid, block, lds, st, __VA_ARGS__);                                      \
        }                                                                                                  \
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
