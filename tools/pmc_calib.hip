// pmc_calib.hip -- known-byte-count kernels in the access patterns this library uses, to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths uncalibrated).
// Each kernel moves exactly N bytes (N = 1 GiB, far beyond the 256 MiB Infinity Cache) once.
//   rd4    4 B per lane, coalesced rows                 (k_threshold_fixed's dword loads, most list reads)
//   rd1    1 B per lane, coalesced                      (k_threshold_stream's producer wave)
//   rd16   16 B per lane, coalesced                     (k_find_starts' tile loads)
//   rdlds  16 B per lane through LDS-DMA (global_load_lds_dwordx4)   (the walkers' window refills)
//   wr4 / wr16   4 / 16 B per lane stores               (mask words, contour points)
// build + run on the GPU box: tools/gpu_pmc3.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ void rd1(const uint8_t *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ void rd4(const uint32_t *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ void rd16(const uint4 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ void rdlds(const uint4 *p, size_t n, uint32_t *sink)
{
    __shared__ uint4 buf[256];
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // one 16-byte LDS-DMA request per lane: lane l of wave w lands at buf[w * 64 + l]
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + i),
                                         (__attribute__((address_space(3))) void *)(buf + (threadIdx.x & ~63u)), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[threadIdx.x].x;
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ void wr4(uint32_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void wr16(uint4 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main()
{
    const size_t N = (size_t)1 << 30;
    void *a = nullptr, *sink = nullptr;
    if (hipMalloc(&a, N) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(a, 1, N);
    hipDeviceSynchronize();
    const dim3 g(2048), b(256);
    for (int rep = 0; rep < 2; rep++) {
        rd1<<<g, b>>>((const uint8_t *)a, N, (uint32_t *)sink);
        rd4<<<g, b>>>((const uint32_t *)a, N / 4, (uint32_t *)sink);
        rd16<<<g, b>>>((const uint4 *)a, N / 16, (uint32_t *)sink);
        rdlds<<<g, b>>>((const uint4 *)a, N / 16, (uint32_t *)sink);
        wr4<<<g, b>>>((uint32_t *)a, N / 4);
        wr16<<<g, b>>>((uint4 *)a, N / 16);
        hipDeviceSynchronize();
    }
    printf("pmc_calib: every kernel moved %zu bytes, twice\n", N);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
