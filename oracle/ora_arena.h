/*
 * ora_arena.h -- per-thread arena behind malloc / calloc / realloc / free inside aruco_detect_oracle.c.  TEST INFRASTRUCTURE
 * (the oracle is the checker and the reported CPU baseline, never part of the product).
 *
 * Why: the restatement allocates and frees tens of MB per frame (row sums, padded label images, contour lists); with one
 * process per host core glibc hands those back to the kernel every time and 64 - 256 processes then spend their time in page
 * faults and mmap_sem instead of computing (round 2: 64 processes gave 12.9 x one core, median per-frame time 370 ms in the
 * pool against 89 ms alone).  Here every thread keeps ONE region for the life of the process: a bump allocator with block
 * headers; free() marks a block and pops whatever is free at the top, so the region's high-water mark is the working set of
 * one frame and its pages stay mapped and warm.  All outputs of the oracle's entry points are caller-provided, every internal
 * allocation is freed before an entry point returns: the region is empty between calls.
 * -DORA_NO_ARENA switches it off (the sanitizer builds do: they want the real allocator's red zones).
 */
#ifndef ORA_ARENA_H
#define ORA_ARENA_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifndef ORA_NO_ARENA
#include <sys/mman.h>

typedef struct ora_blk {
    size_t size;           /* payload bytes (multiple of 32) */
    struct ora_blk *prev;  /* block below this one */
    size_t freed;
    size_t pad;
} ora_blk;

static __thread char *ora_a_base, *ora_a_top;
static __thread size_t ora_a_cap;
static __thread ora_blk *ora_a_last;

static inline int ora_a_owns(const void *p) { return ora_a_base && (const char *)p >= ora_a_base && (const char *)p < ora_a_base + ora_a_cap; }

static void *ora_a_malloc(size_t n)
{
    if (!ora_a_base) {
        const size_t cap = (size_t)4 << 30; /* address space only: pages are touched as the working set grows */
        void *m = mmap(NULL, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) return malloc(n);
        ora_a_base = ora_a_top = (char *)m;
        ora_a_cap = cap;
    }
    n = (n + 31) & ~(size_t)31;
    if ((size_t)(ora_a_base + ora_a_cap - ora_a_top) < n + sizeof(ora_blk)) return malloc(n); /* (never seen: 4 GB per thread) */
    ora_blk *b = (ora_blk *)ora_a_top;
    b->size = n;
    b->prev = ora_a_last;
    b->freed = 0;
    ora_a_last = b;
    ora_a_top += sizeof(ora_blk) + n;
    return b + 1;
}

static void ora_a_free(void *p)
{
    if (!p) return;
    if (!ora_a_owns(p)) {
        free(p);
        return;
    }
    ((ora_blk *)p - 1)->freed = 1;
    while (ora_a_last && ora_a_last->freed) { /* pop everything that is free at the top */
        ora_a_top = (char *)ora_a_last;
        ora_a_last = ora_a_last->prev;
    }
}

static void *ora_a_calloc(size_t k, size_t n)
{
    void *p = ora_a_malloc(k * n);
    if (p) memset(p, 0, k * n);
    return p;
}

static void *ora_a_realloc(void *p, size_t n)
{
    if (!p) return ora_a_malloc(n);
    if (!ora_a_owns(p)) return realloc(p, n);
    ora_blk *b = (ora_blk *)p - 1;
    const size_t n32 = (n + 31) & ~(size_t)31;
    if (b == ora_a_last && (size_t)(ora_a_base + ora_a_cap - (char *)p) >= n32) { /* the top block grows / shrinks in place */
        b->size = n32;
        ora_a_top = (char *)p + n32;
        return p;
    }
    if (n32 <= b->size) return p;
    void *q = ora_a_malloc(n);
    if (q) {
        memcpy(q, p, b->size);
        ora_a_free(p);
    }
    return q;
}

#define malloc ora_a_malloc
#define calloc ora_a_calloc
#define realloc ora_a_realloc
#define free ora_a_free
#endif /* ORA_NO_ARENA */
#endif
