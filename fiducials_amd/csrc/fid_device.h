// fid_device.h -- device-side data layout shared by the kernels and the C-ABI host code.
//
// HBM layout (all per context, sized at fid_create for max_batch frames F of W x H):
//   gray      u8   [F][H][W]                  the image the detector sees (aliases the caller's
//                                              buffer for mono8 device input with stride == W)
//   masks     u32  [F][S][TR][TC][16]         13 bit-packed adaptive-threshold masks, TILED: one tile =
//                                              32 px x 16 rows = one 64-byte line, so that the 3x3
//                                              neighbourhoods border following reads stay inside one or
//                                              two lines.  Pixel x of row y is bit (x & 31) of word
//                                              (yy & 15) of tile (yy >> 4, MASK_PADW + (x >> 5)), yy = y + 1;
//                                              pad rows / tile columns are zero so border following never
//                                              bounds-checks
//   starts    uint2 [F][max_starts]           border-following start candidates (all scales)
//   surv1/surv uint2 [F][max_starts]          the starts that survive the short / the long probe pass, compacted
//   contours  uint4 [F][max_contours]         contour slots: start, meta, length (0 = dropped), discovery key
//   chunk_tab u32  [F][max_contours][maxPerim/64+1]  pool chunk of every 64 points of a contour
//   pool      u32  [F][max_chunks][64]        contour points x | y << 16, written while the border is followed
//   cands     DevCand [F][max_cands]          quads leaving _findMarkerContours
//   sorted / filtered DevCand [F][max_cands]  OpenCV order; after reorder + too-close filter
//   near      u32  [F][max_cands][max_cands/32]
//   ident     DevIdent [F][max_cands]
//   markers   fid_marker [F][max_markers]     (pre- and post-subpix)
//   poses     fid_pose_out [F][max_markers]
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// FID_LAUNCH_LOG=<file>: every distinct (kernel, block, dynamic LDS) a launch site with DYNAMIC shared memory asks for, one line
// each, appended as "kernel block lds_bytes".  rocprofv3's kernel trace reports the STATIC group segment only, so this is where
// tools/occupancy.py gets the rest of a workgroup's LDS footprint from.  Off (one load of a flag) unless the variable is set.
static inline void fid_launch_log(const char *kernel, unsigned block, size_t lds)
{
    static const char *path = getenv("FID_LAUNCH_LOG");
    if (!path) return;
    struct Seen { char name[48]; unsigned block; size_t lds; };
    static Seen seen[256];
    static int nseen = 0;
    static volatile int lock = 0;
    while (__sync_lock_test_and_set(&lock, 1)) {}
    bool have = false;
    for (int i = 0; i < nseen && !have; i++) have = seen[i].block == block && seen[i].lds == lds && !strncmp(seen[i].name, kernel, 47);
    if (!have && nseen < 256) {
        strncpy(seen[nseen].name, kernel, 47);
        seen[nseen].name[47] = 0;
        seen[nseen].block = block;
        seen[nseen].lds = lds;
        nseen++;
        if (FILE *fh = fopen(path, "a")) {
            fprintf(fh, "%s %u %zu\n", kernel, block, lds);
            fclose(fh);
        }
    }
    __sync_lock_release(&lock);
}

#define MASK_PADW 1  // zero tile columns in front of every tile row
#define MT_ROWS 16   // rows per mask tile
#define FID_MAX_SCALES 32
#define FID_MAX_CELLS 9  // marker_size + 2*border <= 9 (7x7 dictionaries)

struct DevParams {
    int W, H, gstride, WW, TC, TR, nscales, nframes;  // WW mask words per image row; TC x TR mask tiles per plane
    int win[FID_MAX_SCALES];       // odd window sizes
    int idelta;                    // cvFloor(adaptiveThreshConstant) (THRESH_BINARY_INV)
    int rmax;                      // max window radius
    int minPerim, maxPerim;        // (unsigned)(rate * max(W,H))
    double polyAcc, minCornerDistRate, minMarkerDistRate;
    int minDistToBorder;
    int markerSize, borderBits, cellSize, cellMargin;  // dictionary n, markerBorderBits, px per cell, margin px
    double minOtsuStdDev;
    int maxBorderErr, maxCorr, nMarkers, nbytes;
    int subpixWin, subpixMaxIter;
    double subpixEps;  // squared
    int refine;
    int maxStarts, maxContours, maxCands, maxMarkers;  // per-frame capacities
    int maxChunks;                                     // per-frame pool of CK-point contour chunks
    int seedShift;                                     // tracing seeds: grid spacing G = 8 << seedShift pixels
    int seedHashCap;                                   // entries of the per-frame state -> seed index hash table (power of two)
    int seedGen;                                       // this call's generation number in that table (1..1023)
};

// word index of padded row yy, word column wi inside one (frame, scale) mask plane of TC tile columns
__host__ __device__ inline long long mask_word(int TC, int yy, int wi)
{
    return ((long long)(yy >> 4) * TC + wi) * MT_ROWS + (yy & (MT_ROWS - 1));
}

struct DevCand {
    int scale;
    int size;       // contour.size()
    int sx, sy;     // first contour point
    int hole;
    unsigned key;   // discovery position in the padded raster (outer: start pixel, hole: pixel right of it)
    float c[8];
    unsigned pref;  // where the contour's points are: offset into the frame's dense point array, or 0x80000000 | table row
                    // (whole-border walk) -- CORNER_REFINE_CONTOUR reads them again (k_refine_contour)
    unsigned pad;
};

struct DevIdent {
    int id, rot;
    unsigned char bits[FID_MAX_CELLS * FID_MAX_CELLS + 3];
};

// ---- seed-accelerated contour tracing.  A border-following state is (pixel, direction d back to the previous pixel).
// SEED states are states a walker can recognise from what it holds anyway and that k_find_starts can enumerate from the bit
// planes: the state's pixel lies on a GRID LINE it has just stepped onto -- a column x = 0 (mod G) entered with a horizontal
// component (d in E, NE, NW, W, SW, SE), or a row y = 0 (mod G) entered with a vertical component (d in NE, N, NW, SW, S, SE) --
// and the neighbour X(d) that border following must have found empty on the way in (the one "to the right of travel":
// direction d+2 for an axis move, d+1 for a diagonal one) is background.  Every seed follows its border only to the next
// seed state (a SEGMENT); a probe survivor follows its border only to the first seed state and the rest of the border is read
// off the segment chain.
// Why grid lines: a border cannot move G pixels in x or in y without stepping onto one, so the seed-free stretches are
// BOUNDED (about 2 G steps for anything but a border that curls up inside one G x G cell) -- the longest sequential piece of
// the whole tracing.  (Round 1 / early round 2 used a thinning lattice, one class per pixel: same seed count, but the gaps
// were geometrically distributed, 1434 steps at worst in the 256-frame bench batch against a mean of 110, and the walker
// kernels lasted as long as that one walker.)  Several states of one pixel can be seeds, so a seed is identified by
// (pixel, d): seed records carry d, and the map state -> seed index is a per-frame hash table (SeedHash).
// G = 8 << DevParams::seedShift: 128 px for batches, down to 32 px for single frames (the spacing trades the longest stretch
// against the number of segments).  Any rule yields the same contours.
#define SEED_SHIFT_MIN 2
#define SEED_SHIFT_MAX 5
#define SEED_DIRS_COL 0xBBu  // back directions with a horizontal component: 0 E, 1 NE, 3 NW, 4 W, 5 SW, 7 SE
#define SEED_DIRS_ROW 0xEEu  // back directions with a vertical component:   1 NE, 2 N, 3 NW, 5 SW, 6 S, 7 SE
__host__ __device__ inline bool seed_state(int x, int y, int d, int gmask /* G - 1 */)
{
    return (((x & gmask) == 0) && ((SEED_DIRS_COL >> d) & 1u)) || (((y & gmask) == 0) && ((SEED_DIRS_ROW >> d) & 1u));
}
// state key of a seed / of the state a walker stopped in front of: x | y << 13 | d << 26
__host__ __device__ inline uint32_t seed_key(int x, int y, int d) { return (uint32_t)x | ((uint32_t)y << 13) | ((uint32_t)d << 26); }
// the neighbour direction that is empty when a state with back direction d was entered
__host__ __device__ inline int seed_empty_dir(int d) { return (d + ((d & 1) ? 1 : 2)) & 7; }
#define SEG_INVALID 0xffffffffu
struct DevSeg {           // one per seed
    uint32_t next_key;    // the seed state the segment ran into: x | y << 13 | d << 26 (same scale)
    uint32_t next_idx;    // ... and that seed's index (filled in by k_seg_link)
    uint32_t n;           // states in the segment (SEG_INVALID: longer than maxPerimeterPixels / pool exhausted)
    uint32_t mout;        // min raster index (pidx) over the segment's pixels
    uint32_t mhole;       // min raster index over the background 4-neighbours its searches passed over
    uint32_t pad[3];
};
// Trace mode 2 ("cycle tracing"): the segment record of a seed when the contours are read off the seed cycles alone (no probe
// survivor needed for a border that has a seed).  Same size as DevSeg: the two share their buffer.
//   ko / kh: the smallest discovery key among the segment's states that START a border the way cvFindNextContour /
//   icvFetchContour would (outer: W neighbour background and back direction = first foreground clockwise from NW, key = raster
//   index of the pixel; hole: E neighbour background and back direction = first foreground clockwise from SE, key = raster index
//   of that E neighbour), and where in the segment that state is.  Around a cycle, min ko < min kh means an outer border whose
//   first point is the state with min ko; otherwise a hole border that starts at the state with min kh (k_seg_cycles).
struct DevSegC {
    // first 16 bytes: all that a hop of k_seg_cycles' walk round a cycle reads (one load per hop)
    uint32_t next_idx;    // seed index of next_key (k_seg_link2)
    uint32_t n;           // states in the segment (SEG_INVALID: abandoned)
    uint32_t ko, kh;      // 0xffffffff: none
    // second 16 bytes
    uint32_t next_key;    // the seed state the segment ran into: x | y << 13 | d << 26 (same scale)
    uint32_t pos;         // position of the ko state | position of the kh state << 16
    uint32_t linked;      // some segment runs into this one (k_seg_link2): only such a seed can lie on a cycle
    uint32_t chunk0;      // the pool chunk of the segment's first 64 chain codes (the next 64: chunk0 + 1; beyond: chunk_tab)
};
struct DevPend {          // one per probe survivor that stopped in front of a seed state
    uint32_t p;           // states it walked itself (0 = not stopped: the survivor closed or died on its own)
    uint32_t next_key;    // that seed state: x | y << 13 | d << 26
    uint32_t next_idx;    // its seed index (k_seg_link)
    uint32_t pad;
};

// per-frame counters
struct DevCounts {
    int ncand;      // candidates emitted by k_approx
    int nfilt;      // after too-close filter
    int nacc;       // identified
    int nmark;      // after _filterDetectedMarkers
    int overflow;   // bit0 cands, bit1 markers, bit2 CORNER_REFINE_CONTOUR met a side of fewer than two points (the reference throws)
    int nstarts;    // border-following start candidates found by k_find_starts
    int ncontours;  // contour slots handed out by the full walk pass (dropped walks leave count == 0)
    int nsurv;      // starts that survived the probe pass
    int npool;      // point chunks handed out by the full walk pass
    int nwalk;      // survivors handed to walker waves so far (work queue head of the full pass)
    int nsurv1;     // starts that survived the first (short) probe pass
    int ndense;     // contour points copied to the dense point array so far
    int nseeds;     // seeds found by k_find_starts<true>
    int nwalk2;     // work queue head of the seed walker
    int nrec;       // copy records written by k_seg_chain / k_seg_cycles (pieces of accepted contours)
    int ncontours2; // trace mode 2: contours of the borders WITHOUT a seed (second list: [maxContours / 2, maxContours) of the frame's
    int nrec2;      //   contour arrays, second half of its record array), traced on the auxiliary stream beside the seed cycles
    int pad[15];
};

// global counters
struct DevGlobal {
    unsigned overflow;  // bit0 starts/survivors, bit1 contours, bit2 approxPolyDP stack, bit3 point pool
    unsigned pad[3];
    unsigned long long dbg[32];  // FID_DEBUG_STATS builds: walk-loop statistics
};
