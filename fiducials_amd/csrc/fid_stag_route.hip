// fid_stag_route.hip -- STag row s4: edge routing (JoinAnchorPointsUsingSortedAnchors), sequential and component-parallel.
// Part of the fid_stag.hip translation unit (included there; not compiled on its own).
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

// the x index of a workgroup as a kernel body sees it: blockIdx.x, or blockIdx.z when a group of frames is launched with the frames
// in blockIdx.x (fid_stag_batch.h, kFrameMinor)
template <bool FM>
__device__ __forceinline__ unsigned STAG_BX()
{
    return FM ? blockIdx.z : blockIdx.x;
}

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
    // (fid_stag_batch.h: every member in declaration order -- a group's launch rebuilds frame f's copy from frame 0's)
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(grad); v(dir); v(edge); v(W); v(H); v(pix); v(stack); v(chains); v(chainNos); v(capPix); v(capStack); v(capChains); v(capNos);
        v(outpix); v(segs); v(capOut); v(capSegs); v(counters);
    }
};

__device__ __forceinline__ bool sr_near(int2 a, int2 b)
{
    int dr = a.x - b.x, dc = a.y - b.y;
    dr = dr < 0 ? -dr : dr;
    dc = dc < 0 ? -dc : dc;
    return dr <= 1 && dc <= 1;
}

// A value that every lane of the wave holds alike, SAID SO to the compiler (round 5).  The wave-wide walkers below run the same
// scalar program in all 64 lanes; what they load (LDS tile words, stack entries, component records) comes back in vector
// registers, and everything computed from it was vector work under EXEC-mask branches: ~170 VALU instructions and 25 masked
// branches per walked pixel.  Through v_readfirstlane the loaded words are scalar values: the step's decisions become s_cmp /
// s_cselect / s_cbranch_scc on the scalar unit, the loop state lives in SGPRs.
__device__ __forceinline__ int sr_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// (a pointer that can only be LDS: two roads that differ in the memory they read are never merged into one generic access)
typedef __attribute__((address_space(3))) int *SrLdsInt;
struct SrLdsInt4 {  // int4 entries behind such a pointer
    SrLdsInt p;
    __device__ __forceinline__ int4 get(int i) const { return make_int4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]); }
    __device__ __forceinline__ void put(int i, int4 v) const { p[4 * i] = v.x; p[4 * i + 1] = v.y; p[4 * i + 2] = v.z; p[4 * i + 3] = v.w; }
};
template <class T>
__device__ __forceinline__ T sr_uni_struct(const T &v)
{
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T o;
    const int *src = reinterpret_cast<const int *>(&v);
    int *dst = reinterpret_cast<int *>(&o);
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; k++) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
    return o;
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
    // component-parallel extraction with the current block staged in LDS (R.outpix then points INTO LDS, shifted so that the
    // block's absolute indices still apply): the one read in front of the block goes to the arena in global memory
    const int2 *gout = nullptr;
#ifdef RW_TIMING
    unsigned long long ex_t[4] = {0, 0, 0, 0};  // longest() calls, copies of the first segment, the remaining chains' loop
#endif
    int wlane = -1;  // >= 0: a whole wave runs the extraction (identical scalar work in every lane, copies spread over the lanes)
    int wl_len = 0, wl_dup = 0, wl_chains = 0;  // what walk_anchor() left behind

    __device__ int2 cpx(int ch, int i) const
    {
        const int k = sr_uni(R.chains[ch].pix) + i;
        return k >= 0 ? sr_uni_struct(R.pix[k]) : make_int2(-1000, -1000);  // (the reference reads in front of its array there)
    }
    __device__ int2 seg(int i) const
    {
        const int k = segbase + i;
        if (par && k < blk0) return (prev_valid && k >= 0) ? sr_uni_struct(gout[k]) : make_int2(-1000, -1000);  // in front of this anchor's block
        return k >= 0 ? sr_uni_struct(R.outpix[k]) : make_int2(-1000, -1000);
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
        const int2 *src = R.pix + sr_uni(R.chains[cn].pix);
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
        if (root == -1 || sr_uni(ch[root].len) == 0) return 0;
        int sp = 0, ret = 0;
        R.stack[sp++] = make_int4(root, 0, 0, 0);
        while (sp > 0) {
            int4 e = sr_uni_struct(R.stack[sp - 1]);
            const int node = e.x;
            if (e.y == 0) {
                e.y = 1;
                R.stack[sp - 1] = e;
                const int c = sr_uni(ch[node].child[0]);
                if (c != -1 && sr_uni(ch[c].len) != 0) {
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
                const int c = sr_uni(ch[node].child[1]);
                if (c != -1 && sr_uni(ch[c].len) != 0) {
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
            ret = sr_uni(ch[node].len) + mx;
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
            const int c0 = sr_uni(R.chains[root].child[0]);
            root = c0 != -1 ? c0 : sr_uni(R.chains[root].child[1]);
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
            const int cn = sr_uni(R.chainNos[k]);
            trim_tail(cpx(cn, 0));
            int start = 0;
            const int L = sr_uni(ch[cn].len);
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

    // ---- the walk.  One body for two kinds of memory (what a step reads and writes is the same in both):
    //   MemGlobal  one lane, plain loads and stores in the image arrays (the sequential road)
    //   MemWave    a whole wave: control flow and bookkeeping wave-uniform (every lane computes them, lane 0 stores them); what
    //              a step needs -- edge / gradient / direction of the three pixels ahead, edge of the two beside -- is
    //              fetched by eleven lanes at once, one round trip per step instead of a chain of dependent loads
    //   (Tile / SparseTile: a copy of the component's bounding box in LDS, one 16-bit word per pixel -- gradient 11 bits,
    //              direction 1 bit, edge state 2 bits -- have a walker of their own, walk_anchor_tile6 below)
    // None of the five pixels a step reads is written in the same step (the current pixel and the two beside it are not among
    // the three ahead), so fetching them together sees exactly what the reference's one-by-one reads see.
    struct Ahead {           // A = ahead - p, B = ahead, C = ahead + p (p = one pixel across the walking direction)
        int eA, eB, eC, gA, gB, gC, dA, dB, dC;
        int s1, s2;          // edge values of the pixels beside the current one (+p, -p)
    };
    struct MemGlobal {
        const int16_t *grad; const uint8_t *dir; uint8_t *edge; int W;
        int4 *gstk;  // the walk's stack of pending branches (R.stack)
        static constexpr int STRIDE = 1;
        __device__ void stk_put(int i, int4 v, bool writer) const { if (writer) gstk[i] = v; }
        __device__ int4 stk_get(int i) const { return gstk[i]; }
        __device__ int edge_at(int r, int c) const { return edge[r * W + c]; }
        __device__ int dir_at(int r, int c) const { return dir[r * W + c]; }
        __device__ void fetch(int r, int c, int ar, int ac, int pr, int pc, int, Ahead &n) const
        {
            const int q = (r + ar) * W + (c + ac), p = pr * W + pc;
            n.eA = edge[q - p]; n.eB = edge[q]; n.eC = edge[q + p];
            n.gA = grad[q - p]; n.gB = grad[q]; n.gC = grad[q + p];
            n.dA = dir[q - p]; n.dB = dir[q]; n.dC = dir[q + p];
            n.s1 = edge[r * W + c + p]; n.s2 = edge[r * W + c - p];
        }
        __device__ void mark(int r, int c, int pr, int pc, const Ahead &n, bool writer) const
        {
            if (!writer) return;
            const int i = r * W + c, p = pr * W + pc;
            edge[i] = STAG_EDGE_PIXEL;
            if (n.s1 == STAG_ANCHOR_PIXEL) edge[i + p] = 0;
            if (n.s2 == STAG_ANCHOR_PIXEL) edge[i - p] = 0;
        }
        __device__ void erase(int r, int c) const { edge[r * W + c] = 0; }
    };
    // the wave-wide forms keep the first `lcap` stack entries in LDS: a chain begins with a pop of what the chain before has
    // just pushed, and through global memory that was a store, a wait for it and a load -- one memory round trip and a half per
    // chain, a few hundred chains per marker: most of what the walk kernel took (round 5)
    struct MemWave : MemGlobal {
        SrLdsInt4 lstk; int lcap;
        static constexpr int STRIDE = 64;
        __device__ void stk_put(int i, int4 v, bool writer) const
        {
            if (!writer) return;
            if (i < lcap) lstk.put(i, v);
            else gstk[i] = v;
        }
        // (plain vector values on this road: with its one memory round trip per step the scalar form's extra taken branches cost
        //  more than its cheaper arithmetic saves -- measured, round 5)
        __device__ int4 stk_get(int i) const { return i < lcap ? lstk.get(i) : gstk[i]; }
        __device__ void fetch(int r, int c, int ar, int ac, int pr, int pc, int lane, Ahead &n) const
        {
            // lane -> (array, pixel): 0-2 edge, 3-5 grad, 6-8 dir of A, B, C; 9, 10 edge beside
            const int nr = r + ar, nc = c + ac;
            const int side = lane % 3 - 1;
            const int q = (nr + side * pr) * W + (nc + side * pc);
            int v = 0;
            if (lane < 3) v = edge[q];
            else if (lane < 6) v = grad[q];
            else if (lane < 9) v = dir[q];
            else if (lane == 9) v = edge[(r + pr) * W + (c + pc)];
            else if (lane == 10) v = edge[(r - pr) * W + (c - pc)];
            n.eA = __builtin_amdgcn_readlane(v, 0); n.eB = __builtin_amdgcn_readlane(v, 1); n.eC = __builtin_amdgcn_readlane(v, 2);
            n.gA = __builtin_amdgcn_readlane(v, 3); n.gB = __builtin_amdgcn_readlane(v, 4); n.gC = __builtin_amdgcn_readlane(v, 5);
            n.dA = __builtin_amdgcn_readlane(v, 6); n.dB = __builtin_amdgcn_readlane(v, 7); n.dC = __builtin_amdgcn_readlane(v, 8);
            n.s1 = __builtin_amdgcn_readlane(v, 9); n.s2 = __builtin_amdgcn_readlane(v, 10);
        }
    };
    struct Tile {
        uint16_t *t;
        int r0, c0, tw;  // origin (row, column) of the tile in the image, words per tile row
        int4 *gstk; SrLdsInt4 lstk; int lcap;  // the stack: as MemWave
        static constexpr int STRIDE = 64;
        static constexpr bool SPARSE = false;
        __device__ void stk_put(int i, int4 v, bool writer) const
        {
            if (!writer) return;
            if (i < lcap) lstk.put(i, v);
            else gstk[i] = v;
        }
        __device__ int4 stk_get(int i) const { return sr_uni_struct(i < lcap ? lstk.get(i) : gstk[i]); }
        __device__ int idx(int r, int c) const { return (r - r0) * tw + (c - c0); }
        __device__ static int edge_of(uint16_t w) { const int st = w >> 12; return st ? 253 + st : 0; }
        __device__ static int grad_of(uint16_t w) { return w & 0x7ff; }
        __device__ static int dir_of(uint16_t w) { return (w & 0x800) ? STAG_EDGE_VERTICAL : STAG_EDGE_HORIZONTAL; }
        // (every lane reads the same word: the value is the wave's, see sr_uni)
        __device__ int word_at(int r, int c) const { return sr_uni(t[idx(r, c)]); }
        __device__ int edge_at(int r, int c) const { return edge_of((uint16_t)word_at(r, c)); }
        __device__ int dir_at(int r, int c) const { return dir_of((uint16_t)word_at(r, c)); }
        __device__ void set_state(int i, int st) const { t[i] = (uint16_t)((t[i] & 0x0fff) | (st << 12)); }
        __device__ void erase(int r, int c) const { set_state(idx(r, c), 0); }  // (a pixel listed twice gets the same word twice)
    };

    // A component whose bounding box does not fit the LDS as a dense tile -- the ring around a marker seen from close: 289 x 288
    // pixels of box for 813 pixels of edge -- as 4 x 4 BLOCKS: a table over the box (one 16-bit slot number per block, 0 = the
    // all-empty block) and the blocks that hold a pixel of the component or a neighbour of one.  Two dependent LDS reads per
    // access instead of one; the walk through global memory it replaces cost a store, a wait and a load per step (2 200 cycles
    // a pixel, and that one workgroup was the whole kernel's duration: round 5).
    struct SparseTile {
        uint16_t *tab, *blk;
        int r0, c0, bw;   // origin of the box in the image (as Tile), blocks per table row
        int4 *gstk; SrLdsInt4 lstk; int lcap;
        static constexpr int STRIDE = 64;
        static constexpr bool SPARSE = true;
        __device__ void stk_put(int i, int4 v, bool writer) const
        {
            if (!writer) return;
            if (i < lcap) lstk.put(i, v);
            else gstk[i] = v;
        }
        __device__ int4 stk_get(int i) const { return sr_uni_struct(i < lcap ? lstk.get(i) : gstk[i]); }
        // word offset in blk of the pixel (rr, cc) of the box
        static constexpr int SH = 2, BM = (1 << SH) - 1;  // blocks of 4 x 4 pixels: a one-pixel-wide outline fills a quarter of each
        __device__ int off(int rr, int cc) const
        {
            const int slot = tab[(rr >> SH) * bw + (cc >> SH)];
            return (slot << (2 * SH)) | ((rr & BM) << SH) | (cc & BM);
        }
        __device__ int word_at(int r, int c) const { return sr_uni(blk[off(r - r0, c - c0)]); }
        __device__ int dir_at(int r, int c) const { return Tile::dir_of((uint16_t)word_at(r, c)); }
        __device__ void erase(int r, int c) const
        {
            const int o = off(r - r0, c - c0);
            blk[o] = (uint16_t)(blk[o] & 0x0fff);
        }
    };

    // the walk from one anchor: true if it produced a path that is kept (the chain tree is then in R.chains / R.pix)
    template <class Mem>
    __device__ bool walk_anchor_t(int r0, int c0, int grad_thresh, const Mem &M, int lane)
    {
        const bool L0 = lane == 0;
        StagChain *ch = R.chains;
        const int capChains = R.capChains, capPix = R.capPix, capStack = R.capStack;
        if (L0) {
            ch[0].dir = 0; ch[0].len = 0; ch[0].parent = -1; ch[0].child[0] = ch[0].child[1] = -1; ch[0].pix = -1;
        }
        int noChains = 1, len = 0, dup = 0, top = -1;
        const bool vert0 = M.dir_at(r0, c0) == STAG_EDGE_VERTICAL;
        M.stk_put(0, make_int4(r0, c0, vert0 ? SR_DOWN : SR_RIGHT, 0), L0);
        M.stk_put(1, make_int4(r0, c0, vert0 ? SR_UP : SR_LEFT, 0), L0);
        top = 1;
        while (top >= 0) {
            const int4 e = M.stk_get(top);
            top--;
            int r = e.x, c = e.y;
            const int d = e.z, parent = e.w;
            if (noChains >= capChains || len + 2 >= capPix || top + 3 >= capStack) {
                overflow |= 16;
                break;
            }
            if (M.edge_at(r, c) != STAG_EDGE_PIXEL) dup++;
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
            const int pr = horiz ? 1 : 0, pc = horiz ? 0 : 1;            // across the walking direction
            const int fs = (d == SR_LEFT || d == SR_UP) ? -1 : 1;         // which diagonal is looked at first
            const int slot = (d == SR_LEFT || d == SR_UP) ? 0 : 1;
            bool stopped = false;
            int curdir = M.dir_at(r, c);
            while (curdir == need) {
                Ahead n;
                M.fetch(r, c, ar, ac, pr, pc, lane, n);
                M.mark(r, c, pr, pc, n, L0);
                const int eF1 = fs < 0 ? n.eA : n.eC, eF2 = fs < 0 ? n.eC : n.eA;  // the diagonal looked at first / second
                int side;
                if (n.eB >= STAG_ANCHOR_PIXEL) side = 0;
                else if (eF1 >= STAG_ANCHOR_PIXEL) side = fs;
                else if (eF2 >= STAG_ANCHOR_PIXEL) side = -fs;
                else {
                    side = 0;
                    if (n.gA > n.gB) side = n.gA > n.gC ? -1 : 1;
                    else if (n.gC > n.gB) side = 1;
                }
                r = r + ar + side * pr;
                c = c + ac + side * pc;
                const int en = side < 0 ? n.eA : side > 0 ? n.eC : n.eB, gn = side < 0 ? n.gA : side > 0 ? n.gC : n.gB;
                curdir = side < 0 ? n.dA : side > 0 ? n.dC : n.dB;
                if (en == STAG_EDGE_PIXEL || gn < grad_thresh) {
                    if (L0) {
                        ch[cur].len = (uint16_t)chainLen;
                        ch[parent].child[slot] = (int16_t)cur;
                    }
                    noChains++;
                    stopped = true;
                    break;
                }
                if (len + 2 >= capPix) { overflow |= 16; stopped = true; break; }
                if (L0) R.pix[len] = make_int2(r, c);
                len++;
                chainLen++;
            }
            if (stopped) continue;
            // the edge turns here: branch both ways across, this chain ends in front of the turning pixel
            M.stk_put(top + 1, make_int4(r, c, horiz ? SR_DOWN : SR_RIGHT, cur), L0);
            M.stk_put(top + 2, make_int4(r, c, horiz ? SR_UP : SR_LEFT, cur), L0);
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
            for (int k = lane; k < len; k += Mem::STRIDE) M.erase(R.pix[k].x, R.pix[k].y);
            return false;
        }
        return true;
    }
    // The walk on the LDS tile, lane-parallel (round 5).  A lone wave issues one instruction every ~4.5 cycles whatever the unit, and
    // the one-body-for-all-memories walk above spends ~200 instructions on a pixel (900 cycles measured: 1 000 pixels and 300
    // chains of a marker's component took 1.2 M cycles, scalar or vector alike).  Here SIX LANES hold the six tile words of a
    // step -- lane 0 / 1 / 2 the pixels ahead (A, B, C = ahead - p, ahead, ahead + p), lane 3 / 4 the pixels beside (+p, -p),
    // lane 5 the pixel the walk stands on -- so that one ds_read, one shift, two compares (ballots) and one masked ds_write do
    // what five decodes and three read-modify-writes did; the decision runs on the scalar unit from the ballots and a packed
    // table, the chosen pixel's word comes over by v_readlane.  The same reads, writes and decisions in the same order as
    // walk_anchor_t (tests: every segment of every road against the reference's).
    template <class TT>
    __device__ bool walk_anchor_tile6(int r0, int c0, int grad_thresh, int lane, const TT &T)
    {
        constexpr bool SP = TT::SPARSE;
        StagChain *ch = R.chains;
        const bool L0 = lane == 0;
        const int capChains = sr_uni(R.capChains), capPix = sr_uni(R.capPix), capStack = sr_uni(R.capStack);
        int tw = 0;
        uint16_t *t;
        if constexpr (SP) t = T.blk;
        else {
            t = T.t;
            tw = T.tw;
        }
        // (a chain record is 16 bytes: dir, len | parent, child[0] | child[1], pad | pix -- begun with ONE store)
        static_assert(sizeof(StagChain) == 16, "one 16-byte store per chain record");
        if (L0) *reinterpret_cast<uint4 *>(&ch[0]) = make_uint4(0u, 0xffffffffu, 0xffffu, 0xffffffffu);  // dir 0, len 0, no parent, no children, pix -1
        int noChains = 1, len = 0, dup = 0, top = -1;
        const bool vert0 = T.dir_at(r0, c0) == STAG_EDGE_VERTICAL;
        T.stk_put(0, make_int4(r0, c0, vert0 ? SR_DOWN : SR_RIGHT, 0), L0);
        T.stk_put(1, make_int4(r0, c0, vert0 ? SR_UP : SR_LEFT, 0), L0);
        top = 1;
        const int orc = lane == 5 ? (2 << 12) : 0;  // the pixel stood on becomes an edge pixel, the anchors beside it are cleared
        // this lane's pixel relative to the one stood on = ka x (one ahead) + kp x (one across)
        const int ka = lane < 3 ? 1 : 0, kp = (lane == 2 || lane == 3) ? 1 : (lane == 0 || lane == 4) ? -1 : 0;
        while (top >= 0) {
            const int4 e = T.stk_get(top);
            top--;
            int r = e.x, c = e.y;
            const int d = e.z, parent = e.w;
            if (noChains >= capChains || len + 2 >= capPix || top + 3 >= capStack) {
                overflow |= 16;
                break;
            }
            const bool horiz = d == SR_LEFT || d == SR_RIGHT;
            const int ar = d == SR_UP ? -1 : d == SR_DOWN ? 1 : 0, ac = d == SR_LEFT ? -1 : d == SR_RIGHT ? 1 : 0;
            const int a = ar * tw + ac, p = horiz ? tw : 1;  // one pixel ahead / across, in tile words
            const bool low = d == SR_LEFT || d == SR_UP;    // fs = -1: the diagonal A is looked at first
            const int slot = low ? 0 : 1;
            // which way for every pattern of non-empty pixels ahead (bit 0 A, 1 B, 2 C), two bits each: 0 / 1 / 2 = side -1 / 0 / +1,
            // 3 = none, the gradients decide
            const unsigned tab = 21075u + (low ? 0u : 2048u);
            // this lane's pixel relative to the one stood on: as a tile offset (dense), as (rows, columns) (blocks)
            const int voff = ka * a + kp * p;
            const int pr = horiz ? 1 : 0, pc = horiz ? 0 : 1;
            const int ldr = ka * ar + kp * pr, ldc = ka * ac + kp * pc;
            int i = 0, o = 0;  // where this lane reads (and, lanes 3 - 5, writes)
            if constexpr (SP) o = T.off(r - T.r0 + ldr, c - T.c0 + ldc);
            else {
                i = T.idx(r, c);
                o = i + voff;
            }
            int w = t[o];
            const int w0 = __builtin_amdgcn_readlane(w, 5);
            if ((w0 >> 12) != 2) dup++;
            const int cur = noChains;
            if (L0) {
                // (len is written again where the chain ends; the two bytes of padding are nobody's)
                *reinterpret_cast<uint4 *>(&ch[cur]) = make_uint4((unsigned)d & 0xffffu, ((unsigned)parent & 0xffffu) | 0xffff0000u, 0xffffu, (unsigned)len);
                R.pix[len] = make_int2(r, c);
            }
            len++;
            int chainLen = 1;
            int why = 0;  // why the chain ends: 0 the edge turns, 1 an edge pixel or a weak gradient ahead, 2 no room for its pixels
            bool along = ((w0 & 0x800) != 0) == !horiz;  // the pixel's edge direction is the chain's
            while (along) {
                const int st = w >> 12;
                const unsigned nz = (unsigned)__ballot(st != 0);
                // mark: lanes 3, 4 clear their anchor, lane 5 writes the edge pixel -- from the words just read
                if (lane == 5 || ((lane == 3 || lane == 4) && st == 1)) t[o] = (uint16_t)((w & 0x0fff) | orc);
                int side = (int)((tab >> (2 * (nz & 7u))) & 3u) - 1;
                if (__builtin_expect(side == 2, 0)) {
                    const int gA = __builtin_amdgcn_readlane(w, 0) & 0x7ff, gB = __builtin_amdgcn_readlane(w, 1) & 0x7ff,
                              gC = __builtin_amdgcn_readlane(w, 2) & 0x7ff;
                    side = gA > gB ? (gA > gC ? -1 : 1) : (gC > gB ? 1 : 0);
                }
                const int wn = __builtin_amdgcn_readlane(w, side + 1);
                if constexpr (!SP) i += a + side * p;
                r += ar + (horiz ? side : 0);
                c += ac + (horiz ? 0 : side);
                along = ((wn & 0x800) != 0) == !horiz;
                why = ((wn >> 12) == 2 || (wn & 0x7ff) < grad_thresh) ? 1 : (len + 2 >= capPix) ? 2 : 0;
                if (__builtin_expect(why != 0, 0)) break;
                if (L0) R.pix[len] = make_int2(r, c);
                len++;
                chainLen++;
                if constexpr (SP) o = T.off(r - T.r0 + ldr, c - T.c0 + ldc);
                else o = i + voff;
                w = t[o];
            }
            if (why == 2) {
                overflow |= 16;
                continue;
            }
            if (why == 1) {
                if (L0) {
                    ch[cur].len = (uint16_t)chainLen;
                    ch[parent].child[slot] = (int16_t)cur;
                }
                noChains++;
                continue;
            }
            // the edge turns here: branch both ways across, this chain ends in front of the turning pixel
            T.stk_put(top + 1, make_int4(r, c, horiz ? SR_DOWN : SR_RIGHT, cur), L0);
            T.stk_put(top + 2, make_int4(r, c, horiz ? SR_UP : SR_LEFT, cur), L0);
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
            for (int k = lane; k < len; k += 64) T.erase(R.pix[k].x, R.pix[k].y);
            return false;
        }
        return true;
    }
    __device__ bool walk_anchor(int r0, int c0, int grad_thresh)
    {
        MemGlobal M;
        M.grad = R.grad; M.dir = R.dir; M.edge = R.edge; M.W = R.W; M.gstk = R.stack;
        return walk_anchor_t(r0, c0, grad_thresh, M, 0);
    }
    __device__ bool walk_anchor_wave(int r0, int c0, int grad_thresh, int lane, SrLdsInt4 lstk, int lcap)
    {
        MemWave M;
        M.grad = R.grad; M.dir = R.dir; M.edge = R.edge; M.W = R.W; M.gstk = R.stack; M.lstk = lstk; M.lcap = lcap;
        return walk_anchor_t(r0, c0, grad_thresh, M, lane);
    }

    __device__ void route_anchor(int r0, int c0, int grad_thresh)
    {
        if (walk_anchor(r0, c0, grad_thresh)) extract_anchor(wl_chains);
    }


    // the chain tree -> segments
    __device__ void extract_anchor(int noChains)
    {
        StagChain *ch = R.chains;
        blk0 = totalPixels;
        segbase = totalPixels;
        nsp = 0;
        const int back = sr_uni(ch[0].child[1]);
#ifdef RW_TIMING
        const unsigned long long ex_a = __builtin_readcyclecounter();
#endif
        int totalLen = longest(back);
#ifdef RW_TIMING
        const unsigned long long ex_b = __builtin_readcyclecounter();
        ex_t[0] += ex_b - ex_a;
#endif
        if (totalLen > 0) {  // the path behind the anchor, copied backwards so that the segment runs through the anchor
            const int count = retrieve(back);
            for (int k = count - 1; k >= 0; k--) {
                const int cn = sr_uni(R.chainNos[k]);
                int L = sr_uni(ch[cn].len);
                trim_tail(cpx(cn, L - 1));
                if (L > 1 && sr_near(cpx(cn, L - 2), seg(nsp - 1))) L--;
                seg_copy(cn, L - 1, -1, L);
                ch[cn].len = 0;
            }
        }
        const int fwd = sr_uni(ch[0].child[0]);
#ifdef RW_TIMING
        const unsigned long long ex_c = __builtin_readcyclecounter();
        ex_t[1] += ex_c - ex_b;
#endif
        totalLen = longest(fwd);
#ifdef RW_TIMING
        const unsigned long long ex_d = __builtin_readcyclecounter();
        ex_t[0] += ex_d - ex_c;
#endif
        if (totalLen > 1) {
            const int count = retrieve(fwd);
            const int first = sr_uni(R.chainNos[0]);  // its first pixel is the anchor again
            ch[first].pix++;
            ch[first].len--;
            append_forward(count);
        }
        close_segment(true);
#ifdef RW_TIMING
        const unsigned long long ex_e = __builtin_readcyclecounter();
        ex_t[1] += ex_e - ex_d;
#endif
        for (int k = 2; k < noChains; k++) {  // what is left of the tree
            if (sr_uni(ch[k].len) < 2) continue;
            totalLen = longest(k);
            if (totalLen >= 10) {
                segbase = totalPixels;
                nsp = 0;
                append_forward(retrieve(k));
                close_segment(false);
            }
        }
#ifdef RW_TIMING
        ex_t[2] += __builtin_readcyclecounter() - ex_e;
#endif
    }
};

__device__ __forceinline__ void k_stag_route_seq_impl(StagRoute R, const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors,
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
__global__ __launch_bounds__(64) void k_stag_route_seq(StagRoute R, const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors, int grad_thresh)
{
    k_stag_route_seq_impl(R, sorted, n_anchors, grad_thresh);
}
struct k_stag_route_seq_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(StagRoute R, const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors, int grad_thresh) const { k_stag_route_seq_impl(R, sorted, n_anchors, grad_thresh); }
};

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
    int minr, minc, maxr, maxc;   // bounding box of the component's pixels
};

struct StagRec {  // one producing anchor
    int rank;
    int pix_off, len;       // its chain-tree pixels inside the component's scratch arena
    int chain_off, nchains;
    int out_off, out_len;   // its block inside the component's output arena
    int seg_off, nsegs;     // its segments inside the component's segment arena
};

// several buffers filled by one launch (every fill is a dispatch of its own otherwise, and the hardware takes them one at a time)
struct StagFills {
    uint4 *p[6];
    unsigned n16[3];  // 16-byte words of the first three (sized by the context: the same for every frame of a group)
    unsigned na16;    // ... and of the last three, sized by the frame's anchor count: ONE scalar that differs between the frames of a
                      // group, which is what a merged launch can carry per frame (fid_stag_batch.h)
    unsigned v[6];    // the 32-bit pattern
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&vis)
    {
        vis(p); vis(n16); vis(na16); vis(v);
    }
};
__device__ __forceinline__ void k_stag_fills_impl(StagFills F)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (i < (k < 3 ? F.n16[k] : F.na16)) F.p[k][i] = make_uint4(F.v[k], F.v[k], F.v[k], F.v[k]);
}
__global__ __launch_bounds__(256) void k_stag_fills(StagFills F)
{
    k_stag_fills_impl(F);
}
struct k_stag_fills_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(StagFills F) const { k_stag_fills_impl(F); }
};

// A frame QUEUED AHEAD of its own counts (fid_stag.hip, stag_advance_impl): the host sizes every launch by what the context's last
// frame needed (with a margin) and does not wait for the counts the device produces on the way.  This kernel stands where the host
// used to read such a count: a count above what the following launches were sized for (or an overflow flag, capacity 0) raises the
// frame's flag and ZEROES the counts the rest of the frame goes by, so that nothing downstream ever works on a table that was
// cleared, filled or covered by a grid only in part; the host sees the flag at the frame's one wait and runs the frame again on
// the counted road.  The true counts have gone to the pinned block in front of this kernel.
struct StagGuard {
    int *cnt[4];
    int cap[4];
    int *kill[4];
    int ncnt, nkill, reset;
    int *bad;
    // the true counts go to the context's pinned block (through its device alias) from here: one stream operation per point
    // instead of a copy or two and this kernel
    int *mdst[3];
    const int *msrc[3];
    int mn[3];
    int *bad_host;  // the flag's place in the pinned block
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(cnt); v(cap); v(kill); v(ncnt); v(nkill); v(reset); v(bad); v(mdst); v(msrc); v(mn); v(bad_host);
    }
};
__device__ __forceinline__ void k_stag_spec_guard_impl(StagGuard g)
{
    if (blockIdx.x != 0) return;
    for (int k = 0; k < 3; k++)
        if (g.mdst[k] && (int)threadIdx.x < g.mn[k]) g.mdst[k][threadIdx.x] = g.msrc[k][threadIdx.x];
    __syncthreads();  // (the counts are read above before thread 0 may zero them below)
    if (threadIdx.x != 0) return;
    bool b = !g.reset && *g.bad != 0;
    for (int i = 0; i < g.ncnt; i++) b = b || (unsigned)*g.cnt[i] > (unsigned)g.cap[i];
    if (g.reset || b) *g.bad = b ? 1 : 0;
    if (g.bad_host) *g.bad_host = b ? 1 : 0;
    if (b)
        for (int i = 0; i < g.nkill; i++) *g.kill[i] = 0;
}
__global__ __launch_bounds__(64) void k_stag_spec_guard(StagGuard g)
{
    k_stag_spec_guard_impl(g);
}
struct k_stag_spec_guard_fn {
    static constexpr int kBounds = 64;
    __device__ __forceinline__ void operator()(StagGuard g) const { k_stag_spec_guard_impl(g); }
};

__device__ __forceinline__ int ccl_find(const int *L, int a)
{
    while (true) {
        const int p = L[a];
        if (p == a) return a;
        a = p;
    }
}

// Connected components of {grad >= thresh} (8-connectivity) in two steps: every 64 x 16 tile labels itself in LDS (union-find,
// hooks by atomicMin), writes the global index of each pixel's tile-local root, and the pixels on the tile borders then
// merge the tiles through the same union-find in global memory.  Which neighbours have to be united: the one above if it is
// foreground (it holds its own left and right neighbours), else the two diagonal ones above; and the left one.  (A single
// pass over all pixels with all four preceding neighbours in global memory took 176 us of the whole GPU per frame.)
#define CCL_TW 64
#define CCL_TH 16
__device__ __forceinline__ void ccl_union(int *L, int a, int b)
{
    while (true) {
        a = ccl_find(L, a);
        b = ccl_find(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);  // hook the larger root under the smaller one
        if (old == a) return;
        a = old;
    }
}

__device__ __forceinline__ void k_stag_ccl_tile_impl(const int16_t *__restrict__ grad, int W, int H, int thresh, int *__restrict__ label,
                                                       int *__restrict__ csize, int *__restrict__ canch, int4 *__restrict__ cbox, uint8_t *__restrict__ tilefg)
{
    __shared__ int L[CCL_TW * CCL_TH];
    const int x0 = blockIdx.x * CCL_TW, y0 = blockIdx.y * CCL_TH;
    int any = 0;
    for (int k = threadIdx.x; k < CCL_TW * CCL_TH; k += 256) {
        const int x = x0 + (k % CCL_TW), y = y0 + (k / CCL_TW);
        const bool fg = x < W && y < H && grad[y * W + x] >= thresh;
        L[k] = fg ? k : -1;
        any |= fg;
    }
    // (round 6) a tile without a foreground pixel -- 60 % of the tiles of the bench frames -- has nothing to merge: its labels are
    // -1, and k_stag_ccl_flatten does not come back to it (tilefg)
    if (!__syncthreads_or(any)) {
        if (threadIdx.x == 0) tilefg[blockIdx.y * ((W + CCL_TW - 1) / CCL_TW) + blockIdx.x] = 0;
        for (int k = threadIdx.x; k < CCL_TW * CCL_TH; k += 256) {
            const int x = x0 + (k % CCL_TW), y = y0 + (k / CCL_TW);
            if (x < W && y < H) label[y * W + x] = -1;
        }
        return;
    }
    if (threadIdx.x == 0) tilefg[blockIdx.y * ((W + CCL_TW - 1) / CCL_TW) + blockIdx.x] = 1;
    for (int k = threadIdx.x; k < CCL_TW * CCL_TH; k += 256) {
        if (L[k] < 0) continue;
        const int lx = k % CCL_TW, ly = k / CCL_TW;
        if (ly > 0) {
            if (L[k - CCL_TW] >= 0) {
                ccl_union(L, k, k - CCL_TW);
            } else {
                if (lx > 0 && L[k - CCL_TW - 1] >= 0) ccl_union(L, k, k - CCL_TW - 1);
                if (lx < CCL_TW - 1 && L[k - CCL_TW + 1] >= 0) ccl_union(L, k, k - CCL_TW + 1);
            }
        }
        if (lx > 0 && L[k - 1] >= 0) ccl_union(L, k, k - 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < CCL_TW * CCL_TH; k += 256) {
        const int x = x0 + (k % CCL_TW), y = y0 + (k / CCL_TW);
        if (x >= W || y >= H) continue;
        int v = -1;
        if (L[k] >= 0) {
            const int r = ccl_find(L, k);
            v = (y0 + r / CCL_TW) * W + x0 + (r % CCL_TW);
            if (r == k) {
                // a component's final root is the smallest index in it, hence the root of its piece in ITS tile: the per-root counters
                // only have to be clean at the tiles' local roots (until round 6: at every foreground pixel, 24 bytes each)
                csize[y * W + x] = 0;
                canch[y * W + x] = 0;
                cbox[y * W + x] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);  // min row, min column, max row, max column
            }
        }
        label[y * W + x] = v;
    }
}
__global__ __launch_bounds__(256) void k_stag_ccl_tile(const int16_t *__restrict__ grad, int W, int H, int thresh, int *__restrict__ label, int *__restrict__ csize, int *__restrict__ canch, int4 *__restrict__ cbox, uint8_t *__restrict__ tilefg)
{
    k_stag_ccl_tile_impl(grad, W, H, thresh, label, csize, canch, cbox, tilefg);
}
struct k_stag_ccl_tile_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int16_t *__restrict__ grad, int W, int H, int thresh, int *__restrict__ label, int *__restrict__ csize, int *__restrict__ canch, int4 *__restrict__ cbox, uint8_t *__restrict__ tilefg) const { k_stag_ccl_tile_impl(grad, W, H, thresh, label, csize, canch, cbox, tilefg); }
};

// the pixels of a tile's first row, first column and last column against their neighbours in the adjacent tiles
__device__ __forceinline__ void k_stag_ccl_border_impl(int W, int H, int *label)
{
    const int t = threadIdx.x;
    int lx, ly;
    if (t < CCL_TW) { lx = t; ly = 0; }
    else if (t < CCL_TW + CCL_TH) { lx = 0; ly = t - CCL_TW; }
    else if (t < CCL_TW + 2 * CCL_TH) { lx = CCL_TW - 1; ly = t - CCL_TW - CCL_TH; }
    else return;
    if (t >= CCL_TW && ly == 0) return;  // (the corners belong to the row)
    const int c = blockIdx.x * CCL_TW + lx, r = blockIdx.y * CCL_TH + ly;
    if (c >= W || r >= H) return;
    const int i = r * W + c;
    if (label[i] < 0) return;
    const bool up = r > 0, left = c > 0, right = c < W - 1;
    if (up && label[i - W] >= 0) {
        if (ly == 0) ccl_union(label, i, i - W);
    } else if (up) {
        if (left && (lx == 0 || ly == 0) && label[i - W - 1] >= 0) ccl_union(label, i, i - W - 1);
        if (right && (lx == CCL_TW - 1 || ly == 0) && label[i - W + 1] >= 0) ccl_union(label, i, i - W + 1);
    }
    if (left && lx == 0 && label[i - 1] >= 0) ccl_union(label, i, i - 1);
}
__global__ __launch_bounds__(128) void k_stag_ccl_border(int W, int H, int *label)
{
    k_stag_ccl_border_impl(W, H, label);
}
struct k_stag_ccl_border_fn {
    static constexpr int kBounds = 128;
    __device__ __forceinline__ void operator()(int W, int H, int *label) const { k_stag_ccl_border_impl(W, H, label); }
};

// Every foreground pixel to its final root, and per root: pixels, anchors, bounding box.  A workgroup = one 64 x 16 tile (the tiles
// of k_stag_ccl_tile), a wave = one tile row per round.  The per-root sums are collected in a small LDS table first -- one slot per
// root that occurs in the tile -- and leave as ONE set of global atomics per (tile, root).  (Until round 6 a wave of 64 consecutive
// pixels sent its own set per root: a marker's outline of 3 000 pixels meant ~2 000 sets of six atomics on the same six words, and
// that traffic jam was 10 of this kernel's 16 us per frame; timed with the statistics compiled out, -DFLATTEN_NO_STATS.)
__device__ __forceinline__ void k_stag_ccl_flatten_impl(int W, int H, int *label, const uint8_t *__restrict__ anchors, int *__restrict__ csize,
                                                          int *__restrict__ canch, int4 *__restrict__ cbox, int *__restrict__ roots, int *__restrict__ cursors, const uint8_t *__restrict__ tilefg)
{
    constexpr int SLOTS = 32;
    __shared__ int s_root[SLOTS], s_cnt[SLOTS], s_anch[SLOTS], s_minr[SLOTS], s_minc[SLOTS], s_maxr[SLOTS], s_maxc[SLOTS];
    if (!tilefg[blockIdx.y * ((W + CCL_TW - 1) / CCL_TW) + blockIdx.x]) return;  // (no foreground pixel in this tile: k_stag_ccl_tile said so)
    const int tid = threadIdx.x, lane = tid & 63;
    const int x0 = blockIdx.x * CCL_TW, y0 = blockIdx.y * CCL_TH;
    if (tid < SLOTS) {
        s_root[tid] = -1;
        s_cnt[tid] = s_anch[tid] = 0;
        s_minr[tid] = s_minc[tid] = 0x7fffffff;
        s_maxr[tid] = s_maxc[tid] = -1;
    }
    __syncthreads();
    const int c = x0 + lane;
#pragma unroll
    for (int j = 0; j < CCL_TH / 4; j++) {
        const int r = y0 + (tid >> 6) + 4 * j;  // (wave-uniform: a wave takes one tile row per round)
        const int i = r * W + c;
        int root = -1;
        bool anch = false;
        if (c < W && r < H && label[i] >= 0) {
            root = ccl_find(label, i);
            label[i] = root;
            anch = anchors[i] == STAG_ANCHOR_PIXEL;
            // the frame's ROOTS as a list (cursors[13] counts them): k_stag_comp_alloc goes by it instead of asking every pixel of the
            // image whether it is one (a few hundred roots among two million pixels).  8-connected components are at most
            // ceil(W / 2) * ceil(H / 2), which is what `roots` holds.
            if (root == i) roots[atomicAdd(&cursors[13], 1)] = i;
        }
        unsigned long long pending = __ballot(root >= 0);
#ifdef FLATTEN_NO_STATS
        pending = 0;
#endif
        while (pending) {
            const int lead = __builtin_ctzll(pending);
            const int r0 = __builtin_amdgcn_readlane(root, lead);
            const bool mine = root == r0;
            const unsigned long long m = __ballot(mine), ma = __ballot(mine && anch);
            if (lane == lead) {
                const int cnt = (int)__builtin_popcountll(m), na = (int)__builtin_popcountll(ma);
                const int mnc = x0 + (int)__builtin_ctzll(m), mxc = x0 + 63 - (int)__builtin_clzll(m);
                // the root's slot: open addressing on the root's index, SLOTS tries (a tile that holds more roots than slots -- noise --
                // sends the surplus straight to global memory)
                int slot = -1;
                unsigned h = ((unsigned)r0 * 2654435761u) >> 27;
                for (int t = 0; t < SLOTS; t++, h = (h + 1) & (SLOTS - 1)) {
                    const int old = atomicCAS(&s_root[h], -1, r0);
                    if (old == -1 || old == r0) {
                        slot = (int)h;
                        break;
                    }
                }
                if (slot >= 0) {
                    atomicAdd(&s_cnt[slot], cnt);
                    if (na) atomicAdd(&s_anch[slot], na);
                    atomicMin(&s_minr[slot], r);
                    atomicMin(&s_minc[slot], mnc);
                    atomicMax(&s_maxr[slot], r);
                    atomicMax(&s_maxc[slot], mxc);
                } else {
                    atomicAdd(&csize[r0], cnt);
                    if (na) atomicAdd(&canch[r0], na);
                    int *bx = reinterpret_cast<int *>(cbox + r0);
                    atomicMin(bx + 0, r);
                    atomicMin(bx + 1, mnc);
                    atomicMax(bx + 2, r);
                    atomicMax(bx + 3, mxc);
                }
            }
            pending &= ~m;
        }
    }
    __syncthreads();
    if (tid < SLOTS && s_root[tid] >= 0) {
        const int r0 = s_root[tid];
        atomicAdd(&csize[r0], s_cnt[tid]);
        if (s_anch[tid]) atomicAdd(&canch[r0], s_anch[tid]);
        int *bx = reinterpret_cast<int *>(cbox + r0);
        atomicMin(bx + 0, s_minr[tid]);
        atomicMin(bx + 1, s_minc[tid]);
        atomicMax(bx + 2, s_maxr[tid]);
        atomicMax(bx + 3, s_maxc[tid]);
    }
}
__global__ __launch_bounds__(256) void k_stag_ccl_flatten(int W, int H, int *label, const uint8_t *__restrict__ anchors, int *__restrict__ csize, int *__restrict__ canch, int4 *__restrict__ cbox, int *__restrict__ roots, int *__restrict__ cursors, const uint8_t *__restrict__ tilefg)
{
    k_stag_ccl_flatten_impl(W, H, label, anchors, csize, canch, cbox, roots, cursors, tilefg);
}
struct k_stag_ccl_flatten_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(int W, int H, int *label, const uint8_t *__restrict__ anchors, int *__restrict__ csize, int *__restrict__ canch, int4 *__restrict__ cbox, int *__restrict__ roots, int *__restrict__ cursors, const uint8_t *__restrict__ tilefg) const { k_stag_ccl_flatten_impl(W, H, label, anchors, csize, canch, cbox, roots, cursors, tilefg); }
};

// cursors: [0] components [1] anchor slots [2] scratch pixels [3] stack [4] chains [5] output pixels [6] segments [7] overflow
//          [8] overflow flags of the routing kernels [9] most anchors in one component [10] largest tile [11] big / [12] small
//          components in the walk's order list [13] roots (k_stag_ccl_flatten's list)
__device__ __forceinline__ void k_stag_comp_alloc_impl(const int *__restrict__ roots, const int *__restrict__ csize,
                                                         const int *__restrict__ canch, const int4 *__restrict__ cbox, int *__restrict__ cursors, int max_comps,
                                                         const int *caps, StagComp *__restrict__ comps, int *__restrict__ cidmap)
{
    const int nroots = cursors[13];
    const int lane = threadIdx.x & 63;
    // (round 6) a WAVE asks for the places of its 64 roots at once: one atomic per cursor and wave, the lanes' shares by prefix sums.
    // A thread per root sent eight returning atomics of its own to the same seven words -- a frame's ~8 k roots queued up there:
    // 19 us.  Which component gets which place was never defined (the atomics' order); every component still gets its own.
    for (int k0 = blockIdx.x * 256 + (int)(threadIdx.x & ~63u); k0 < nroots; k0 += gridDim.x * 256) {
        const int k = k0 + lane;
        const bool in = k < nroots;
        const int i = in ? roots[k] : 0;  // (one thread per ROOT; until round 6: one per pixel, asking label[i] == i)
        if (in) cidmap[i] = -1;
        const int na = in ? canch[i] : 0, sz = in ? csize[i] : 0;
        const bool has = na > 0;
        const unsigned long long hm = __ballot(has);
        if (!hm) continue;
        StagComp C;
        C.root = i; C.size = sz; C.nanch = na; C.nrec = 0;
        if (has) {
            const int4 bx = cbox[i];
            C.minr = bx.x; C.minc = bx.y; C.maxr = bx.z; C.maxc = bx.w;
        }
        int p2 = 1;
        while (p2 < na) p2 <<= 1;
        C.anch_cap = has ? p2 : 0;
        C.pix_cap = has ? 2 * sz + 12 * na + 64 : 0;
        C.stack_cap = has ? (sz + 2 * na + 64) + (sz / 4 + 64) : 0;  // pending branches + (in its tail) the chain lists of the extraction
        C.chain_cap = has ? sz + 2 * na + 64 : 0;
        C.out_cap = C.pix_cap;
        C.seg_cap = has ? C.pix_cap / 8 + na + 8 : 0;
        const int wmax = wave_max_i32(na);
        const int s1 = wave_iscan(C.anch_cap), s2 = wave_iscan(C.pix_cap), s3 = wave_iscan(C.stack_cap), s4 = wave_iscan(C.chain_cap),
                  s6 = wave_iscan(C.seg_cap);
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0;
        if (lane == 63) {  // (the lane that holds the totals)
            atomicMax(&cursors[9], wmax);
            b0 = atomicAdd(&cursors[0], (int)__builtin_popcountll(hm));
            b1 = atomicAdd(&cursors[1], s1);
            b2 = atomicAdd(&cursors[2], s2);
            b3 = atomicAdd(&cursors[3], s3);
            b4 = atomicAdd(&cursors[4], s4);
            b5 = atomicAdd(&cursors[5], s2);  // (out_cap = pix_cap)
            b6 = atomicAdd(&cursors[6], s6);
        }
        b0 = __builtin_amdgcn_readlane(b0, 63); b1 = __builtin_amdgcn_readlane(b1, 63); b2 = __builtin_amdgcn_readlane(b2, 63);
        b3 = __builtin_amdgcn_readlane(b3, 63); b4 = __builtin_amdgcn_readlane(b4, 63); b5 = __builtin_amdgcn_readlane(b5, 63);
        b6 = __builtin_amdgcn_readlane(b6, 63);
        if (!has) continue;
        const int cid = b0 + (int)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
        if (cid >= max_comps) {
            atomicOr(&cursors[7], 1);
            continue;
        }
        C.anch_base = b1 + s1 - C.anch_cap;
        C.pix_base = b2 + s2 - C.pix_cap;
        C.stack_base = b3 + s3 - C.stack_cap;
        C.chain_base = b4 + s4 - C.chain_cap;
        C.out_base = b5 + s2 - C.out_cap;
        C.seg_base = b6 + s6 - C.seg_cap;
        if (C.anch_base + C.anch_cap > caps[1] || C.pix_base + C.pix_cap > caps[2] || C.stack_base + C.stack_cap > caps[3] ||
            C.chain_base + C.chain_cap > caps[4] || C.out_base + C.out_cap > caps[5] || C.seg_base + C.seg_cap > caps[6]) {
            atomicOr(&cursors[7], 2);
            C.nanch = 0;  // not processed; the call reports FID_E_CAPACITY
        }
        comps[cid] = C;
        cidmap[i] = cid;
    }
}
__global__ __launch_bounds__(256) void k_stag_comp_alloc(const int *__restrict__ roots, const int *__restrict__ csize, const int *__restrict__ canch, const int4 *__restrict__ cbox, int *__restrict__ cursors, int max_comps, const int *caps, StagComp *__restrict__ comps, int *__restrict__ cidmap)
{
    k_stag_comp_alloc_impl(roots, csize, canch, cbox, cursors, max_comps, caps, comps, cidmap);
}
struct k_stag_comp_alloc_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int *__restrict__ roots, const int *__restrict__ csize, const int *__restrict__ canch, const int4 *__restrict__ cbox, int *__restrict__ cursors, int max_comps, const int *caps, StagComp *__restrict__ comps, int *__restrict__ cidmap) const { k_stag_comp_alloc_impl(roots, csize, canch, cbox, cursors, max_comps, caps, comps, cidmap); }
};

// (64 consecutive ranks hold many anchors of the frame's big components: one atomic per (wave, component), the anchors of
//  a group placed in rank order)
__device__ __forceinline__ void k_stag_comp_fill_impl(const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors,
                                                        const int *__restrict__ label, const int *__restrict__ cidmap, const StagComp *__restrict__ comps,
                                                        int *__restrict__ fill, int *__restrict__ aslots)
{
    const int r = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    int cid = -1;
    if (r < (int)*n_anchors) {
        cid = cidmap[label[sorted[r]]];
        if (cid >= 0 && comps[cid].nanch == 0) cid = -1;
    }
    // (round 6) first every lane learns its group -- the lane that leads it, its place in it, its size: ballots only -- then the
    // leaders ask for their groups' places with ONE wave instruction.  Before, the atomic sat inside the loop and the wave waited
    // for it once per component: 64 consecutive ranks of the gradient order belong to 30 - 60 components, 35 us for a single frame.
    unsigned long long pending = __ballot(cid >= 0);
    int lead = 0, rank = 0, cnt = 0;
    while (pending) {
        const int l0 = __builtin_ctzll(pending);
        const int c0 = __builtin_amdgcn_readlane(cid, l0);
        const unsigned long long m = __ballot(cid == c0);
        if (cid == c0) {
            lead = l0;
            rank = (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            cnt = (int)__builtin_popcountll(m);
        }
        pending &= ~m;
    }
    int base = 0;
    if (cid >= 0 && lane == lead) base = atomicAdd(&fill[cid], cnt);
    base = __shfl(base, lead, 64);
    if (cid >= 0) aslots[comps[cid].anch_base + base + rank] = r;
}
__global__ __launch_bounds__(256) void k_stag_comp_fill(const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors, const int *__restrict__ label, const int *__restrict__ cidmap, const StagComp *__restrict__ comps, int *__restrict__ fill, int *__restrict__ aslots)
{
    k_stag_comp_fill_impl(sorted, n_anchors, label, cidmap, comps, fill, aslots);
}
struct k_stag_comp_fill_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const int32_t *__restrict__ sorted, const unsigned *__restrict__ n_anchors, const int *__restrict__ label, const int *__restrict__ cidmap, const StagComp *__restrict__ comps, int *__restrict__ fill, int *__restrict__ aslots) const { k_stag_comp_fill_impl(sorted, n_anchors, label, cidmap, comps, fill, aslots); }
};

// cursors[10] = the largest LDS tile (bytes) a component would need (the boxes come from k_stag_ccl_flatten)
// ... and the order the walk and the extraction take the components in: the big ones (a marker's outline and inside: hundreds of
// pixels, a walk of a millisecond) from the FRONT of `order`, the small ones from its back, so that a launch starts every long walk at
// once and fills the gaps with the short ones (cursors[11] = big components, cursors[12] = small ones; which of two equal
// components comes first is up to the atomics -- the components are independent, every one writes its own arenas and records)
#define STAG_BIG_COMP 192     // pixels
#define STAG_SMALL_TILE 8192  // bytes: what a "small" component's box needs at most as a dense LDS tile
__device__ __forceinline__ int stag_comp_by_rank(const int *__restrict__ order, const int *__restrict__ cursors, int rank)
{
    const int nbig = cursors[11], ncomp = cursors[0];
    return rank < nbig ? order[rank] : order[ncomp - 1 - (rank - nbig)];  // (small ones: slot s of the back = order[ncomp - 1 - s])
}
__device__ __forceinline__ void k_stag_comp_tilemax_impl(const StagComp *__restrict__ comps, int *cursors, int lds_cap, int *__restrict__ order, int max_comps)
{
    const int cid = blockIdx.x * 256 + threadIdx.x;
    // (k_stag_comp_alloc counts past max_comps when the table is full -- the overflow flag is up and the frame takes the sequential
    //  road -- but `order` and `comps` end at max_comps)
    const int ncomp = cursors[0] < max_comps ? cursors[0] : max_comps;
    if (cid >= ncomp) return;
    const StagComp C = comps[cid];
    // (every component gets a place, the empty ones among the small: nbig + nsmall = ncomp, front and back never meet)
    const int bytes = (C.maxr - C.minr + 3) * (C.maxc - C.minc + 3) * 2;
    if (C.nanch > 0 && (C.size >= STAG_BIG_COMP || bytes > STAG_SMALL_TILE)) order[atomicAdd(&cursors[11], 1)] = cid;
    else order[ncomp - 1 - atomicAdd(&cursors[12], 1)] = cid;
    if (C.nanch == 0) return;
    atomicMax(&cursors[10], bytes <= lds_cap ? bytes : lds_cap);  // (beyond the cap: the whole of it, for the component's blocks)
}
__global__ __launch_bounds__(256) void k_stag_comp_tilemax(const StagComp *__restrict__ comps, int *cursors, int lds_cap, int *__restrict__ order, int max_comps)
{
    k_stag_comp_tilemax_impl(comps, cursors, lds_cap, order, max_comps);
}
struct k_stag_comp_tilemax_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const StagComp *__restrict__ comps, int *cursors, int lds_cap, int *__restrict__ order, int max_comps) const { k_stag_comp_tilemax_impl(comps, cursors, lds_cap, order, max_comps); }
};

// ranks of one component, descending: bitonic sort of the (padded, -1 filled) slice, one wave per component; slices of up to
// 2048 entries are sorted in LDS
#define STAG_SORT_LDS 2048
#define STAG_SORT_WAVE 256   // slices up to here: a wave each (k_stag_comp_sort); up to STAG_SORT_BIG: a workgroup of 1024 each
#define STAG_SORT_BIG 16384
__device__ __forceinline__ void k_stag_comp_sort_impl(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots)
{
    __shared__ int s_buf[4][STAG_SORT_LDS];
    const int wv = threadIdx.x >> 6, cid = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (cid >= cursors[0]) return;
    const StagComp C = comps[cid];
    if (C.nanch < 2) return;
    int *g = aslots + C.anch_base;
    const int P = C.anch_cap;
    if (P > STAG_SORT_WAVE && P <= STAG_SORT_BIG) return;  // k_stag_comp_sort_big's
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
__global__ __launch_bounds__(256) void k_stag_comp_sort(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots)
{
    k_stag_comp_sort_impl(comps, cursors, aslots);
}
struct k_stag_comp_sort_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots) const { k_stag_comp_sort_impl(comps, cursors, aslots); }
};

// the same for the slices of 257 .. 16384 entries (a frame's larger components): one workgroup of 1024 per component, the
// slice in up to 64 KB of LDS (one wave took 118 us for a slice of 2048: 66 passes of 32 rounds each)
__device__ __forceinline__ void k_stag_comp_sort_big_impl(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots)
{
    extern __shared__ int s_big[];
    // (a workgroup goes through the components blockIdx.x, + gridDim.x, ...: a frame has a few dozen slices of this size among
    //  its components, and a 1 024-thread workgroup with 64 KB of LDS per COMPONENT -- most of them returning at once -- waited for
    //  16 free wave slots on one CU each: 112 us alone, 680 us beside the other groups' kernels)
    const int ncomp = cursors[0];
    for (int cid = blockIdx.x; cid < ncomp; cid += gridDim.x) {
    const StagComp C = comps[cid];
    const int P = C.anch_cap;
    if (C.nanch < 2 || P <= STAG_SORT_WAVE || P > STAG_SORT_BIG) continue;
    __syncthreads();  // (the slice of the component before this one has left the LDS)
    int *g = aslots + C.anch_base;
    for (int i = threadIdx.x; i < P; i += (int)blockDim.x) s_big[i] = g[i];
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += (int)blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const int x = s_big[i], y = s_big[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? x < y : x > y) {
                        s_big[i] = y;
                        s_big[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < P; i += (int)blockDim.x) g[i] = s_big[i];
    }
}
__global__ __launch_bounds__(1024) void k_stag_comp_sort_big(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots)
{
    k_stag_comp_sort_big_impl(comps, cursors, aslots);
}
struct k_stag_comp_sort_big_fn {
    static constexpr int kBounds = 1024;
    __device__ __forceinline__ void operator()(const StagComp *__restrict__ comps, const int *__restrict__ cursors, int *aslots) const { k_stag_comp_sort_big_impl(comps, cursors, aslots); }
};

struct StagArenas {
    int2 *pix;
    int4 *stack;
    StagChain *chains;
    int2 *out;
    int2 *segs;
    StagRec *recs;  // indexed like the anchor slots
    template <class V>
    __host__ __device__ __forceinline__ void visit(V &&v)
    {
        v(pix); v(stack); v(chains); v(out); v(segs); v(recs);
    }
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
    S.gout = S.R.outpix;
    S.R.capOut = C.out_cap;
    S.R.segs = A.segs + C.seg_base;
    S.R.capSegs = C.seg_cap;
    S.par = true;
}

template <bool FM = false>
__device__ __forceinline__ void k_stag_route_walk_impl(StagRoute G, StagArenas A, StagComp *__restrict__ comps, const int *__restrict__ cursors,
                                                        const int *__restrict__ order, const int32_t *__restrict__ sorted, const int *__restrict__ aslots,
                                                        const int *__restrict__ label,
                                                        int grad_thresh, int lds_bytes, int *__restrict__ prodflag, int *__restrict__ ovf, int cls)
{
    extern __shared__ uint16_t s_tile[];
    constexpr int WSTACK = 128;  // (deeper than that: the arena in global memory)
    __shared__ int4 s_wstack[WSTACK];  // the first entries of the walk's stack (StagRouter::MemWave)
    // one workgroup per component: four waves move the tile in and out, wave 0 walks
    // (longest first: k_stag_comp_tilemax's order list -- big components from the front, small ones from the back of `order`)
    // cls 0: every component; 1: the big ones only (ranks below cursors[11]); 2: the small ones only -- a group of frames launches the
    // two classes one after the other: the big ones with the large tile, all of them resident at once, the small ones with
    // STAG_SMALL_TILE of LDS, so that the hundreds of short walks are not queueing for the one slot per CU the long walks leave free
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int rank = (int)STAG_BX<FM>() + (cls == 2 ? cursors[11] : 0);
    if (rank >= (cls == 1 ? cursors[11] : cursors[0])) return;
    const int cid = stag_comp_by_rank(order, cursors, rank);
    const StagComp C = sr_uni_struct(comps[cid]);
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
    const int W = G.W;
    // the component's bounding box (+1 all around) as an LDS tile, if it fits
    StagRouter::Tile T;
    T.t = s_tile;
    T.r0 = C.minr - 1;
    T.c0 = C.minc - 1;
    T.tw = C.maxc - C.minc + 3;
    T.gstk = S.R.stack; T.lstk.p = (SrLdsInt)(int *)s_wstack; T.lcap = WSTACK;
    const int th = C.maxr - C.minr + 3;
    const int lds_real = lds_bytes & ~3;  // (bit 0: an experiment switch)
    const bool tiled = T.tw * th * 2 <= lds_real;
#ifdef RW_TIMING
    const unsigned long long rw_tl = __builtin_readcyclecounter();
#endif
    // ... or as 4 x 4 blocks (StagRouter::SparseTile), if those fit
    __shared__ int s_nblk;
    StagRouter::SparseTile P;
    bool sparse = false;
    if (!tiled && !(lds_bytes & 1)) {  // (bit 0 of lds_bytes: no blocks -- FID_STAG_SPARSE=0, an experiment switch)
        constexpr int SH = StagRouter::SparseTile::SH, BM = StagRouter::SparseTile::BM, BW = 1 << (2 * SH);  // (words per block)
        const int bw = (T.tw + BM) >> SH, bh = (th + BM) >> SH, ntab = (bw * bh + 63) & ~63;
        const int maxblk = (lds_real / 2 - ntab) / BW - 1;  // (block 0 is the empty one)
        P.tab = s_tile; P.blk = s_tile + ntab; P.r0 = T.r0; P.c0 = T.c0; P.bw = bw;
        P.gstk = S.R.stack; P.lstk.p = (SrLdsInt)(int *)s_wstack; P.lcap = WSTACK;
        if (maxblk >= 16 && maxblk < 65535) {  // (uniform over the workgroup: the barriers below are safe)
            for (int i = tid; i < ntab; i += 256) s_tile[i] = 0;
            if (tid < BW) P.blk[tid] = 0;
            if (tid == 0) s_nblk = 0;
            __syncthreads();
            // the blocks within one pixel of a pixel of the component (a wave per row, its lanes along the row; the box's own
            // border ring holds no pixel of the component)
            for (int r4 = 1 + (tid >> 6) * 16; r4 < th - 1; r4 += 64)
                for (int cc = 1 + lane; cc < T.tw - 1; cc += 64) {
                    int lb[16];
#pragma unroll
                    for (int u = 0; u < 16; u++)  // (sixteen rows in flight per thread: the pass is a chain of memory round trips)
                        if (r4 + u < th - 1) lb[u] = label[(T.r0 + r4 + u) * W + T.c0 + cc];
#pragma unroll
                    for (int u = 0; u < 16; u++)
                        if (r4 + u < th - 1 && lb[u] == C.root) {
                            const int rr = r4 + u;
                            const int b0 = ((rr - 1) >> SH) * bw, b1 = ((rr + 1) >> SH) * bw, q0 = (cc - 1) >> SH, q1 = (cc + 1) >> SH;
                            s_tile[b0 + q0] = 0xffff; s_tile[b0 + q1] = 0xffff; s_tile[b1 + q0] = 0xffff; s_tile[b1 + q1] = 0xffff;
                        }
                }
            __syncthreads();
            for (int i = tid; i < bw * bh; i += 256)
                if (s_tile[i] == 0xffff) {
                    const int slot = atomicAdd(&s_nblk, 1) + 1;
                    s_tile[i] = (uint16_t)(slot <= maxblk ? slot : 0);
                }
            __syncthreads();
            sparse = s_nblk <= maxblk;
            if (sparse) {
                for (int r4 = (tid >> 6) * 8; r4 < th; r4 += 32)
                    for (int cc = lane; cc < T.tw; cc += 64) {
                        int slot[8], e[8], gr[8], dr[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            slot[u] = r4 + u < th ? s_tile[((r4 + u) >> SH) * bw + (cc >> SH)] : 0;
                            if (slot[u]) {
                                const int g = (T.r0 + r4 + u) * W + T.c0 + cc;
                                e[u] = G.edge[g]; gr[u] = G.grad[g]; dr[u] = G.dir[g];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (slot[u]) {
                                const int st = e[u] == STAG_EDGE_PIXEL ? 2 : e[u] == STAG_ANCHOR_PIXEL ? 1 : 0;
                                P.blk[(slot[u] << (2 * SH)) | (((r4 + u) & BM) << SH) | (cc & BM)] =
                                    (uint16_t)((gr[u] & 0x7ff) | (dr[u] == STAG_EDGE_VERTICAL ? 0x800 : 0) | (st << 12));
                            }
                    }
            }
            __syncthreads();
        }
    }
    if (tiled) {
        // a wave per row, its lanes along the row (no division per pixel), four rows in flight per thread
        for (int r4 = (tid >> 6) * 4; r4 < th; r4 += 16)
            for (int cc = lane; cc < T.tw; cc += 64) {
                int e[4], gr[4], dr[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (r4 + u < th) {
                        const int g = (T.r0 + r4 + u) * W + T.c0 + cc;
                        e[u] = G.edge[g]; gr[u] = G.grad[g]; dr[u] = G.dir[g];
                    }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (r4 + u < th) {
                        const int st = e[u] == STAG_EDGE_PIXEL ? 2 : e[u] == STAG_ANCHOR_PIXEL ? 1 : 0;
                        s_tile[(r4 + u) * T.tw + cc] = (uint16_t)((gr[u] & 0x7ff) | (dr[u] == STAG_EDGE_VERTICAL ? 0x800 : 0) | (st << 12));
                    }
            }
        __syncthreads();
    }
#ifdef RW_TIMING
    const unsigned long long rw_t0 = __builtin_readcyclecounter();
    const unsigned long long rw_load = rw_t0 - rw_tl;
    unsigned long long rw_walk = 0;
    int rw_walks = 0, rw_live = 0, rw_pix = 0, rw_chains = 0;
#endif
    // (which wave of the workgroup walks makes no difference: rotating it over the four, so that the walkers of the workgroups that
    //  share a CU in a group of frames sit on different SIMDs, left the group's kernel where it was -- measured, round 5)
    if (tid < 64) {
    for (int k0 = 0; k0 < C.nanch; k0 += 64) {
        // which of the next 64 anchors are still anchors?  (a walk can only turn anchors OFF, so a stale "on" is re-checked)
        const int kk = k0 + lane;
        int my_off = -1, my_rank = -1;
        if (kk < C.nanch) {
            my_rank = aslots[C.anch_base + kk];
            my_off = sorted[my_rank];
        }
        bool on = false;
        if (my_off >= 0) {
            const int orr = my_off / W, occ = my_off - orr * W;
            on = tiled    ? StagRouter::Tile::edge_of(s_tile[T.idx(orr, occ)]) == STAG_ANCHOR_PIXEL
                 : sparse ? (P.blk[P.off(orr - P.r0, occ - P.c0)] >> 12) == 1
                          : G.edge[my_off] == STAG_ANCHOR_PIXEL;
        }
        unsigned long long live = __ballot(on);
        while (live) {
            const int j = __builtin_ctzll(live);
            live &= live - 1;
            const int rank = __builtin_amdgcn_readlane(my_rank, j), off = __builtin_amdgcn_readlane(my_off, j);  // (no second trip to memory)
            const int ar = off / W, ac = off - ar * W;
            const bool still = tiled    ? StagRouter::Tile::edge_of((uint16_t)sr_uni(s_tile[T.idx(ar, ac)])) == STAG_ANCHOR_PIXEL
                               : sparse ? (P.word_at(ar, ac) >> 12) == 1
                                        : sr_uni(G.edge[off]) == STAG_ANCHOR_PIXEL;
#ifdef RW_TIMING
            rw_live++;
#endif
            if (!still) continue;
            S.R.pix = pix0 + pix_used;
            S.R.capPix = capPix0 - pix_used;
            S.R.chains = chain0 + chain_used;
            const int left = capChain0 - chain_used;
            S.R.capChains = left < 32767 ? left : 32767;
            if (S.R.capPix < 16 || S.R.capChains < 4) {
                S.overflow |= 32;
                break;
            }
#ifdef RW_TIMING
            const unsigned long long rw_a = __builtin_readcyclecounter();
#endif
            const bool keep = tiled    ? S.walk_anchor_tile6(ar, ac, grad_thresh, lane, T)
                              : sparse ? S.walk_anchor_tile6(ar, ac, grad_thresh, lane, P)
                                       : S.walk_anchor_wave(ar, ac, grad_thresh, lane, SrLdsInt4{(SrLdsInt)(int *)s_wstack}, WSTACK);
#ifdef RW_TIMING
            rw_walk += __builtin_readcyclecounter() - rw_a;
            rw_walks++;
            rw_pix += S.wl_len;
            rw_chains += S.wl_chains;
#endif
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
#ifdef RW_TIMING
    if (lane == 0 && C.nanch > 200)
        printf("walk comp %d: size %d anchors %d tile %dx%d tiled %d | load %llu cycles, anchor loop %llu (walks %llu in %d walks of %d live, %d pixels, %d chains, %d kept)\n", cid, C.size, C.nanch,
               T.tw, th, (int)tiled + 2 * (int)sparse, rw_load, (unsigned long long)(__builtin_readcyclecounter() - rw_t0), rw_walk, rw_walks, rw_live, rw_pix, rw_chains, nrec);
#endif
    }  // wave 0
    if (tiled) {  // the component's own pixels back into the edge image
        __syncthreads();
        for (int r4 = 1 + (tid >> 6) * 4; r4 < th - 1; r4 += 16)
            for (int cc = 1 + lane; cc < T.tw - 1; cc += 64) {
                int lb[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (r4 + u < th - 1) lb[u] = label[(T.r0 + r4 + u) * W + T.c0 + cc];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (r4 + u < th - 1 && lb[u] == C.root)
                        G.edge[(T.r0 + r4 + u) * W + T.c0 + cc] = (uint8_t)StagRouter::Tile::edge_of(s_tile[(r4 + u) * T.tw + cc]);
            }
    } else if (sparse) {
        __syncthreads();
        // block by block (not the box again: its label image is 4 bytes a pixel): a wave per block, sixteen lanes its pixels
        constexpr int SH = StagRouter::SparseTile::SH, BW = 1 << (2 * SH);
        const int nb = P.bw * ((th + StagRouter::SparseTile::BM) >> SH);
        for (int b0 = (tid >> 6) * 64; b0 < nb; b0 += 256) {  // 64 table entries per wave and round, the blocks among them one by one
            const int myslot = b0 + lane < nb ? P.tab[b0 + lane] : 0;
            unsigned long long have = __ballot(myslot != 0);
            while (have) {
                const int j = __builtin_ctzll(have);
                have &= have - 1;
                const int slot = __builtin_amdgcn_readlane(myslot, j), b = b0 + j;
                const int brow = b / P.bw, bcol = b - brow * P.bw;
                const int rr = (brow << SH) + (lane >> SH), cc = (bcol << SH) + (lane & ((1 << SH) - 1));
                if (lane < BW && rr >= 1 && rr < th - 1 && cc >= 1 && cc < T.tw - 1) {
                    const int g = (T.r0 + rr) * W + T.c0 + cc;
                    if (label[g] == C.root) G.edge[g] = (uint8_t)StagRouter::Tile::edge_of(P.blk[(slot << (2 * SH)) + lane]);
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_stag_route_walk(StagRoute G, StagArenas A, StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int32_t *__restrict__ sorted, const int *__restrict__ aslots, const int *__restrict__ label, int grad_thresh, int lds_bytes, int *__restrict__ prodflag, int *__restrict__ ovf, int cls)
{
    k_stag_route_walk_impl(G, A, comps, cursors, order, sorted, aslots, label, grad_thresh, lds_bytes, prodflag, ovf, cls);
}
struct k_stag_route_walk_fn {
    static constexpr int kBounds = 256;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(StagRoute G, StagArenas A, StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int32_t *__restrict__ sorted, const int *__restrict__ aslots, const int *__restrict__ label, int grad_thresh, int lds_bytes, int *__restrict__ prodflag, int *__restrict__ ovf, int cls) const { k_stag_route_walk_impl<true>(G, A, comps, cursors, order, sorted, aslots, label, grad_thresh, lds_bytes, prodflag, ovf, cls); }
};

// next[r] = the smallest producing rank > r, or -1 (one workgroup, chunks of 1024 from the top)
__device__ __forceinline__ void k_stag_next_above_impl(const int *__restrict__ prodflag, const unsigned *__restrict__ n_anchors, int *__restrict__ next)
{
    // next[r] = the smallest producing rank above r (-1: none): a running minimum over the ranks taken from the top down,
    // 32 consecutive ranks per thread, wave scans by shuffles, one barrier pair per 32 768 ranks (1 024 threads)
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = (int)*n_anchors;
    constexpr int PER = 32;  // (round 6: 32 ranks a thread, whole aligned runs read and written as 16-byte vectors -- a trip costs its
                             //  memory round trips and barriers whatever it carries: five trips for a frame's ~40 k anchors were 38 us)
    int carry = INT_MAX;
    const int NT = (int)blockDim.x, NWV = NT >> 6;  // (any block size up to 1 024; launched with 256 since round 6, see k_stag_scan_counts)
    const int ntop = (n + PER - 1) / PER * PER;      // (the ranks n .. ntop - 1 do not exist: no flag, nothing written)
    for (int top = ntop; top > 0; top -= NT * PER) {
        const int lo = top - (tid + 1) * PER;        // this thread's ranks: lo + PER - 1 down to lo (lo is a multiple of PER)
        const bool whole = lo >= 0 && lo + PER <= n;
        int f[PER], loc = INT_MAX;                   // f[k]: rank lo + k if it produces, else INT_MAX
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) {
                const int4 q = *reinterpret_cast<const int4 *>(prodflag + lo + k);
                f[k] = q.x ? lo + k : INT_MAX;
                f[k + 1] = q.y ? lo + k + 1 : INT_MAX;
                f[k + 2] = q.z ? lo + k + 2 : INT_MAX;
                f[k + 3] = q.w ? lo + k + 3 : INT_MAX;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int r = lo + k;
                f[k] = (r >= 0 && r < n && prodflag[r]) ? r : INT_MAX;
            }
        }
#pragma unroll
        for (int k = 0; k < PER; k++) loc = min(loc, f[k]);
        int incl = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl = min(incl, o);
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wpre = INT_MAX, tot = INT_MAX;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = k < NWV ? s_w[k] : INT_MAX;
            if (k < wv) wpre = min(wpre, t);
            tot = min(tot, t);
        }
        const int left = __shfl_up(incl, 1, 64);
        int run = min(min(carry, wpre), lane > 0 ? left : INT_MAX);  // the smallest producing rank above this thread's ranks
#pragma unroll
        for (int k = PER - 1; k >= 0; k--) {  // from the top down: f[k] -> next[lo + k]
            const int t = f[k];
            f[k] = run == INT_MAX ? -1 : run;
            run = min(run, t);
        }
        if (whole) {
#pragma unroll
            for (int k = 0; k < PER; k += 4) *reinterpret_cast<int4 *>(next + lo + k) = make_int4(f[k], f[k + 1], f[k + 2], f[k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int r = lo + k;
                if (r >= 0 && r < n) next[r] = f[k];
            }
        }
        carry = min(carry, tot);
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_stag_next_above(const int *__restrict__ prodflag, const unsigned *__restrict__ n_anchors, int *__restrict__ next)
{
    k_stag_next_above_impl(prodflag, n_anchors, next);
}
struct k_stag_next_above_fn {
    static constexpr int kBounds = 1024;
    __device__ __forceinline__ void operator()(const int *__restrict__ prodflag, const unsigned *__restrict__ n_anchors, int *__restrict__ next) const { k_stag_next_above_impl(prodflag, n_anchors, next); }
};

template <bool FM = false, int EX_CHAINS = 512, int EX_PIX = 1024, int NWV = 4>
__device__ __forceinline__ void k_stag_route_extract_impl(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors,
                                                            const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors,
                                                            int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where,
                                                            int *__restrict__ ovf, int cls)
{
    // one wave per component: every lane runs the same scalar steps (same values, same stores); pixel runs are copied by all lanes.
    // The chain tree of the anchor being extracted (and the stack of its tree walk) sits in LDS when it has <= EX_CHAINS chains:
    // the extraction prunes and empties chains as it goes, none of which has to survive it.
    // (round 5) so do the anchor's pixels and the block of output pixels it produces, when the walk left <= EX_PIX pixels: every
    // look at a chain's first pixels or at the segment's tail (trim_tail, append_forward) was a round trip to global memory, some
    // of them behind the stores just issued -- half of this kernel's time on a marker's component
    // (EX_CHAINS / EX_PIX are template arguments since round 6: 512 / 1 024 = 32 KB a wave, one workgroup per CU -- what a marker's
    //  component needs; a group of frames takes its SMALL components (< STAG_BIG_COMP pixels) through an instance of 128 / 256 = 8 KB a
    //  wave, four workgroups per CU, in a launch of its own: cls as in k_stag_route_walk_impl.  Which instance a component meets
    //  only decides where its chain tree lives while it is taken apart -- the arithmetic is the same.)
    // (NWV waves per workgroup, a component each: 4 for a frame on its own and for a group's small components; ONE for a group's big
    //  components -- a 4-wave workgroup of the 32 KB-a-wave instance needs a CU with 128 KB of LDS free, and beside the other groups'
    //  kernels it waited for one longer than it then ran: 479 us alone, 1 110 us in the batch)
    __shared__ StagChain s_chains[NWV][EX_CHAINS];
    __shared__ int4 s_stack[NWV][EX_CHAINS];
    __shared__ int2 s_pix[NWV][EX_PIX + 1], s_out[NWV][EX_PIX + 1];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rank = (int)STAG_BX<FM>() * NWV + wv + (cls == 2 ? cursors[11] : 0);
    if (rank >= (cls == 1 ? cursors[11] : cursors[0])) return;
    const int cid = stag_comp_by_rank(order, cursors, rank);  // (longest first, as the walk took them)
    const StagComp C = sr_uni_struct(comps[cid]);
    if (C.nanch == 0 || C.nrec == 0) return;
    StagRouter S;
    stag_bind(S, G, A, C);
    S.wlane = lane;
    S.noSegments = S.totalPixels = S.overflow = 0;
    S.segbase = S.nsp = 0;
    StagRec *recs = A.recs + C.anch_base;
    int2 *pix0 = S.R.pix;
    int2 *out0p = S.R.outpix;
    StagChain *chain0 = S.R.chains;
    int4 *stack0 = S.R.stack;
    const int capStack0 = S.R.capStack, capOut0 = S.R.capOut;
#ifdef RW_TIMING
    const unsigned long long ex_t0 = __builtin_readcyclecounter();
#endif
    const int n = (int)*n_anchors;
    int prev_rank = -1;
    for (int k = 0; k < C.nrec; k++) {
        StagRec r = sr_uni_struct(recs[k]);
        S.R.pix = pix0 + r.pix_off;
        // the block the reference wrote just before this one: ours only if no other component produced in between
        S.prev_valid = k > 0 && sr_uni(next[r.rank]) == prev_rank;
        const int seg0 = S.noSegments, out0 = S.totalPixels;
        // (the tree pointers are set and used INSIDE each branch on purpose: merged in front of one call site they are "LDS or
        //  global", i.e. generic, and every access to the chain tree was a flat_load / flat_store -- 63 of them in this kernel --
        //  that waits like an LDS and a memory operation at once; per branch the compiler knows which it is: ds_* in the common case)
        if (r.nchains <= EX_CHAINS && r.len <= EX_PIX) {
            for (int i = lane; i < r.nchains; i += 64) s_chains[wv][i] = chain0[r.chain_off + i];
            for (int i = lane; i <= r.len; i += 64) s_pix[wv][i] = pix0[r.pix_off + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            S.R.chains = s_chains[wv];
            S.R.stack = s_stack[wv];
            S.R.capStack = EX_CHAINS;
            S.R.pix = s_pix[wv];
            S.R.outpix = s_out[wv] - out0;  // (absolute indices: the block begins at out0)
            // (a block is a subset of the anchor's <= EX_PIX pixels; should that ever not hold the overflow flag goes up -- the
            //  frame then takes the sequential road -- instead of a write behind the buffer)
            S.R.capOut = capOut0 < out0 + EX_PIX ? capOut0 : out0 + EX_PIX;
            S.extract_anchor(r.nchains);
            S.R.capOut = capOut0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            {  // the block to its place in the component's arena
                const int nout = S.totalPixels - out0, room = capOut0 - out0;
                int ncopy = nout < room ? nout : room;
                ncopy = ncopy < EX_PIX ? ncopy : EX_PIX;  // (beyond: the arena overflowed, the frame takes the sequential road)
                for (int i = lane; i < ncopy; i += 64) out0p[out0 + i] = s_out[wv][i];
            }
        } else if (r.nchains <= EX_CHAINS) {
            for (int i = lane; i < r.nchains; i += 64) s_chains[wv][i] = chain0[r.chain_off + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            S.R.chains = s_chains[wv];
            S.R.stack = s_stack[wv];
            S.R.capStack = EX_CHAINS;
            S.R.outpix = out0p;
            S.extract_anchor(r.nchains);
        } else {
            S.R.chains = chain0 + r.chain_off;
            S.R.stack = stack0;
            S.R.capStack = capStack0;
            S.R.outpix = out0p;
            S.extract_anchor(r.nchains);
        }
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
#ifdef RW_TIMING
    if (lane == 0 && C.nrec > 8)
        printf("extract comp %d: recs %d | total %llu cycles: longest %llu, first segment %llu, other chains %llu\n", cid, C.nrec,
               (unsigned long long)(__builtin_readcyclecounter() - ex_t0), S.ex_t[0], S.ex_t[1], S.ex_t[2]);
#endif
    if (S.overflow && lane == 0) atomicOr(ovf, S.overflow);
}
__global__ __launch_bounds__(256) void k_stag_route_extract(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls)
{
    k_stag_route_extract_impl(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls);
}
struct k_stag_route_extract_fn {
    static constexpr int kBounds = 256;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls) const { k_stag_route_extract_impl<true>(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls); }
};
// (a group's big components: one wave, 32 KB, per workgroup)
__global__ __launch_bounds__(64) void k_stag_route_extract_big(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls)
{
    k_stag_route_extract_impl<false, 512, 1024, 1>(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls);
}
struct k_stag_route_extract_big_fn {
    static constexpr int kBounds = 64;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls) const { k_stag_route_extract_impl<true, 512, 1024, 1>(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls); }
};
// (the small-footprint instance: group mode only)
__global__ __launch_bounds__(256) void k_stag_route_extract_small(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls)
{
    k_stag_route_extract_impl<false, 128, 256>(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls);
}
struct k_stag_route_extract_small_fn {
    static constexpr int kBounds = 256;
    static constexpr bool kFrameMinor = true;
    __device__ __forceinline__ void operator()(StagRoute G, StagArenas A, const StagComp *__restrict__ comps, const int *__restrict__ cursors, const int *__restrict__ order, const int *__restrict__ next, const unsigned *__restrict__ n_anchors, int *__restrict__ blk_pix, int *__restrict__ blk_segs, int2 *__restrict__ blk_where, int *__restrict__ ovf, int cls) const { k_stag_route_extract_impl<true, 128, 256>(G, A, comps, cursors, order, next, n_anchors, blk_pix, blk_segs, blk_where, ovf, cls); }
};

// blk_pix / blk_segs hold exclusive prefix sums by now: copy every block to its place in the global order
__device__ __forceinline__ void k_stag_route_gather_impl(int per_wave, StagArenas A, const StagComp *__restrict__ comps, const unsigned *__restrict__ n_anchors,
                                                           const int *__restrict__ prodflag, const int *__restrict__ blk_pix,
                                                           const int *__restrict__ blk_segs, const int2 *__restrict__ blk_where,
                                                           int2 *__restrict__ outpix, int2 *__restrict__ segs, int capOut, int capSegs, int *__restrict__ ovf)
{
    // a LANE per anchor asks whether it produced (most did not: one wave per anchor was 7 500 workgroups a frame, nearly all of
    // them one load and an exit), the wave then copies the blocks of the ones that did, one after the other
    const int lane = threadIdx.x & 63;
    const int n = (int)*n_anchors;
    const int q0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * per_wave;  // (per_wave: 64, or 1 = a wave per anchor)
    if (q0 >= n) return;
    const int qm = q0 + lane;
    unsigned long long todo = __ballot(lane < per_wave && qm < n && prodflag[n - 1 - qm] != 0);
    while (todo) {
        const int q = q0 + __builtin_ctzll(todo);
        todo &= todo - 1;
        const int2 w = blk_where[q];
        const StagComp C = comps[w.x];
        const StagRec r = (A.recs + C.anch_base)[w.y];
        const int po = blk_pix[q], so = blk_segs[q];
        if (po + r.out_len > capOut || so + r.nsegs > capSegs) {
            if (lane == 0) atomicOr(ovf, 64);
            continue;
        }
        const int2 *src = A.out + C.out_base + r.out_off;
        for (int i = lane; i < r.out_len; i += 64) outpix[po + i] = src[i];
        const int2 *sg = A.segs + C.seg_base + r.seg_off;
        for (int i = lane; i < r.nsegs; i += 64) segs[so + i] = make_int2(sg[i].x - r.out_off + po, sg[i].y);
    }
}
__global__ __launch_bounds__(256) void k_stag_route_gather(int per_wave, StagArenas A, const StagComp *__restrict__ comps, const unsigned *__restrict__ n_anchors, const int *__restrict__ prodflag, const int *__restrict__ blk_pix, const int *__restrict__ blk_segs, const int2 *__restrict__ blk_where, int2 *__restrict__ outpix, int2 *__restrict__ segs, int capOut, int capSegs, int *__restrict__ ovf)
{
    k_stag_route_gather_impl(per_wave, A, comps, n_anchors, prodflag, blk_pix, blk_segs, blk_where, outpix, segs, capOut, capSegs, ovf);
}
struct k_stag_route_gather_fn {
    static constexpr int kBounds = 256;
    __device__ __forceinline__ void operator()(int per_wave, StagArenas A, const StagComp *__restrict__ comps, const unsigned *__restrict__ n_anchors, const int *__restrict__ prodflag, const int *__restrict__ blk_pix, const int *__restrict__ blk_segs, const int2 *__restrict__ blk_where, int2 *__restrict__ outpix, int2 *__restrict__ segs, int capOut, int capSegs, int *__restrict__ ovf) const { k_stag_route_gather_impl(per_wave, A, comps, n_anchors, prodflag, blk_pix, blk_segs, blk_where, outpix, segs, capOut, capSegs, ovf); }
};
