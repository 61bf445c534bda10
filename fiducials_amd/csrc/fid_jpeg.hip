// fid_jpeg.hip -- JPEG ingest on the device (include/fid_abi.h: fid_jpeg_*).  Part of the fid_api.hip translation unit.
//
// Where it sits: with `transport:=compressed` (aruco_detect/launch/aruco_detect.launch:6, the launch file's default) every
// frame reaches FiducialsNode::imageCallback (aruco_detect.cpp:332) through image_transport's compressed subscriber, which
// calls cv::imdecode -- libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy upsampling, JFIF YCbCr -> RGB.  The arithmetic
// is libjpeg's (a third-party dependency, not in the reference tree); oracle/jpeg_oracle.c restates it and is pinned bit for
// bit on libjpeg-turbo's own output, and the parity tests compare every stage of this file with that oracle.
//
// Stages (all integer work, bit-exact):
//   host   header parse (markers, quantisation and Huffman tables), 16-bit code lookup tables (cached per table contents)
//   J1  k_jpeg_huff<0/1/2>  entropy decoding (jdhuff.c decode_mcu).  A Huffman stream can only be read from the front -- but it
//       re-synchronises: a decoder started at a wrong bit falls into step with the true symbol boundaries after a few
//       symbols.  The scan is cut into sub-sequences of JP_SUB bytes, one lane each:
//         <0> every lane decodes its sub-sequence from its first bit (speculative: only sub-sequence 0 starts on a symbol)
//             and leaves its EXIT STATE: bit position, block within the MCU, zig-zag index;
//         <1> round r: lane i decodes sub-sequence i again from the exit state of lane i-1; a lane whose entry did not change
//             in the previous round keeps its result.  Sub-sequence 0 is exact from the start, so exactness spreads at least
//             one sub-sequence per round -- in practice everything agrees after five to seven rounds of 128-byte
//             sub-sequences (the block phase takes longer to agree than the symbol boundaries; restart markers are where
//             all decoders meet).  When no exit state changed, every exit state is the true one;
//         scan of the blocks completed per sub-sequence -> where each lane's output goes;
//         <2> the same decode once more, now writing coefficients (sparse: the buffer was cleared).
//       Every pass first unstuffs the workgroup's bytes into LDS (FF 00, fill bytes, markers), so that the symbol loop reads
//       plain bits; states are exchanged as RAW bit offsets that never point into a removed byte (equal logical positions
//       compare equal whoever computed them).
//   J2  k_jpeg_scan   per sub-sequence: blocks completed before it and, per component, the DC value its first block continues
//                     from (DC prediction: segmented prefix sums of the differences, segments = restart intervals) -- the
//                     writing pass then stores absolute DC values straight away
//   J3  k_jpeg_idct   dequantise + jidctint.c (13-bit constants, two passes), eight lanes per block, transposed through LDS
//   J4  k_jpeg_color  jdsample.c fancy upsampling (h2v1 / h2v2, edge rows and columns as jdmainct.c replicates them) +
//                     jdcolor.c YCbCr -> RGB, written as BGR (cv::imdecode) or straight as the gray image
//                     cvtColor(BGR2GRAY) makes of it (K0's formula) -- the detector's input, without the colour image
#include <unordered_map>

#define JP_SUB 128         // bytes per sub-sequence (a decoder that starts out of step needs a few hundred bytes to fall into step with
                           // the block phase as well as the symbol boundaries: 64-byte sub-sequences took 11-15 rounds, 256-byte ones 3)
#define JP_TPB 256         // lanes (sub-sequences) per workgroup
#define JP_LOOK 10         // bits of the LDS first-level code tables

struct JpImage {
    unsigned long long scan_off;   // into the packed scan bytes of the call
    unsigned long long coef_base;  // int16 units: this frame's coefficient area
    unsigned long long plane_base; // bytes: this frame's plane area
    uint32_t scan_len, sub_base, nsub, nblocks;
    int32_t w, h, ncomp, hs, vs, bpm, mcux, nmcu, restart;
    int32_t bw[3], bh[3];
    uint32_t coef_off[3], plane_off[3];  // within the frame's areas (int16 units / bytes)
    int32_t lut_dc[3], lut_ac[3];        // slots of the 16-bit code tables
    uint16_t q[3][64];                   // quantisation tables, natural order
};

namespace {

__constant__ uint8_t c_jp_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- the workgroup's bytes, unstuffed.  Every lane cleans one sub-sequence (the last lane's is the next workgroup's first: it
// only serves the lane in front of it, which may end a symbol behind its own): FF 00 ->
// FF, fill bytes and markers dropped, and writes what is left into one packed LDS buffer, so that the symbol loop below reads
// plain bits.  What has to survive of the raw layout:
//   s_R[j], s_rm[j]   removed bytes before sub-sequence j / which of its 128 bytes were removed: raw <-> packed positions
//                     (states are exchanged as RAW bit positions: they are the same whoever computes them)
//   s_bnd             packed byte positions in front of which a restart marker stood: every decoder starts afresh there
//   s_end             packed byte position of the end of the data (EOI, another marker, or the end of the file)
#define JP_NSTG JP_TPB          // sub-sequences a workgroup unstuffs: one per lane ...
#define JP_OWN (JP_TPB - 1)      // ... of which it decodes all but the last (that one only serves the lane in front of it)
struct JpStage {
    uint8_t cmp[JP_NSTG * JP_SUB + 16];
    uint32_t rm[JP_NSTG][4];
    uint32_t R[JP_NSTG + 1];
    uint32_t bnd[JP_NSTG * JP_SUB / 32 + 2];
    uint32_t end;
    int wsum[JP_TPB / 64];
};

__device__ __forceinline__ int jp_popc_below(const uint32_t m[4], int o)  // set bits of the 128-bit mask below bit o
{
    int n = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int lo = w * 32;
        if (o >= lo + 32) n += __popc(m[w]);
        else if (o > lo) n += __popc(m[w] & ((1u << (o - lo)) - 1u));
    }
    return n;
}
// raw byte (relative to the workgroup's first byte) -> packed byte
__device__ __forceinline__ uint32_t jp_raw_to_cmp(const JpStage &S, uint32_t rawk)
{
    const uint32_t j = rawk / JP_SUB, o = rawk % JP_SUB;
    return rawk - S.R[j] - (uint32_t)jp_popc_below(S.rm[j], (int)o);
}
// packed byte -> raw byte (relative); j: a sub-sequence at or before the one that holds it
__device__ __forceinline__ uint32_t jp_cmp_to_raw(const JpStage &S, uint32_t cb, uint32_t j, uint32_t nstg)
{
    while (j + 1 < nstg && (j + 1) * JP_SUB - S.R[j + 1] <= cb) j++;
    uint32_t n = cb - (j * JP_SUB - S.R[j]);  // its index among the kept bytes of sub-sequence j
    for (int w = 0; w < 4; w++) {
        uint32_t kept = ~S.rm[j][w];
        const uint32_t cnt = (uint32_t)__popc(kept);
        if (n < cnt) {
            for (; n > 0; n--) kept &= kept - 1;
            return j * JP_SUB + w * 32 + (uint32_t)(__ffs(kept) - 1);
        }
        n -= cnt;
    }
    return (j + 1) * JP_SUB;  // (behind the last kept byte)
}

// ---- the bit reader over the packed bytes: MSB first, 32 bits at a time
struct JpReader {
    const uint8_t *s;        // packed bytes
    uint32_t bytepos;        // next packed byte to load
    unsigned long long acc;  // the low `nbits` bits are valid
    int nbits;
    __device__ __forceinline__ void start(uint32_t bitpos)
    {
        bytepos = bitpos >> 3;
        acc = 0;
        nbits = 0;
        const int drop = (int)(bitpos & 7u);
        if (drop) {
            acc = s[bytepos++];
            nbits = 8 - drop;
        }
    }
    // a symbol takes at most 27 bits
    __device__ __forceinline__ void fill()
    {
        if (nbits < 32) {
            const uint32_t a = bytepos & ~3u, sh = (bytepos & 3u) * 8u;
            const uint32_t w0 = *reinterpret_cast<const uint32_t *>(s + a), w1 = *reinterpret_cast<const uint32_t *>(s + a + 4);
            const uint32_t le = sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;  // the four bytes at bytepos, first byte lowest
            const uint32_t be = __builtin_bswap32(le);
            acc = (acc << 32) | be;
            nbits += 32;
            bytepos += 4;
        }
    }
    __device__ __forceinline__ uint32_t pos() const { return bytepos * 8u - (uint32_t)nbits; }
    __device__ __forceinline__ unsigned peek(int n) const { return (unsigned)((acc >> (nbits - n)) & ((1ull << n) - 1ull)); }
    __device__ __forceinline__ void drop(int n) { nbits -= n; }
};

struct JpState {
    uint32_t p;   // raw bit position
    uint32_t cz;  // block within the MCU | zig-zag index << 8 | end-of-image << 16
};
__device__ __forceinline__ bool operator!=(const JpState &a, const JpState &b) { return a.p != b.p || a.cz != b.cz; }

// MODE 0: speculative first pass, 1: synchronisation round, 2: writing pass
template <int MODE>
__global__ __launch_bounds__(JP_TPB) void k_jpeg_huff(const JpImage *__restrict__ imgs, const uint8_t *__restrict__ scan, const uint16_t *__restrict__ luts, const uint16_t *__restrict__ lut1,
                                                     const JpState *__restrict__ st_in, JpState *__restrict__ st_out,
                                                     const uint8_t *__restrict__ chg_in, uint8_t *__restrict__ chg_out, int4 *__restrict__ nblk,
                                                     const int4 *__restrict__ blkbase, int16_t *__restrict__ coefs, unsigned *__restrict__ any_changed)
{
    __shared__ JpStage S;
    __shared__ uint16_t s_lut[4][1 << JP_LOOK];  // [dc luma, ac luma, dc chroma, ac chroma] first-level tables
    __shared__ uint8_t s_zz[64];
    const JpImage &I = imgs[blockIdx.y];
    const uint32_t first = blockIdx.x * JP_OWN;
    if (first >= I.nsub) return;
    const uint8_t *g = scan + I.scan_off;
    const uint32_t s_lo = first * JP_SUB;
    const uint32_t tid = threadIdx.x, i = first + tid;
    const bool decoder = tid < JP_OWN && i < I.nsub;
    const uint32_t gsub = I.sub_base + (i < I.nsub ? i : 0u);
    // ---- does this workgroup have anything to do in this round?
    bool active = decoder;
    if (MODE == 1 && active) active = i > 0 && chg_in[gsub - 1];
    if (MODE == 1) {
        if (decoder && !active) {  // nothing new to start from: the previous result stands
            st_out[gsub] = st_in[gsub];
            chg_out[gsub] = 0;
        }
        if (!__syncthreads_or(active ? 1 : 0)) return;
    }
    // ---- first-level code tables
    const int lc = I.ncomp > 1 ? 1 : 0;  // the chroma component that names the second pair of tables
    const int slot[4] = {I.lut_dc[0], I.lut_ac[0], I.lut_dc[lc], I.lut_ac[lc]};
    // (Cb and Cr may name different tables: then component 2 goes through the 16-bit tables only)
    const bool cr_same = I.ncomp < 3 || (I.lut_dc[2] == I.lut_dc[1] && I.lut_ac[2] == I.lut_ac[1]);
    for (int k = tid; k < 4 << JP_LOOK; k += JP_TPB) {
        const int t = k >> JP_LOOK, q = k & ((1 << JP_LOOK) - 1);
        s_lut[t][q] = lut1[(size_t)slot[t] * (1 << JP_LOOK) + q];
    }
    for (int k = tid; k < JP_NSTG * JP_SUB / 32 + 2; k += JP_TPB) S.bnd[k] = 0u;
    if (MODE == 2 && tid < 64) s_zz[tid] = c_jp_zigzag[tid];
    if (tid == 0) S.end = 0xffffffffu;
    // ---- unstuff: sub-sequences first .. first + nstg - 1 (the last one only serves the lane in front of it)
    const uint32_t nstg = (I.nsub - first) >= JP_NSTG ? JP_NSTG : (I.nsub - first);
    uint32_t w[1][32];   // the raw bytes of this lane's sub-sequence
    uint32_t rmv[1][4], bndm[1][4], endm[1][4];
    int nrem[1] = {0};
    {
        constexpr int t = 0;
        const uint32_t j = tid;
        const bool mine = tid < nstg;
#pragma unroll
        for (int q = 0; q < 4; q++) rmv[t][q] = bndm[t][q] = endm[t][q] = 0u;
        if (mine) {
        const uint32_t r0 = s_lo + j * JP_SUB;  // first raw byte
        uint32_t F[4] = {0, 0, 0, 0}, Z[4] = {0, 0, 0, 0}, D[4] = {0, 0, 0, 0};  // byte is FF / 00 / D0..D7
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r0 + q * 16 + 16 <= I.scan_len) {
                v = *reinterpret_cast<const uint4 *>(g + r0 + q * 16);
            } else {
                uint32_t tmp[4] = {0, 0, 0, 0};
                for (int u = 0; u < 16; u++)
                    if (r0 + q * 16 + u < I.scan_len) tmp[u >> 2] |= (uint32_t)g[r0 + q * 16 + u] << ((u & 3) * 8);
                v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t x = vv[u];
                w[t][q * 4 + u] = x;
                // one bit per byte: the byte equals FF / equals 00 / lies in D0..D7
                uint32_t f = 0, z0 = 0, d = 0;
#pragma unroll
                for (int bb = 0; bb < 4; bb++) {
                    const uint32_t by = (x >> (8 * bb)) & 0xffu;
                    f |= (by == 0xffu ? 1u : 0u) << bb;
                    z0 |= (by == 0u ? 1u : 0u) << bb;
                    d |= ((by & 0xf8u) == 0xd0u ? 1u : 0u) << bb;
                }
                const int bit = (q * 4 + u) * 4;
                F[bit >> 5] |= f << (bit & 31);
                Z[bit >> 5] |= z0 << (bit & 31);
                D[bit >> 5] |= d << (bit & 31);
            }
        }
        // bytes behind the end of the data count as an end marker (FF D9)
        const uint32_t valid = r0 >= I.scan_len ? 0u : (I.scan_len - r0 >= JP_SUB ? JP_SUB : I.scan_len - r0);
        const unsigned prev = r0 > 0 ? g[r0 - 1] : 0u;
        const unsigned next = r0 + JP_SUB < I.scan_len ? g[r0 + JP_SUB] : 0xD9u;
        const uint32_t pF = prev == 0xffu, nF = next == 0xffu, nZ = next == 0u, nD = (next & 0xf8u) == 0xd0u;
        // neighbours: previous byte is FF / next byte is 00 / FF / D0..D7
        uint32_t prevF[4], nextZ[4], nextF[4], nextD[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            prevF[q] = (F[q] << 1) | (q ? F[q - 1] >> 31 : pF);
            nextZ[q] = (Z[q] >> 1) | ((q < 3 ? Z[q + 1] & 1u : nZ) << 31);
            nextF[q] = (F[q] >> 1) | ((q < 3 ? F[q + 1] & 1u : nF) << 31);
            nextD[q] = (D[q] >> 1) | ((q < 3 ? D[q + 1] & 1u : nD) << 31);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // in range?
            uint32_t in = 0xffffffffu;
            if (valid < (uint32_t)(q * 32 + 32)) in = valid > (uint32_t)(q * 32) ? (1u << (valid - q * 32)) - 1u : 0u;
            const uint32_t second = prevF[q] & ~F[q];             // the byte behind an FF: a stuffed zero or a marker's code
            const uint32_t lead = F[q] & ~nextZ[q];               // an FF that is not data: a fill byte or a marker's first byte
            const uint32_t mark = lead & ~nextF[q];               // ... a marker's first byte
            rmv[t][q] = ((second | lead) & in) | ~in;             // (bytes behind the end are "removed" as well)
            bndm[t][q] = mark & nextD[q] & in;
            endm[t][q] = mark & ~nextD[q] & in;
            nrem[t] += __popc(rmv[t][q] & in) + (32 - __popc(in));
        }
        // (the byte behind the end of the data is an end marker too)
        if (valid < JP_SUB && r0 + valid == I.scan_len) endm[t][valid >> 5] |= 1u << (valid & 31);
        }
    }
    // removed bytes before every sub-sequence: scan over the lanes (+ the extra one of lane 0 at the end)
    {
        const int incl = wave_iscan(nrem[0]);
        if ((tid & 63u) == 63u) S.wsum[tid >> 6] = incl;
        __syncthreads();
        int wb = 0;
        for (uint32_t k = 0; k < (tid >> 6); k++) wb += S.wsum[k];
        if (tid < nstg) S.R[tid] = (uint32_t)(wb + incl - nrem[0]);
        if (tid == JP_TPB - 1) S.R[JP_TPB] = (uint32_t)(wb + incl);
        if (tid < nstg)
            for (int q = 0; q < 4; q++) S.rm[tid][q] = rmv[0][q];
    }
    __syncthreads();
    // packed bytes, restart boundaries, end of data
    if (tid < nstg) {
        constexpr int t = 0;
        const uint32_t j = tid;
        uint32_t dst = j * JP_SUB - S.R[j];
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const uint32_t x = w[t][q];
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const int k = q * 4 + bb;
                const uint32_t bit = 1u << (k & 31);
                if (bndm[t][k >> 5] & bit) atomicOr(&S.bnd[dst >> 5], 1u << (dst & 31));
                if (endm[t][k >> 5] & bit) atomicMin(&S.end, dst);
                if (!(rmv[t][k >> 5] & bit)) S.cmp[dst++] = (uint8_t)(x >> (8 * bb));
            }
        }
        if (j + 1 == nstg)  // (zeros behind the last packed byte: the reader looks a few bytes ahead)
            for (int u = 0; u < 16; u++) S.cmp[dst + u] = 0;
    }
    __syncthreads();
    // ---- decode.  MODE 0 / 1 go round INSIDE the kernel (round 4): a lane whose predecessor sits in the same workgroup takes that
    // lane's new exit state from LDS and decodes again, until nothing in the workgroup changes -- the bytes stay unstuffed in
    // LDS, the tables stay loaded, no launch and no host look in between.  Only a workgroup's first lane waits for another
    // launch (its predecessor is the previous workgroup's last lane).  A 256-frame batch used to take a speculative pass and
    // six or seven synchronisation launches with a host look after every second one; now: the speculative launch (which
    // settles every workgroup within itself), one launch that carries the true states across the workgroup boundaries, one
    // that finds nothing left to change.  Same decodes from the same entry states: the settled exit states are the true ones.
    __shared__ JpState s_exit[JP_TPB];
    const int nl = I.hs * I.vs;  // luma blocks per MCU
    const uint32_t cstart = tid < nstg ? tid * JP_SUB - S.R[tid] : 0u;                                              // packed: my first byte
    const uint32_t cend = tid < nstg ? (tid + 1 < nstg ? (tid + 1) * JP_SUB - S.R[tid + 1] : (tid + 1) * JP_SUB - S.R[tid] - (uint32_t)nrem[0]) : 0u;
    const uint32_t end_bit = cend * 8u;  // this lane's part ends with the first symbol that starts at or behind it
    const uint32_t end_data = S.end == 0xffffffffu ? 0xffffffffu : S.end * 8u;
    // the next restart boundary / the end of the data at or behind a packed bit position
    auto next_boundary = [&](uint32_t bitpos) -> uint32_t {
        uint32_t q = (bitpos + 7u) >> 3;  // first whole byte position at or behind it
        const uint32_t lim = cend + 8u;   // (a lane never gets further than a symbol behind its end)
        uint32_t wd = q >> 5;
        uint32_t m = S.bnd[wd] & ~((1u << (q & 31)) - 1u);
        while (!m && wd * 32 + 32 <= lim) m = S.bnd[++wd];
        uint32_t bq = m ? wd * 32 + (uint32_t)__ffs(m) - 1u : 0xffffffffu;
        uint32_t bb = bq == 0xffffffffu ? 0xffffffffu : bq * 8u;
        return bb < end_data ? bb : end_data;
    };
    // what a lane has to show: its exit state and what the scan needs of it (blocks done, DC sums); `e_used` = the entry state
    // that result was decoded from
    JpState e_used, o_state;
    int4 o_nblk = make_int4(0, 0, 0, 0);
    bool have = false;          // this launch decoded this lane at least once
    bool live = decoder;        // decodes in the coming round
    JpState e;
    e.p = 0;
    e.cz = 0;
    bool spec = false;          // the coming decode starts at the lane's own first byte (MODE 0's first round)
    if (MODE == 0) {
        spec = i > 0;           // (sub-sequence 0 starts on a symbol: exact from the start)
        // the state this start stands for, should the predecessor happen to end exactly here
        e.p = (s_lo + tid * JP_SUB) * 8u;
        e.cz = 0;
        if (i == 0) e.p = 0;
    } else if (MODE == 1) {
        live = decoder && active;
        if (decoder && i > 0) e = st_in[gsub - 1];
        // (a lane that is not active has, in st_in[gsub], the result of decoding from st_in[gsub - 1])
        if (decoder && !live) {
            o_state = st_in[gsub];
        }
    } else if (decoder && i > 0) {
        e = st_in[gsub - 1];
    }
    e_used = e;
    for (int round = 0;; round++) {
      if (live) {
        int c = (int)(e.cz & 0xffu), z = (int)((e.cz >> 8) & 0xffu);
        bool eoi = (e.cz >> 16) & 1u;
        uint32_t done = 0;  // blocks completed by this lane
        uint32_t entry_bit = cstart * 8u;
        if (!spec && i > 0 && !eoi) {
            // (a predecessor that has met the end of the image needs no entry position -- and its p may lie in front of this
            //  workgroup's staged bytes: untrusted input must not turn that into an index.  Clamped for the same reason.)
            const uint32_t pb = e.p >> 3;
            uint32_t rawk = pb > s_lo ? pb - s_lo : 0u;
            rawk = rawk < (uint32_t)nstg * JP_SUB ? rawk : (uint32_t)nstg * JP_SUB;
            entry_bit = jp_raw_to_cmp(S, rawk) * 8u + (e.p & 7u);
        }
        JpReader R;
        R.s = S.cmp;
        R.start(entry_bit);
        uint32_t Mb = next_boundary(entry_bit);
        // MODE 2: where the block being decoded goes.  (block in the MCU, MCU column, MCU row) are counted along; only the start
        // needs divisions.
        int16_t *cblk = nullptr;
        uint32_t B = 0, mx = 0, my = 0;
        auto bind_block = [&]() {
            if (MODE != 2) return;
            cblk = nullptr;
            if (B >= I.nblocks) return;  // (more blocks than the frame has: damaged data, dropped)
            const int comp = c < nl ? 0 : 1 + (c - nl);
            const int v = comp == 0 ? c / I.hs : 0, h = comp == 0 ? c - v * I.hs : 0;
            const int hsc = comp == 0 ? I.hs : 1, vsc = comp == 0 ? I.vs : 1;
            cblk = coefs + I.coef_base + I.coef_off[comp] + ((size_t)(my * vsc + v) * I.bw[comp] + (mx * hsc + h)) * 64;
        };
        if (MODE == 2) {
            B = (uint32_t)blkbase[gsub].x;
            const uint32_t m = B / (uint32_t)I.bpm;
            // (for a sound file B mod bpm == c; a damaged one may disagree: the entry state decides the tables, B the place)
            mx = m % (uint32_t)I.mcux;
            my = m / (uint32_t)I.mcux;
            bind_block();
        }
        // DC prediction: the decoding passes sum the differences per component (since the last restart marker this lane met), the
        // writing pass starts from what the scan made of those sums and stores absolute values
        int dc0 = 0, dc1 = 0, dc2 = 0, had_reset = 0;
        if (MODE == 2) {
            const int4 bb = blkbase[gsub];
            dc0 = bb.y;
            dc1 = bb.z;
            dc2 = bb.w;
        }
        uint32_t p_exit_c = entry_bit;  // packed bit position where this lane stops
        if (!eoi) {
            for (;;) {
                R.fill();
                const uint32_t pos = R.pos();
                // ---- at a restart boundary (or within its padding of one bits), or at the end of the data?
                if (pos + 8u > Mb) {
                    const int rem = (int)Mb - (int)pos;
                    const bool pad = rem <= 0 || R.peek(rem) == (1u << rem) - 1u;
                    if (pad) {
                        if (Mb == end_data) {
                            eoi = true;
                            p_exit_c = Mb;
                            break;
                        }
                        // every decoder starts afresh behind a restart marker
                        R.start(Mb);
                        // (a block or MCU cut short by the marker is abandoned: sound data ends intervals on MCU boundaries)
                        if (MODE == 2 && (c != 0 || z != 0)) {
                            B += (uint32_t)(I.bpm - c);
                            if (++mx == (uint32_t)I.mcux) {
                                mx = 0;
                                my++;
                            }
                        }
                        c = 0;
                        z = 0;
                        dc0 = dc1 = dc2 = 0;  // (the prediction starts from zero behind a restart marker)
                        had_reset = 1;
                        bind_block();
                        const uint32_t at = Mb;
                        Mb = next_boundary(at + 1u);
                        if (at >= end_bit) {
                            p_exit_c = at;
                            break;
                        }
                        continue;
                    }
                }
                if (pos >= end_bit) {
                    p_exit_c = pos;
                    break;
                }
                // ---- one symbol: the DC difference (z == 0) or an AC run / size pair, through one instruction stream
                const int comp = c < nl ? 0 : 1 + (c - nl);
                const bool isdc = z == 0;
                uint16_t en = (comp < 2 || cr_same) ? s_lut[(comp ? 2 : 0) + (isdc ? 0 : 1)][R.peek(JP_LOOK)] : (uint16_t)0;
                if (!(en >> 8)) en = luts[(size_t)(isdc ? I.lut_dc[comp] : I.lut_ac[comp]) * 65536 + R.peek(16)];  // (a long code: rare)
                int len = en >> 8, sym = en & 0xff;
                if (len == 0) {  // no such code (only while out of step): one bit, nothing
                    len = 1;
                    sym = 0;
                }
                R.drop(len);
                const int r = isdc ? 0 : sym >> 4, sz = sym & 15;
                const int raw = sz ? (int)R.peek(sz) : 0;
                R.drop(sz);
                const int val = (sz && raw < (1 << (sz - 1))) ? raw - (1 << sz) + 1 : raw;
                int dcv = 0;
                if (isdc) {
                    dc0 += comp == 0 ? val : 0;
                    dc1 += comp == 1 ? val : 0;
                    dc2 += comp == 2 ? val : 0;
                    dcv = comp == 0 ? dc0 : (comp == 1 ? dc1 : dc2);
                }
                const int zi = isdc ? 0 : z + r;  // where the value goes (zig-zag order)
                const bool has = isdc || sz != 0;
                if (MODE == 2 && has && zi < 64 && cblk) cblk[s_zz[zi]] = (int16_t)(isdc ? dcv : val);
                z = isdc ? 1 : (sz ? (zi < 64 ? zi + 1 : 64) : (r == 15 ? z + 16 : 64));
                if (z >= 64) {
                    z = 0;
                    done++;
                    if (++c == I.bpm) {
                        c = 0;
                        if (MODE == 2 && ++mx == (uint32_t)I.mcux) {
                            mx = 0;
                            my++;
                        }
                    }
                    if (MODE == 2) {
                        B++;
                        bind_block();
                    }
                }
            }
        }
        if (MODE != 2) {
            JpState o;
            // back to a raw position (the same whoever computed it)
            o.p = eoi ? e.p : (s_lo + jp_cmp_to_raw(S, p_exit_c >> 3, tid, nstg)) * 8u + (p_exit_c & 7u);
            if (eoi && !((e.cz >> 16) & 1u)) o.p = (s_lo + jp_cmp_to_raw(S, p_exit_c >> 3, tid, nstg)) * 8u;
            o.cz = (uint32_t)c | ((uint32_t)z << 8) | (eoi ? 1u << 16 : 0u);
            o_state = o;
            o_nblk = make_int4((int)(done | (had_reset ? 0x80000000u : 0u)), dc0, dc1, dc2);
            have = true;
            e_used = e;
        }
      }
      if (MODE == 2) break;  // (the writing pass decodes once, from the settled states)
      // ---- who goes again: a lane whose predecessor in this workgroup now ends somewhere else than where this lane started
      if (decoder) s_exit[tid] = o_state;
      __syncthreads();
      live = false;
      spec = false;
      if (decoder && tid > 0 && round < JP_TPB) {
          const JpState pe = s_exit[tid - 1];
          if (pe != e_used) {
              e = pe;
              live = true;
          }
      }
      if (!__syncthreads_or(live ? 1 : 0)) break;
    }
    if (MODE != 2 && decoder) {
        if (have) nblk[gsub] = o_nblk;
        if (MODE == 0) {
            st_out[gsub] = o_state;
            chg_out[gsub] = 1;
        } else {
            const bool ch = have && o_state != st_in[gsub];
            st_out[gsub] = o_state;
            chg_out[gsub] = ch ? 1 : 0;
            if (ch) atomicOr(any_changed, 1u);
        }
    }
}

// What every sub-sequence starts from: blocks completed before it (a plain exclusive scan) and the DC predictions it
// continues (per component: the sum of the differences since the last restart marker -- a SEGMENTED scan: an element that met
// a restart marker starts the sums afresh).  One workgroup per image, 8 items per lane, the operator through shuffles.
struct JpSeg {
    int n, f, s0, s1, s2;
};
__device__ __forceinline__ JpSeg jp_seg_combine(const JpSeg &a, const JpSeg &b)
{
    JpSeg r;
    r.n = a.n + b.n;
    r.f = a.f | b.f;
    r.s0 = b.f ? b.s0 : a.s0 + b.s0;
    r.s1 = b.f ? b.s1 : a.s1 + b.s1;
    r.s2 = b.f ? b.s2 : a.s2 + b.s2;
    return r;
}
__device__ __forceinline__ JpSeg jp_seg_shfl_up(const JpSeg &v, int d)
{
    JpSeg o;
    o.n = __shfl_up(v.n, d, 64);
    o.f = __shfl_up(v.f, d, 64);
    o.s0 = __shfl_up(v.s0, d, 64);
    o.s1 = __shfl_up(v.s1, d, 64);
    o.s2 = __shfl_up(v.s2, d, 64);
    return o;
}
__global__ __launch_bounds__(1024) void k_jpeg_scan_blocks(const JpImage *__restrict__ imgs, const int4 *__restrict__ nblk, int4 *__restrict__ blkbase)
{
    __shared__ JpSeg s_w[16];
    const JpImage &I = imgs[blockIdx.x];
    const int4 *in = nblk + I.sub_base;
    int4 *out = blkbase + I.sub_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = (int)I.nsub;
    constexpr int PER = 8;
    const JpSeg zero = {0, 0, 0, 0, 0};
    JpSeg carry = zero;
    for (int base = 0; base < n; base += 1024 * PER) {
        const int i0 = base + tid * PER;
        JpSeg pre[PER], acc = zero;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            pre[k] = acc;
            JpSeg it = zero;
            if (i0 + k < n) {
                const int4 v = in[i0 + k];
                it.n = v.x & 0x7fffffff;
                it.f = (v.x >> 31) & 1;
                it.s0 = v.y;
                it.s1 = v.z;
                it.s2 = v.w;
            }
            acc = jp_seg_combine(acc, it);
        }
        JpSeg incl = acc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const JpSeg o = jp_seg_shfl_up(incl, d);
            if (lane >= d) incl = jp_seg_combine(o, incl);
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        JpSeg wpre = zero, tot = zero;
        for (int k = 0; k < 16; k++) {
            if (k == wv) wpre = tot;
            tot = jp_seg_combine(tot, s_w[k]);
        }
        JpSeg lex = jp_seg_shfl_up(incl, 1);
        if (lane == 0) lex = zero;
        const JpSeg mine = jp_seg_combine(carry, jp_seg_combine(wpre, lex));
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (i0 + k < n) {
                const JpSeg o = jp_seg_combine(mine, pre[k]);
                out[i0 + k] = make_int4(o.n, o.s0, o.s1, o.s2);
            }
        }
        carry = jp_seg_combine(carry, tot);
        __syncthreads();
    }
}

// ---- jidctint.c, one 1-D pass (dequantised inputs)
#define JP_CONST_BITS 13
#define JP_PASS1_BITS 2
__device__ __forceinline__ void jp_idct_1d(const int in[8], int out[8], int shift)
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    z2 = in[0];
    z3 = in[4];
    int tmp0 = (z2 + z3) * (1 << JP_CONST_BITS);
    int tmp1 = (z2 - z3) * (1 << JP_CONST_BITS);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446;
    tmp1 *= 16819;
    tmp2 *= 25172;
    tmp3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 *= -16069;
    z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    out[0] = (tmp10 + tmp3 + rnd) >> shift;
    out[7] = (tmp10 - tmp3 + rnd) >> shift;
    out[1] = (tmp11 + tmp2 + rnd) >> shift;
    out[6] = (tmp11 - tmp2 + rnd) >> shift;
    out[2] = (tmp12 + tmp1 + rnd) >> shift;
    out[5] = (tmp12 - tmp1 + rnd) >> shift;
    out[3] = (tmp13 + tmp0 + rnd) >> shift;
    out[4] = (tmp13 - tmp0 + rnd) >> shift;
}

// eight lanes per 8 x 8 block: a lane fetches one ROW of coefficients (one 16-byte load; eight 2-byte loads down a column kept the
// kernel at 2.2 TB/s), the block is turned through LDS, the lane takes one column (pass 1), the block is transposed through LDS
// again, the lane takes one row (pass 2) and stores its eight samples as one 8-byte word.
// clear != 0: a row that held anything is zeroed behind the read -- the coefficient array is then all zeros again when the call
// ends, and the next call's entropy decoder (which only stores non-zero coefficients) needs no 1.6 GB fill in front of it
// (256 frames 1920 x 1080 4:2:0; most rows of most blocks are zeros at camera qualities and cost no store).
__global__ __launch_bounds__(256) void k_jpeg_idct(const JpImage *__restrict__ imgs, int16_t *__restrict__ coefs, uint8_t *__restrict__ planes, int clear)
{
    __shared__ int s_ws[32][64 + 8];
    __shared__ uint4 s_in[32][9];  // (144 bytes per block: the column reads of the eight blocks of a wave fall into different banks)
    const JpImage &I = imgs[blockIdx.y];
    const uint32_t nb0 = (uint32_t)I.bw[0] * I.bh[0], nb1 = I.ncomp > 1 ? (uint32_t)I.bw[1] * I.bh[1] : 0u;
    const uint32_t total = nb0 + 2u * nb1;
    const int jb = threadIdx.x >> 3, col = threadIdx.x & 7;
    const uint32_t b = blockIdx.x * 32u + (uint32_t)jb;
    const bool live = b < total;
    int comp = 0;
    uint32_t bi = b;
    if (live && b >= nb0) {
        comp = b - nb0 >= nb1 ? 2 : 1;
        bi = b - nb0 - (comp == 2 ? nb1 : 0u);
    }
    if (live) {
        uint4 *cf = reinterpret_cast<uint4 *>(coefs + I.coef_base + I.coef_off[comp] + (size_t)bi * 64) + col;  // row `col` of the block
        const uint4 v = *cf;
        if (clear && (v.x | v.y | v.z | v.w)) *cf = make_uint4(0u, 0u, 0u, 0u);
        s_in[jb][col] = v;
    }
    __syncthreads();
    if (live) {
        const int16_t *si = reinterpret_cast<const int16_t *>(s_in[jb]);
        const uint16_t *q = I.q[comp];
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = (int)si[r * 8 + col] * (int)q[r * 8 + col];
        jp_idct_1d(in, out, JP_CONST_BITS - JP_PASS1_BITS);
#pragma unroll
        for (int r = 0; r < 8; r++) s_ws[jb][r * 8 + col] = out[r];
    }
    __syncthreads();
    if (live) {
        const int row = col;
        int in[8], out[8];
#pragma unroll
        for (int k = 0; k < 8; k++) in[k] = s_ws[jb][row * 8 + k];
        jp_idct_1d(in, out, JP_CONST_BITS + JP_PASS1_BITS + 3);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int v = out[k] + 128;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            if (k < 4) lo |= (unsigned)v << (8 * k);
            else hi |= (unsigned)v << (8 * (k - 4));
        }
        const int bw = I.bw[comp];
        const uint32_t by = bi / (uint32_t)bw, bx = bi - by * (uint32_t)bw;
        uint8_t *dst = planes + I.plane_base + I.plane_off[comp] + ((size_t)by * 8 + row) * ((size_t)bw * 8) + (size_t)bx * 8;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
    }
}

// one lane per output pixel: chroma by jdsample.c's fancy filters (or as it is for 4:4:4), jdcolor.c, then BGR or gray
__device__ __forceinline__ int jp_chroma(const uint8_t *pl, int pitch, int cw, int ch, int hs, int vs, int x, int y)
{
    if (hs == 1 && vs == 1) return pl[(size_t)y * pitch + x];
    const int i = x >> 1;
    if (vs == 1) {
        const uint8_t *in = pl + (size_t)y * pitch;
        if (cw <= 2) return in[i];  // (too narrow for the filter: libjpeg replicates)
        if (x & 1) return i == cw - 1 ? in[i] : (in[i] * 3 + in[i + 1] + 2) >> 2;
        return i == 0 ? in[0] : (in[i] * 3 + in[i - 1] + 1) >> 2;
    }
    const int cy = y >> 1;
    if (cw <= 2) return pl[(size_t)cy * pitch + i];
    int yf = (y & 1) ? cy + 1 : cy - 1;
    yf = yf < 0 ? 0 : (yf > ch - 1 ? ch - 1 : yf);
    const uint8_t *in0 = pl + (size_t)cy * pitch, *in1 = pl + (size_t)yf * pitch;
    const int cur = in0[i] * 3 + in1[i];
    if (x & 1) {
        if (i == cw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (in0[i + 1] * 3 + in1[i + 1]) + 7) >> 4;
    }
    if (i == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (in0[i - 1] * 3 + in1[i - 1]) + 8) >> 4;
}

// The 4:2:0 / 4:2:2 three-component picture whose planes are word-aligned (every picture this decoder lays out) and whose width is
// a multiple of 8: a lane takes EIGHT pixels of a row (x0 = 8 k).  Their chroma columns i0 .. i0 + 3 (i0 = 4 k) are one aligned
// word per row and plane; the two columns beside them come from the neighbour lanes' words (DPP-free: __shfl) when those lanes
// hold the same row, otherwise as single bytes.  Per eight pixels: one 8-byte luma load, four chroma word loads, one 8-byte gray
// store -- the four-pixel kernel below issues seventeen loads per four pixels and was bound by them (1.7 ms for 256 frames).
__global__ __launch_bounds__(256) void k_jpeg_color8(const JpImage *__restrict__ imgs, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out,
                                                     long long out_fstride, int gray_out)
{
    const JpImage &I = imgs[blockIdx.y];
    const int W = I.w, H = I.h, W8 = W >> 3;
    const uint8_t *P = planes + I.plane_base;
    uint8_t *dst = out + (long long)blockIdx.y * out_fstride;
    const int cw = (W + 1) >> 1, ch = (H + I.vs - 1) / I.vs;
    const int p0 = I.bw[0] * 8, p1 = I.bw[1] * 8;
    const unsigned total8 = (unsigned)W8 * (unsigned)H, step8 = gridDim.x * blockDim.x;
    const int lane = (int)(threadIdx.x & 63u);
    for (unsigned base = blockIdx.x * blockDim.x; base < total8; base += step8) {  // (workgroup-uniform trip count: the shuffles below want every lane)
        const unsigned idx = base + threadIdx.x;
        const bool live = idx < total8;
        const unsigned idc = live ? idx : total8 - 1u;
        const int y = (int)(idc / (unsigned)W8), k8 = (int)(idc - (unsigned)y * (unsigned)W8);
        const int x0 = k8 * 8, i0 = k8 * 4;
        const uint2 yw = *reinterpret_cast<const uint2 *>(P + I.plane_off[0] + (size_t)y * p0 + x0);
        const int cy = I.vs == 2 ? y >> 1 : y;
        int yf = cy;
        if (I.vs == 2) {
            yf = (y & 1) ? cy + 1 : cy - 1;
            yf = yf < 0 ? 0 : (yf > ch - 1 ? ch - 1 : yf);
        }
        // the lane to the left / right holds the eight pixels before / behind mine when it is in the wave and on my row
        const bool left_in = lane > 0 && k8 > 0, right_in = lane < 63 && k8 + 1 < W8 && idx + 1u < total8;
        int cbv[8], crv[8];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t *bp = P + I.plane_off[1 + pl];
            const uint8_t *in0 = bp + (size_t)cy * p1, *in1 = bp + (size_t)yf * p1;
            const uint32_t w0 = *reinterpret_cast<const uint32_t *>(in0 + i0);
            const uint32_t w1 = I.vs == 2 ? *reinterpret_cast<const uint32_t *>(in1 + i0) : 0u;
            int col[6];  // columns i0 - 1 .. i0 + 4: 3 * near row + far row (h2v2) or the row itself (h2v1)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a = (int)((w0 >> (8 * k)) & 0xffu), b = (int)((w1 >> (8 * k)) & 0xffu);
                col[1 + k] = I.vs == 2 ? a * 3 + b : a;
            }
            const int from_l = __shfl_up(col[4], 1, 64), from_r = __shfl_down(col[1], 1, 64);
            {
                const int il = i0 > 0 ? i0 - 1 : 0, ir = i0 + 4 < cw ? i0 + 4 : cw - 1;
                col[0] = left_in ? from_l : (I.vs == 2 ? in0[il] * 3 + in1[il] : in0[il]);
                col[5] = right_in ? from_r : (I.vs == 2 ? in0[ir] * 3 + in1[ir] : in0[ir]);
            }
            int *o = pl ? crv : cbv;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + (k >> 1);  // this pixel's chroma column; col[1 + (k >> 1)] is its sample
                const int cur = col[1 + (k >> 1)];
                int v;
                if (I.vs == 2) {
                    if (k & 1) v = i >= cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + col[2 + (k >> 1)] + 7) >> 4;
                    else v = i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + col[k >> 1] + 8) >> 4;
                } else {
                    if (k & 1) v = i >= cw - 1 ? cur : (cur * 3 + col[2 + (k >> 1)] + 2) >> 2;
                    else v = i == 0 ? cur : (cur * 3 + col[k >> 1] + 1) >> 2;
                }
                o[k] = v;
            }
        }
        uint32_t gw[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int Y = (int)(((k < 4 ? yw.x : yw.y) >> (8 * (k & 3))) & 0xffu);
            const int cb = cbv[k] - 128, cr = crv[k] - 128;
            int r = Y + ((91881 * cr + 32768) >> 16);
            int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
            int b = Y + ((116130 * cb + 32768) >> 16);
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            g = g < 0 ? 0 : (g > 255 ? 255 : g);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            if (gray_out) {
                gw[k >> 2] |= (uint32_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15) << (8 * (k & 3));  // K0: cvtColor(BGR2GRAY)
            } else if (live) {
                uint8_t *o = dst + ((long long)y * W + x0 + k) * 3;
                o[0] = (uint8_t)b;
                o[1] = (uint8_t)g;
                o[2] = (uint8_t)r;
            }
        }
        if (gray_out && live) *reinterpret_cast<uint2 *>(dst + (long long)y * W + x0) = make_uint2(gw[0], gw[1]);
    }
}

// a lane takes four pixels of a row (x = 4 k .. 4 k + 3): one division per four pixels, the chroma samples they share are
// loaded once, the gray result leaves as one 4-byte store
__global__ __launch_bounds__(256) void k_jpeg_color(const JpImage *__restrict__ imgs, const uint8_t *__restrict__ planes, uint8_t *__restrict__ out,
                                                    long long out_fstride, int gray_out)
{
    const JpImage &I = imgs[blockIdx.y];
    const int W = I.w, H = I.h, W4 = (W + 3) >> 2;
    const uint8_t *P = planes + I.plane_base;
    uint8_t *dst = out + (long long)blockIdx.y * out_fstride;
    const int cw = (W + I.hs - 1) / I.hs, ch = (H + I.vs - 1) / I.vs;
    const int p0 = I.bw[0] * 8, p1 = I.ncomp == 3 ? I.bw[1] * 8 : 0;
    const bool fancy = I.ncomp == 3 && I.hs == 2 && cw > 2;
    // (32-bit index arithmetic: W4 * H < 2^31 for every supported size; the 64-bit division this loop used to start with was ~90
    //  VALU instructions per four pixels, more than the colour arithmetic itself)
    const unsigned total4 = (unsigned)W4 * (unsigned)H, step4 = gridDim.x * blockDim.x;
    const bool y_aligned = ((I.plane_base + I.plane_off[0]) & 3) == 0 && (p0 & 3) == 0 && ((uintptr_t)planes & 3) == 0;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += step4) {
        const int y = (int)(idx / (unsigned)W4), x0 = (int)(idx - (unsigned)y * (unsigned)W4) * 4;
        const uint8_t *yr = P + I.plane_off[0] + (size_t)y * p0 + x0;  // (rows of the planes are multiples of 8 wide: four bytes are there)
        int Y[4], cbv[4], crv[4];
        if (y_aligned) {
            const uint32_t yw = *reinterpret_cast<const uint32_t *>(yr);
#pragma unroll
            for (int k = 0; k < 4; k++) Y[k] = (int)((yw >> (8 * k)) & 0xffu);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) Y[k] = yr[k];
        }
        if (I.ncomp == 3) {
            if (fancy) {
                // chroma columns i0 - 1 .. i0 + 2 serve the four pixels (x0 = 2 i0); column sums 3 * near row + far row for
                // h2v2, the row itself (times 4, to share the arithmetic below) for h2v1
                const int i0 = x0 >> 1;
                const int cy = I.vs == 2 ? y >> 1 : y;
                int yf = cy;
                if (I.vs == 2) {
                    yf = (y & 1) ? cy + 1 : cy - 1;
                    yf = yf < 0 ? 0 : (yf > ch - 1 ? ch - 1 : yf);
                }
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const uint8_t *base = P + I.plane_off[1 + pl];
                    const uint8_t *in0 = base + (size_t)cy * p1, *in1 = base + (size_t)yf * p1;
                    int col[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int i = i0 - 1 + k;
                        i = i < 0 ? 0 : (i > cw - 1 ? cw - 1 : i);
                        col[k] = I.vs == 2 ? in0[i] * 3 + in1[i] : in0[i];
                    }
                    int *o = pl ? crv : cbv;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int i = i0 + (k >> 1);  // this pixel's chroma column; col[1 + (k >> 1)] is its sample
                        const int cur = col[1 + (k >> 1)];
                        int v;
                        if (I.vs == 2) {
                            if (k & 1) v = i >= cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + col[2 + (k >> 1)] + 7) >> 4;
                            else v = i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + col[k >> 1] + 8) >> 4;
                        } else {
                            if (k & 1) v = i >= cw - 1 ? cur : (cur * 3 + col[2 + (k >> 1)] + 2) >> 2;
                            else v = i == 0 ? cur : (cur * 3 + col[k >> 1] + 1) >> 2;
                        }
                        o[k] = v;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = x0 + k < W ? x0 + k : W - 1;
                    cbv[k] = jp_chroma(P + I.plane_off[1], p1, cw, ch, I.hs, I.vs, x, y);
                    crv[k] = jp_chroma(P + I.plane_off[2], p1, cw, ch, I.hs, I.vs, x, y);
                }
            }
        }
        uint32_t gw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int r = Y[k], g = Y[k], b = Y[k];
            if (I.ncomp == 3) {
                const int cb = cbv[k] - 128, cr = crv[k] - 128;
                r = Y[k] + ((91881 * cr + 32768) >> 16);
                g = Y[k] + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                b = Y[k] + ((116130 * cb + 32768) >> 16);
                r = r < 0 ? 0 : (r > 255 ? 255 : r);
                g = g < 0 ? 0 : (g > 255 ? 255 : g);
                b = b < 0 ? 0 : (b > 255 ? 255 : b);
            }
            if (gray_out) {
                gw |= (uint32_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15) << (8 * k);  // K0: cvtColor(BGR2GRAY)
            } else if (x0 + k < W) {
                uint8_t *o = dst + ((long long)y * W + x0 + k) * 3;
                o[0] = (uint8_t)b;
                o[1] = (uint8_t)g;
                o[2] = (uint8_t)r;
            }
        }
        if (gray_out) {
            uint8_t *o = dst + (long long)y * W + x0;
            if (x0 + 4 <= W && (((long long)y * W + x0) & 3) == 0) {
                *reinterpret_cast<uint32_t *>(o) = gw;
            } else {
                for (int k = 0; k < 4 && x0 + k < W; k++) o[k] = (uint8_t)(gw >> (8 * k));
            }
        }
    }
}

// ---- host side: header parse
struct JpHuffSpec {
    uint8_t bits[17];
    uint8_t vals[256];
    int nvals = 0;
    bool present = false;
};
struct JpHeader {
    int w = 0, h = 0, ncomp = 0;
    int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
    uint16_t q[4][64];
    bool qpresent[4] = {false, false, false, false};
    JpHuffSpec dc[4], ac[4];
    int restart = 0;
    size_t scan_off = 0, scan_len = 0;
};
const uint8_t kJpZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

fid_status jp_parse(const uint8_t *d, size_t n, JpHeader *H, const char **why)
{
#define JP_FAIL(code, text) \
    do {                    \
        *why = text;        \
        return code;        \
    } while (0)
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) JP_FAIL(FID_E_INVALID_ARG, "not a JPEG file (no SOI)");
    size_t p = 2;
    bool have_sof = false, adobe_rgb = false;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) JP_FAIL(FID_E_INVALID_ARG, "marker expected");
        while (p < n && d[p] == 0xFF) p++;
        if (p >= n) break;
        const int m = d[p++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) JP_FAIL(FID_E_INVALID_ARG, "end of image before the scan");
        if (p + 2 > n) break;
        const size_t len = ((size_t)d[p] << 8) | d[p + 1];
        if (len < 2 || p + len > n) JP_FAIL(FID_E_INVALID_ARG, "truncated segment");
        const uint8_t *s = d + p + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) JP_FAIL(FID_E_INVALID_ARG, "short frame header");
            if (s[0] != 8) JP_FAIL(FID_E_UNSUPPORTED, "sample precision other than 8 bits");
            H->h = (s[1] << 8) | s[2];
            H->w = (s[3] << 8) | s[4];
            H->ncomp = s[5];
            if (H->ncomp != 1 && H->ncomp != 3) JP_FAIL(FID_E_UNSUPPORTED, "component count other than 1 or 3");
            if (H->w < 1 || H->h < 1 || sl < 6 + 3 * (size_t)H->ncomp) JP_FAIL(FID_E_INVALID_ARG, "bad frame header");
            for (int c = 0; c < H->ncomp; c++) {
                H->cid[c] = s[6 + 3 * c];
                H->hs[c] = s[7 + 3 * c] >> 4;
                H->vs[c] = s[7 + 3 * c] & 15;
                H->tq[c] = s[8 + 3 * c] & 3;
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            JP_FAIL(FID_E_UNSUPPORTED, "not a baseline file (progressive, lossless or arithmetic coding)");
        } else if (m == 0xEE) {
            // APP14 "Adobe": transform 0 with three components means the planes ARE R, G, B (jdapimin.c default_decompress_parms);
            // libjpeg / cv::imdecode then skip the YCbCr conversion that k_jpeg_color always applies -- refuse rather than
            // hand out another image
            if (sl >= 12 && !memcmp(s, "Adobe", 5) && s[11] == 0) adobe_rgb = true;
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                o++;
                if (tq > 3 || o + (pq ? 128u : 64u) > sl) JP_FAIL(FID_E_INVALID_ARG, "bad quantisation table");
                for (int k = 0; k < 64; k++) {
                    H->q[tq][kJpZigzag[k]] = (uint16_t)(pq ? ((s[o] << 8) | s[o + 1]) : s[o]);
                    o += pq ? 2 : 1;
                }
                H->qpresent[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) JP_FAIL(FID_E_INVALID_ARG, "bad Huffman table header");
                JpHuffSpec &t = tc ? H->ac[th] : H->dc[th];
                int cnt = 0;
                t.bits[0] = 0;
                for (int l = 1; l <= 16; l++) {
                    t.bits[l] = s[o + l];
                    cnt += s[o + l];
                }
                o += 17;
                if (cnt > 256 || o + cnt > sl) JP_FAIL(FID_E_INVALID_ARG, "bad Huffman table");
                memcpy(t.vals, s + o, (size_t)cnt);
                t.nvals = cnt;
                t.present = true;
                o += cnt;
            }
        } else if (m == 0xDD) {
            if (sl < 2) JP_FAIL(FID_E_INVALID_ARG, "bad restart interval");
            H->restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof) JP_FAIL(FID_E_INVALID_ARG, "scan before the frame header");
            if (sl < 1 || s[0] != H->ncomp || sl < 1 + 2 * (size_t)H->ncomp + 3) JP_FAIL(FID_E_UNSUPPORTED, "the components are not in one interleaved scan");
            for (int c = 0; c < H->ncomp; c++) {
                if (s[1 + 2 * c] != H->cid[c]) JP_FAIL(FID_E_UNSUPPORTED, "scan components not in frame order");
                H->td[c] = s[2 + 2 * c] >> 4;
                H->ta[c] = s[2 + 2 * c] & 15;
                if (H->td[c] > 3 || H->ta[c] > 3 || !H->dc[H->td[c]].present || !H->ac[H->ta[c]].present || !H->qpresent[H->tq[c]])
                    JP_FAIL(FID_E_INVALID_ARG, "a table the scan names is missing");
            }
            const uint8_t *e = s + 1 + 2 * H->ncomp;
            if (e[0] != 0 || e[1] != 63 || e[2] != 0) JP_FAIL(FID_E_UNSUPPORTED, "spectral selection / successive approximation");
            H->scan_off = p + len;
            H->scan_len = n - H->scan_off;
            break;
        }
        p += len;
    }
    if (!H->scan_off) JP_FAIL(FID_E_INVALID_ARG, "no scan");
    if (H->ncomp == 3 && (adobe_rgb || (H->cid[0] == 'R' && H->cid[1] == 'G' && H->cid[2] == 'B')))
        JP_FAIL(FID_E_UNSUPPORTED, "three-component file that is not YCbCr (Adobe transform 0 / component ids R, G, B)");
    if (H->ncomp == 1) {
        H->hs[0] = H->vs[0] = 1;  // a one-component scan is not interleaved: 8 x 8 MCUs whatever the factors say
    } else {
        if (H->hs[1] != 1 || H->vs[1] != 1 || H->hs[2] != 1 || H->vs[2] != 1) JP_FAIL(FID_E_UNSUPPORTED, "chroma sampling factors other than 1 x 1");
        if (!((H->hs[0] == 1 && H->vs[0] == 1) || (H->hs[0] == 2 && H->vs[0] == 1) || (H->hs[0] == 2 && H->vs[0] == 2)))
            JP_FAIL(FID_E_UNSUPPORTED, "luma sampling other than 1x1, 2x1, 2x2");
    }
    return FID_OK;
#undef JP_FAIL
}

// 16-bit code table: entry = code length << 8 | symbol for every 16-bit window that starts with a code; 0 where none does
bool jp_build_lut(const JpHuffSpec &t, uint16_t *lut)
{
    memset(lut, 0, 65536 * sizeof(uint16_t));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < t.bits[l]; i++) {
            if (code >= (1u << l) || k >= t.nvals) return false;
            const unsigned lo = code << (16 - l), hi = lo + (1u << (16 - l));
            const uint16_t e = (uint16_t)((l << 8) | t.vals[k]);
            for (unsigned w = lo; w < hi; w++) lut[w] = e;
            code++;
            k++;
        }
        code <<= 1;
    }
    return true;
}

unsigned long long jp_hash(const JpHuffSpec &t)
{
    unsigned long long h = 1469598103934665603ull;
    for (int l = 1; l <= 16; l++) h = (h ^ t.bits[l]) * 1099511628211ull;
    for (int i = 0; i < t.nvals; i++) h = (h ^ t.vals[i]) * 1099511628211ull;
    return h ^ ((unsigned long long)t.nvals << 56);
}

}  // namespace

struct fid_jpeg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int maxW = 0, maxH = 0, maxB = 0;
    size_t max_scan = 0, max_sub = 0, max_blocks = 0;  // per image
    // device
    uint8_t *d_scan = nullptr, *d_planes = nullptr, *d_out = nullptr;
    int16_t *d_coefs = nullptr;
    bool keep_coefs = false;   // FID_JPEG_KEEP_COEFS=1 at fid_jpeg_create: the coefficients of a call stay readable (FID_JPEG_TAP_COEFS)
    bool coefs_clean = false;  // d_coefs is all zeros
    JpImage *d_imgs = nullptr;
    uint16_t *d_luts = nullptr, *d_lut1 = nullptr;  // 16-bit code tables; their first-level (JP_LOOK bit) extracts, contiguous
    JpState *d_state[2] = {nullptr, nullptr};
    uint8_t *d_chg[2] = {nullptr, nullptr};
    int4 *d_nblk = nullptr, *d_blkbase = nullptr;  // per sub-sequence: blocks completed | met a restart marker, DC sums / what it starts from
    unsigned *d_flag = nullptr;
    // pinned host
    uint8_t *h_scan = nullptr;
    JpImage *h_imgs = nullptr;
    unsigned *h_flag = nullptr;
    uint16_t *h_lut = nullptr;
    std::unordered_map<unsigned long long, int> lut_slot;
    std::vector<JpHuffSpec> lut_spec;  // the table behind every slot: a hash hit is only a hit if the table is the same
    int lut_next = 0, lut_cap = 0;  // cached 16-bit code tables: room for four new ones per frame of a call and 16 more
    // the last decode
    int last_n = 0, last_w = 0, last_h = 0, last_enc = 0, last_rounds = 0;
    std::vector<JpImage> last_imgs;
    std::string last_error;
};

namespace {
#define JPCHK(ctx, expr)                                                               \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);     \
            return e_ == hipErrorOutOfMemory ? FID_E_OUT_OF_MEMORY : FID_E_HIP;        \
        }                                                                              \
    } while (0)

int jp_lut_slot(fid_jpeg_ctx *c, const JpHuffSpec &t, fid_status *rc)
{
    const unsigned long long h = jp_hash(t);
    auto it = c->lut_slot.find(h);
    if (it != c->lut_slot.end()) {
        const JpHuffSpec &o = c->lut_spec[(size_t)it->second];
        if (o.nvals == t.nvals && !memcmp(o.bits, t.bits, sizeof o.bits) && !memcmp(o.vals, t.vals, (size_t)t.nvals)) return it->second;
        // (a 64-bit hash collision -- constructible on a network topic: not the same table, so not the same slot)
        *rc = FID_E_UNSUPPORTED;
        c->last_error = "two different Huffman tables with the same hash in one call";
        return 0;
    }
    if (c->lut_next >= c->lut_cap) {
        *rc = FID_E_UNSUPPORTED;
        c->last_error = "more distinct Huffman tables in one call than the table cache holds";
        return 0;
    }
    if (!jp_build_lut(t, c->h_lut)) {
        *rc = FID_E_INVALID_ARG;
        c->last_error = "inconsistent Huffman table";
        return 0;
    }
    const int slot = c->lut_next++;
    // (synchronous: the pinned staging table is reused for the next one)
    uint16_t first_level[1 << JP_LOOK];
    for (int q = 0; q < (1 << JP_LOOK); q++) {
        const uint16_t e = c->h_lut[(size_t)q << (16 - JP_LOOK)];
        first_level[q] = (e >> 8) <= JP_LOOK ? e : (uint16_t)0;
    }
    if (hipMemcpy(c->d_lut1 + (size_t)slot * (1 << JP_LOOK), first_level, sizeof(first_level), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_luts + (size_t)slot * 65536, c->h_lut, 65536 * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) {
        *rc = FID_E_HIP;
        c->last_error = "upload of a code table failed";
        return 0;
    }
    c->lut_slot[h] = slot;
    if (c->lut_spec.size() <= (size_t)slot) c->lut_spec.resize((size_t)slot + 1);
    c->lut_spec[(size_t)slot] = t;
    return slot;
}
}  // namespace

extern "C" {

fid_status fid_jpeg_probe(const uint8_t *data, int64_t nbytes, fid_jpeg_info *info)
{
    if (!data || nbytes <= 0 || !info) return FID_E_INVALID_ARG;
    JpHeader H;
    const char *why = "";
    const fid_status rc = jp_parse(data, (size_t)nbytes, &H, &why);
    if (rc != FID_OK) return rc;
    memset(info, 0, sizeof(*info));
    info->width = H.w;
    info->height = H.h;
    info->components = H.ncomp;
    info->h_samp = H.hs[0];
    info->v_samp = H.vs[0];
    info->restart_interval = H.restart;
    const int mcux = (H.w + 8 * H.hs[0] - 1) / (8 * H.hs[0]), mcuy = (H.h + 8 * H.vs[0] - 1) / (8 * H.vs[0]);
    for (int c = 0; c < H.ncomp; c++) {
        info->blocks_w[c] = mcux * (c == 0 ? H.hs[0] : 1);
        info->blocks_h[c] = mcuy * (c == 0 ? H.vs[0] : 1);
    }
    info->scan_bytes = (int64_t)H.scan_len;
    return FID_OK;
}

fid_status fid_jpeg_create(int32_t device, int32_t max_width, int32_t max_height, int32_t max_batch, fid_jpeg_ctx **out)
{
    if (!out || max_width < 1 || max_height < 1 || max_batch < 1 || max_width > 16384 || max_height > 16384) return FID_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FID_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return FID_E_INVALID_ARG;
    fid_jpeg_ctx *c = new (std::nothrow) fid_jpeg_ctx();
    if (!c) return FID_E_OUT_OF_MEMORY;
    c->device = device;
    c->maxW = max_width;
    c->maxH = max_height;
    c->maxB = max_batch;
    c->lut_cap = 16 + 4 * max_batch;
    const size_t F = (size_t)max_batch;
    const size_t mw = ((size_t)max_width + 15) / 16 * 16, mh = ((size_t)max_height + 15) / 16 * 16;
    c->max_blocks = mw * mh / 64 * 3;               // 4:4:4 is the largest
    c->max_scan = mw * mh * 2 + 65536;              // entropy-coded bytes per image this context takes (MCU-padded size: narrow images too)
    c->max_sub = (c->max_scan + JP_SUB - 1) / JP_SUB;
    // the decoder's kernels are few waves that wait for memory: beside a detector that fills the chip (a stream of compressed batches)
    // they go first (FID_JPEG_PRIO=0: a stream of the default priority)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char *pe = getenv("FID_JPEG_PRIO");
    const int prio = pe && atoi(pe) == 0 ? 0 : prio_hi;
    c->keep_coefs = getenv("FID_JPEG_KEEP_COEFS") != nullptr;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_scan, F * c->max_scan + 64) == hipSuccess && hipMalloc((void **)&c->d_coefs, F * c->max_blocks * 64 * sizeof(int16_t)) == hipSuccess &&
         hipMalloc((void **)&c->d_planes, F * c->max_blocks * 64) == hipSuccess && hipMalloc((void **)&c->d_out, F * (size_t)max_width * max_height * 3) == hipSuccess &&
         hipMalloc((void **)&c->d_imgs, F * sizeof(JpImage)) == hipSuccess &&
         hipMalloc((void **)&c->d_luts, (size_t)c->lut_cap * 65536 * sizeof(uint16_t)) == hipSuccess &&
         hipMalloc((void **)&c->d_lut1, (size_t)c->lut_cap * (1 << JP_LOOK) * sizeof(uint16_t)) == hipSuccess &&
         hipMalloc((void **)&c->d_state[0], F * c->max_sub * sizeof(JpState)) == hipSuccess && hipMalloc((void **)&c->d_state[1], F * c->max_sub * sizeof(JpState)) == hipSuccess &&
         hipMalloc((void **)&c->d_chg[0], F * c->max_sub) == hipSuccess && hipMalloc((void **)&c->d_chg[1], F * c->max_sub) == hipSuccess &&
         hipMalloc((void **)&c->d_nblk, F * c->max_sub * sizeof(int4)) == hipSuccess && hipMalloc((void **)&c->d_blkbase, F * c->max_sub * sizeof(int4)) == hipSuccess &&
         hipMalloc((void **)&c->d_flag, 8) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_scan, F * c->max_scan + 64) == hipSuccess && hipHostMalloc((void **)&c->h_imgs, F * sizeof(JpImage)) == hipSuccess &&
         hipHostMalloc((void **)&c->h_flag, 8) == hipSuccess && hipHostMalloc((void **)&c->h_lut, 65536 * sizeof(uint16_t)) == hipSuccess;
    if (!ok) {
        fid_jpeg_destroy(c);
        return FID_E_OUT_OF_MEMORY;
    }
    *out = c;
    return FID_OK;
}

void fid_jpeg_destroy(fid_jpeg_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_scan, c->d_coefs, c->d_planes, c->d_out, c->d_imgs, c->d_luts, c->d_lut1, c->d_state[0], c->d_state[1], c->d_chg[0], c->d_chg[1],
                   c->d_nblk, c->d_blkbase, c->d_flag};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    void *host[] = {c->h_scan, c->h_imgs, c->h_flag, c->h_lut};
    for (void *p : host)
        if (p) (void)hipHostFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

fid_status fid_jpeg_decode(fid_jpeg_ctx *c, const uint8_t *const *files, const int64_t *nbytes, int32_t n, fid_encoding out_enc, uint8_t *host_out,
                           int64_t host_frame_stride)
{
    if (!c || !files || !nbytes || n < 1 || n > c->maxB) return FID_E_INVALID_ARG;
    if (out_enc != FID_ENC_BGR8 && out_enc != FID_ENC_MONO8) return FID_E_INVALID_ARG;
    JPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->last_n = 0;
    static const bool jp_timing = getenv("FID_JPEG_TIMING") != nullptr;  // host-side phases of a call to stderr
    const auto jp_t0 = std::chrono::steady_clock::now();
    double jp_ms[5] = {0, 0, 0, 0, 0};
    auto jp_mark = [&](int k) {
        if (jp_timing) jp_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jp_t0).count();
    };
    if (c->lut_next + 4 * n > c->lut_cap) {  // a stream of files with ever new tables: start the cache again (between calls only)
        c->lut_slot.clear();
        c->lut_next = 0;
    }
    // ---- headers, tables, packing of the entropy-coded bytes
    size_t scan_at = 0, sub_at = 0;
    std::vector<const uint8_t *> pack_src((size_t)n);
    uint32_t max_nsub = 0, max_blocks = 0;
    int W = 0, Hh = 0;
    for (int f = 0; f < n; f++) {
        JpHeader H;
        const char *why = "";
        fid_status rc = (files[f] && nbytes[f] > 0) ? jp_parse(files[f], (size_t)nbytes[f], &H, &why) : FID_E_INVALID_ARG;
        if (rc != FID_OK) {
            c->last_error = std::string("frame ") + std::to_string(f) + ": " + why;
            return rc;
        }
        if (f == 0) {
            W = H.w;
            Hh = H.h;
        }
        if (H.w != W || H.h != Hh) {
            c->last_error = "the files of one call must have one image size";
            return FID_E_INVALID_ARG;
        }
        if (W > c->maxW || Hh > c->maxH) {
            c->last_error = "image larger than the context was created for";
            return FID_E_INVALID_ARG;
        }
        if (H.scan_len > c->max_scan || H.scan_len >= (1ull << 28)) {
            c->last_error = "entropy-coded data larger than the context takes";
            return FID_E_CAPACITY;
        }
        JpImage &I = c->h_imgs[f];
        memset(&I, 0, sizeof(I));
        I.w = W;
        I.h = Hh;
        I.ncomp = H.ncomp;
        I.hs = H.hs[0];
        I.vs = H.vs[0];
        I.bpm = H.ncomp == 1 ? 1 : I.hs * I.vs + 2;
        I.mcux = (W + 8 * I.hs - 1) / (8 * I.hs);
        const int mcuy = (Hh + 8 * I.vs - 1) / (8 * I.vs);
        I.nmcu = I.mcux * mcuy;
        I.restart = H.restart;
        uint32_t co = 0, po = 0;
        for (int k = 0; k < H.ncomp; k++) {
            I.bw[k] = I.mcux * (k == 0 ? I.hs : 1);
            I.bh[k] = mcuy * (k == 0 ? I.vs : 1);
            I.coef_off[k] = co;
            I.plane_off[k] = po;
            co += (uint32_t)I.bw[k] * I.bh[k] * 64u;
            po += (uint32_t)I.bw[k] * I.bh[k] * 64u;
            memcpy(I.q[k], H.q[H.tq[k]], sizeof(I.q[k]));
            I.lut_dc[k] = jp_lut_slot(c, H.dc[H.td[k]], &rc);
            I.lut_ac[k] = jp_lut_slot(c, H.ac[H.ta[k]], &rc);
            if (rc != FID_OK) return rc;
        }
        I.nblocks = (uint32_t)I.nmcu * (uint32_t)I.bpm;
        I.coef_base = (unsigned long long)f * c->max_blocks * 64;
        I.plane_base = (unsigned long long)f * c->max_blocks * 64;
        I.scan_off = scan_at;
        I.scan_len = (uint32_t)H.scan_len;
        I.sub_base = (uint32_t)sub_at;
        I.nsub = (uint32_t)((H.scan_len + JP_SUB - 1) / JP_SUB);
        if (I.nsub == 0) {
            c->last_error = "empty scan";
            return FID_E_INVALID_ARG;
        }
        pack_src[(size_t)f] = files[f] + H.scan_off;
        scan_at = (scan_at + H.scan_len + 15) & ~(size_t)15;  // (every image's bytes start on a 16-byte boundary)
        sub_at += I.nsub;
        max_nsub = I.nsub > max_nsub ? I.nsub : max_nsub;
        max_blocks = co / 64 > max_blocks ? co / 64 : max_blocks;
    }
    jp_mark(0);
    // the entropy-coded bytes into pinned memory (a few host threads for a large batch: one thread moves ~10 GB/s), then one copy
    {
        const int nt = n >= 32 ? 8 : (n >= 8 ? 4 : 1);
        auto pack = [&](int t) {
            for (int f = t; f < n; f += nt) memcpy(c->h_scan + c->h_imgs[f].scan_off, pack_src[(size_t)f], c->h_imgs[f].scan_len);
        };
        if (nt == 1) {
            pack(0);
        } else {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(pack, t);
            pack(0);
            for (auto &x : th) x.join();
        }
    }
    jp_mark(1);
    JPCHK(c, hipMemcpyAsync(c->d_scan, c->h_scan, scan_at, hipMemcpyHostToDevice, st));
    JPCHK(c, hipMemcpyAsync(c->d_imgs, c->h_imgs, (size_t)n * sizeof(JpImage), hipMemcpyHostToDevice, st));
    // the coefficient array: all zeros when a call begins -- k_jpeg_idct leaves it so (it zeroes what it has read); filled here only
    // when the tap wants the coefficients kept (FID_JPEG_KEEP_COEFS) or a call ended between the entropy decoder and the IDCT
    if (c->keep_coefs) JPCHK(c, hipMemsetAsync(c->d_coefs, 0, (size_t)n * c->max_blocks * 64 * sizeof(int16_t), st));
    else if (!c->coefs_clean) JPCHK(c, hipMemsetAsync(c->d_coefs, 0, (size_t)c->maxB * c->max_blocks * 64 * sizeof(int16_t), st));
    c->coefs_clean = false;
    // ---- J1: entropy decoding
    const dim3 hgrid((max_nsub + JP_OWN - 1) / JP_OWN, n);
    hipLaunchKernelGGL(k_jpeg_huff<0>, hgrid, dim3(JP_TPB), 0, st, c->d_imgs, c->d_scan, c->d_luts, c->d_lut1, (const JpState *)nullptr, c->d_state[0], (const uint8_t *)nullptr,
                       c->d_chg[0], c->d_nblk, (const int4 *)nullptr, (int16_t *)nullptr, c->d_flag);
    int cur = 0, rounds = 0;
    for (;;) {
        // two rounds per look at the flag (a look costs a host round trip; a round in which nothing changes costs ~10 us)
        JPCHK(c, hipMemsetAsync(c->d_flag, 0, 8, st));
        for (int k = 0; k < 2; k++) {
            hipLaunchKernelGGL(k_jpeg_huff<1>, hgrid, dim3(JP_TPB), 0, st, c->d_imgs, c->d_scan, c->d_luts, c->d_lut1, c->d_state[cur], c->d_state[cur ^ 1], c->d_chg[cur],
                               c->d_chg[cur ^ 1], c->d_nblk, (const int4 *)nullptr, (int16_t *)nullptr, c->d_flag + k);
            cur ^= 1;
            rounds++;
        }
        JPCHK(c, hipMemcpyAsync(c->h_flag, c->d_flag, 8, hipMemcpyDeviceToHost, st));
        JPCHK(c, hipStreamSynchronize(st));
        if (!c->h_flag[1]) {  // the second round of the pair changed nothing: settled
            if (!c->h_flag[0]) rounds--;  // (so did the first)
            break;
        }
        if (rounds > (int)max_nsub + 4) {  // (cannot happen: exactness spreads one sub-sequence per round at least)
            c->last_error = "entropy decoding did not settle";
            return FID_E_HIP;
        }
    }
    c->last_rounds = rounds;
    jp_mark(2);
    hipLaunchKernelGGL(k_jpeg_scan_blocks, dim3(n), dim3(1024), 0, st, c->d_imgs, c->d_nblk, c->d_blkbase);
    hipLaunchKernelGGL(k_jpeg_huff<2>, hgrid, dim3(JP_TPB), 0, st, c->d_imgs, c->d_scan, c->d_luts, c->d_lut1, c->d_state[cur], (JpState *)nullptr, (const uint8_t *)nullptr,
                       (uint8_t *)nullptr, (int4 *)nullptr, c->d_blkbase, c->d_coefs, c->d_flag);
    // ---- J2 .. J4
    hipLaunchKernelGGL(k_jpeg_idct, dim3((max_blocks + 31) / 32, n), dim3(256), 0, st, c->d_imgs, c->d_coefs, c->d_planes, c->keep_coefs ? 0 : 1);
    JPCHK(c, hipGetLastError());
    c->coefs_clean = !c->keep_coefs;
    const int bpp = out_enc == FID_ENC_MONO8 ? 1 : 3;
    const long long fstride = (long long)W * Hh * bpp;
    {
        long long px = (long long)((W + 3) / 4) * Hh;  // four pixels per lane
        int blocks = (int)((px + 255) / 256);
        blocks = blocks > 4096 ? 4096 : blocks;
        // (every picture of a call has one size; the eight-pixel kernel wants three components, 2:1 horizontal subsampling in every
        //  picture, a width that is a multiple of 8 and more than two chroma columns -- what a camera's JPEG stream is)
        bool wide = (W & 7) == 0 && W >= 16 && ((uintptr_t)c->d_out & 7) == 0 && (fstride & 7) == 0 && !getenv("FID_JPEG_COLOR4");
        for (int f = 0; f < n && wide; f++) wide = c->h_imgs[f].ncomp == 3 && c->h_imgs[f].hs == 2;
        if (wide) {
            const long long px8 = (long long)(W / 8) * Hh;
            int b8 = (int)((px8 + 255) / 256);
            b8 = b8 > 4096 ? 4096 : b8;
            hipLaunchKernelGGL(k_jpeg_color8, dim3(b8, n), dim3(256), 0, st, c->d_imgs, c->d_planes, c->d_out, fstride, out_enc == FID_ENC_MONO8 ? 1 : 0);
        } else {
            hipLaunchKernelGGL(k_jpeg_color, dim3(blocks, n), dim3(256), 0, st, c->d_imgs, c->d_planes, c->d_out, fstride, out_enc == FID_ENC_MONO8 ? 1 : 0);
        }
    }
    JPCHK(c, hipGetLastError());
    if (host_out) {
        if (host_frame_stride < fstride) return FID_E_INVALID_ARG;
        if (host_frame_stride == fstride) {
            JPCHK(c, hipMemcpyAsync(host_out, c->d_out, (size_t)fstride * n, hipMemcpyDeviceToHost, st));
        } else {
            for (int f = 0; f < n; f++)
                JPCHK(c, hipMemcpyAsync(host_out + (size_t)f * host_frame_stride, c->d_out + (size_t)f * fstride, (size_t)fstride, hipMemcpyDeviceToHost, st));
        }
    }
    JPCHK(c, hipStreamSynchronize(st));
    jp_mark(3);
    if (jp_timing)
        fprintf(stderr, "fid_jpeg_decode n=%d: headers %.2f ms, pack %.2f, copy + entropy rounds %.2f, coefficients + idct + colour %.2f\n", n, jp_ms[0],
                jp_ms[1] - jp_ms[0], jp_ms[2] - jp_ms[1], jp_ms[3] - jp_ms[2]);
    c->last_n = n;
    c->last_w = W;
    c->last_h = Hh;
    c->last_enc = (int)out_enc;
    c->last_imgs.assign(c->h_imgs, c->h_imgs + n);
    return FID_OK;
}

const void *fid_jpeg_device_ptr(fid_jpeg_ctx *c, int32_t *width, int32_t *height, int32_t *stride_bytes, int64_t *frame_stride_bytes)
{
    if (!c || c->last_n <= 0) return nullptr;
    const int bpp = c->last_enc == FID_ENC_MONO8 ? 1 : 3;
    if (width) *width = c->last_w;
    if (height) *height = c->last_h;
    if (stride_bytes) *stride_bytes = c->last_w * bpp;
    if (frame_stride_bytes) *frame_stride_bytes = (int64_t)c->last_w * c->last_h * bpp;
    return c->d_out;
}

int64_t fid_jpeg_tap_bytes(fid_jpeg_ctx *c, fid_jpeg_tap which, int32_t frame)
{
    if (!c || frame < 0 || frame >= c->last_n) return 0;
    const JpImage &I = c->last_imgs[(size_t)frame];
    int64_t nb = 0;
    for (int k = 0; k < I.ncomp; k++) nb += (int64_t)I.bw[k] * I.bh[k];
    return which == FID_JPEG_TAP_COEFS ? nb * 64 * 2 : (which == FID_JPEG_TAP_PLANES ? nb * 64 : 0);
}

fid_status fid_jpeg_tap_read(fid_jpeg_ctx *c, fid_jpeg_tap which, int32_t frame, void *dst, int64_t dst_bytes)
{
    const int64_t need = fid_jpeg_tap_bytes(c, which, frame);
    if (!c || !dst || need <= 0 || dst_bytes < need) return FID_E_INVALID_ARG;
    JPCHK(c, hipSetDevice(c->device));
    const JpImage &I = c->last_imgs[(size_t)frame];
    if (which == FID_JPEG_TAP_COEFS && !c->keep_coefs) {
        c->last_error = "the coefficients are consumed by the IDCT: create the context with FID_JPEG_KEEP_COEFS=1 in the environment to read them";
        return FID_E_UNSUPPORTED;
    }
    const void *src = which == FID_JPEG_TAP_COEFS ? (const void *)(c->d_coefs + I.coef_base) : (const void *)(c->d_planes + I.plane_base);
    JPCHK(c, hipMemcpy(dst, src, (size_t)need, hipMemcpyDeviceToHost));
    return FID_OK;
}

int32_t fid_jpeg_last_rounds(fid_jpeg_ctx *c) { return c ? c->last_rounds : 0; }
const char *fid_jpeg_last_error(fid_jpeg_ctx *c) { return c ? c->last_error.c_str() : "null context"; }

}  // extern "C"
