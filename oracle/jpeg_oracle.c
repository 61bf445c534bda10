/* jpeg_oracle.c -- TEST INFRASTRUCTURE (never linked into the product): plain-C restatement of the baseline JPEG decode that
 * stands in front of the node when `image_transport` runs compressed (aruco_detect.launch:6 `transport` default "compressed";
 * the subscriber plugin calls cv::imdecode, i.e. libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy upsampling, JFIF
 * YCbCr -> RGB).  libjpeg is a third-party dependency that is not in /root/reference; what is restated here is its published
 * algorithm (jdhuff.c, jidctint.c, jdsample.c h2v1/h2v2 fancy, jdcolor.c, jdmainct.c edge rows).  PINNED: the decoder is
 * checked bit for bit against libjpeg-turbo itself (Pillow in the authoring container) on the reference's own JPEG fixtures
 * (fiducial_slam/test/test_images/403.jpg, the CompressedImage frames of fiducial_slam/test/aruco_images.bag) and on generated
 * files of every supported layout (tests/test_oracle_jpeg.py, fixtures under tests/golden/jpeg/).
 *
 * Supported: baseline sequential DCT (SOF0 / SOF1 with 8-bit samples, Huffman), 1 or 3 components in one interleaved scan, luma
 * sampling 1x1, 2x1 or 2x2 with 1x1 chroma, restart intervals.  Everything else returns a negative status.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JO_OK 0
#define JO_E_FORMAT -1       /* not a JPEG / truncated / corrupt */
#define JO_E_UNSUPPORTED -2  /* progressive, arithmetic, 12-bit, CMYK, other sampling layouts, multiple scans */

typedef struct {
    uint8_t bits[17];
    uint8_t vals[256];
    int present;
    /* decoding by code length: mincode / maxcode / valptr as in jdhuff.c jpeg_make_d_derived_tbl */
    int32_t maxcode[18];
    int32_t valoffset[17];
} jo_huff;

typedef struct {
    int w, h, ncomp;
    int hs[3], vs[3], tq[3], td[3], ta[3], cid[3];
    uint16_t q[4][64]; /* natural order */
    int qpresent[4];
    jo_huff dc[4], ac[4];
    int restart;
    const uint8_t *scan;
    size_t scan_len;
    int hmax, vmax, mcux, mcuy;
    int bw[3], bh[3]; /* blocks per component row / column (MCU padded) */
} jo_hdr;

static const uint8_t jo_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int jo_build(jo_huff *t)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valoffset[l] = k - code;
        k += t->bits[l];
        code += t->bits[l];
        t->maxcode[l] = t->bits[l] ? code - 1 : -1;
        if (code > (1 << l)) return JO_E_FORMAT;
        code <<= 1;
    }
    t->maxcode[17] = 0xfffff;
    return k <= 256 ? JO_OK : JO_E_FORMAT;
}

static int jo_parse(const uint8_t *d, size_t n, jo_hdr *H)
{
    memset(H, 0, sizeof(*H));
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return JO_E_FORMAT;
    size_t p = 2;
    int have_sof = 0;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return JO_E_FORMAT;
        while (p < n && d[p] == 0xFF) p++;
        if (p >= n) return JO_E_FORMAT;
        const int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return JO_E_FORMAT;
        if (p + 2 > n) return JO_E_FORMAT;
        const size_t len = ((size_t)d[p] << 8) | d[p + 1];
        if (len < 2 || p + len > n) return JO_E_FORMAT;
        const uint8_t *s = d + p + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return JO_E_UNSUPPORTED;
            H->h = (s[1] << 8) | s[2];
            H->w = (s[3] << 8) | s[4];
            H->ncomp = s[5];
            if ((H->ncomp != 1 && H->ncomp != 3) || H->w < 1 || H->h < 1 || sl < 6 + 3 * (size_t)H->ncomp) return JO_E_UNSUPPORTED;
            for (int c = 0; c < H->ncomp; c++) {
                H->cid[c] = s[6 + 3 * c];
                H->hs[c] = s[7 + 3 * c] >> 4;
                H->vs[c] = s[7 + 3 * c] & 15;
                H->tq[c] = s[8 + 3 * c] & 3;
            }
            have_sof = 1;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return JO_E_UNSUPPORTED;
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3) return JO_E_FORMAT;
                o++;
                if (o + (pq ? 128 : 64) > sl) return JO_E_FORMAT;
                for (int k = 0; k < 64; k++) {
                    const int v = pq ? ((s[o] << 8) | s[o + 1]) : s[o];
                    o += pq ? 2 : 1;
                    H->q[tq][jo_zigzag[k]] = (uint16_t)v;
                }
                H->qpresent[tq] = 1;
            }
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) return JO_E_FORMAT;
                jo_huff *t = tc ? &H->ac[th] : &H->dc[th];
                int cnt = 0;
                t->bits[0] = 0;
                for (int l = 1; l <= 16; l++) {
                    t->bits[l] = s[o + l];
                    cnt += s[o + l];
                }
                o += 17;
                if (cnt > 256 || o + cnt > sl) return JO_E_FORMAT;
                memcpy(t->vals, s + o, (size_t)cnt);
                o += cnt;
                t->present = 1;
                if (jo_build(t) != JO_OK) return JO_E_FORMAT;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return JO_E_FORMAT;
            H->restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1 || s[0] != H->ncomp || sl < 1 + 2 * (size_t)H->ncomp + 3) return JO_E_UNSUPPORTED;
            for (int c = 0; c < H->ncomp; c++) {
                int ci = -1;
                for (int k = 0; k < H->ncomp; k++)
                    if (H->cid[k] == s[1 + 2 * c]) ci = k;
                if (ci != c) return JO_E_UNSUPPORTED; /* components in frame order */
                H->td[c] = s[2 + 2 * c] >> 4;
                H->ta[c] = s[2 + 2 * c] & 15;
                if (H->td[c] > 3 || H->ta[c] > 3 || !H->dc[H->td[c]].present || !H->ac[H->ta[c]].present || !H->qpresent[H->tq[c]]) return JO_E_FORMAT;
            }
            const uint8_t *e = s + 1 + 2 * H->ncomp;
            if (e[0] != 0 || e[1] != 63 || e[2] != 0) return JO_E_UNSUPPORTED;
            H->scan = d + p + len;
            H->scan_len = n - (p + len);
            break;
        }
        p += len;
    }
    if (!H->scan) return JO_E_FORMAT;
    if (H->ncomp == 1) {
        H->hs[0] = H->vs[0] = 1; /* a single-component scan is never interleaved: 8 x 8 MCUs whatever the factors say */
    } else {
        if (H->hs[1] != 1 || H->vs[1] != 1 || H->hs[2] != 1 || H->vs[2] != 1) return JO_E_UNSUPPORTED;
        if (!((H->hs[0] == 1 && H->vs[0] == 1) || (H->hs[0] == 2 && H->vs[0] == 1) || (H->hs[0] == 2 && H->vs[0] == 2))) return JO_E_UNSUPPORTED;
    }
    H->hmax = H->hs[0];
    H->vmax = H->vs[0];
    H->mcux = (H->w + 8 * H->hmax - 1) / (8 * H->hmax);
    H->mcuy = (H->h + 8 * H->vmax - 1) / (8 * H->vmax);
    for (int c = 0; c < H->ncomp; c++) {
        H->bw[c] = H->mcux * H->hs[c];
        H->bh[c] = H->mcuy * H->vs[c];
    }
    return JO_OK;
}

/* ---- entropy decoding (jdhuff.c): MSB-first bit reader over the scan with FF00 unstuffing; markers end the data */
typedef struct {
    const uint8_t *p, *end;
    uint32_t acc;
    int nbits;
    int marker; /* a marker was met: further bits read as zero */
} jo_bits;

static void jo_fill(jo_bits *b)
{
    while (b->nbits <= 24) {
        int c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xFF) {
                if (b->p < b->end && *b->p == 0) {
                    b->p++;
                } else {
                    b->marker = 1;
                    b->p--;
                    c = 0;
                }
            }
        } else {
            b->marker = 1;
        }
        b->acc |= (uint32_t)c << (24 - b->nbits);
        b->nbits += 8;
    }
}
static int jo_get(jo_bits *b, int n)
{
    if (n == 0) return 0;
    if (b->nbits < n) jo_fill(b);
    const int v = (int)(b->acc >> (32 - n));
    b->acc <<= n;
    b->nbits -= n;
    return v;
}
static int jo_sym(jo_bits *b, const jo_huff *t)
{
    if (b->nbits < 16) jo_fill(b);
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (int)(b->acc >> (32 - l));
        if (code <= t->maxcode[l]) {
            b->acc <<= l;
            b->nbits -= l;
            return t->vals[(code + t->valoffset[l]) & 255];
        }
    }
    b->acc <<= 16;
    b->nbits -= 16;
    return 0; /* (corrupt data) */
}
static int jo_extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

/* coefs: per component a [bh][bw][64] array of QUANTISED coefficients in natural order, DC prediction undone */
static int jo_entropy(const jo_hdr *H, int16_t *coef[3])
{
    jo_bits b = {H->scan, H->scan + H->scan_len, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    int todo = H->restart;
    for (int my = 0; my < H->mcuy; my++)
        for (int mx = 0; mx < H->mcux; mx++) {
            if (H->restart && todo == 0) {
                /* byte-align, expect RSTn */
                b.acc = 0;
                b.nbits = 0;
                if (b.marker) {
                    if (b.p + 1 < b.end && b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7) b.p += 2;
                    b.marker = 0;
                } else {
                    /* the marker has not been reached by the bit reader yet: find it */
                    while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
                    if (b.p + 1 < b.end) b.p += 2;
                }
                pred[0] = pred[1] = pred[2] = 0;
                todo = H->restart;
            }
            for (int c = 0; c < H->ncomp; c++)
                for (int v = 0; v < H->vs[c]; v++)
                    for (int h = 0; h < H->hs[c]; h++) {
                        int16_t *blk = coef[c] + ((size_t)(my * H->vs[c] + v) * H->bw[c] + (mx * H->hs[c] + h)) * 64;
                        memset(blk, 0, 64 * sizeof(int16_t));
                        int t = jo_sym(&b, &H->dc[H->td[c]]);
                        int diff = t ? jo_extend(jo_get(&b, t), t) : 0;
                        pred[c] += diff;
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = jo_sym(&b, &H->ac[H->ta[c]]);
                            const int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) break; /* (corrupt data) */
                            blk[jo_zigzag[k]] = (int16_t)jo_extend(jo_get(&b, s), s);
                            k++;
                        }
                    }
            if (H->restart) todo--;
        }
    return JO_OK;
}

/* ---- jidctint.c (JDCT_ISLOW), dequantisation folded in as the reference does */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static uint8_t jo_clamp(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static void jo_idct_1d(const int32_t in[8], int32_t out[8], int shift)
{
    int32_t z2 = in[2], z3 = in[6];
    int32_t z1 = (z2 + z3) * FIX_0_541196100;
    int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
    int32_t tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0];
    z3 = in[4];
    int32_t tmp0 = (z2 + z3) * (1 << CONST_BITS);
    int32_t tmp1 = (z2 - z3) * (1 << CONST_BITS);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336;
    tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026;
    tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223;
    z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560;
    z4 *= -FIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = DESCALE(tmp10 + tmp3, shift);
    out[7] = DESCALE(tmp10 - tmp3, shift);
    out[1] = DESCALE(tmp11 + tmp2, shift);
    out[6] = DESCALE(tmp11 - tmp2, shift);
    out[2] = DESCALE(tmp12 + tmp1, shift);
    out[5] = DESCALE(tmp12 - tmp1, shift);
    out[3] = DESCALE(tmp13 + tmp0, shift);
    out[4] = DESCALE(tmp13 - tmp0, shift);
}

static void jo_idct_block(const int16_t *coef, const uint16_t *q, uint8_t *dst, int pitch)
{
    int32_t ws[64];
    for (int c = 0; c < 8; c++) {
        int32_t in[8], out[8];
        for (int r = 0; r < 8; r++) in[r] = (int32_t)coef[r * 8 + c] * (int32_t)q[r * 8 + c];
        jo_idct_1d(in, out, CONST_BITS - PASS1_BITS); /* (the all-zero-AC shortcut of the reference gives the same values) */
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = out[r];
    }
    for (int r = 0; r < 8; r++) {
        int32_t out[8];
        jo_idct_1d(ws + r * 8, out, CONST_BITS + PASS1_BITS + 3);
        for (int c = 0; c < 8; c++) dst[r * pitch + c] = jo_clamp(out[c] + 128);
    }
}

/* ---- jdsample.c fancy upsampling.  src: component plane (pitch sp), real size cw x ch; dst: full-resolution plane */
static void jo_h2v1_fancy(const uint8_t *src, int sp, int cw, int ch, uint8_t *dst, int dp)
{
    for (int y = 0; y < ch; y++) {
        const uint8_t *in = src + (size_t)y * sp;
        uint8_t *out = dst + (size_t)y * dp;
        if (cw == 1) {
            out[0] = out[1] = in[0];
            continue;
        }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int x = 1; x < cw - 1; x++) {
            out[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
            out[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
        }
        out[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
        out[2 * cw - 1] = in[cw - 1];
    }
}
static void jo_h2v2_fancy(const uint8_t *src, int sp, int cw, int ch, uint8_t *dst, int dp)
{
    for (int y = 0; y < ch; y++)
        for (int v = 0; v < 2; v++) {
            /* the nearer row counts 3, the farther 1; rows outside the real data repeat the edge row (jdmainct.c) */
            int yf = v == 0 ? y - 1 : y + 1;
            yf = yf < 0 ? 0 : (yf > ch - 1 ? ch - 1 : yf);
            const uint8_t *in0 = src + (size_t)y * sp, *in1 = src + (size_t)yf * sp;
            uint8_t *out = dst + (size_t)(2 * y + v) * dp;
            if (cw == 1) {
                const int t = in0[0] * 3 + in1[0];
                out[0] = (uint8_t)((t * 4 + 8) >> 4);
                out[1] = (uint8_t)((t * 4 + 7) >> 4);
                continue;
            }
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
            out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
            for (int x = 1; x < cw - 1; x++) {
                nextcol = in0[x + 1] * 3 + in1[x + 1];
                out[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                out[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol;
                thiscol = nextcol;
            }
            out[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            out[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
}

/* ---- public entry points (ctypes) */
int jpeg_oracle_info(const uint8_t *data, size_t n, int32_t *info /* w h ncomp hmax vmax bw0 bh0 bw1 bh1 restart */)
{
    jo_hdr H;
    const int rc = jo_parse(data, n, &H);
    if (rc != JO_OK) return rc;
    info[0] = H.w; info[1] = H.h; info[2] = H.ncomp; info[3] = H.hmax; info[4] = H.vmax;
    info[5] = H.bw[0]; info[6] = H.bh[0]; info[7] = H.ncomp == 3 ? H.bw[1] : 0; info[8] = H.ncomp == 3 ? H.bh[1] : 0;
    info[9] = H.restart;
    return JO_OK;
}

/* everything at once; any output may be NULL.
 *   coefs:  component after component, [bh][bw][64] int16, quantised, natural order
 *   planes: component after component, [bh*8][bw*8] u8 (IDCT output, MCU padded)
 *   bgr:    [h][w][3] u8 as cv::imdecode(IMREAD_COLOR) hands it to the node (B = G = R for a one-component file) */
int jpeg_oracle_decode(const uint8_t *data, size_t n, int16_t *coefs, uint8_t *planes, uint8_t *bgr)
{
    jo_hdr H;
    int rc = jo_parse(data, n, &H);
    if (rc != JO_OK) return rc;
    int16_t *coef[3] = {0, 0, 0};
    uint8_t *pl[3] = {0, 0, 0}, *up[3] = {0, 0, 0};
    size_t nb[3];
    for (int c = 0; c < H.ncomp; c++) {
        nb[c] = (size_t)H.bw[c] * H.bh[c];
        coef[c] = (int16_t *)malloc(nb[c] * 64 * sizeof(int16_t));
        pl[c] = (uint8_t *)malloc(nb[c] * 64);
        if (!coef[c] || !pl[c]) return JO_E_FORMAT;
    }
    rc = jo_entropy(&H, coef);
    for (int c = 0; c < H.ncomp && rc == JO_OK; c++) {
        const int pitch = H.bw[c] * 8;
        for (int by = 0; by < H.bh[c]; by++)
            for (int bx = 0; bx < H.bw[c]; bx++)
                jo_idct_block(coef[c] + ((size_t)by * H.bw[c] + bx) * 64, H.q[H.tq[c]], pl[c] + (size_t)by * 8 * pitch + bx * 8, pitch);
    }
    if (rc == JO_OK && coefs) {
        size_t o = 0;
        for (int c = 0; c < H.ncomp; c++) {
            memcpy(coefs + o, coef[c], nb[c] * 64 * sizeof(int16_t));
            o += nb[c] * 64;
        }
    }
    if (rc == JO_OK && planes) {
        size_t o = 0;
        for (int c = 0; c < H.ncomp; c++) {
            memcpy(planes + o, pl[c], nb[c] * 64);
            o += nb[c] * 64;
        }
    }
    if (rc == JO_OK && bgr) {
        const int W = H.w, Hh = H.h;
        if (H.ncomp == 1) {
            for (int y = 0; y < Hh; y++)
                for (int x = 0; x < W; x++) {
                    const uint8_t v = pl[0][(size_t)y * H.bw[0] * 8 + x];
                    uint8_t *o = bgr + ((size_t)y * W + x) * 3;
                    o[0] = o[1] = o[2] = v;
                }
        } else {
            /* chroma to full resolution */
            const int cw = (W + H.hmax - 1) / H.hmax, ch = (Hh + H.vmax - 1) / H.vmax; /* downsampled_width / _height */
            const int fp = cw * H.hmax;
            for (int c = 1; c < 3; c++) {
                up[c] = (uint8_t *)malloc((size_t)fp * ch * H.vmax + 16);
                const int sp = H.bw[c] * 8;
                if (H.hmax == 1 && H.vmax == 1) {
                    for (int y = 0; y < ch; y++) memcpy(up[c] + (size_t)y * fp, pl[c] + (size_t)y * sp, (size_t)cw);
                } else if (H.hmax == 2 && H.vmax == 1) {
                    if (cw > 2) jo_h2v1_fancy(pl[c], sp, cw, ch, up[c], fp);
                    else
                        for (int y = 0; y < ch; y++)
                            for (int x = 0; x < fp; x++) up[c][(size_t)y * fp + x] = pl[c][(size_t)y * sp + x / 2];
                } else {
                    if (cw > 2) jo_h2v2_fancy(pl[c], sp, cw, ch, up[c], fp);
                    else
                        for (int y = 0; y < ch * 2; y++)
                            for (int x = 0; x < fp; x++) up[c][(size_t)y * fp + x] = pl[c][(size_t)(y / 2) * sp + x / 2];
                }
            }
            /* jdcolor.c ycc_rgb_convert */
            for (int y = 0; y < Hh; y++)
                for (int x = 0; x < W; x++) {
                    const int Y = pl[0][(size_t)y * H.bw[0] * 8 + x];
                    const int cb = up[1][(size_t)y * fp + x] - 128, cr = up[2][(size_t)y * fp + x] - 128;
                    const int r = Y + ((91881 * cr + 32768) >> 16);
                    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                    const int bl = Y + ((116130 * cb + 32768) >> 16);
                    uint8_t *o = bgr + ((size_t)y * W + x) * 3;
                    o[0] = jo_clamp(bl);
                    o[1] = jo_clamp(g);
                    o[2] = jo_clamp(r);
                }
        }
    }
    for (int c = 0; c < 3; c++) {
        free(coef[c]);
        free(pl[c]);
        free(up[c]);
    }
    return rc;
}
