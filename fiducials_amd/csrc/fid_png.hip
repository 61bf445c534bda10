// fid_png.hip -- frames that arrive as PNG (compressed_image_transport with `format: png`: cv::imencode(".png") of the camera
// image), SURVEY section 8 row f1.  Part of the fid_api.hip translation unit (included there; not compiled on its own).
//
// Host code on purpose.  A PNG is a zlib stream -- a sequential bit stream whose every symbol's position depends on all symbols
// before it, with back-references of up to 32 KB -- followed by a per-row prediction filter; the reference decodes it on the
// CPU too (cv::imdecode inside the subscriber plugin, in front of imageCallback, aruco_detect.cpp:332).  Inflate is zlib's; what
// is written here is the container (chunks, CRC), the row filters and the conversion to what cv::imdecode(IMREAD_COLOR) hands
// the node: 8-bit BGR (alpha dropped, palettes expanded, 1 / 2 / 4-bit gray scaled to 8 bits, 16-bit samples cut to their high
// byte).  The decoded frame then takes the normal road: fid_detect with FID_ENC_BGR8 / FID_ENC_MONO8.  Interlaced (Adam7) files,
// which cv::imencode never writes, are refused with FID_E_UNSUPPORTED -- never a wrong image.
#include <zlib.h>

namespace {

thread_local std::string g_png_error;

inline uint32_t png_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct PngHeader {
    int32_t width = 0, height = 0, bit_depth = 0, color_type = 0, interlace = 0;
    int channels() const { return color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 3 ? 1 : color_type == 4 ? 2 : 4; }
};

// IHDR and the chunk walk: IDAT payloads are appended to `idat`, the palette (if any) copied out
fid_status png_parse(const uint8_t *data, int64_t n, PngHeader *h, std::vector<uint8_t> *idat, uint8_t pal[256][3], int *npal)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!data || n < 8 + 25 || memcmp(data, sig, 8)) {
        g_png_error = "not a PNG file";
        return FID_E_INVALID_ARG;
    }
    int64_t off = 8;
    bool have_ihdr = false, have_end = false;
    if (npal) *npal = 0;
    while (off + 12 <= n && !have_end) {
        const uint32_t len = png_be32(data + off);
        const uint8_t *type = data + off + 4, *body = data + off + 8;
        if ((int64_t)len > n - off - 12) {
            g_png_error = "truncated chunk";
            return FID_E_INVALID_ARG;
        }
        const uint32_t crc = png_be32(body + len);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) {
            g_png_error = "chunk CRC mismatch";
            return FID_E_INVALID_ARG;
        }
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) {
                g_png_error = "bad IHDR";
                return FID_E_INVALID_ARG;
            }
            h->width = (int32_t)png_be32(body);
            h->height = (int32_t)png_be32(body + 4);
            h->bit_depth = body[8];
            h->color_type = body[9];
            h->interlace = body[12];
            const int bd = h->bit_depth, ct = h->color_type;
            const bool ok = (ct == 0 && (bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16)) || (ct == 3 && (bd == 1 || bd == 2 || bd == 4 || bd == 8)) ||
                            ((ct == 2 || ct == 4 || ct == 6) && (bd == 8 || bd == 16));
            if (!ok || body[10] != 0 || body[11] != 0 || h->width < 1 || h->height < 1 || h->width > 16384 || h->height > 16384) {
                g_png_error = "IHDR values outside the PNG specification (or an image beyond 16384 pixels a side)";
                return FID_E_INVALID_ARG;
            }
            have_ihdr = true;
        } else if (!have_ihdr) {
            g_png_error = "first chunk is not IHDR";
            return FID_E_INVALID_ARG;
        } else if (!memcmp(type, "PLTE", 4)) {
            if (len % 3 || len > 768) {
                g_png_error = "bad PLTE";
                return FID_E_INVALID_ARG;
            }
            if (pal) memcpy(pal, body, len);
            if (npal) *npal = (int)(len / 3);
        } else if (!memcmp(type, "IDAT", 4)) {
            if (idat) idat->insert(idat->end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            have_end = true;
        } else if (!(type[0] & 0x20)) {  // a critical chunk this decoder does not know
            g_png_error = "unknown critical chunk";
            return FID_E_UNSUPPORTED;
        }
        off += 12 + (int64_t)len;
    }
    if (!have_ihdr || !have_end) {
        g_png_error = "IHDR / IEND missing";
        return FID_E_INVALID_ARG;
    }
    return FID_OK;
}

inline int png_paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

extern "C" {

static fid_status png_probe_impl(const uint8_t *data, int64_t nbytes, fid_png_info *info)
{
    if (!info) return FID_E_INVALID_ARG;
    PngHeader h;
    const fid_status rc = png_parse(data, nbytes, &h, nullptr, nullptr, nullptr);
    if (rc != FID_OK) return rc;
    info->width = h.width;
    info->height = h.height;
    info->bit_depth = h.bit_depth;
    info->color_type = h.color_type;
    info->interlace = h.interlace;
    info->gray = (h.color_type == 0 || h.color_type == 4) ? 1 : 0;
    return FID_OK;
}

static fid_status png_decode_impl(const uint8_t *data, int64_t nbytes, fid_encoding out_enc, uint8_t *out, int64_t out_bytes, fid_png_info *info)
{
    if (!out || (out_enc != FID_ENC_BGR8 && out_enc != FID_ENC_MONO8)) return FID_E_INVALID_ARG;
    PngHeader h;
    std::vector<uint8_t> idat;
    uint8_t pal[256][3];
    int npal = 0;
    memset(pal, 0, sizeof pal);
    fid_status rc = png_parse(data, nbytes, &h, &idat, pal, &npal);
    if (rc != FID_OK) return rc;
    if (info) {
        info->width = h.width;
        info->height = h.height;
        info->bit_depth = h.bit_depth;
        info->color_type = h.color_type;
        info->interlace = h.interlace;
        info->gray = (h.color_type == 0 || h.color_type == 4) ? 1 : 0;
    }
    if (h.interlace) {
        g_png_error = "interlaced (Adam7) PNG";
        return FID_E_UNSUPPORTED;
    }
    if (h.color_type == 3 && npal == 0) {
        g_png_error = "palette image without PLTE";
        return FID_E_INVALID_ARG;
    }
    const int W = h.width, H = h.height, ch = h.channels(), bd = h.bit_depth;
    const int opx = out_enc == FID_ENC_BGR8 ? 3 : 1;
    if (out_bytes < (int64_t)W * H * opx) {
        g_png_error = "output buffer too small";
        return FID_E_CAPACITY;
    }
    const size_t rowbytes = ((size_t)W * ch * bd + 7) / 8, bpp = (size_t)(ch * bd + 7) / 8;  // filter unit: whole bytes per pixel, >= 1
    // sizes come from an untrusted header: a tiny file may announce 16384 x 16384 RGBA16.  No more than what a camera frame can be
    // (the pixel limit of cv::imdecode, CV_IO_MAX_IMAGE_PIXELS = 2^30), and deflate cannot expand by more than ~1032 : 1
    if ((uint64_t)W * (uint64_t)H > (1ull << 30) || (rowbytes + 1) * (size_t)H > (size_t)idat.size() * 1040 + 4096) {
        g_png_error = "image data shorter than the header says";
        return FID_E_INVALID_ARG;
    }
    std::vector<uint8_t> raw((rowbytes + 1) * (size_t)H);
    {
        uLongf got = (uLongf)raw.size();
        const int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
        if (zr != Z_OK || got != raw.size()) {
            g_png_error = zr == Z_OK ? "image data shorter than the header says" : "zlib stream is damaged";
            return FID_E_INVALID_ARG;
        }
    }
    // ---- the row filters (PNG specification, section 9): in place, row by row
    std::vector<uint8_t> zero(rowbytes, 0);
    for (int y = 0; y < H; y++) {
        uint8_t *row = raw.data() + (size_t)y * (rowbytes + 1);
        const int ft = row[0];
        uint8_t *cur = row + 1;
        const uint8_t *up = y ? row - rowbytes : zero.data();  // (the row above, already reconstructed: its bytes start one past its filter byte)
        switch (ft) {
        case 0: break;
        case 1:
            for (size_t i = bpp; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + up[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < rowbytes; i++)
                cur[i] = (uint8_t)(cur[i] + png_paeth(i >= bpp ? cur[i - bpp] : 0, up[i], i >= bpp ? up[i - bpp] : 0));
            break;
        default:
            g_png_error = "unknown row filter";
            return FID_E_INVALID_ARG;
        }
    }
    // ---- to what cv::imdecode(IMREAD_COLOR) returns (BGR), or its cvtColor(BGR2GRAY) (the constants of k_to_gray)
    const int step = bd == 16 ? 2 : 1;  // 16-bit samples: the high byte (png_set_strip_16)
    for (int y = 0; y < H; y++) {
        const uint8_t *cur = raw.data() + (size_t)y * (rowbytes + 1) + 1;
        uint8_t *o = out + (size_t)y * W * opx;
        for (int x = 0; x < W; x++) {
            int r, g, b;
            if (h.color_type == 0 || h.color_type == 3) {
                int v;
                if (bd >= 8) {
                    v = cur[(size_t)x * step * ch];
                } else {
                    const int per = 8 / bd, sh = (per - 1 - x % per) * bd;
                    v = (cur[x / per] >> sh) & ((1 << bd) - 1);
                    if (h.color_type == 0) v = v * (255 / ((1 << bd) - 1));  // png_set_expand_gray_1_2_4_to_8
                }
                if (h.color_type == 3) {
                    if (v >= npal) {
                        g_png_error = "palette index beyond PLTE";
                        return FID_E_INVALID_ARG;
                    }
                    r = pal[v][0];
                    g = pal[v][1];
                    b = pal[v][2];
                } else {
                    r = g = b = v;
                }
            } else if (h.color_type == 4) {
                r = g = b = cur[(size_t)x * 2 * step];
            } else {
                const uint8_t *p = cur + (size_t)x * ch * step;
                r = p[0];
                g = p[step];
                b = p[2 * step];
            }
            if (opx == 3) {
                o[3 * x] = (uint8_t)b;
                o[3 * x + 1] = (uint8_t)g;
                o[3 * x + 2] = (uint8_t)r;
            } else {
                o[x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
            }
        }
    }
    return FID_OK;
}

// nothing throws across the boundary (fid_abi.h): allocation failures on sizes a damaged file announces come back as a status
fid_status fid_png_probe(const uint8_t *data, int64_t nbytes, fid_png_info *info)
{
    try {
        return png_probe_impl(data, nbytes, info);
    } catch (const std::bad_alloc &) {
        g_png_error = "out of memory";
        return FID_E_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_png_error = std::string("damaged file: ") + e.what();
        return FID_E_INVALID_ARG;
    }
}

fid_status fid_png_decode(const uint8_t *data, int64_t nbytes, fid_encoding out_enc, uint8_t *out, int64_t out_bytes, fid_png_info *info)
{
    try {
        return png_decode_impl(data, nbytes, out_enc, out, out_bytes, info);
    } catch (const std::bad_alloc &) {
        g_png_error = "out of memory";
        return FID_E_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_png_error = std::string("damaged file: ") + e.what();
        return FID_E_INVALID_ARG;
    }
}

const char *fid_png_last_error(void) { return g_png_error.c_str(); }

}  // extern "C"
