// fid_dict.hip -- aruco::getPredefinedDictionary(dicno) (aruco_detect.cpp:671) from a table file the DEPLOYER has.  Part of the
// fid_api.hip translation unit; host code, no device.
//
// OpenCV's dictionary tables are third-party data: they ship neither with the reference (it links libopencv_aruco) nor with
// this repository (fiducials_amd/data/ holds the codewords the reference's fixtures pin plus labelled fillers).  A deployment
// that links OpenCV hands Dictionary::bytesList straight to fid_create; one that does not points this loader at
//   (a) OpenCV's modules/aruco/src/predefined_dictionaries.hpp as text: `static unsigned char DICT_6X6_1000_BYTES[][4][5] =
//       { { { b, .. }, { .. }, { .. }, { .. } }, ... }` -- per marker four rotations of nbytes = ceil(n^2 / 8) bytes, exactly
//       bytesList; the 50 / 100 / 250-marker dictionaries are the first rows of the 1000-marker table of their size
//       (dictionary.cpp), DICT_ARUCO_ORIGINAL is DICT_ARUCO_BYTES (1024 markers, 5 x 5);
//   (b) a cv::FileStorage YAML as aruco::Dictionary::writeDictionary / the contrib sample `create_dictionary` writes it:
//       nmarkers, markersize, maxCorrectionBits, marker_<i>: "<n*n bits, row-major, 1 = white>" (custom dictionaries: dicno -1);
//   (c) this repository's own dict_*.txt (index, P | F, hexadecimal codeword).
#include <stdio.h>

namespace {

thread_local std::string g_dict_error;

struct DictRow {
    int n, count, maxc;
};
// enum value -> (marker size, nMarkers, maxCorrectionBits): dictionary.cpp's predefined Dictionary objects
bool dict_row(int dicno, DictRow *r)
{
    static const DictRow rows[17] = {{4, 50, 1},   {4, 100, 1},  {4, 250, 1},  {4, 1000, 0}, {5, 50, 3},  {5, 100, 3}, {5, 250, 2}, {5, 1000, 2}, {6, 50, 6},
                                     {6, 100, 5},  {6, 250, 5},  {6, 1000, 4}, {7, 50, 9},   {7, 100, 8}, {7, 250, 8}, {7, 1000, 6}, {5, 1024, 0}};
    if (dicno < 0 || dicno > 16) return false;
    *r = rows[dicno];
    return true;
}

// Dictionary::getByteListFromBits: rotation r = r quarter turns counter-clockwise, bits packed MSB first
void dict_bytes_from_bits(const std::vector<int> &bits0, int n, uint8_t *out /* 4 * nbytes */)
{
    const int nbytes = (n * n + 7) / 8;
    std::vector<int> bits = bits0, rot((size_t)n * n);
    memset(out, 0, (size_t)4 * nbytes);
    for (int r = 0; r < 4; r++) {
        uint8_t *o = out + (size_t)r * nbytes;
        int cur = 0;
        for (int i = 0; i < n * n; i++) {
            o[cur] = (uint8_t)((o[cur] << 1) | bits[i]);
            if (i % 8 == 7) cur++;
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) rot[(size_t)i * n + j] = bits[(size_t)j * n + (n - 1 - i)];
        bits = rot;
    }
}

std::string dict_strip_comments(const std::string &s)
{
    std::string o;
    o.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        if (s.compare(i, 2, "//") == 0) {
            while (i < s.size() && s[i] != '\n') i++;
        } else if (s.compare(i, 2, "/*") == 0) {
            const size_t e = s.find("*/", i + 2);
            i = e == std::string::npos ? s.size() : e + 2;
        } else {
            o.push_back(s[i++]);
        }
    }
    return o;
}

fid_status dict_load_impl(const char *path, int32_t dicno, uint8_t *bytes, int64_t cap, fid_dict *out)
{
    if (!path || !out) return FID_E_INVALID_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) {
        g_dict_error = std::string("cannot open ") + path;
        return FID_E_INVALID_ARG;
    }
    std::string text;
    char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) {
        text.append(buf, got);
        if (text.size() > (64u << 20)) {  // (OpenCV's header is 1.3 MB) -- refused, not cut off in the middle of a table
            fclose(f);
            g_dict_error = "file larger than 64 MB: not a dictionary table";
            return FID_E_INVALID_ARG;
        }
    }
    fclose(f);
    DictRow row = {0, 0, 0};
    const bool have_row = dict_row(dicno, &row);
    if (dicno != -1 && !have_row) {
        g_dict_error = "dictionary enum value outside 0..16 (and not -1 = what the file says)";
        return FID_E_INVALID_ARG;
    }
    int n = 0, count = 0, maxc = 0;
    std::vector<uint8_t> table;  // count x 4 x nbytes
    // ---- (a) predefined_dictionaries.hpp
    const bool is_hpp = text.find("_BYTES") != std::string::npos && text.find('{') != std::string::npos;
    const bool is_yaml = !is_hpp && text.find("markersize") != std::string::npos;
    if (is_hpp) {
        if (!have_row) {
            g_dict_error = "an OpenCV header holds many tables: say which with the enum value (0..16)";
            return FID_E_INVALID_ARG;
        }
        n = row.n;
        const int nbytes = (n * n + 7) / 8;
        char name[64];
        if (dicno == 16) snprintf(name, sizeof name, "DICT_ARUCO_BYTES");
        else snprintf(name, sizeof name, "DICT_%dX%d_1000_BYTES", n, n);
        const std::string src = dict_strip_comments(text);
        size_t p = src.find(name);
        if (p == std::string::npos) {
            g_dict_error = std::string("array ") + name + " not found in the file";
            return FID_E_INVALID_ARG;
        }
        p = src.find('=', p);
        const size_t open = p == std::string::npos ? std::string::npos : src.find('{', p);
        if (open == std::string::npos) {
            g_dict_error = std::string("array ") + name + " has no initialiser";
            return FID_E_INVALID_ARG;
        }
        int depth = 0;
        std::vector<long> vals;
        size_t i = open;
        for (; i < src.size(); i++) {
            const char ch = src[i];
            if (ch == '{') depth++;
            else if (ch == '}') {
                if (--depth == 0) break;
            } else if (ch >= '0' && ch <= '9') {
                char *end = nullptr;
                const long v = strtol(src.c_str() + i, &end, 0);
                if (v < 0 || v > 255) {
                    g_dict_error = "a table entry is not a byte";
                    return FID_E_INVALID_ARG;
                }
                vals.push_back(v);
                i = (size_t)(end - src.c_str()) - 1;
            }
        }
        if (depth != 0 || vals.empty() || vals.size() % ((size_t)4 * nbytes) != 0) {
            g_dict_error = std::string("array ") + name + " does not hold whole markers of 4 x " + std::to_string(nbytes) + " bytes";
            return FID_E_INVALID_ARG;
        }
        const int have = (int)(vals.size() / ((size_t)4 * nbytes));
        if (have < row.count) {
            g_dict_error = std::string("array ") + name + " holds " + std::to_string(have) + " markers, the dictionary needs " + std::to_string(row.count);
            return FID_E_INVALID_ARG;
        }
        count = row.count;
        maxc = row.maxc;
        table.resize((size_t)count * 4 * nbytes);
        for (size_t k = 0; k < table.size(); k++) table[k] = (uint8_t)vals[k];
    } else if (is_yaml) {
        // ---- (b) FileStorage YAML: scalar keys and marker_<i> strings, whatever the order
        auto scalar = [&](const char *key, int *v) {
            const size_t p = text.find(key);
            if (p == std::string::npos) return false;
            const size_t c = text.find(':', p);
            if (c == std::string::npos) return false;
            *v = atoi(text.c_str() + c + 1);
            return true;
        };
        if (!scalar("nmarkers", &count) || !scalar("markersize", &n) || count < 1 || n < 1 || n > 16) {
            g_dict_error = "nmarkers / markersize missing";
            return FID_E_INVALID_ARG;
        }
        if (!scalar("maxCorrectionBits", &maxc)) maxc = 0;
        if (have_row && (row.n != n || row.count > count)) {
            g_dict_error = "the file's markersize / nmarkers do not fit the enum value";
            return FID_E_INVALID_ARG;
        }
        if (have_row) {
            count = row.count;
            maxc = row.maxc;
        }
        const int nbytes = (n * n + 7) / 8;
        // every marker needs its n * n bit characters in the file: a count the file cannot hold is refused before anything is
        // allocated for it ("nmarkers: 2000000000" used to zero-fill tens of GB first)
        if ((uint64_t)count * (uint64_t)(n * n) > (uint64_t)text.size()) {
            g_dict_error = "nmarkers is larger than the file can hold";
            return FID_E_INVALID_ARG;
        }
        table.assign((size_t)count * 4 * nbytes, 0);
        std::vector<int> bits((size_t)n * n);
        size_t from = 0;  // FileStorage writes the markers in order: the search resumes behind the previous one (and starts over
                          // from the top once for a file that does not)
        for (int m = 0; m < count; m++) {
            const std::string key = "marker_" + std::to_string(m) + ":";
            size_t p = text.find(key, from);
            if (p == std::string::npos && from > 0) p = text.find(key);
            if (p == std::string::npos) {
                g_dict_error = "marker_" + std::to_string(m) + " missing";
                return FID_E_INVALID_ARG;
            }
            p += key.size();
            from = p;
            int k = 0;
            for (; p < text.size() && text[p] != '\n' && k < n * n; p++)
                if (text[p] == '0' || text[p] == '1') bits[k++] = text[p] - '0';
            if (k != n * n) {
                g_dict_error = "marker_" + std::to_string(m) + " does not hold markersize^2 bits";
                return FID_E_INVALID_ARG;
            }
            dict_bytes_from_bits(bits, n, &table[(size_t)m * 4 * nbytes]);
        }
    } else {
        // ---- (c) dict_*.txt: "<index> <P|F> <hex codeword>" lines; the header comment names the marker size
        if (!have_row) {
            g_dict_error = "a dict_*.txt table needs the enum value (its lines do not say the marker size)";
            return FID_E_INVALID_ARG;
        }
        n = row.n;
        count = row.count;
        maxc = row.maxc;
        const int nbytes = (n * n + 7) / 8;
        table.assign((size_t)count * 4 * nbytes, 0);
        std::vector<int> bits((size_t)n * n);
        int have = 0;
        size_t pos = 0;
        while (have < count && pos < text.size()) {
            size_t e = text.find('\n', pos);
            if (e == std::string::npos) e = text.size();
            const std::string line = text.substr(pos, e - pos);
            pos = e + 1;
            if (line.empty() || line[0] == '#') continue;
            int idx = -1;
            char flag[8] = {0}, hex[64] = {0};
            if (sscanf(line.c_str(), "%d %7s %63s", &idx, flag, hex) != 3 || idx != have) {
                g_dict_error = "not a dictionary table (line " + std::to_string(have) + ")";
                return FID_E_INVALID_ARG;
            }
            const unsigned long long word = strtoull(hex, nullptr, 16);
            for (int k = 0; k < n * n; k++) bits[k] = (int)((word >> (n * n - 1 - k)) & 1ull);
            dict_bytes_from_bits(bits, n, &table[(size_t)have * 4 * nbytes]);
            have++;
        }
        if (have != count) {
            g_dict_error = "dictionary table too short";
            return FID_E_INVALID_ARG;
        }
    }
    out->marker_size = n;
    out->max_correction_bits = maxc;
    out->n_markers = count;
    out->reserved0 = 0;
    out->bytes = nullptr;
    if (!bytes || cap < (int64_t)table.size()) {
        g_dict_error = "byte buffer too small: " + std::to_string(table.size()) + " bytes needed";
        return FID_E_CAPACITY;
    }
    memcpy(bytes, table.data(), table.size());
    out->bytes = bytes;
    return FID_OK;
}

}  // namespace

extern "C" {

fid_status fid_dict_load_file(const char *path, int32_t dicno, uint8_t *bytes, int64_t bytes_cap, fid_dict *out)
{
    try {
        return dict_load_impl(path, dicno, bytes, bytes_cap, out);
    } catch (const std::bad_alloc &) {
        g_dict_error = "out of memory";
        return FID_E_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_dict_error = std::string("damaged file: ") + e.what();
        return FID_E_INVALID_ARG;
    }
}

const char *fid_dict_last_error(void) { return g_dict_error.c_str(); }

}  // extern "C"
