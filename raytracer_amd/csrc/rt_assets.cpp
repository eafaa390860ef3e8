// Host asset decoders for the renderer: Wavefront OBJ -> triangle soup, PNG -> 8-bit pixels.
//
// They produce what the reference's loaders produce for the same files:
//   * OBJ: LoadOBJ (Object.cpp:349-435): v / vt / vn / f records, 1-based and negative (relative) indices, fan
//     triangulation, face normal when a corner names no normal.  Every field yields the value the reference's parser
//     gives it (strtof on the same characters), so every float has the same bits.
//   * PNG: the stb_image calls of DataUtility.cpp:17-28,98-109 — rows flipped vertically, a requested channel count.
//     Decoding is done here with zlib's inflate and the PNG filters (PNG spec, sections 9 and 8.2 Adam7).
#include "rt_host.h"

#include <zlib.h>

#include <algorithm>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

// ------------------------------------------------------------------------------------------------- OBJ
// The file is read whole and scanned in place: lines, blank-separated words, and inside a face word the '/'-separated
// index fields.  What has to agree with the reference's parser (formatted stream extraction into floats, atoi on index
// fields) is the VALUE every field yields, including the odd cases:
//   * a number is [sign] digits [. digits] [e|E [sign] digits] as far as it goes; the text is converted by strtof, and
//     a field that is not a number as a whole ("1e", "-", "abc") yields 0 and ends the record: the fields after it
//     keep their defaults (x y z u v i j k = 0, w = 1), as do fields the line does not have;
//   * an index field is an optional sign and digits; anything else (or nothing) counts as 0 = "not given";
//   * negative indices count back from the end of what has been read so far; a face with more than three corners is a
//     fan about its first corner; a corner without a normal gets the face's own (Object.cpp:349-435).
namespace {
struct Text {
    const char* at;
    const char* end;
    bool done() const { return at >= end; }
};
inline bool blank(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
// the next line of `file` (without its '\n'); false at the end of the file
bool next_line(Text& file, Text& line) {
    if (file.done()) return false;
    const char* nl = static_cast<const char*>(memchr(file.at, '\n', (size_t)(file.end - file.at)));
    line.at = file.at;
    line.end = nl ? nl : file.end;
    file.at = nl ? nl + 1 : file.end;
    return true;
}
// the next blank-separated word of `line`; false when only blanks are left
bool next_word(Text& line, Text& word) {
    while (!line.done() && blank(*line.at)) line.at++;
    if (line.done()) return false;
    word.at = line.at;
    while (!line.done() && !blank(*line.at)) line.at++;
    word.end = line.at;
    return true;
}
bool word_is(const Text& w, const char* name) {
    const size_t n = strlen(name);
    return (size_t)(w.end - w.at) == n && memcmp(w.at, name, n) == 0;
}
// Up to `n` numbers from the rest of `line` into out[0..n), stopping at the first field that is missing or malformed.
void take_numbers(Text line, float* out, int n) {
    for (int k = 0; k < n; k++) {
        while (!line.done() && blank(*line.at)) line.at++;
        if (line.done()) return;                       // nothing left: this field and the rest keep their defaults
        char text[64];
        size_t len = 0;
        bool digits = false, point = false, expo = false;
        const char* p = line.at;
        if (p < line.end && (*p == '+' || *p == '-')) text[len++] = *p++;
        while (p < line.end && len + 1 < sizeof text) {
            const char c = *p;
            if (c >= '0' && c <= '9') digits = true;
            else if (c == '.' && !point && !expo) point = true;
            else if ((c == 'e' || c == 'E') && !expo && digits) {
                expo = true;
                text[len++] = c; p++;
                if (p < line.end && (*p == '+' || *p == '-') && len + 1 < sizeof text) { text[len++] = *p++; }
                continue;
            } else break;
            text[len++] = c; p++;
        }
        text[len] = 0;
        line.at = p;
        char* stop = text;
        float v = len ? strtof(text, &stop) : 0.0f;
        const bool whole = len && stop == text + len;
        if (!whole) { out[k] = 0.0f; return; }         // not a number: zero, and the record ends here
        if (std::isinf(v)) { out[k] = v > 0 ? FLT_MAX : -FLT_MAX; return; }   // out of range: clamped, and the record ends
        out[k] = v;
    }
}
// an index field [at, end): optional sign and digits, 0 for anything else
int index_field(const char* at, const char* end) {
    while (at < end && blank(*at)) at++;
    bool neg = false;
    if (at < end && (*at == '+' || *at == '-')) neg = *at++ == '-';
    long long v = 0;
    while (at < end && *at >= '0' && *at <= '9') { v = v * 10 + (*at++ - '0'); if (v > 0x7FFFFFFFll) v = 0x7FFFFFFFll; }
    return (int)(neg ? -v : v);
}
}  // namespace

bool load_obj(const std::string& path, std::vector<float>& verts8, std::string& err) {
    std::vector<char> bytes;
    {
        FILE* fp = fopen(path.c_str(), "rb");
        if (!fp) { err = "cannot open OBJ file " + path; return false; }
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, fp)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
        fclose(fp);
    }
    // tables are 1-based: slot 0 is the "not given" entry a zero index names
    std::vector<float> pos(3, 0.0f), tex(2, 0.0f), nrm(3, 0.0f);
    struct Corner { int p, t, n; };
    std::vector<Corner> corners;
    verts8.clear();
    Text file{bytes.data(), bytes.data() + bytes.size()}, line, word;
    while (next_line(file, line)) {
        if (!next_word(line, word)) continue;
        if (word_is(word, "v")) {
            float f[4] = {0, 0, 0, 1};
            take_numbers(line, f, 4);
            pos.insert(pos.end(), f, f + 3);
        } else if (word_is(word, "vt")) {
            float f[3] = {0, 0, 0};
            take_numbers(line, f, 3);
            tex.insert(tex.end(), f, f + 2);
        } else if (word_is(word, "vn")) {
            float f[3] = {0, 0, 0};
            take_numbers(line, f, 3);
            const Vec3 n = normalize(Vec3{f[0], f[1], f[2]});
            nrm.push_back(n.x); nrm.push_back(n.y); nrm.push_back(n.z);
        } else if (word_is(word, "f")) {
            const int n_pos = (int)pos.size() / 3, n_tex = (int)tex.size() / 2, n_nrm = (int)nrm.size() / 3;
            corners.clear();
            while (next_word(line, word)) {
                int idx[3] = {0, 0, 0};
                const char* at = word.at;
                for (int k = 0; k < 3 && at <= word.end; k++) {
                    const char* slash = static_cast<const char*>(memchr(at, '/', (size_t)(word.end - at)));
                    const char* stop = slash ? slash : word.end;
                    idx[k] = index_field(at, stop);
                    at = stop + 1;
                    if (!slash) break;
                }
                Corner c{idx[0], idx[1], idx[2]};
                if (c.p < 0) c.p += n_pos;
                if (c.t < 0) c.t += n_tex;
                if (c.n < 0) c.n += n_nrm;
                if (c.p < 0 || c.p >= n_pos || c.t < 0 || c.t >= n_tex || c.n < 0 || c.n >= n_nrm) { err = "OBJ index out of range in " + path; return false; }
                corners.push_back(c);
            }
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
                const float *a = &pos[3 * (size_t)tri[0].p], *b = &pos[3 * (size_t)tri[1].p], *c = &pos[3 * (size_t)tri[2].p];
                const Vec3 face = normalize(cross(Vec3{b[0] - a[0], b[1] - a[1], b[2] - a[2]}, Vec3{c[0] - a[0], c[1] - a[1], c[2] - a[2]}));
                for (const Corner& q : tri) {
                    const float* pp = &pos[3 * (size_t)q.p];
                    const float* tt = &tex[2 * (size_t)q.t];
                    const Vec3 n = q.n ? Vec3{nrm[3 * (size_t)q.n], nrm[3 * (size_t)q.n + 1], nrm[3 * (size_t)q.n + 2]} : face;
                    const float row[8] = {pp[0], pp[1], pp[2], tt[0], tt[1], n.x, n.y, n.z};
                    verts8.insert(verts8.end(), row, row + 8);
                }
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------- PNG
namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}
// undo the scanline filters of one (sub)image in place; `data` holds (1 + stride) bytes per row
bool unfilter(uint8_t* data, size_t avail, int rows, size_t stride, int bpp, std::vector<uint8_t>& out) {
    if (avail < (size_t)rows * (stride + 1)) return false;
    out.assign((size_t)rows * stride, 0);
    for (int y = 0; y < rows; y++) {
        const uint8_t* src = data + (size_t)y * (stride + 1);
        const int ft = src[0];
        src++;
        uint8_t* cur = out.data() + (size_t)y * stride;
        const uint8_t* up = y ? cur - stride : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            cur[x] = (uint8_t)v;
        }
    }
    return true;
}
}  // namespace

bool load_png(const std::string& path, int req_channels, bool flip, std::vector<uint8_t>& out, int& w, int& h,
              int& file_channels, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open image " + path; return false; }
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) { err = path + " is not a PNG file"; return false; }
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false;
    while (p + 12 <= file.size()) {
        uint32_t len = be32(&file[p]);
        const uint8_t* type = &file[p + 4];
        if (p + 12 + (size_t)len > file.size()) { err = "truncated PNG " + path; return false; }
        const uint8_t* d = &file[p + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad PNG header in " + path; return false; }
            const uint32_t uw = be32(d), uh = be32(d + 4);
            // bounds that keep every size computed below far from overflow and from absurd allocations: 2^15 texels per
            // side is what the texture lookups can address with their 16-bit sizes anyway
            if (uw == 0 || uh == 0 || uw > 32768u || uh > 32768u) { err = "PNG dimensions out of range (1..32768) in " + path; return false; }
            w = (int)uw; h = (int)uh; depth = d[8]; ctype = d[9]; interlace = d[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0) { err = "bad PNG header in " + path; return false; }
    if (depth != 8) { err = "only 8-bit PNG files are supported: " + path; return false; }
    int chans;
    switch (ctype) {
    case 0: chans = 1; break;
    case 2: chans = 3; break;
    case 3: chans = 1; break;
    case 4: chans = 2; break;
    case 6: chans = 4; break;
    default: err = "bad PNG colour type in " + path; return false;
    }
    // The filtered data the header announces: one filter byte and the samples of every row, for the whole image or for
    // each of the seven Adam7 passes.
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t need = 0;
    if (!interlace) need = (size_t)h * ((size_t)w * chans + 1);
    else
        for (int pass = 0; pass < 7; pass++) {
            const int pw = (w - xs[pass] + dx[pass] - 1) / dx[pass], ph = (h - ys[pass] + dy[pass] - 1) / dy[pass];
            if (pw > 0 && ph > 0) need += (size_t)ph * ((size_t)pw * chans + 1);
        }
    // Deflate cannot expand by more than 1032 : 1, so a file too short for its header is turned away before anything of
    // the announced size (up to 4 GiB) is allocated.
    if (idat.size() > ((size_t)1 << 40) || idat.size() * 1032 + 1024 < need) { err = "corrupt PNG data in " + path; return false; }
    // inflate: both sides are fed to zlib in pieces that fit its 32-bit counters
    std::vector<uint8_t> raw;
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; }
        raw.resize(need);
        const size_t piece = (size_t)1 << 30;
        size_t in_at = 0, out_at = 0;
        int rc = Z_OK;
        while (rc == Z_OK) {
            if (zs.avail_in == 0 && in_at < idat.size()) {
                const size_t n_in = std::min(piece, idat.size() - in_at);
                zs.next_in = idat.data() + in_at; zs.avail_in = (uInt)n_in; in_at += n_in;
            }
            if (zs.avail_out == 0 && out_at < raw.size()) {
                const size_t n_out = std::min(piece, raw.size() - out_at);
                zs.next_out = raw.data() + out_at; zs.avail_out = (uInt)n_out; out_at += n_out;
            }
            if (zs.avail_out == 0) break;                          // more data than the header announced: ignored
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_at == idat.size()) break;   // the stream ends early
        }
        const size_t got = out_at - zs.avail_out;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) { err = "zlib inflate failed on " + path; return false; }
        if (got < need) { err = "corrupt PNG data in " + path; return false; }
    }
    std::vector<uint8_t> px((size_t)w * h * chans);
    if (!interlace) {
        std::vector<uint8_t> img;
        if (!unfilter(raw.data(), raw.size(), h, (size_t)w * chans, chans, img)) { err = "corrupt PNG data in " + path; return false; }
        px.swap(img);
    } else {
        size_t off = 0;
        for (int pass = 0; pass < 7; pass++) {
            int pw = (w - xs[pass] + dx[pass] - 1) / dx[pass], ph = (h - ys[pass] + dy[pass] - 1) / dy[pass];
            if (pw <= 0 || ph <= 0) continue;
            std::vector<uint8_t> img;
            if (off > raw.size() || !unfilter(raw.data() + off, raw.size() - off, ph, (size_t)pw * chans, chans, img)) { err = "corrupt PNG data in " + path; return false; }
            off += (size_t)ph * ((size_t)pw * chans + 1);
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++)
                    memcpy(&px[((size_t)(ys[pass] + y * dy[pass]) * w + xs[pass] + x * dx[pass]) * chans], &img[((size_t)y * pw + x) * chans], chans);
        }
    }
    // expand to the file's logical channel count
    std::vector<uint8_t> logical;
    if (ctype == 3) {
        file_channels = trns.empty() ? 3 : 4;
        logical.resize((size_t)w * h * file_channels);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            unsigned idx = px[i];
            for (int c = 0; c < 3; c++) logical[i * file_channels + c] = idx * 3 + c < plte.size() ? plte[idx * 3 + c] : 0;
            if (file_channels == 4) logical[i * 4 + 3] = idx < trns.size() ? trns[idx] : 255;
        }
    } else {
        file_channels = chans;
        logical.swap(px);
        if (!trns.empty() && (ctype == 0 || ctype == 2)) {   // colour-key transparency adds an alpha channel
            int fc = chans + 1;
            std::vector<uint8_t> withA((size_t)w * h * fc);
            for (size_t i = 0; i < (size_t)w * h; i++) {
                bool key = true;
                for (int c = 0; c < chans; c++) {
                    withA[i * fc + c] = logical[i * chans + c];
                    if (trns.size() >= (size_t)(2 * c + 2) && logical[i * chans + c] != trns[2 * c + 1]) key = false;
                }
                withA[i * fc + chans] = key ? 0 : 255;
            }
            logical.swap(withA);
            file_channels = fc;
        }
    }
    // convert to the requested channel count the way stb_image's stbi__convert_format does
    if (req_channels <= 0) req_channels = file_channels;
    out.resize((size_t)w * h * req_channels);
    const int fc = file_channels;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* s = &logical[i * fc];
        uint8_t r, g, b, a;
        if (fc >= 3) { r = s[0]; g = s[1]; b = s[2]; a = fc == 4 ? s[3] : 255; }
        else { r = g = b = s[0]; a = fc == 2 ? s[1] : 255; }
        uint8_t* d = &out[i * req_channels];
        const uint8_t y = fc >= 3 ? (uint8_t)(((r * 77) + (g * 150) + (29 * b)) >> 8) : s[0];
        switch (req_channels) {
        case 1: d[0] = y; break;
        case 2: d[0] = y; d[1] = a; break;
        case 3: d[0] = r; d[1] = g; d[2] = b; break;
        default: d[0] = r; d[1] = g; d[2] = b; d[3] = a; break;
        }
    }
    if (flip) {
        const size_t row = (size_t)w * req_channels;
        std::vector<uint8_t> tmp(row);
        for (int y = 0; y < h / 2; y++) {
            memcpy(tmp.data(), &out[(size_t)y * row], row);
            memcpy(&out[(size_t)y * row], &out[(size_t)(h - 1 - y) * row], row);
            memcpy(&out[(size_t)(h - 1 - y) * row], tmp.data(), row);
        }
    }
    return true;
}

}  // namespace rt
