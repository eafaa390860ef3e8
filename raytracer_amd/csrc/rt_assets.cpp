// Host asset decoders for the renderer: Wavefront OBJ -> triangle soup, PNG -> 8-bit pixels.
//
// They produce what the reference's loaders produce for the same files:
//   * OBJ: LoadOBJ (Object.cpp:349-435): v / vt / vn / f records, 1-based and negative (relative) indices, fan
//     triangulation, face normal when a corner names no normal.  Numbers are parsed with the same iostream
//     extraction the reference uses, so every float has the same bits.
//   * PNG: the stb_image calls of DataUtility.cpp:17-28,98-109 — rows flipped vertically, a requested channel count.
//     Decoding is done here with zlib's inflate and the PNG filters (PNG spec, sections 9 and 8.2 Adam7).
#include "rt_host.h"

#include <zlib.h>

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace rt {

// ------------------------------------------------------------------------------------------------- OBJ
bool load_obj(const std::string& path, std::vector<float>& verts8, std::string& err) {
    std::ifstream in(path);
    if (!in) { err = "cannot open OBJ file " + path; return false; }
    struct Ref { int v, vt, vn; };
    struct P4 { float x, y, z, w; };
    std::vector<P4> positions(1, P4{0, 0, 0, 0});
    std::vector<Vec3> texcoords(1, Vec3{0, 0, 0});
    std::vector<Vec3> normals(1, Vec3{0, 0, 0});
    std::string line;
    verts8.clear();
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string type;
        ls >> type;
        if (type == "v") {
            float x = 0, y = 0, z = 0, w = 1;
            ls >> x >> y >> z >> w;
            positions.push_back(P4{x, y, z, w});
        } else if (type == "vt") {
            float u = 0, v = 0, w = 0;
            ls >> u >> v >> w;
            texcoords.push_back(Vec3{u, v, w});
        } else if (type == "vn") {
            float i = 0, j = 0, k = 0;
            ls >> i >> j >> k;
            normals.push_back(normalize(Vec3{i, j, k}));
        } else if (type == "f") {
            std::vector<Ref> refs;
            std::string tok;
            while (ls >> tok) {
                std::istringstream rs(tok);
                std::string a, b, c;
                std::getline(rs, a, '/');
                std::getline(rs, b, '/');
                std::getline(rs, c, '/');
                int v = atoi(a.c_str()), vt = atoi(b.c_str()), vn = atoi(c.c_str());
                v = v >= 0 ? v : (int)positions.size() + v;
                vt = vt >= 0 ? vt : (int)texcoords.size() + vt;
                vn = vn >= 0 ? vn : (int)normals.size() + vn;
                refs.push_back(Ref{v, vt, vn});
            }
            for (size_t i = 1; i + 1 < refs.size(); ++i) {
                const Ref* p[3] = {&refs[0], &refs[i], &refs[i + 1]};
                for (int j = 0; j < 3; j++)
                    if (p[j]->v < 0 || p[j]->v >= (int)positions.size() || p[j]->vt < 0 || p[j]->vt >= (int)texcoords.size() ||
                        p[j]->vn < 0 || p[j]->vn >= (int)normals.size()) { err = "OBJ index out of range in " + path; return false; }
                const P4 &a = positions[p[0]->v], &b = positions[p[1]->v], &c = positions[p[2]->v];
                Vec3 U{b.x - a.x, b.y - a.y, b.z - a.z}, V{c.x - a.x, c.y - a.y, c.z - a.z};
                Vec3 face = normalize(cross(U, V));
                for (int j = 0; j < 3; j++) {
                    const P4& q = positions[p[j]->v];
                    const Vec3& t = texcoords[p[j]->vt];
                    Vec3 n = p[j]->vn != 0 ? normals[p[j]->vn] : face;
                    const float row[8] = {q.x, q.y, q.z, t.x, t.y, n.x, n.y, n.z};
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
