// tr_loaders.cpp -- asset loading behind the C ABI (tr_load_obj, tr_load_tga_rgb8).
//
// Counterparts of what src/app.rs:87-131 gets from third-party crates:
//   obj-rs 0.7.0 `parse_obj`           -> positions / tex_coords / normals / polygons
//   image 0.24.5 `open(..).into_rgb8()` -> tightly packed rgb8, row 0 = top
// Only what the path consumes is produced: the first three v/vt/vn triples of a face
// (scene.rs:224-226), xyz of a position (util.rs:25-31).  Faces without all of v, vt and vn are
// reported as TR_E_BAD_POLYGON, where the reference panics (scene.rs:218).
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "tiny_renderer.h"
#include "tr_error.h"

namespace {

struct MeshOwner {
    tr_mesh pub;
    std::vector<float> pos, tex, nrm;
    std::vector<uint32_t> idx;
};

bool read_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) {
        fclose(f);
        return false;
    }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// Parses up to `want` floats from s (strtof: correctly rounded, like Rust's str::parse::<f32>).
int parse_floats(const char *s, float *dst, int want)
{
    int n = 0;
    while (n < want) {
        char *end = nullptr;
        errno = 0;
        float v = strtof(s, &end);
        if (end == s) break;
        dst[n++] = v;
        s = end;
    }
    return n;
}

// One "v/vt/vn" group.  Returns false unless all three indices are present.
bool parse_ptn(const char *&s, long count[3], uint32_t out[3])
{
    while (*s == ' ' || *s == '\t') s++;
    if (*s == '\0' || *s == '\n' || *s == '\r') return false;
    for (int k = 0; k < 3; k++) {
        char *end = nullptr;
        long v = strtol(s, &end, 10);
        if (end == s) return false;
        // OBJ indices are 1-based; negative ones are relative to the end of the list
        long z = v > 0 ? v - 1 : count[k] + v;
        if (v == 0 || z < 0 || z >= count[k]) return false;
        out[k] = (uint32_t)z;
        s = end;
        if (k < 2) {
            if (*s != '/') return false;
            s++;
        }
    }
    return true;
}

}  // namespace

extern "C" int tr_load_obj(const char *path, tr_mesh **out)
{
    if (!path || !out) return tr::fail(TR_E_INVALID, "tr_load_obj: null argument");
    *out = nullptr;
    std::vector<uint8_t> data;
    if (!read_file(path, data)) return tr::fail(TR_E_IO, std::string("cannot read ") + path);
    data.push_back('\n');
    data.push_back('\0');

    MeshOwner *m = new MeshOwner();
    int status = TR_OK;
    std::string why;
    char *p = reinterpret_cast<char *>(data.data());
    long line_no = 0;
    while (*p) {
        char *eol = strchr(p, '\n');
        *eol = '\0';
        line_no++;
        const char *s = p;
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float v[4] = { 0, 0, 0, 1 };
            if (parse_floats(s + 1, v, 4) < 3) {
                status = TR_E_FORMAT;
                why = "bad vertex position";
            }
            m->pos.insert(m->pos.end(), v, v + 3);
        } else if (s[0] == 'v' && s[1] == 't' && (s[2] == ' ' || s[2] == '\t')) {
            float v[3] = { 0, 0, 0 };
            if (parse_floats(s + 2, v, 3) < 1) {
                status = TR_E_FORMAT;
                why = "bad texture coordinate";
            }
            m->tex.insert(m->tex.end(), v, v + 3);
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            float v[3] = { 0, 0, 0 };
            if (parse_floats(s + 2, v, 3) < 3) {
                status = TR_E_FORMAT;
                why = "bad vertex normal";
            }
            m->nrm.insert(m->nrm.end(), v, v + 3);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            long count[3] = { (long)(m->pos.size() / 3), (long)(m->tex.size() / 3), (long)(m->nrm.size() / 3) };
            const char *q = s + 1;
            uint32_t tri[9];
            bool ok = true;
            for (int k = 0; k < 3 && ok; k++) ok = parse_ptn(q, count, &tri[3 * k]);
            if (!ok) {
                status = TR_E_BAD_POLYGON;
                why = "face is not three or more v/vt/vn groups with valid indices";
            } else {
                m->idx.insert(m->idx.end(), tri, tri + 9);
            }
        }
        if (status != TR_OK) {
            why += " (line " + std::to_string(line_no) + " of " + path + ")";
            break;
        }
        p = eol + 1;
    }
    if (status != TR_OK) {
        delete m;
        return tr::fail(status, why);
    }
    m->pub.pos = m->pos.data();
    m->pub.tex = m->tex.data();
    m->pub.nrm = m->nrm.data();
    m->pub.idx = m->idx.data();
    m->pub.n_pos = (uint32_t)(m->pos.size() / 3);
    m->pub.n_tex = (uint32_t)(m->tex.size() / 3);
    m->pub.n_nrm = (uint32_t)(m->nrm.size() / 3);
    m->pub.n_tri = (uint32_t)(m->idx.size() / 9);
    *out = &m->pub;  // pub is the first member: the owner is recovered by a cast in tr_free_mesh
    return TR_OK;
}

extern "C" void tr_free_mesh(tr_mesh *mesh)
{
    if (mesh) delete reinterpret_cast<MeshOwner *>(mesh);
}

// TGA: image types 1 (colour-mapped), 2 (true colour), 3 (grey) and 9 / 10 / 11 (their run-length
// forms); 8-bit indices into a 24- or 32-bit palette, 24- or 32-bit true colour, 8-bit grey; either
// origin.  15/16-bit pixels or palettes are rejected.
extern "C" int tr_load_tga_rgb8(const char *path, tr_image_rgb8 *out)
{
    if (!path || !out) return tr::fail(TR_E_INVALID, "tr_load_tga_rgb8: null argument");
    out->rgb = nullptr;
    out->w = out->h = 0;
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return tr::fail(TR_E_IO, std::string("cannot read ") + path);
    if (d.size() < 18) return tr::fail(TR_E_FORMAT, std::string("truncated TGA header: ") + path);
    const uint32_t id_len = d[0], cmap_type = d[1], type = d[2];
    const uint32_t cmap_len = d[5] | (d[6] << 8), cmap_bits = d[7];
    const uint32_t w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bpp = d[16], desc = d[17];
    const bool rle = (type == 9 || type == 10 || type == 11), grey = (type == 3 || type == 11);
    const bool mapped = (type == 1 || type == 9);
    if (!(type == 1 || type == 2 || type == 3 || type == 9 || type == 10 || type == 11))
        return tr::fail(TR_E_FORMAT, std::string("unsupported TGA image type in ") + path);
    if (((grey || mapped) && bpp != 8) || (!grey && !mapped && bpp != 24 && bpp != 32))
        return tr::fail(TR_E_FORMAT, std::string("unsupported TGA pixel depth in ") + path);
    if (mapped && (cmap_type != 1 || (cmap_bits != 24 && cmap_bits != 32)))
        return tr::fail(TR_E_FORMAT, std::string("unsupported TGA colour map in ") + path);
    const size_t bytes_pp = bpp / 8, npx = (size_t)w * h;
    const uint32_t cmap_first = d[3] | (d[4] << 8);
    const size_t cmap_entry = (cmap_bits + 7) / 8, cmap_pos = 18 + id_len;
    size_t pos = cmap_pos + (cmap_type ? (size_t)cmap_len * cmap_entry : 0);
    if (pos > d.size()) return tr::fail(TR_E_FORMAT, std::string("truncated TGA colour map: ") + path);

    std::vector<uint8_t> raw(npx * bytes_pp);
    if (!rle) {
        if (pos + raw.size() > d.size()) return tr::fail(TR_E_FORMAT, std::string("truncated TGA data: ") + path);
        if (!raw.empty()) memcpy(raw.data(), &d[pos], raw.size());
    } else {
        size_t px = 0;
        while (px < npx) {
            if (pos >= d.size()) return tr::fail(TR_E_FORMAT, std::string("truncated TGA run: ") + path);
            const uint8_t hd = d[pos++];
            size_t run = (size_t)(hd & 0x7F) + 1;
            if (px + run > npx) run = npx - px;
            if (hd & 0x80) {
                if (pos + bytes_pp > d.size()) return tr::fail(TR_E_FORMAT, std::string("truncated TGA run: ") + path);
                for (size_t i = 0; i < run; i++) memcpy(&raw[(px + i) * bytes_pp], &d[pos], bytes_pp);
                pos += bytes_pp;
            } else {
                if (pos + run * bytes_pp > d.size()) return tr::fail(TR_E_FORMAT, std::string("truncated TGA run: ") + path);
                memcpy(&raw[px * bytes_pp], &d[pos], run * bytes_pp);
                pos += run * bytes_pp;
            }
            px += run;
        }
    }

    uint8_t *rgb = (uint8_t *)malloc(npx ? npx * 3 : 1);
    if (!rgb) return tr::fail(TR_E_NOMEM, "tr_load_tga_rgb8: out of memory");
    const bool top_origin = (desc & 0x20) != 0, right_origin = (desc & 0x10) != 0;
    for (uint32_t y = 0; y < h; y++) {
        const uint32_t sy = top_origin ? y : h - 1 - y;
        for (uint32_t x = 0; x < w; x++) {
            const uint32_t sx = right_origin ? w - 1 - x : x;
            const uint8_t *s = &raw[((size_t)sy * w + sx) * bytes_pp];
            uint8_t *t = &rgb[((size_t)y * w + x) * 3];
            if (grey) {
                t[0] = t[1] = t[2] = s[0];
            } else if (mapped) {
                // palette entries are stored b, g, r(, a) like true-colour pixels
                const uint32_t k = s[0] >= cmap_first ? s[0] - cmap_first : cmap_len;
                if (k >= cmap_len) {
                    free(rgb);
                    return tr::fail(TR_E_FORMAT, std::string("TGA palette index out of range in ") + path);
                }
                const uint8_t *c = &d[cmap_pos + (size_t)k * cmap_entry];
                t[0] = c[2];
                t[1] = c[1];
                t[2] = c[0];
            } else {
                t[0] = s[2];
                t[1] = s[1];
                t[2] = s[0];
            }
        }
    }
    out->rgb = rgb;
    out->w = w;
    out->h = h;
    return TR_OK;
}

// Frame writer (the reference has none: it shows frames in a window): uncompressed 24-bit
// true-colour TGA, top-left origin -- the layout of get_frame_buffer.
extern "C" int tr_save_tga_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h)
{
    if (!path || (!rgb && w && h)) return tr::fail(TR_E_INVALID, "tr_save_tga_rgb8: null argument");
    if (w > 65535u || h > 65535u) return tr::fail(TR_E_INVALID, "tr_save_tga_rgb8: a TGA side is at most 65535");
    FILE *f = fopen(path, "wb");
    if (!f) return tr::fail(TR_E_IO, std::string("cannot write ") + path);
    uint8_t hd[18] = { 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint8_t)(w & 0xFF), (uint8_t)(w >> 8),
                       (uint8_t)(h & 0xFF), (uint8_t)(h >> 8), 24, 0x20 };
    bool ok = fwrite(hd, 1, sizeof hd, f) == sizeof hd;
    std::vector<uint8_t> row((size_t)w * 3);
    for (uint32_t y = 0; y < h && ok; y++) {
        const uint8_t *s = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            row[3 * x] = s[3 * x + 2];
            row[3 * x + 1] = s[3 * x + 1];
            row[3 * x + 2] = s[3 * x];
        }
        ok = row.empty() || fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) return tr::fail(TR_E_IO, std::string("short write to ") + path);
    return TR_OK;
}

// PNG frame writer: 8-bit RGB, filter 0, the image data in stored (uncompressed) deflate blocks --
// a valid PNG any viewer opens, written without a compression library.
namespace {
uint32_t crc32_update(uint32_t c, const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t k = i;
            for (int j = 0; j < 8; j++) k = (k & 1u) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
            table[i] = k;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
bool write_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc32_update(0xFFFFFFFFu, head.data() + 4, 4);
    if (!data.empty()) c = crc32_update(c, data.data(), data.size());
    std::vector<uint8_t> tail;
    put_be32(tail, c ^ 0xFFFFFFFFu);
    return fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}
}  // namespace

extern "C" int tr_save_png_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h)
{
    if (!path || (!rgb && w && h)) return tr::fail(TR_E_INVALID, "tr_save_png_rgb8: null argument");
    if (w == 0 || h == 0 || w > 65535u || h > 65535u) return tr::fail(TR_E_INVALID, "tr_save_png_rgb8: sides must be within 1..65535");
    FILE *f = fopen(path, "wb");
    if (!f) return tr::fail(TR_E_IO, std::string("cannot write ") + path);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    bool ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    const uint8_t fmt[5] = { 8, 2, 0, 0, 0 };  // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), fmt, fmt + 5);
    ok = ok && write_chunk(f, "IHDR", ihdr);
    // zlib stream: header, stored blocks of at most 65535 bytes over the filtered scanlines, adler32
    const size_t stride = (size_t)w * 3 + 1, total = stride * h;
    std::vector<uint8_t> z;
    z.reserve(total + total / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    std::vector<uint8_t> raw(total);
    for (uint32_t y = 0; y < h; y++) {
        raw[y * stride] = 0;  // filter type 0
        memcpy(&raw[y * stride + 1], rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    for (size_t off = 0; off < total; off += 65535) {
        const size_t n = total - off < 65535 ? total - off : 65535;
        z.push_back(off + n == total ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    }
    for (size_t i = 0; i < total;) {  // adler32 in runs short enough not to overflow 32 bits
        const size_t n = total - i < 5552 ? total - i : 5552;
        for (size_t k = 0; k < n; k++) { a += raw[i + k]; b += a; }
        a %= 65521u; b %= 65521u;
        i += n;
    }
    put_be32(z, (b << 16) | a);
    ok = ok && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", std::vector<uint8_t>());
    ok = (fclose(f) == 0) && ok;
    if (!ok) return tr::fail(TR_E_IO, std::string("short write to ") + path);
    return TR_OK;
}

extern "C" void tr_free_image(tr_image_rgb8 *img)
{
    if (img && img->rgb) {
        free(const_cast<uint8_t *>(img->rgb));
        img->rgb = nullptr;
        img->w = img->h = 0;
    }
}
