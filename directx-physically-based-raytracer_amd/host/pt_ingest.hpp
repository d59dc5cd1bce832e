// pt_ingest.hpp -- scene-JSON + glTF 2.0 ingest for the C++ host, in the reference's schema and conventions.
//
// What it mirrors (host side of the path, upstream of the acceleration-structure build):
//   scene descriptor   Source/MyScene.ixx:33-90, Source/JSONConverters.ixx:12-33, Source/Scene.ixx:33-73
//                      {Camera{Position,Rotation}, EnvironmentLight{Color,Rotation,Texture}, Models{name:path},
//                       RenderObjects[{Name,Transform{Translation,Rotation,Scale},IsVisible,Model}]}
//   glTF loader        Source/GLTFHelpers.ixx:142-537 (ProcessPrimitive, LoadModel): triangles only, the index array written BACKWARDS
//                      (flipWindingOrder is always set, Scene.ixx:90), u16 indices iff count <= 65535, tangents ALWAYS recomputed when
//                      NORMAL + TEXCOORD_0 exist, material factors incl. KHR_materials_emissive_strength / ior / transmission
//   instance transform Source/Scene.ixx:195-231: world = GlobalTransform * Scale(1,1,-1) * RenderObject.Transform() (row vectors),
//                      AffineTransform = Scale * Rotation * Translation (Math.ixx:17-19)
// Same conventions, same arithmetic order as the test harness's ingest.py, so that both hosts hand the library the same bytes
// (tests/test_host_cpp.py compares vertex / index buffers, transforms and materials of the committed fixture, and the rendered frame).
// Textures: 8-bit PNG images (embedded or referenced) are decoded here (a small inflate + the PNG filters, checked against PIL); the reference
// decodes files with DirectXTex / stb behind TextureHelpers.ixx, this image has neither, so JPEG / DDS / 16-bit references are listed in
// MeshData::SkippedTextures and the material keeps its factors (a host with a codec fills those slots through pt_heap_set_texture).
// [DirectXMesh spec] ComputeTangentFrame is an un-vendored dependency: restated as Lengyel's per-vertex accumulation with Gram-Schmidt
// against the normal (as in ingest.py). Header-only, C++20, no dependency beyond the standard library and include/ptamd.h.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptamd.h"

namespace ptamd::ingest {

// ------------------------------------------------------------------------------------------------
// a small JSON reader (objects, arrays, strings, numbers, true / false / null)
// ------------------------------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false; double num = 0.0; std::string str;
    std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const std::string& key) const
    {
        if (kind != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { const Json* j = find(key); return j && j->kind != Null; }
    const Json& at(const std::string& key) const { const Json* j = find(key); if (!j) throw std::runtime_error("JSON: missing key " + key); return *j; }
    const Json& at(size_t i) const { if (kind != Array || i >= arr.size()) throw std::runtime_error("JSON: index out of range"); return arr[i]; }
    double number(const std::string& key, double def) const { const Json* j = find(key); return j && j->kind == Number ? j->num : def; }
    size_t size() const { return kind == Array ? arr.size() : obj.size(); }
};

class JsonParser {
public:
    explicit JsonParser(const std::string& text) : s(text) {}
    Json parse() { Json v = value(); ws(); if (p != s.size()) fail("trailing characters"); return v; }

private:
    const std::string& s; size_t p = 0;
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("JSON: ") + what + " at byte " + std::to_string(p)); }
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++; }
    bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { p++; return true; } return false; }
    Json value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end");
        Json v;
        const char c = s[p];
        if (c == '{') {
            p++; v.kind = Json::Object;
            if (eat('}')) return v;
            do { ws(); Json k = value(); if (k.kind != Json::String) fail("object key is not a string"); if (!eat(':')) fail("':' expected"); v.obj.emplace_back(k.str, value()); } while (eat(','));
            if (!eat('}')) fail("'}' expected");
        } else if (c == '[') {
            p++; v.kind = Json::Array;
            if (eat(']')) return v;
            do v.arr.push_back(value()); while (eat(','));
            if (!eat(']')) fail("']' expected");
        } else if (c == '"') {
            p++; v.kind = Json::String;
            while (p < s.size() && s[p] != '"') {
                if (s[p] == '\\') {
                    if (++p >= s.size()) fail("bad escape");
                    switch (s[p]) {
                    case 'n': v.str += '\n'; break; case 't': v.str += '\t'; break; case 'r': v.str += '\r'; break;
                    case 'b': v.str += '\b'; break; case 'f': v.str += '\f'; break;
                    case 'u': {                                            // BMP code point -> UTF-8 (names and paths only)
                        if (p + 4 >= s.size()) fail("bad \\u escape");
                        const unsigned cp = (unsigned)std::stoul(s.substr(p + 1, 4), nullptr, 16); p += 4;
                        if (cp < 0x80) v.str += (char)cp;
                        else if (cp < 0x800) { v.str += (char)(0xC0 | (cp >> 6)); v.str += (char)(0x80 | (cp & 0x3F)); }
                        else { v.str += (char)(0xE0 | (cp >> 12)); v.str += (char)(0x80 | ((cp >> 6) & 0x3F)); v.str += (char)(0x80 | (cp & 0x3F)); }
                        break; }
                    default: v.str += s[p];
                    }
                    p++;
                } else v.str += s[p++];
            }
            if (p >= s.size()) fail("unterminated string");
            p++;
        } else if (!s.compare(p, 4, "true")) { p += 4; v.kind = Json::Bool; v.b = true; }
        else if (!s.compare(p, 5, "false")) { p += 5; v.kind = Json::Bool; v.b = false; }
        else if (!s.compare(p, 4, "null")) { p += 4; v.kind = Json::Null; }
        else {
            char* end = nullptr;
            v.num = std::strtod(s.c_str() + p, &end);                     // correctly rounded, like Python's float()
            if (end == s.c_str() + p) fail("value expected");
            p = (size_t)(end - s.c_str()); v.kind = Json::Number;
        }
        return v;
    }
};

inline std::string read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
inline std::string dir_of(const std::string& path) { const size_t k = path.find_last_of('/'); return k == std::string::npos ? std::string(".") : path.substr(0, k); }
inline std::string resolve(const std::string& base, const std::string& p) { return (p.empty() || p[0] == '/') ? p : base + "/" + p; }

inline std::string base64_decode(const std::string& in)
{
    std::string out; unsigned acc = 0; int bits = 0;
    for (unsigned char c : in) {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else continue;     // '=' padding and whitespace are skipped
        acc = (acc << 6) | (unsigned)v; bits += 6;
        if (bits >= 8) { bits -= 8; out += (char)((acc >> bits) & 0xFFu); }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// PNG (the lossless half of what glTF embeds; JPEG / DDS need a codec this image does not have: such textures are listed and skipped)
// ------------------------------------------------------------------------------------------------
// RFC 1951 inflate (stored, fixed and dynamic Huffman blocks) behind the RFC 1950 zlib header; canonical-code decoding by counts per length.
class Inflate {
public:
    static std::vector<uint8_t> zlib(const uint8_t* src, size_t n)
    {
        if (n < 6 || (src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) throw std::runtime_error("PNG: bad zlib header");
        Inflate z(src + 2, n - 2);
        z.run();
        return std::move(z.out);
    }

private:
    Inflate(const uint8_t* s, size_t n) : in(s), size(n) {}
    const uint8_t* in; size_t size, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0;
    std::vector<uint8_t> out;
    struct Huffman { uint16_t count[16]; uint16_t symbol[288]; };

    uint32_t bits(int need)
    {
        uint32_t v = bitbuf;
        while (bitcnt < need) { if (pos >= size) throw std::runtime_error("PNG: deflate stream ends early"); v |= (uint32_t)in[pos++] << bitcnt; bitcnt += 8; }
        bitbuf = need < 32 ? v >> need : 0; bitcnt -= need;
        return need < 32 ? v & ((1u << need) - 1u) : v;
    }
    static void build(Huffman& h, const uint8_t* lengths, int n)
    {
        for (int i = 0; i < 16; i++) h.count[i] = 0;
        for (int i = 0; i < n; i++) h.count[lengths[i]]++;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + h.count[i]);
        for (int i = 0; i < n; i++) if (lengths[i]) h.symbol[offs[lengths[i]]++] = (uint16_t)i;
        h.count[0] = 0;
    }
    int decode(const Huffman& h)
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)bits(1);
            const int count = h.count[len];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count; first += count; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("PNG: bad Huffman code");
    }
    void codes(const Huffman& lencode, const Huffman& distcode)
    {
        static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
        static const uint16_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
        static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
        static const uint16_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
        for (;;) {
            int sym = decode(lencode);
            if (sym < 256) out.push_back((uint8_t)sym);
            else if (sym == 256) return;
            else {
                sym -= 257;
                if (sym >= 29) throw std::runtime_error("PNG: bad length symbol");
                const int len = lbase[sym] + (int)bits(lext[sym]);
                const int ds = decode(distcode);
                if (ds >= 30) throw std::runtime_error("PNG: bad distance symbol");
                const size_t dist = dbase[ds] + bits(dext[ds]);
                if (dist > out.size()) throw std::runtime_error("PNG: distance beyond the window");
                for (int k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
            }
        }
    }
    void run()
    {
        for (bool last = false; !last;) {
            last = bits(1) != 0;
            const uint32_t type = bits(2);
            if (type == 0) {
                bitbuf = 0; bitcnt = 0;
                if (pos + 4 > size) throw std::runtime_error("PNG: stored block ends early");
                const uint32_t len = in[pos] | (in[pos + 1] << 8), nlen = in[pos + 2] | (in[pos + 3] << 8);
                pos += 4;
                if ((len ^ 0xFFFFu) != nlen || pos + len > size) throw std::runtime_error("PNG: bad stored block");
                out.insert(out.end(), in + pos, in + pos + len); pos += len;
            } else if (type == 1) {
                uint8_t l[288];
                for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                Huffman lc, dc; build(lc, l, 288);
                uint8_t d[30]; for (int i = 0; i < 30; i++) d[i] = 5;
                build(dc, d, 30);
                codes(lc, dc);
            } else if (type == 2) {
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
                if (nlen > 286 || ndist > 30) throw std::runtime_error("PNG: bad dynamic block header");
                uint8_t l[320] = { 0 };
                for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)bits(3);
                Huffman cl; build(cl, l, 19);
                uint8_t lengths[320]; int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode(cl);
                    if (sym < 16) lengths[idx++] = (uint8_t)sym;
                    else {
                        uint8_t prev = 0; int rep;
                        if (sym == 16) { if (!idx) throw std::runtime_error("PNG: repeat without a previous length"); prev = lengths[idx - 1]; rep = 3 + (int)bits(2); }
                        else if (sym == 17) rep = 3 + (int)bits(3); else rep = 11 + (int)bits(7);
                        if (idx + rep > nlen + ndist) throw std::runtime_error("PNG: too many code lengths");
                        while (rep--) lengths[idx++] = prev;
                    }
                }
                Huffman lc, dc; build(lc, lengths, nlen); build(dc, lengths + nlen, ndist);
                codes(lc, dc);
            } else throw std::runtime_error("PNG: bad block type");
        }
    }
};

struct Image { uint32_t Width = 0, Height = 0; std::vector<uint8_t> RGBA; };       // 8 bits per channel, rows top to bottom

inline bool is_png(const std::string& d) { return d.size() >= 8 && !std::memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8); }

// 8-bit grey / grey+alpha / RGB / RGBA / palette PNG without interlacing -> RGBA8, as PIL's Image.open(...).convert("RGBA") delivers it
inline Image decode_png(const std::string& d)
{
    if (!is_png(d)) throw std::runtime_error("not a PNG");
    auto be32 = [&](size_t o) { return ((uint32_t)(uint8_t)d[o] << 24) | ((uint32_t)(uint8_t)d[o + 1] << 16) | ((uint32_t)(uint8_t)d[o + 2] << 8) | (uint32_t)(uint8_t)d[o + 3]; };
    Image im; int depth = 0, ctype = 0, interlace = 0;
    std::string idat; std::vector<uint8_t> palette, trns;
    for (size_t o = 8; o + 12 <= d.size();) {
        const uint32_t len = be32(o); const std::string type = d.substr(o + 4, 4);
        if (o + 12 + (size_t)len > d.size()) throw std::runtime_error("PNG: chunk reaches beyond the file");
        if (type == "IHDR") { im.Width = be32(o + 8); im.Height = be32(o + 12); depth = (uint8_t)d[o + 16]; ctype = (uint8_t)d[o + 17]; interlace = (uint8_t)d[o + 20]; }
        else if (type == "PLTE") palette.assign(d.begin() + (long)o + 8, d.begin() + (long)o + 8 + len);
        else if (type == "tRNS") trns.assign(d.begin() + (long)o + 8, d.begin() + (long)o + 8 + len);
        else if (type == "IDAT") idat.append(d, o + 8, len);
        else if (type == "IEND") break;
        o += 12 + (size_t)len;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || depth != 8 || interlace || !im.Width || !im.Height) throw std::runtime_error("PNG: only 8-bit non-interlaced images are decoded here");
    const std::vector<uint8_t> raw = Inflate::zlib((const uint8_t*)idat.data(), idat.size());
    const size_t bpp = (size_t)channels, stride = (size_t)im.Width * bpp;
    if (raw.size() < (stride + 1) * im.Height) throw std::runtime_error("PNG: image data too short");
    std::vector<uint8_t> px(stride * im.Height);
    for (uint32_t y = 0; y < im.Height; y++) {                                      // undo the scanline filters (PNG spec 9.2)
        const uint8_t* src = raw.data() + (stride + 1) * y; const int f = src[0]; src++;
        uint8_t* cur = px.data() + stride * y; const uint8_t* up = y ? cur - stride : nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            if (f == 1) v += a; else if (f == 2) v += b; else if (f == 3) v += (a + b) >> 1;
            else if (f == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (f != 0) throw std::runtime_error("PNG: bad filter type");
            cur[x] = (uint8_t)v;
        }
    }
    im.RGBA.resize((size_t)im.Width * im.Height * 4);
    for (size_t i = 0; i < (size_t)im.Width * im.Height; i++) {
        uint8_t* o = im.RGBA.data() + 4 * i; const uint8_t* s = px.data() + bpp * i;
        if (ctype == 6) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
        else if (ctype == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; }
        else if (ctype == 0) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
        else if (ctype == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
        else { const size_t k = s[0]; if (3 * k + 3 > palette.size()) throw std::runtime_error("PNG: palette index out of range");
               o[0] = palette[3 * k]; o[1] = palette[3 * k + 1]; o[2] = palette[3 * k + 2]; o[3] = k < trns.size() ? trns[k] : 255; }
    }
    return im;
}

// ------------------------------------------------------------------------------------------------
// SimpleMath / DirectXMath conventions (row vectors: v' = v M), in double like ingest.py
// ------------------------------------------------------------------------------------------------
struct M4 { double m[4][4]; };
inline M4 identity() { M4 r{}; for (int i = 0; i < 4; i++) r.m[i][i] = 1.0; return r; }
inline M4 mul(const M4& a, const M4& b)
{
    M4 r{};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0.0; for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
    return r;
}
inline M4 transpose(const M4& a) { M4 r{}; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i]; return r; }
inline M4 rot_x(double a) { const double c = std::cos(a), s = std::sin(a); M4 r = identity(); r.m[1][1] = c; r.m[1][2] = s; r.m[2][1] = -s; r.m[2][2] = c; return r; }
inline M4 rot_y(double a) { const double c = std::cos(a), s = std::sin(a); M4 r = identity(); r.m[0][0] = c; r.m[0][2] = -s; r.m[2][0] = s; r.m[2][2] = c; return r; }
inline M4 rot_z(double a) { const double c = std::cos(a), s = std::sin(a); M4 r = identity(); r.m[0][0] = c; r.m[0][1] = s; r.m[1][0] = -s; r.m[1][1] = c; return r; }
inline M4 matrix_from_quaternion(double x, double y, double z, double w)
{
    M4 r = identity();
    r.m[0][0] = 1 - 2 * (y * y + z * z); r.m[0][1] = 2 * (x * y + z * w); r.m[0][2] = 2 * (x * z - y * w);
    r.m[1][0] = 2 * (x * y - z * w); r.m[1][1] = 1 - 2 * (x * x + z * z); r.m[1][2] = 2 * (y * z + x * w);
    r.m[2][0] = 2 * (x * z + y * w); r.m[2][1] = 2 * (y * z - x * w); r.m[2][2] = 1 - 2 * (x * x + y * y);
    return r;
}
inline double radians(double deg) { return deg * (M_PI / 180.0); }

// JSONConverters.ixx:18-26: {Yaw,Pitch,Roll} in degrees -> CreateFromYawPitchRoll(yaw, -pitch, -roll) (roll about Z first, then pitch about X,
// then yaw about Y); all three zero -> raw quaternion {X,Y,Z,W} (default identity)
inline M4 rotation_from_json(const Json* j)
{
    if (!j || j->kind != Json::Object) return identity();
    const double yaw = j->number("Yaw", 0), pitch = j->number("Pitch", 0), roll = j->number("Roll", 0);
    if (yaw == 0 && pitch == 0 && roll == 0) {
        double q[4] = { j->number("X", 0), j->number("Y", 0), j->number("Z", 0), j->number("W", 1) };
        double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (n == 0.0) n = 1.0;
        return matrix_from_quaternion(q[0] / n, q[1] / n, q[2] / n, q[3] / n);
    }
    return mul(mul(rot_z(radians(-roll)), rot_x(radians(-pitch))), rot_y(radians(yaw)));
}
inline void vec3_from_json(const Json* j, const double def[3], double out[3])
{
    out[0] = def[0]; out[1] = def[1]; out[2] = def[2];
    if (j && j->kind == Json::Object) { out[0] = j->number("X", def[0]); out[1] = j->number("Y", def[1]); out[2] = j->number("Z", def[2]); }
}
// Math::AffineTransform::operator(): Scale * Rotation * Translation (row-vector)
inline M4 affine_from_json(const Json* j)
{
    const double one[3] = { 1, 1, 1 }, zero[3] = { 0, 0, 0 };
    double s[3], t[3];
    vec3_from_json(j ? j->find("Scale") : nullptr, one, s); vec3_from_json(j ? j->find("Translation") : nullptr, zero, t);
    M4 sc = identity(); sc.m[0][0] = s[0]; sc.m[1][1] = s[1]; sc.m[2][2] = s[2];
    M4 m = mul(sc, rotation_from_json(j ? j->find("Rotation") : nullptr));
    M4 tr = identity(); tr.m[3][0] = t[0]; tr.m[3][1] = t[1]; tr.m[3][2] = t[2];
    return mul(m, tr);
}
// XMStoreFloat3x4: the column-vector affine [R|t] = top 3 rows of the transpose
inline void store_float3x4(const M4& rowMajor, float out[12]) { for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (float)rowMajor.m[j][i]; }

// ------------------------------------------------------------------------------------------------
// vertex packing (VertexPositionNormalTangentTexture, Source/Vertex.ixx:38-50)
// ------------------------------------------------------------------------------------------------
struct Vertex { float Position[3]; int16_t Normal[3], Tangent[3]; uint16_t TexCoord[2][2]; };
static_assert(sizeof(Vertex) == 32, "layout");

inline int16_t encode_snorm16(double v)
{
    v = std::fmin(std::fmax(v, -1.0), 1.0) * 32767.0;
    return (int16_t)(v >= 0 ? std::floor(v + 0.5) : std::ceil(v - 0.5));
}
inline uint16_t f32_to_f16(float f)                                      // round to nearest even, like numpy's astype(float16)
{
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, a = x & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);              // NaN
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);             // >= 65536: infinity
    if (a < 0x38800000u) {                                               // below 2^-14: a subnormal half (or zero) = round(|f| * 2^24)
        float af; std::memcpy(&af, &a, 4);
        return (uint16_t)(sign | (uint32_t)std::nearbyintf(af * 16777216.0f));     // exact scaling; nearbyint rounds to nearest even
    }
    const uint32_t mant = a & 0x7FFFFFu, rem = mant & 0x1FFFu;
    uint32_t h = (((a >> 23) - 112u) << 10) | (mant >> 13);              // exponent rebias 127 -> 15
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;              // a carry runs into the exponent, up to infinity: still the right code
    return (uint16_t)(sign | h);
}

// ------------------------------------------------------------------------------------------------
// model: glTF 2.0 (.gltf with data: / external buffers, .glb)
// ------------------------------------------------------------------------------------------------
struct MeshData {
    std::vector<Vertex> Vertices;
    std::vector<uint8_t> Indices; uint32_t IndexStride = 2, IndexCount = 0;      // u16 iff count <= 65535 (GLTFHelpers.ixx:183-188)
    bool HasNormals = false, HasTangents = false, HasUV[2] = { false, false };
    bool HasMaterial = false; PtMaterial Material{};
    // texture slots in the order of Material.ixx:22-33 (BaseColor, EmissiveColor, Metallic, Roughness, MetallicRoughness, Transmission, Normal);
    // textures are shared between the meshes of a model by (image, sRGB) like the reference's and the harness's loaders share them
    std::shared_ptr<struct Texture> Textures[7]; uint32_t TextureCoordinateIndex[7] = { 0, 0, 0, 0, 0, 0, 0 };
    std::vector<std::string> SkippedTextures;                               // slots whose image this host cannot decode (anything but 8-bit PNG)
};
struct Texture { Image Texels; bool SRGB = false; };                        // base-colour / emissive textures are created as *_UNORM_SRGB (GLTFHelpers.ixx:375-391)
enum TextureSlot { BaseColor = 0, EmissiveColor, Metallic, Roughness, MetallicRoughness, Transmission, Normal };
struct MeshNode { std::vector<MeshData> Meshes; M4 GlobalTransform; };        // GlobalTransform reinterpreted as a row-vector matrix (LoadModel)

inline PtMaterial default_material()
{
    PtMaterial m{};                                                            // Material() defaults, Source/Material.ixx:13-19
    m.BaseColor[3] = 1; m.EmissiveStrength = 1; m.Roughness = 0.5f; m.IOR = 1.5f; m.AlphaCutoff = 0.5f;
    return m;
}

class Asset {
public:
    explicit Asset(const std::string& path) : dir(dir_of(path))
    {
        const std::string raw = read_file(path);
        if (raw.size() >= 12 && !raw.compare(0, 4, "glTF")) {
            uint32_t length; std::memcpy(&length, raw.data() + 8, 4);
            size_t off = 12;
            while (off + 8 <= length && off + 8 <= raw.size()) {
                uint32_t clen, ctype; std::memcpy(&clen, raw.data() + off, 4); std::memcpy(&ctype, raw.data() + off + 4, 4);
                if (ctype == 0x4E4F534Au) j = JsonParser(jsonText = raw.substr(off + 8, clen)).parse();
                else if (ctype == 0x004E4942u) { binChunk = raw.substr(off + 8, clen); haveBin = true; }
                off += 8 + (size_t)clen;
            }
        } else j = JsonParser(jsonText = raw).parse();
    }
    Json j;

    const std::string& buffer(size_t i)
    {
        auto it = buffers.find(i);
        if (it != buffers.end()) return it->second;
        const Json& b = j.at("buffers").at(i);
        if (!b.has("uri")) { if (!haveBin) throw std::runtime_error("glTF: buffer without uri and no BIN chunk"); return buffers[i] = binChunk; }
        const std::string& uri = b.at("uri").str;
        if (!uri.compare(0, 5, "data:")) return buffers[i] = base64_decode(uri.substr(uri.find(',') + 1));
        return buffers[i] = read_file(resolve(dir, uri));
    }
    // element (i, c) of an accessor as double (integers as they are; normalised integers divided by their maximum, in float like ingest.py)
    struct Accessor { const uint8_t* data; size_t stride, count; int ncomp, ctype; bool normalized; };
    Accessor accessor(size_t ai)
    {
        const Json& a = j.at("accessors").at(ai);
        const Json& v = j.at("bufferViews").at((size_t)a.at("bufferView").num);
        const std::string& b = buffer((size_t)v.at("buffer").num);
        Accessor r;
        r.ctype = (int)a.at("componentType").num;
        const std::string& t = a.at("type").str;
        r.ncomp = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : t == "MAT4" ? 16 : 0;
        if (!r.ncomp) throw std::runtime_error("glTF: accessor type " + t);
        const size_t elem = (size_t)component_size(r.ctype) * r.ncomp, vstride = (size_t)v.number("byteStride", 0);
        r.stride = (vstride == 0 || vstride == elem) ? elem : vstride;
        r.count = (size_t)a.at("count").num;
        const size_t off = (size_t)v.number("byteOffset", 0) + (size_t)a.number("byteOffset", 0);
        if (off + (r.count ? (r.count - 1) * r.stride + elem : 0) > b.size()) throw std::runtime_error("glTF: accessor reaches beyond its buffer");
        r.data = (const uint8_t*)b.data() + off;
        const Json* n = a.find("normalized");
        r.normalized = n && n->kind == Json::Bool && n->b;
        return r;
    }
    static int component_size(int ctype)
    {
        switch (ctype) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; }
        throw std::runtime_error("glTF: component type " + std::to_string(ctype));
    }
    static double element(const Accessor& a, size_t i, int c)
    {
        const uint8_t* p = a.data + i * a.stride + (size_t)c * component_size(a.ctype);
        double v; float maxv = 1.0f;
        switch (a.ctype) {
        case 5120: { int8_t x; std::memcpy(&x, p, 1); v = x; maxv = 127.0f; break; }
        case 5121: { uint8_t x; std::memcpy(&x, p, 1); v = x; maxv = 255.0f; break; }
        case 5122: { int16_t x; std::memcpy(&x, p, 2); v = x; maxv = 32767.0f; break; }
        case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = x; maxv = 65535.0f; break; }
        case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = x; maxv = 4294967295.0f; break; }
        default: { float x; std::memcpy(&x, p, 4); return x; }
        }
        return a.normalized ? (double)((float)v / maxv) : v;
    }

    // the encoded bytes of image ii (data: / file uri, or a buffer view)
    std::string image_bytes(size_t ii)
    {
        const Json& im = j.at("images").at(ii);
        if (im.has("uri")) {
            const std::string& uri = im.at("uri").str;
            return !uri.compare(0, 5, "data:") ? base64_decode(uri.substr(uri.find(',') + 1)) : read_file(resolve(dir, uri));
        }
        const Json& v = j.at("bufferViews").at((size_t)im.at("bufferView").num);
        const std::string& b = buffer((size_t)v.at("buffer").num);
        const size_t off = (size_t)v.number("byteOffset", 0), len = (size_t)v.at("byteLength").num;
        if (off + len > b.size()) throw std::runtime_error("glTF: image reaches beyond its buffer");
        return b.substr(off, len);
    }

private:
    std::string dir, jsonText, binChunk; bool haveBin = false;
    std::map<size_t, std::string> buffers;
};

// [DirectXMesh spec] ComputeTangentFrame (tangent output only): same operations in the same order as ingest.py's compute_tangents
inline std::vector<std::array<double, 3>> compute_tangents(const std::vector<std::array<float, 3>>& pos, const std::vector<std::array<float, 3>>& nrm,
                                                           const std::vector<std::array<float, 2>>& uv, const std::vector<uint32_t>& idx)
{
    std::vector<std::array<double, 3>> tan(pos.size(), { 0.0, 0.0, 0.0 });
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
        const uint32_t a = idx[t], b = idx[t + 1], c = idx[t + 2];
        double e1[3], e2[3];
        for (int k = 0; k < 3; k++) { e1[k] = (double)pos[b][k] - (double)pos[a][k]; e2[k] = (double)pos[c][k] - (double)pos[a][k]; }
        const double du1 = (double)uv[b][0] - (double)uv[a][0], dv1 = (double)uv[b][1] - (double)uv[a][1];
        const double du2 = (double)uv[c][0] - (double)uv[a][0], dv2 = (double)uv[c][1] - (double)uv[a][1];
        const double det = du1 * dv2 - du2 * dv1;
        if (std::fabs(det) < 1e-20) continue;
        for (int k = 0; k < 3; k++) { const double s = (e1[k] * dv2 - e2[k] * dv1) / det; tan[a][k] += s; tan[b][k] += s; tan[c][k] += s; }
    }
    for (size_t i = 0; i < pos.size(); i++) {
        const double n[3] = { nrm[i][0], nrm[i][1], nrm[i][2] };
        const double d = (n[0] * tan[i][0] + n[1] * tan[i][1]) + n[2] * tan[i][2];
        double t[3] = { tan[i][0] - n[0] * d, tan[i][1] - n[1] * d, tan[i][2] - n[2] * d };
        const double ln = std::sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        double f[3] = { n[1] * 0.0 - n[2] * 1.0, n[2] * 0.0 - n[0] * 0.0, n[0] * 1.0 - n[1] * 0.0 };   // cross(n, (0, 1, 0))
        const double fl = std::sqrt((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
        if (fl > 1e-8) { const double dd = std::fmax(fl, 1e-30); f[0] /= dd; f[1] /= dd; f[2] /= dd; } else { f[0] = 1.0; f[1] = 0.0; f[2] = 0.0; }
        if (ln > 1e-12) { const double dd = std::fmax(ln, 1e-30); tan[i] = { t[0] / dd, t[1] / dd, t[2] / dd }; } else tan[i] = { f[0], f[1], f[2] };
    }
    return tan;
}

// glTF column-vector local matrix of a node
inline M4 node_matrix(const Json& node)
{
    if (node.has("matrix")) { M4 c{}; const Json& m = node.at("matrix"); for (int i = 0; i < 16; i++) c.m[i / 4][i % 4] = m.at((size_t)i).num; return transpose(c); }   // column-major list
    M4 t = identity(), r = identity(), s = identity();
    if (node.has("translation")) for (int k = 0; k < 3; k++) t.m[k][3] = node.at("translation").at((size_t)k).num;
    if (node.has("rotation")) { const Json& q = node.at("rotation"); r = transpose(matrix_from_quaternion(q.at(0).num, q.at(1).num, q.at(2).num, q.at(3).num)); }
    if (node.has("scale")) for (int k = 0; k < 3; k++) s.m[k][k] = node.at("scale").at((size_t)k).num;
    return mul(mul(t, r), s);
}

// GLTFHelpers::LoadModel: one MeshNode per glTF node that carries a mesh, with its global transform
inline std::vector<std::shared_ptr<MeshNode>> load_model(const std::string& path, bool flipWindingOrder = true)
{
    Asset asset(path);
    const Json& j = asset.j;
    std::map<std::pair<size_t, bool>, std::shared_ptr<Texture>> textureCache;          // (image, forced sRGB) -> texture; nullptr: not decodable here
    auto texture_for = [&](const Json& info, bool forceSrgb) -> std::shared_ptr<Texture> {
        const size_t src = (size_t)j.at("textures").at((size_t)info.at("index").num).at("source").num;
        const auto key = std::make_pair(src, forceSrgb);
        auto it = textureCache.find(key);
        if (it != textureCache.end()) return it->second;
        std::shared_ptr<Texture> t;
        const std::string bytes = asset.image_bytes(src);
        if (is_png(bytes)) { try { t = std::make_shared<Texture>(); t->Texels = decode_png(bytes); t->SRGB = forceSrgb; } catch (const std::exception&) { t.reset(); } }
        return textureCache[key] = t;
    };
    auto process_primitive = [&](const Json& prim, MeshData& mesh) -> bool {
        if ((int)prim.number("mode", 4) != 4 || !prim.has("attributes") || !prim.at("attributes").has("POSITION") || !prim.has("indices")) return false;   // :150-152,169-171,191-193
        const Json& attrs = prim.at("attributes");
        const Asset::Accessor pa = asset.accessor((size_t)attrs.at("POSITION").num);
        if (pa.ncomp < 3) throw std::runtime_error("glTF: POSITION is not a VEC3");
        std::vector<std::array<float, 3>> pos(pa.count);
        for (size_t i = 0; i < pa.count; i++) for (int k = 0; k < 3; k++) pos[i][k] = (float)Asset::element(pa, i, k);
        const Asset::Accessor ia = asset.accessor((size_t)prim.at("indices").num);
        std::vector<uint32_t> idx(ia.count);
        for (size_t i = 0; i < ia.count; i++) { idx[i] = (uint32_t)Asset::element(ia, i, 0); if (idx[i] >= pa.count) throw std::runtime_error("glTF: vertex index beyond the POSITION accessor"); }
        if (flipWindingOrder) for (size_t i = 0; i < idx.size() / 2; i++) std::swap(idx[i], idx[idx.size() - 1 - i]);        // slot count-1-i <- index i (:179)
        std::vector<std::array<float, 2>> uv[2];
        for (int s = 0; s < 2; s++) {
            const std::string name = "TEXCOORD_" + std::to_string(s);
            if (!attrs.has(name)) continue;
            const Asset::Accessor ua = asset.accessor((size_t)attrs.at(name).num);
            if (ua.count != pa.count || ua.ncomp < 2) throw std::runtime_error("glTF: " + name + " does not match POSITION");
            uv[s].resize(ua.count);
            for (size_t i = 0; i < ua.count; i++) for (int k = 0; k < 2; k++) uv[s][i][k] = (float)Asset::element(ua, i, k);
            mesh.HasUV[s] = true;
        }
        std::vector<std::array<float, 3>> nrm; std::vector<std::array<double, 3>> tan;
        if (attrs.has("NORMAL")) {
            const Asset::Accessor na = asset.accessor((size_t)attrs.at("NORMAL").num);
            if (na.count != pa.count || na.ncomp < 3) throw std::runtime_error("glTF: NORMAL does not match POSITION");
            nrm.resize(na.count);
            for (size_t i = 0; i < na.count; i++) for (int k = 0; k < 3; k++) nrm[i][k] = (float)Asset::element(na, i, k);
            mesh.HasNormals = true;
            if (mesh.HasUV[0]) { tan = compute_tangents(pos, nrm, uv[0], idx); mesh.HasTangents = true; }     // "Tangent" is never found -> always recomputed (:251-275)
        }
        mesh.Vertices.assign(pos.size(), Vertex{});
        for (size_t i = 0; i < pos.size(); i++) {
            Vertex& v = mesh.Vertices[i];
            for (int k = 0; k < 3; k++) {
                v.Position[k] = pos[i][k];
                if (mesh.HasNormals) v.Normal[k] = encode_snorm16((double)nrm[i][k]);
                if (mesh.HasTangents) v.Tangent[k] = encode_snorm16(tan[i][k]);
            }
            for (int s = 0; s < 2; s++) if (mesh.HasUV[s]) for (int k = 0; k < 2; k++) v.TexCoord[s][k] = f32_to_f16(uv[s][i][k]);
        }
        mesh.IndexCount = (uint32_t)idx.size(); mesh.IndexStride = idx.size() <= 65535 ? 2u : 4u;
        mesh.Indices.resize((size_t)mesh.IndexCount * mesh.IndexStride);
        for (size_t i = 0; i < idx.size(); i++) {
            if (mesh.IndexStride == 2) { const uint16_t x = (uint16_t)idx[i]; std::memcpy(mesh.Indices.data() + 2 * i, &x, 2); }
            else std::memcpy(mesh.Indices.data() + 4 * i, &idx[i], 4);
        }
        if (prim.has("material")) {
            const Json& m = j.at("materials").at((size_t)prim.at("material").num);
            static const Json empty = [] { Json e; e.kind = Json::Object; return e; }();
            const Json& pbr = m.has("pbrMetallicRoughness") ? m.at("pbrMetallicRoughness") : empty;
            const Json& ext = m.has("extensions") ? m.at("extensions") : empty;
            PtMaterial mat = default_material();
            if (pbr.has("baseColorFactor")) for (int k = 0; k < 4; k++) mat.BaseColor[k] = (float)pbr.at("baseColorFactor").at((size_t)k).num;
            else for (int k = 0; k < 4; k++) mat.BaseColor[k] = 1.0f;
            mat.EmissiveStrength = ext.has("KHR_materials_emissive_strength") ? (float)ext.at("KHR_materials_emissive_strength").number("emissiveStrength", 1.0) : 1.0f;
            for (int k = 0; k < 3; k++) mat.EmissiveColor[k] = m.has("emissiveFactor") ? (float)m.at("emissiveFactor").at((size_t)k).num : 0.0f;
            mat.Metallic = (float)pbr.number("metallicFactor", 1.0);
            mat.Roughness = (float)pbr.number("roughnessFactor", 1.0);
            mat.IOR = ext.has("KHR_materials_ior") ? (float)ext.at("KHR_materials_ior").number("ior", 1.5) : 1.5f;
            const std::string am = m.has("alphaMode") ? m.at("alphaMode").str : "OPAQUE";
            mat.AlphaMode = am == "MASK" ? 1u : am == "BLEND" ? 2u : 0u;
            mat.AlphaCutoff = (float)m.number("alphaCutoff", 0.5);
            const Json* tr = ext.find("KHR_materials_transmission");
            if (tr && tr->kind == Json::Object) mat.Transmission = (float)tr->number("transmissionFactor", 0.0);
            mesh.Material = mat; mesh.HasMaterial = true;
            if (mesh.HasUV[0] || mesh.HasUV[1]) {                          // :370-428
                auto slot = [&](TextureSlot k, const char* name, const Json* info, bool srgb) {
                    if (!info || info->kind != Json::Object) return;
                    const int tc = (int)info->number("texCoord", 0);
                    if (tc >= 2 || !mesh.HasUV[tc]) return;
                    if (auto t = texture_for(*info, srgb)) { mesh.Textures[k] = t; mesh.TextureCoordinateIndex[k] = (uint32_t)tc; }
                    else mesh.SkippedTextures.push_back(name);
                };
                slot(BaseColor, "BaseColor", pbr.find("baseColorTexture"), true); slot(EmissiveColor, "EmissiveColor", m.find("emissiveTexture"), true);
                slot(MetallicRoughness, "MetallicRoughness", pbr.find("metallicRoughnessTexture"), false);
                slot(Transmission, "Transmission", tr ? tr->find("transmissionTexture") : nullptr, false);
                if (mesh.HasTangents) slot(Normal, "Normal", m.find("normalTexture"), false);   // a normal map needs the tangent frame
            }
        }
        return true;
    };

    std::vector<std::shared_ptr<MeshNode>> out;
    const Json& scene = j.at("scenes").at((size_t)j.number("scene", 0));
    std::function<void(size_t, const M4&)> visit = [&](size_t ni, const M4& parent) {
        const Json& node = j.at("nodes").at(ni);
        const M4 m = mul(parent, node_matrix(node));
        if (node.has("mesh")) {
            const Json& mj = j.at("meshes").at((size_t)node.at("mesh").num);
            if (mj.has("primitives") && mj.at("primitives").size()) {
                auto mn = std::make_shared<MeshNode>();
                for (const Json& p : mj.at("primitives").arr) { MeshData md; if (process_primitive(p, md)) mn->Meshes.push_back(std::move(md)); }
                mn->GlobalTransform = transpose(m);                       // the column-vector global matrix reinterpreted as a row-vector Matrix
                out.push_back(mn);
            }
        }
        if (node.has("children")) for (const Json& c : node.at("children").arr) visit((size_t)c.num, m);
    };
    if (scene.has("nodes")) for (const Json& n : scene.at("nodes").arr) visit((size_t)n.num, identity());
    return out;
}

// ------------------------------------------------------------------------------------------------
// scene descriptor + Scene::Load / Refresh
// ------------------------------------------------------------------------------------------------
struct RenderObject { uint32_t Node; float Transform[12]; bool IsVisible; std::string Name; };   // Transform: column-vector [R|t], as InstanceData.ObjectToWorld wants it
struct Scene {
    double CameraPosition[3] = { 0, 0, 0 }; M4 CameraRotation = identity();
    float EnvironmentLightColor[4] = { 0, 0, 0, -1 }; float EnvironmentLightTransform[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    std::string EnvironmentLightTexture;                                    // path, not decoded here
    std::vector<std::shared_ptr<MeshNode>> Nodes;                           // one bottom level each
    std::vector<RenderObject> Objects;                                      // one instance each
};

inline Scene load_scene(const std::string& path)
{
    const Json j = JsonParser(read_file(path)).parse();
    const std::string base = dir_of(path);
    Scene sc;
    const double zero[3] = { 0, 0, 0 };
    const Json* cam = j.find("Camera");
    vec3_from_json(cam ? cam->find("Position") : nullptr, zero, sc.CameraPosition);
    sc.CameraRotation = rotation_from_json(cam ? cam->find("Rotation") : nullptr);
    const Json* env = j.find("EnvironmentLight");
    if (env && env->kind == Json::Object) {
        if (const Json* col = env->find("Color"); col && col->kind == Json::Object) {
            sc.EnvironmentLightColor[0] = (float)col->number("R", 0); sc.EnvironmentLightColor[1] = (float)col->number("G", 0);
            sc.EnvironmentLightColor[2] = (float)col->number("B", 0); sc.EnvironmentLightColor[3] = (float)col->number("A", -1);
        } else sc.EnvironmentLightColor[3] = -1;
        store_float3x4(rotation_from_json(env->find("Rotation")), sc.EnvironmentLightTransform);     // App.cpp:1019
        if (env->has("Texture")) sc.EnvironmentLightTexture = resolve(base, env->at("Texture").str);
    }
    std::map<std::string, std::string> models;
    if (const Json* m = j.find("Models"); m && m->kind == Json::Object) for (auto& kv : m->obj) models[kv.first] = resolve(base, kv.second.str);
    std::map<std::string, std::vector<std::shared_ptr<MeshNode>>> loaded;
    M4 zflip = identity(); zflip.m[2][2] = -1.0;
    if (const Json* ros = j.find("RenderObjects"); ros && ros->kind == Json::Array)
        for (const Json& ro : ros->arr) {
            const std::string model = ro.has("Model") ? ro.at("Model").str : "", name = ro.has("Name") ? ro.at("Name").str : "";
            if (!model.empty() && !models.count(model))                    // MyScene.ixx:57-70
                throw std::runtime_error(path + ": " + (name.empty() ? std::string("Unnamed RenderObject") : "RenderObject " + name) + ": Models " + model + " not found");
            if (model.empty()) continue;
            if (!loaded.count(model)) loaded[model] = load_model(models[model], true);      // Scene.ixx:90
            const M4 transform = affine_from_json(ro.find("Transform"));
            const Json* vis = ro.find("IsVisible");
            for (auto& mn : loaded[model]) {
                uint32_t ni = 0;
                for (; ni < sc.Nodes.size(); ni++) if (sc.Nodes[ni] == mn) break;
                if (ni == sc.Nodes.size()) sc.Nodes.push_back(mn);
                RenderObject o; o.Node = ni; o.Name = name; o.IsVisible = !(vis && vis->kind == Json::Bool && !vis->b);
                store_float3x4(mul(mul(mn->GlobalTransform, zflip), transform), o.Transform);     // Scene.ixx:199-214
                sc.Objects.push_back(o);
            }
        }
    return sc;
}

} // namespace ptamd::ingest
