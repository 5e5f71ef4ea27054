// hrt_assets.cpp -- on-disk formats feeding the path: Wavefront OBJ + MTL parser and the TGA / BMP
// texture readers, producing the MeshHost arrays that hrth_scene_load_mesh_instance appends to a scene.
// Host code above the C ABI (include/hrt_host.h); follows the behaviour of
//   Engine/MeshLoaderOBJ.cs:67-277  (OBJ statement handling, fan triangulation, material merge, texture binding)
//   Engine/MeshLoaderOBJ.cs:339-443 (MTL statements)      :520-593 (TGA raw / RLE -> BGRA, bottom-origin flip)
//   Engine/Scene.cs:144-149,654-674 (LoadObjInstance's file half)
// including its quirks: statements are recognised only at column 0, tokens are separated by ' ' only,
// a missing vt index means texcoord 0, `usemtl` of an unknown name creates a default material, textures are
// de-duplicated by case-insensitive path, a missing texture clears the material's map flag.
// Differences, all on inputs where the reference throws or depends on the OS:
//   * '\\' in mtllib / map paths is treated as a directory separator (the reference targets Windows, where it is);
//   * images other than .tga go through System.Drawing in the reference; here .png (all colour types at up to 8 bits per sample,
//     transparency, Adam7; own inflate) and uncompressed 24/32-bit .bmp are read natively and any other format (.jpg: a lossy
//     decoder's output is implementation-defined, GDI+'s cannot be reproduced) is an error (HRTH_ERR_FORMAT), never a substitute;
//   * .NET exceptions (FormatException, InvalidDataException, EndOfStreamException) become error codes + a message.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <climits>
#include <string>
#include <vector>
#include <unordered_map>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include "../../include/hrt_host.h"

namespace {

thread_local std::string g_err;

struct ParseError { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw ParseError{m}; }

bool file_exists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// ---- text helpers with .NET semantics --------------------------------------------------------------

// char.IsWhiteSpace for the Latin-1 range (String.Trim); multi-byte UTF-8 white space is left alone
inline bool is_trim_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0; }
// white space accepted around numbers by NumberStyles.AllowLeadingWhite/AllowTrailingWhite
inline bool is_num_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

struct Span {
    const char* b; const char* e;
    size_t size() const { return (size_t)(e - b); }
    Span trim() const
    {
        const char *x = b, *y = e;
        while (x < y && is_trim_ws((unsigned char)*x)) x++;
        while (y > x && is_trim_ws((unsigned char)y[-1])) y--;
        return Span{x, y};
    }
    std::string str() const { return std::string(b, e); }
};

bool starts_with(const std::string& s, const char* p) { size_t n = std::strlen(p); return s.size() >= n && std::memcmp(s.data(), p, n) == 0; }

// float.Parse(span, CultureInfo.InvariantCulture): NumberStyles.Float | AllowThousands.
// [ws][sign](digits[,digits]*[.digits*] | .digits+)[(e|E)[sign]digits+][ws]  or  [sign]Infinity / NaN.
// The validated text is converted by strtof (correctly rounded, like .NET Core 3.0+); overflow gives +-inf, not an error.
float parse_float(Span s, const char* what)
{
    const char *p = s.b, *e = s.e;
    while (p < e && is_num_ws((unsigned char)*p)) p++;
    while (e > p && (is_num_ws((unsigned char)e[-1]) || e[-1] == '\0')) e--;
    std::string clean;
    const char* q = p;
    if (q < e && (*q == '+' || *q == '-')) { clean.push_back(*q); q++; }
    auto ieq = [&](const char* lit) {
        size_t n = std::strlen(lit);
        if ((size_t)(e - q) != n) return false;
        for (size_t i = 0; i < n; i++) { char c = q[i]; if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); if (c != lit[i]) return false; }
        return true;
    };
    if (ieq("infinity")) return clean == "-" ? -__builtin_inff() : __builtin_inff();
    if (ieq("nan")) return __builtin_nanf("");
    int digits = 0;
    while (q < e && ((*q >= '0' && *q <= '9') || (*q == ',' && digits > 0))) { if (*q != ',') { clean.push_back(*q); digits++; } q++; }
    if (q < e && *q == '.')
    {
        clean.push_back('.'); q++;
        while (q < e && *q >= '0' && *q <= '9') { clean.push_back(*q); digits++; q++; }
    }
    if (digits == 0) fail(std::string("not a number in ") + what + ": '" + s.str() + "'");
    if (q < e && (*q == 'e' || *q == 'E'))
    {
        const char* r = q + 1;
        std::string ex = "e";
        if (r < e && (*r == '+' || *r == '-')) { ex.push_back(*r); r++; }
        int ed = 0;
        while (r < e && *r >= '0' && *r <= '9') { ex.push_back(*r); r++; ed++; }
        if (ed > 0) { clean += ex; q = r; }
    }
    if (q != e) fail(std::string("not a number in ") + what + ": '" + s.str() + "'");
    return std::strtof(clean.c_str(), nullptr);
}

// int.Parse(span, CultureInfo.InvariantCulture): [ws][sign]digits+[ws]; out of Int32 range -> OverflowException
int parse_int(Span s, const char* what)
{
    const char *p = s.b, *e = s.e;
    while (p < e && is_num_ws((unsigned char)*p)) p++;
    while (e > p && (is_num_ws((unsigned char)e[-1]) || e[-1] == '\0')) e--;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); p++; }
    if (p == e) fail(std::string("not an integer in ") + what + ": '" + s.str() + "'");
    long long v = 0;
    for (; p < e; p++)
    {
        if (*p < '0' || *p > '9') fail(std::string("not an integer in ") + what + ": '" + s.str() + "'");
        v = v * 10 + (*p - '0');
        if (v > 2147483648LL) fail(std::string("integer out of range in ") + what + ": '" + s.str() + "'");
    }
    if (neg) v = -v;
    if (v > INT_MAX || v < INT_MIN) fail(std::string("integer out of range in ") + what + ": '" + s.str() + "'");
    return (int)v;
}

inline void skip_spaces(Span s, size_t& i) { while (i < s.size() && s.b[i] == ' ') i++; }        // MeshLoaderOBJ.cs:336
inline size_t next_sep(Span s, size_t i) { while (i < s.size() && s.b[i] != ' ') i++; return i; } // :337

void parse_floats(Span s, float* out, int n, const char* what)     // Parse3 / Parse2 :299-315
{
    size_t i0 = 0;
    for (int k = 0; k < n; k++)
    {
        skip_spaces(s, i0);
        size_t i1 = next_sep(s, i0);
        out[k] = parse_float(Span{s.b + i0, s.b + i1}, what);
        i0 = i1;
    }
}

// StreamReader.ReadLine over the whole file: lines end at \n, \r or \r\n; a UTF-8 BOM is skipped
std::vector<std::string> read_lines(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open '" + path + "'");
    std::stringstream ss; ss << f.rdbuf();
    std::string all = ss.str();
    size_t p = 0;
    if (all.size() >= 3 && (unsigned char)all[0] == 0xEF && (unsigned char)all[1] == 0xBB && (unsigned char)all[2] == 0xBF) p = 3;
    std::vector<std::string> lines;
    size_t start = p;
    for (; p < all.size(); p++)
    {
        char c = all[p];
        if (c == '\n' || c == '\r')
        {
            lines.emplace_back(all, start, p - start);
            if (c == '\r' && p + 1 < all.size() && all[p + 1] == '\n') p++;
            start = p + 1;
        }
    }
    if (start < all.size()) lines.emplace_back(all, start, all.size() - start);
    return lines;
}

// Path.GetDirectoryName(Path.GetFullPath(path)) and Path.Combine(baseDir, rel), with '\\' as a separator too
std::string dir_of(const std::string& path)
{
    std::string full = path;
    if (full.empty() || full[0] != '/')
    {
        char buf[4096];
        if (::getcwd(buf, sizeof buf)) full = std::string(buf) + "/" + path;
    }
    size_t k = full.find_last_of('/');
    return k == std::string::npos ? std::string() : (k == 0 ? std::string("/") : full.substr(0, k));
}
std::string combine(const std::string& base, std::string rel)
{
    for (char& c : rel) if (c == '\\') c = '/';
    if (!rel.empty() && rel[0] == '/') return rel;
    if (base.empty()) return rel;
    return base.back() == '/' ? base + rel : base + "/" + rel;
}
std::string lower_ascii(std::string s) { for (char& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }

// ---- images -------------------------------------------------------------------------------------------

struct Image { int w = 0, h = 0; std::vector<uint8_t> bgra; };

struct ByteReader {
    const std::vector<uint8_t>& d; size_t p = 0; const std::string& file;
    uint8_t u8() { if (p >= d.size()) fail("unexpected end of file in '" + file + "'"); return d[p++]; }
    uint16_t u16() { uint16_t a = u8(); uint16_t b = u8(); return (uint16_t)(a | (b << 8)); }
    uint32_t u32() { uint32_t a = u16(); uint32_t b = u16(); return a | (b << 16); }
    void skip(size_t n) { if (p + n > d.size()) fail("unexpected end of file in '" + file + "'"); p += n; }
};

std::vector<uint8_t> read_bytes(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open '" + path + "'");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

Image load_tga(const std::string& file)                             // LoadTgaBGRA, MeshLoaderOBJ.cs:520-593
{
    std::vector<uint8_t> bytes = read_bytes(file);
    ByteReader br{bytes, 0, file};
    uint8_t idLength = br.u8(), colorMapType = br.u8(), imageType = br.u8();
    br.u16(); br.u16(); br.u8();                                    // colour-map spec
    br.u16(); br.u16();                                             // origin
    int w = br.u16(), h = br.u16();
    uint8_t pixelDepth = br.u8(), imageDesc = br.u8();
    if (idLength > 0) br.skip(std::min<size_t>(idLength, bytes.size() - br.p));     // ReadBytes returns short at EOF
    if (colorMapType != 0) fail("TGA colorMapType=" + std::to_string(colorMapType) + " not supported: " + file);
    const bool topOrigin = (imageDesc & 0x20) != 0;
    const int bpp = pixelDepth == 32 ? 4 : (pixelDepth == 24 ? 3 : (pixelDepth == 8 ? 1 : 0));
    if (bpp == 0) fail("TGA pixelDepth=" + std::to_string(pixelDepth) + " not supported: " + file);
    Image img; img.w = w; img.h = h; img.bgra.assign((size_t)w * h * 4, 0);
    auto read_px = [&](uint8_t px[4]) {
        if (bpp == 4) { px[0] = br.u8(); px[1] = br.u8(); px[2] = br.u8(); px[3] = br.u8(); }
        else if (bpp == 3) { px[0] = br.u8(); px[1] = br.u8(); px[2] = br.u8(); px[3] = 255; }
        else { uint8_t y = br.u8(); px[0] = px[1] = px[2] = y; px[3] = 255; }
    };
    auto write_px = [&](int i, const uint8_t px[4]) {
        int x = i % w, y = i / w;
        int yOut = topOrigin ? y : (h - 1 - y);
        std::memcpy(&img.bgra[((size_t)yOut * w + x) * 4], px, 4);
    };
    const int total = w * h;
    uint8_t px[4];
    if (imageType == 2 || imageType == 3)
    {
        for (int i = 0; i < total; i++) { read_px(px); write_px(i, px); }
    }
    else if (imageType == 10)
    {
        int i = 0;
        while (i < total)
        {
            uint8_t packet = br.u8();
            int count = (packet & 0x7F) + 1;
            if (packet & 0x80) { read_px(px); for (int k = 0; k < count && i < total; k++, i++) write_px(i, px); }
            else for (int k = 0; k < count && i < total; k++, i++) { read_px(px); write_px(i, px); }
        }
    }
    else fail("TGA imageType=" + std::to_string(imageType) + " not supported: " + file);
    return img;
}

// Uncompressed 24/32-bit Windows bitmaps, as `new Bitmap(file)` + LockBits(Format32bppArgb) delivers them
// (MeshLoaderOBJ.cs:463-511): rows top-down in memory, B,G,R,A; 24-bit sources get A = 255.
Image load_bmp(const std::string& file)
{
    std::vector<uint8_t> bytes = read_bytes(file);
    ByteReader br{bytes, 0, file};
    if (br.u8() != 'B' || br.u8() != 'M') fail("not a BMP file: " + file);
    br.u32(); br.u32();
    uint32_t dataOff = br.u32(), hdr = br.u32();
    if (hdr < 40) fail("BMP header size " + std::to_string(hdr) + " not supported: " + file);
    int32_t w = (int32_t)br.u32(), hs = (int32_t)br.u32();
    br.u16();
    uint16_t bits = br.u16();
    uint32_t comp = br.u32();
    if (!(bits == 24 || bits == 32) || !(comp == 0 || (comp == 3 && bits == 32)) || w <= 0 || hs == 0)
        fail("BMP " + std::to_string(bits) + " bpp / compression " + std::to_string(comp) + " not supported: " + file);
    bool hasAlpha = false;
    if (comp == 3)
    {   // BI_BITFIELDS: only the B,G,R,A byte order is taken; alpha counts when its mask is present
        br.p = 14 + 40;
        uint32_t mr = br.u32(), mg = br.u32(), mb = br.u32(), ma = hdr >= 56 ? br.u32() : 0u;
        if (mr != 0x00FF0000u || mg != 0x0000FF00u || mb != 0x000000FFu) fail("BMP channel masks not supported: " + file);
        hasAlpha = ma == 0xFF000000u;
    }
    const int h = hs < 0 ? -hs : hs;
    const bool bottomUp = hs > 0;
    const size_t bpp = bits / 8, stride = ((size_t)w * bpp + 3) & ~(size_t)3;
    if ((size_t)dataOff + stride * h > bytes.size()) fail("unexpected end of file in '" + file + "'");
    Image img; img.w = w; img.h = h; img.bgra.resize((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
    {
        const uint8_t* src = &bytes[dataOff + stride * (size_t)(bottomUp ? h - 1 - y : y)];
        uint8_t* dst = &img.bgra[(size_t)y * w * 4];
        for (int x = 0; x < w; x++)
        {
            dst[4 * x + 0] = src[bpp * x + 0]; dst[4 * x + 1] = src[bpp * x + 1]; dst[4 * x + 2] = src[bpp * x + 2];
            dst[4 * x + 3] = hasAlpha ? src[4 * x + 3] : 255;
        }
    }
    return img;
}

// ---- PNG ------------------------------------------------------------------------------------------------
// What `new Bitmap(file)` + LockBits(Format32bppArgb) delivers for a PNG (MeshLoaderOBJ.cs:463-511): rows top-down, B,G,R,A.
// Decoder written from the PNG (ISO/IEC 15948) and DEFLATE / zlib (RFC 1951 / 1950) specifications: all colour types at 1-8 bits
// per sample, palette transparency (tRNS), colour-key transparency of grey / RGB images, Adam7 interlace, CRC-32 and Adler-32
// verified.  16 bits per sample fail loudly (how GDI+ narrows them is not something this repository can pin), ancillary colour
// chunks (gAMA, sRGB, iCCP, cHRM) are ignored: samples are delivered as stored.  PARITY UNPINNED like the rest of the loader.
struct Inflater {
    const uint8_t* in; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; const std::string& file;
    std::vector<uint8_t> out;
    size_t limit = (size_t)-1;          // the scanlines the header announces: a stream that inflates to more is damaged (or a decompression bomb)
    void room(size_t more) { if (out.size() + more > limit) bad("compressed stream holds more than the image's scanlines"); }
    [[noreturn]] void bad(const char* what) { fail(std::string("PNG: ") + what + ": " + file); }
    int bits(int need)
    {
        uint32_t v = bitbuf;
        while (bitcnt < need) { if (pos >= n) bad("compressed stream ends early"); v |= (uint32_t)in[pos++] << bitcnt; bitcnt += 8; }
        bitbuf = need == 32 ? 0 : v >> need; bitcnt -= need;
        return (int)(v & ((need == 32 ? 0u : (1u << need)) - 1u));
    }
    struct Huff { short count[16]; short symbol[288]; };
    static bool build(Huff& h, const short* length, int n)
    {
        for (int i = 0; i < 16; i++) h.count[i] = 0;
        for (int i = 0; i < n; i++) h.count[length[i]]++;
        if (h.count[0] == n) return true;                            // no codes: legal for an unused distance tree
        int left = 1;
        for (int len = 1; len < 16; len++) { left <<= 1; left -= h.count[len]; if (left < 0) return false; }
        short offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (short)(offs[len] + h.count[len]);
        for (int i = 0; i < n; i++) if (length[i] != 0) h.symbol[offs[length[i]]++] = (short)i;
        return true;
    }
    int decode(const Huff& h)
    {   // canonical code, one bit at a time (RFC 1951 3.2.2)
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++)
        {
            code |= bits(1);
            const int count = h.count[len];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count; first += count; first <<= 1; code <<= 1;
        }
        bad("invalid Huffman code");
    }
    void codes(const Huff& lencode, const Huff& distcode)
    {
        static const short lens[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const short dists[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;)
        {
            int sym = decode(lencode);
            if (sym < 256) { room(1); out.push_back((uint8_t)sym); continue; }
            if (sym == 256) return;
            sym -= 257;
            if (sym >= 29) bad("invalid length symbol");
            const int len = lens[sym] + bits(lext[sym]);
            const int ds = decode(distcode);
            if (ds >= 30) bad("invalid distance symbol");
            const size_t dist = (size_t)dists[ds] + (size_t)bits(dext[ds]);
            if (dist > out.size()) bad("distance reaches before the start of the output");
            room((size_t)len);
            for (int k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
        }
    }
    void run()
    {   // zlib wrapper (RFC 1950) around a DEFLATE stream
        if (n < 6) bad("compressed stream too short");
        const int cmf = in[0], flg = in[1];
        if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) bad("not a zlib stream");
        pos = 2;
        int last;
        do
        {
            last = bits(1);
            const int type = bits(2);
            if (type == 0)
            {
                bitbuf = 0; bitcnt = 0;
                if (pos + 4 > n) bad("stored block header ends early");
                const unsigned len = in[pos] | (in[pos + 1] << 8), nlen = in[pos + 2] | (in[pos + 3] << 8);
                pos += 4;
                if ((len ^ 0xFFFFu) != nlen || pos + len > n) bad("stored block is damaged");
                room((size_t)len);
                out.insert(out.end(), in + pos, in + pos + len); pos += len;
            }
            else if (type == 1)
            {
                short l[288]; Huff lc, dc;
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                build(lc, l, 288);
                short d[30]; for (int i = 0; i < 30; i++) d[i] = 5;
                build(dc, d, 30);
                codes(lc, dc);
            }
            else if (type == 2)
            {
                static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = bits(5) + 257, ndist = bits(5) + 1, ncode = bits(4) + 4;
                if (nlen > 286 || ndist > 30) bad("too many codes");
                short l[320];
                for (int i = 0; i < 19; i++) l[i] = 0;
                for (int i = 0; i < ncode; i++) l[order[i]] = (short)bits(3);
                Huff cl;
                if (!build(cl, l, 19)) bad("invalid code-length code");
                int idx = 0;
                while (idx < nlen + ndist)
                {
                    int sym = decode(cl);
                    if (sym < 16) l[idx++] = (short)sym;
                    else
                    {
                        int prev = 0, rep;
                        if (sym == 16) { if (idx == 0) bad("repeat without a previous length"); prev = l[idx - 1]; rep = 3 + bits(2); }
                        else if (sym == 17) rep = 3 + bits(3);
                        else rep = 11 + bits(7);
                        if (idx + rep > nlen + ndist) bad("code lengths overrun");
                        while (rep--) l[idx++] = (short)prev;
                    }
                }
                if (l[256] == 0) bad("no end-of-block code");
                Huff lc, dc;
                if (!build(lc, l, nlen) || !build(dc, l + nlen, ndist)) bad("over-subscribed Huffman code");
                codes(lc, dc);
            }
            else bad("invalid block type");
        } while (!last);
        bitbuf = 0; bitcnt = 0;
        if (pos + 4 > n) bad("Adler-32 missing");
        uint32_t a = 1, b = 0;
        for (uint8_t v : out) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
        const uint32_t want = ((uint32_t)in[pos] << 24) | ((uint32_t)in[pos + 1] << 16) | ((uint32_t)in[pos + 2] << 8) | in[pos + 3];
        if (((b << 16) | a) != want) bad("Adler-32 mismatch");
    }
};

uint32_t crc32_png(const uint8_t* p, size_t n)
{
    static uint32_t table[256]; static bool ready = false;
    if (!ready) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } ready = true; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

Image load_png(const std::string& file)
{
    const std::vector<uint8_t> bytes = read_bytes(file);
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (bytes.size() < 8 || std::memcmp(bytes.data(), sig, 8) != 0) fail("not a PNG file: " + file);
    auto be32 = [&](size_t p) { return ((uint32_t)bytes[p] << 24) | ((uint32_t)bytes[p + 1] << 16) | ((uint32_t)bytes[p + 2] << 8) | bytes[p + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool seenEnd = false;
    for (size_t p = 8; !seenEnd;)
    {
        if (p + 12 > bytes.size()) fail("PNG: chunk list ends early: " + file);
        const uint32_t len = be32(p);
        if ((size_t)len > bytes.size() - p - 12) fail("PNG: chunk runs past the end of the file: " + file);
        const std::string type((const char*)&bytes[p + 4], 4);
        const uint8_t* data = &bytes[p + 8];
        if (crc32_png(&bytes[p + 4], (size_t)len + 4) != be32(p + 8 + len)) fail("PNG: CRC mismatch in chunk " + type + ": " + file);
        if (type == "IHDR")
        {
            if (len != 13) fail("PNG: bad IHDR: " + file);
            w = be32(p + 8); h = be32(p + 12); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) fail("PNG: unknown compression / filter / interlace method: " + file);
        }
        else if (type == "PLTE") plte.assign(data, data + len);
        else if (type == "tRNS") trns.assign(data, data + len);
        else if (type == "IDAT") idat.insert(idat.end(), data, data + len);
        else if (type == "IEND") seenEnd = true;
        else if (!(type[0] & 32)) fail("PNG: unknown critical chunk " + type + ": " + file);
        p += (size_t)len + 12;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depthOk = depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4));
    if (w == 0 || h == 0 || w > 32768u || h > 32768u || channels == 0) fail("PNG: bad header: " + file);
    if (depth == 16) fail("PNG: 16 bits per sample are not supported: " + file);
    if (!depthOk) fail("PNG: bit depth " + std::to_string(depth) + " is not valid for colour type " + std::to_string(ctype) + ": " + file);
    if (ctype == 3 && (plte.empty() || plte.size() % 3 != 0)) fail("PNG: palette image without a valid PLTE chunk: " + file);
    const int bitsPerPixel = channels * depth, bpp = std::max(1, bitsPerPixel / 8);
    // bytes of filtered scanlines the header announces (per Adam7 pass: rows x (1 + stride)); nothing is allocated for the picture
    // before the stream has delivered them, and the stream may not deliver more
    size_t expected = 0;
    {
        static const uint32_t px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
        auto add = [&](uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
            if (x0 >= w || y0 >= h) return;
            const size_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
            expected += ph * (1 + (pw * (size_t)bitsPerPixel + 7) / 8);
        };
        if (interlace) for (int i = 0; i < 7; i++) add(px0[i], py0[i], pdx[i], pdy[i]);
        else add(0, 0, 1, 1);
    }
    if (expected > ((size_t)1 << 30)) fail("PNG: image larger than 1 GiB of scanlines: " + file);
    Inflater inf{idat.data(), idat.size(), 0, 0, 0, file, {}};
    inf.limit = expected;
    inf.run();
    const std::vector<uint8_t>& raw = inf.out;
    if (raw.size() < expected) fail("PNG: image data ends early: " + file);
    Image img; img.w = (int)w; img.h = (int)h; img.bgra.assign((size_t)w * h * 4, 0);
    auto put = [&](uint32_t x, uint32_t y, const uint8_t* line, uint32_t xi) {       // sample(s) of pixel xi of a defiltered scanline
        uint8_t s[4] = {0, 0, 0, 255};
        if (depth == 8) for (int c = 0; c < channels; c++) s[c] = line[(size_t)xi * channels + c];
        else { const int per = 8 / depth; const uint8_t b = line[xi / per]; s[0] = (uint8_t)((b >> ((per - 1 - (int)(xi % per)) * depth)) & ((1 << depth) - 1)); }
        uint8_t* o = &img.bgra[((size_t)y * w + x) * 4];
        if (ctype == 3)
        {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) fail("PNG: palette index out of range: " + file);
            o[2] = plte[s[0] * 3]; o[1] = plte[s[0] * 3 + 1]; o[0] = plte[s[0] * 3 + 2];
            o[3] = s[0] < trns.size() ? trns[s[0]] : 255;
        }
        else if (ctype == 0 || ctype == 4)
        {
            const int maxv = (1 << depth) - 1;
            const uint8_t g = (uint8_t)(depth == 8 ? s[0] : s[0] * 255 / maxv);                // 1 / 2 / 4-bit greys scale to 0..255
            o[0] = o[1] = o[2] = g;
            o[3] = ctype == 4 ? s[1] : ((trns.size() >= 2 && (((unsigned)trns[0] << 8) | trns[1]) == (unsigned)s[0]) ? 0 : 255);
        }
        else
        {
            o[2] = s[0]; o[1] = s[1]; o[0] = s[2];
            if (ctype == 6) o[3] = s[3];
            else o[3] = (trns.size() >= 6 && trns[1] == s[0] && trns[3] == s[1] && trns[5] == s[2] && trns[0] == 0 && trns[2] == 0 && trns[4] == 0) ? 0 : 255;
        }
    };
    size_t at = 0;
    std::vector<uint8_t> prev, cur;
    auto pass = [&](uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
        if (x0 >= w || y0 >= h) return;
        const uint32_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8;
        prev.assign(stride, 0); cur.resize(stride);
        for (uint32_t r = 0; r < ph; r++)
        {
            if (at + 1 + stride > raw.size()) fail("PNG: image data ends early: " + file);
            const int ft = raw[at++];
            if (ft > 4) fail("PNG: unknown filter type: " + file);
            for (size_t i = 0; i < stride; i++)
            {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pred = 0;
                if (ft == 1) pred = a;
                else if (ft == 2) pred = b;
                else if (ft == 3) pred = (a + b) >> 1;
                else if (ft == 4) { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
                cur[i] = (uint8_t)(raw[at + i] + pred);
            }
            at += stride;
            for (uint32_t xi = 0; xi < pw; xi++) put(x0 + xi * dx, y0 + r * dy, cur.data(), xi);
            prev.swap(cur);
        }
    };
    if (interlace == 0) pass(0, 0, 1, 1);
    else
    {
        static const uint32_t px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7; k++) pass(px0[k], py0[k], pdx[k], pdy[k]);
    }
    return img;
}

Image load_texture(const std::string& file)                         // LoadTextureBGRA :456-479
{
    size_t dot = file.find_last_of('.');
    size_t slash = file.find_last_of('/');
    std::string ext = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? lower_ascii(file.substr(dot)) : std::string();
    if (ext == ".tga") return load_tga(file);
    if (ext == ".bmp") return load_bmp(file);
    if (ext == ".png") return load_png(file);
    fail("image format '" + ext + "' is not supported (only .tga, .png at up to 8 bits per sample and uncompressed .bmp): " + file);
}

// ---- OBJ / MTL ------------------------------------------------------------------------------------------

hrt_material default_material()                                     // DefaultMaterial :284-295
{
    hrt_material m; std::memset(&m, 0, sizeof m);
    m.Kd.X = m.Kd.Y = m.Kd.Z = 0.8f;
    m.HasDiffuseMap = 0; m.DiffuseTexIndex = -1; m.Shading = 0; m.IOR = 1.f;
    m.HasAlphaMap = 0; m.AlphaTexIndex = -1; m.TwoSided = 0; m.AlphaCutoff = 0.5f;
    return m;
}

// Dictionary<string, T> as the loader uses it: insertion-ordered enumeration, assignment keeps the slot
template <class V> struct OrderedMap {
    std::vector<std::pair<std::string, V>> items;
    std::unordered_map<std::string, size_t> pos;
    V* find(const std::string& k) { auto it = pos.find(k); return it == pos.end() ? nullptr : &items[it->second].second; }
    void set(const std::string& k, const V& v)
    {
        auto it = pos.find(k);
        if (it == pos.end()) { pos.emplace(k, items.size()); items.emplace_back(k, v); }
        else items[it->second].second = v;
    }
};

struct Mesh {
    std::vector<hrt_float3> positions;
    std::vector<hrt_mesh_tri> triangles;
    std::vector<hrt_float2> texcoords;
    std::vector<hrt_mesh_tri_uv> triUVs;
    std::vector<int32_t> triMaterial;
    std::vector<hrt_material> materials;
    std::vector<std::string> materialNames;
    std::vector<Image> textures;
    std::vector<std::string> texturePaths;
    // flattened views for hrth_mesh_get
    std::vector<int32_t> texW, texH;
    std::vector<uint8_t> texBGRA;
};

int parse_one_index(Span s, int countSoFar)                         // ParseOneIndex :330-334
{
    int v = parse_int(s, "face index");
    return v > 0 ? v - 1 : countSoFar + v;
}

void parse_face_vvt(Span tok, int vCount, int tCount, int& v, int& t)   // ParseFaceVVT :311-328
{
    const char* s1 = (const char*)std::memchr(tok.b, '/', tok.size());
    if (!s1) { v = parse_one_index(tok, vCount); t = 0; return; }
    v = parse_one_index(Span{tok.b, s1}, vCount);
    const char* rest = s1 + 1;
    const char* s2 = (const char*)std::memchr(rest, '/', (size_t)(tok.e - rest));
    if (!s2) t = parse_one_index(Span{rest, tok.e}, tCount);
    else t = (s2 > rest) ? parse_one_index(Span{rest, s2}, tCount) : 0;
}

void load_mtl(const std::string& mtlPath, const std::string& baseDir, OrderedMap<hrt_material>& dict,
              OrderedMap<std::string>& diffusePaths, OrderedMap<std::string>& alphaPaths)        // LoadMtl :339-443
{
    bool haveCur = false; std::string cur;
    hrt_material m = default_material();
    for (const std::string& line : read_lines(mtlPath))
    {
        if (line.empty() || line[0] == '#') continue;
        const char* b = line.data(); const char* e = b + line.size();
        if (starts_with(line, "newmtl "))
        {
            if (haveCur) dict.set(cur, m);
            cur = Span{b + 7, e}.trim().str(); haveCur = true;
            m = default_material();
        }
        else if (starts_with(line, "Kd "))
        {
            float k[3]; parse_floats(Span{b + 3, e}.trim(), k, 3, "Kd");
            m.Kd.X = k[0]; m.Kd.Y = k[1]; m.Kd.Z = k[2];
        }
        else if (starts_with(line, "map_Kd "))
        {
            std::string raw = Span{b + 7, e}.trim().str();
            if (haveCur) diffusePaths.set(cur, combine(baseDir, raw));
            m.HasDiffuseMap = 1;
        }
        else if (starts_with(line, "map_d "))
        {
            std::string raw = Span{b + 6, e}.trim().str();
            if (haveCur) alphaPaths.set(cur, combine(baseDir, raw));
            m.HasAlphaMap = 1; m.TwoSided = 1;
        }
        else if (starts_with(line, "d "))
        {
            float d = parse_float(Span{b + 2, e}.trim(), "d");
            if (d < 0.999f) { m.TwoSided = 1; m.AlphaCutoff = 0.5f; }
        }
        else if (starts_with(line, "Tr "))
        {
            float tr = parse_float(Span{b + 3, e}.trim(), "Tr");
            float d = 1.f - tr;
            if (d < 0.999f) { m.TwoSided = 1; m.AlphaCutoff = 0.5f; }
        }
        else if (starts_with(line, "Ni "))
        {
            Span s = Span{b + 3, e}.trim();
            size_t i0 = 0; skip_spaces(s, i0);
            size_t i1 = next_sep(s, i0);
            m.IOR = parse_float(Span{s.b + i0, s.b + i1}, "Ni");
            if (m.IOR <= 0.f) m.IOR = 1.f;
        }
        else if (starts_with(line, "illum "))
        {
            int model = parse_int(Span{b + 6, e}.trim(), "illum");
            m.Shading = model >= 5 ? 2 : (model >= 3 ? 1 : 0);
        }
    }
    if (haveCur) dict.set(cur, m);
}

Mesh* load_obj(const std::string& path, float scale, bool flipWinding)      // MeshLoaderOBJ.Load :67-277
{
    std::vector<std::string> lines = read_lines(path);
    const std::string baseDir = dir_of(path);
    Mesh* mesh = new Mesh();
    struct Guard { Mesh*& m; bool keep = false; ~Guard() { if (!keep) { delete m; m = nullptr; } } } guard{mesh};

    std::vector<int> faceV, faceT;
    std::string mtlLibPath; bool haveMtlLib = false;
    int currentMtl = -1;
    OrderedMap<int> mtlNameToIndex;

    for (const std::string& line : lines)
    {
        if (line.empty() || line[0] == '#') continue;
        const char* b = line.data(); const char* e = b + line.size();
        if (starts_with(line, "v "))
        {
            float p[3]; parse_floats(Span{b + 2, e}.trim(), p, 3, "v");
            mesh->positions.push_back(hrt_float3{p[0] * scale, p[1] * scale, p[2] * scale});
        }
        else if (starts_with(line, "vt "))
        {
            float t[2]; parse_floats(Span{b + 3, e}.trim(), t, 2, "vt");
            mesh->texcoords.push_back(hrt_float2{t[0], t[1]});
        }
        else if (starts_with(line, "f "))
        {
            faceV.clear(); faceT.clear();
            Span s = Span{b + 2, e}.trim();
            size_t i = 0;
            while (i < s.size())
            {
                while (i < s.size() && s.b[i] == ' ') i++;
                if (i >= s.size()) break;
                size_t j = i; while (j < s.size() && s.b[j] != ' ') j++;
                if (j > i)
                {
                    int vi, ti;
                    parse_face_vvt(Span{s.b + i, s.b + j}, (int)mesh->positions.size(), (int)mesh->texcoords.size(), vi, ti);
                    faceV.push_back(vi); faceT.push_back(ti);
                }
                i = j + 1;
            }
            if (faceV.size() >= 3)
                for (size_t k = 1; k + 1 < faceV.size(); k++)
                {
                    if (!flipWinding) { mesh->triangles.push_back(hrt_mesh_tri{faceV[0], faceV[k], faceV[k + 1]}); mesh->triUVs.push_back(hrt_mesh_tri_uv{faceT[0], faceT[k], faceT[k + 1]}); }
                    else              { mesh->triangles.push_back(hrt_mesh_tri{faceV[0], faceV[k + 1], faceV[k]}); mesh->triUVs.push_back(hrt_mesh_tri_uv{faceT[0], faceT[k + 1], faceT[k]}); }
                    mesh->triMaterial.push_back(currentMtl < 0 ? 0 : currentMtl);
                }
        }
        else if (starts_with(line, "mtllib "))
        {
            std::string rel = Span{b + 7, e}.trim().str();
            if (!rel.empty()) { mtlLibPath = combine(baseDir, rel); haveMtlLib = true; }
        }
        else if (starts_with(line, "usemtl "))
        {
            std::string name = Span{b + 7, e}.trim().str();
            if (!name.empty())
            {
                if (int* idx = mtlNameToIndex.find(name)) currentMtl = *idx;
                else
                {   // Dictionary.TryGetValue leaves `currentMtl` at default(int) before it is reassigned: same end state
                    currentMtl = (int)mesh->materials.size();
                    mtlNameToIndex.set(name, currentMtl);
                    mesh->materials.push_back(default_material());
                }
            }
        }
    }

    // merge the MTL library (:183-205)
    OrderedMap<std::string> materialTexPath, alphaTexPath;          // keyed by decimal material index, insertion-ordered
    if (haveMtlLib && file_exists(mtlLibPath))
    {
        OrderedMap<hrt_material> loaded; OrderedMap<std::string> diffuseMap, alphaMap;
        load_mtl(mtlLibPath, baseDir, loaded, diffuseMap, alphaMap);
        for (auto& kv : loaded.items)
        {
            if (int* idx = mtlNameToIndex.find(kv.first)) mesh->materials[(size_t)*idx] = kv.second;
            else { int ni = (int)mesh->materials.size(); mtlNameToIndex.set(kv.first, ni); mesh->materials.push_back(kv.second); }
        }
        for (auto& kv : diffuseMap.items) if (int* mi = mtlNameToIndex.find(kv.first)) materialTexPath.set(std::to_string(*mi), kv.second);
        for (auto& kv : alphaMap.items)   if (int* mi = mtlNameToIndex.find(kv.first)) alphaTexPath.set(std::to_string(*mi), kv.second);
    }

    // bind textures (:207-262); one local texture per distinct (case-insensitive) path
    std::unordered_map<std::string, int> texPathToIndex;
    auto bind = [&](OrderedMap<std::string>& paths, bool alpha) {
        for (auto& kv : paths.items)
        {
            hrt_material& mr = mesh->materials[(size_t)std::atoi(kv.first.c_str())];
            const std::string& p = kv.second;
            auto it = texPathToIndex.find(lower_ascii(p));
            int texIndex;
            if (it == texPathToIndex.end())
            {
                if (!file_exists(p))
                {
                    if (alpha) { mr.HasAlphaMap = 0; mr.AlphaTexIndex = -1; } else { mr.HasDiffuseMap = 0; mr.DiffuseTexIndex = -1; }
                    continue;
                }
                texIndex = (int)mesh->textures.size();
                mesh->textures.push_back(load_texture(p));
                mesh->texturePaths.push_back(p);
                texPathToIndex.emplace(lower_ascii(p), texIndex);
            }
            else texIndex = it->second;
            if (alpha) { mr.HasAlphaMap = 1; mr.AlphaTexIndex = texIndex; mr.TwoSided = 1; }
            else { mr.HasDiffuseMap = 1; mr.DiffuseTexIndex = texIndex; }
        }
    };
    bind(materialTexPath, false);
    bind(alphaTexPath, true);

    mesh->materialNames.assign(mesh->materials.size(), std::string());
    for (auto& kv : mtlNameToIndex.items) mesh->materialNames[(size_t)kv.second] = kv.first;
    for (const Image& im : mesh->textures)
    {
        mesh->texW.push_back(im.w); mesh->texH.push_back(im.h);
        mesh->texBGRA.insert(mesh->texBGRA.end(), im.bgra.begin(), im.bgra.end());
    }
    guard.keep = true;
    return mesh;
}

template <class F> int guarded(F f)
{
    try { g_err.clear(); return f(); }
    catch (const ParseError& e) { g_err = e.msg; return HRTH_ERR_FORMAT; }
    catch (const std::bad_alloc&) { g_err = "out of host memory"; return HRTH_ERR_FORMAT; }
}

} // namespace

extern "C" {

const char* hrth_last_error(void) { return g_err.c_str(); }

int hrth_mesh_load_obj(const char* path, float scale, int flipWinding, void** out)
{
    if (!out) return HRTH_ERR_ARGUMENT;
    *out = nullptr;
    if (!path || !*path) { g_err = "path is empty"; return HRTH_ERR_ARGUMENT; }
    if (!file_exists(path)) { g_err = std::string("OBJ file not found: ") + path; return HRTH_ERR_NOT_FOUND; }
    return guarded([&] { *out = load_obj(path, scale, flipWinding != 0); return 0; });
}

void hrth_mesh_free(void* mesh) { delete static_cast<Mesh*>(mesh); }

int hrth_mesh_get(void* mesh_, hrth_mesh_desc* d)
{
    Mesh* m = static_cast<Mesh*>(mesh_);
    if (!m || !d) return HRTH_ERR_ARGUMENT;
    std::memset(d, 0, sizeof *d);
    d->positions = m->positions.data(); d->n_positions = (int)m->positions.size();
    d->triangles = m->triangles.data(); d->n_triangles = (int)m->triangles.size();
    d->texcoords = m->texcoords.data(); d->n_texcoords = (int)m->texcoords.size();
    d->tri_uvs = m->triUVs.data();
    d->tri_material_index = m->triMaterial.data(); d->n_tri_material_index = (int)m->triMaterial.size();
    d->materials = m->materials.data(); d->n_materials = (int)m->materials.size();
    d->tex_w = m->texW.data(); d->tex_h = m->texH.data(); d->tex_bgra = m->texBGRA.data(); d->n_textures = (int)m->textures.size();
    return 0;
}

const char* hrth_mesh_material_name(void* mesh_, int i)
{
    Mesh* m = static_cast<Mesh*>(mesh_);
    return (m && i >= 0 && (size_t)i < m->materialNames.size()) ? m->materialNames[(size_t)i].c_str() : nullptr;
}

const char* hrth_mesh_texture_path(void* mesh_, int i)
{
    Mesh* m = static_cast<Mesh*>(mesh_);
    return (m && i >= 0 && (size_t)i < m->texturePaths.size()) ? m->texturePaths[(size_t)i].c_str() : nullptr;
}

int hrth_image_load(const char* path, int* w, int* h, uint8_t** bgra)
{
    if (!path || !w || !h || !bgra) return HRTH_ERR_ARGUMENT;
    *bgra = nullptr; *w = *h = 0;
    if (!file_exists(path)) { g_err = std::string("image file not found: ") + path; return HRTH_ERR_NOT_FOUND; }
    return guarded([&] {
        Image im = load_texture(path);
        *w = im.w; *h = im.h;
        *bgra = static_cast<uint8_t*>(std::malloc(im.bgra.size() ? im.bgra.size() : 1));
        if (!*bgra) throw std::bad_alloc();
        std::memcpy(*bgra, im.bgra.data(), im.bgra.size());
        return 0;
    });
}

void hrth_image_free(uint8_t* bgra) { std::free(bgra); }

int hrth_scene_load_obj_instance(void* scene, const char* objPath, const hrt_affine3x4* objectToWorld, float uniformScale)   // Scene.cs:144-256
{
    if (!scene || !objectToWorld) return HRTH_ERR_ARGUMENT;
    void* mesh = nullptr;
    int rc = hrth_mesh_load_obj(objPath, uniformScale, /*flipWinding=*/0, &mesh);
    if (rc != 0) return rc;
    hrth_mesh_desc d;
    hrth_mesh_get(mesh, &d);
    int inst = hrth_scene_load_mesh_instance(scene, d.positions, d.n_positions, d.triangles, d.n_triangles, d.texcoords, d.n_texcoords,
                                             d.tri_uvs, d.tri_material_index, d.n_tri_material_index, d.materials, d.n_materials,
                                             d.tex_w, d.tex_h, d.tex_bgra, d.n_textures, objectToWorld);
    hrth_mesh_free(mesh);
    if (inst < 0) { g_err = std::string("OBJ has no triangles or indexes vertices / texcoords / materials that do not exist: ") + objPath; return HRTH_ERR_FORMAT; }
    return inst;
}

} // extern "C"
