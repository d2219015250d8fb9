// Image files -> 8-bit RGB texels, for textures (map_Kd) and the six sky box faces.
//
// The reference decodes with stb_image 2.27 asking for three channels (reference framework/src/image.cpp:13-35,
// stbi_load(..., STBI_rgb)) and divides by 255.  A texel therefore IS the byte stb_image produces, and this file
// restates how those bytes come about for the two formats the reference's data uses:
//
//   PNG   lossless, so any correct decoder agrees on the samples; what is stb_image's own is the reduction to RGB8
//         (stb_image.h:4591-4801, 5146-5170, 5199-5219, 1170-1186): 16-bit samples keep their high byte, 1/2/4-bit
//         grey is scaled by 255/85/17, palette entries are looked up, grey is replicated into three channels, and
//         alpha (channel or tRNS) is dropped without compositing.  CRCs and the Adler checksum are not verified there
//         and are not verified here.
//   JPEG  baseline and progressive Huffman, 8 bit (SOF0/1/2).  The bytes depend on the decoder's arithmetic:
//         coefficients times quantiser truncated to 16 bits (stb_image.h:2197,2214,2227,3039-3044), the integer
//         inverse DCT with 12-bit constants and its two rounding steps (:2392-2490), chroma upsampling by the
//         3:1 / 9:3:3:1 "centred" filters for factors 1x2, 2x1 and 2x2 and by replication otherwise (:3402-3473,
//         3592-3601, 3853-3886), and the 20-bit fixed-point YCbCr conversion whose Cb->G product is cut to its
//         upper 16 bits (:3605-3629).  The SSE2 kernels stb_image uses on x86-64 produce the same bytes as these
//         scalar forms.
//
// Not handled (the reference would decode them, this throws): BMP, GIF, PSD, PIC, PNM, HDR, TGA files, Apple's
// CgBI PNG variant, PNG colour types 2/4/6 with fewer than 8 bits per sample (outside the PNG specification).
#include "rt_facade.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace rtamd {
namespace {

[[noreturn]] void fail(const std::string& why) { throw std::runtime_error("image: " + why); }

// --------------------------------------------------------------------------------------------------
// zlib / deflate (RFC 1950, 1951)
// --------------------------------------------------------------------------------------------------
class Inflater {
public:
    Inflater(const uint8_t* data, size_t size) : m_in(data), m_size(size) {}

    std::vector<uint8_t> run(size_t sizeHint)
    {
        if (m_size < 2) fail("zlib stream too short");
        const unsigned cmf = m_in[0], flg = m_in[1];
        if ((cmf * 256u + flg) % 31u != 0) fail("bad zlib header");
        if (flg & 32u) fail("zlib preset dictionary");
        if ((cmf & 15u) != 8u) fail("zlib: not deflate");
        m_pos = 2;
        m_out.reserve(sizeHint);
        bool last;
        do {
            last = bits(1) != 0;
            switch (bits(2)) {
            case 0: stored(); break;
            case 1: fixed_tables(); compressed(); break;
            case 2: dynamic_tables(); compressed(); break;
            default: fail("bad deflate block type");
            }
        } while (!last);
        return std::move(m_out);
    }

private:
    // canonical Huffman code as counts per length and symbols in code order
    struct Code {
        std::array<uint16_t, 16> count {};
        std::array<uint16_t, 288> symbol {};
    };

    const uint8_t* m_in;
    size_t m_size, m_pos = 0;
    uint32_t m_hold = 0;
    int m_held = 0;
    std::vector<uint8_t> m_out;
    Code m_lit, m_dist;

    uint32_t bits(int n)
    {
        while (m_held < n) {
            uint32_t b = 0;
            if (m_pos < m_size) b = m_in[m_pos];
            else if (m_pos > m_size + 8) fail("deflate data ends early"); // a few zero bytes past the end are tolerated
            m_pos++;
            m_hold |= b << m_held;
            m_held += 8;
        }
        const uint32_t v = m_hold & ((1u << n) - 1u);
        m_hold >>= n;
        m_held -= n;
        return v;
    }

    static void build(Code& c, const uint8_t* lengths, int n)
    {
        c.count.fill(0);
        for (int i = 0; i < n; i++) c.count[lengths[i]]++;
        c.count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left = left * 2 - c.count[len];
            if (left < 0) fail("bad deflate code lengths");
        }
        std::array<uint16_t, 16> first {};
        for (int len = 1; len < 15; len++) first[len + 1] = uint16_t(first[len] + c.count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) c.symbol[first[lengths[i]]++] = uint16_t(i);
    }

    int decode(const Code& c)
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= int(bits(1));
            const int count = c.count[len];
            if (code - count < first) return c.symbol[index + (code - first)];
            index += count;
            first = (first + count) << 1;
            code <<= 1;
        }
        fail("bad deflate code");
    }

    void stored()
    {
        m_hold = 0;
        m_held = 0;
        if (m_pos + 4 > m_size) fail("deflate data ends early");
        const unsigned len = m_in[m_pos] | (m_in[m_pos + 1] << 8), nlen = m_in[m_pos + 2] | (m_in[m_pos + 3] << 8);
        m_pos += 4;
        if ((len ^ 0xffffu) != nlen) fail("corrupt stored block");
        if (m_pos + len > m_size) fail("deflate data ends early");
        m_out.insert(m_out.end(), m_in + m_pos, m_in + m_pos + len);
        m_pos += len;
    }

    void fixed_tables()
    {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        build(m_lit, l, 288);
        uint8_t d[30];
        for (int i = 0; i < 30; i++) d[i] = 5;
        build(m_dist, d, 30);
    }

    void dynamic_tables()
    {
        static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
        const int nlit = int(bits(5)) + 257, ndist = int(bits(5)) + 1, ncode = int(bits(4)) + 4;
        if (nlit > 286 || ndist > 30) fail("bad deflate table sizes");
        uint8_t cl[19] = {};
        for (int i = 0; i < ncode; i++) cl[order[i]] = uint8_t(bits(3));
        Code lengthCode;
        build(lengthCode, cl, 19);
        uint8_t l[286 + 30] = {};
        int i = 0;
        while (i < nlit + ndist) {
            const int sym = decode(lengthCode);
            if (sym < 16) {
                l[i++] = uint8_t(sym);
                continue;
            }
            int repeat;
            uint8_t value = 0;
            if (sym == 16) {
                if (i == 0) fail("bad deflate code lengths");
                value = l[i - 1];
                repeat = 3 + int(bits(2));
            } else if (sym == 17) {
                repeat = 3 + int(bits(3));
            } else {
                repeat = 11 + int(bits(7));
            }
            if (i + repeat > nlit + ndist) fail("bad deflate code lengths");
            while (repeat--) l[i++] = value;
        }
        build(m_lit, l, nlit);
        build(m_dist, l + nlit, ndist);
    }

    void compressed()
    {
        static const uint16_t lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
        static const uint8_t lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
        static const uint16_t distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
        static const uint8_t distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
        while (true) {
            int sym = decode(m_lit);
            if (sym < 256) {
                m_out.push_back(uint8_t(sym));
            } else if (sym == 256) {
                return;
            } else {
                sym -= 257;
                if (sym >= 29) fail("bad deflate length code");
                const size_t len = lenBase[sym] + bits(lenExtra[sym]);
                const int ds = decode(m_dist);
                if (ds >= 30) fail("bad deflate distance code");
                const size_t dist = distBase[ds] + bits(distExtra[ds]);
                if (dist > m_out.size()) fail("deflate distance too far back");
                const size_t from = m_out.size() - dist;
                for (size_t k = 0; k < len; k++) m_out.push_back(m_out[from + k]);
            }
        }
    }
};

// --------------------------------------------------------------------------------------------------
// PNG
// --------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

struct PngHeader {
    uint32_t width = 0, height = 0;
    int depth = 0, colour = 0, interlace = 0;
    int channels() const { return colour == 3 ? 1 : ((colour & 2) ? 3 : 1) + ((colour & 4) ? 1 : 0); }
};

// One pass (the whole picture, or one of the seven interlace passes): undoes the row filters in place and stores
// the RGB bytes of pixel (i, j) of the pass at out[(y0 + j * dy) * width + x0 + i * dx].  Returns the bytes consumed.
size_t png_pass(const PngHeader& h, const uint8_t* palette, uint32_t paletteEntries, uint8_t* raw, size_t rawLen, uint32_t w, uint32_t ht, uint32_t x0, uint32_t y0,
    uint32_t dx, uint32_t dy, uint8_t* out)
{
    const int ch = h.channels();
    const size_t rowBytes = (size_t(ch) * w * size_t(h.depth) + 7) >> 3;
    const size_t need = (rowBytes + 1) * ht;
    if (rawLen < need) fail("PNG: not enough pixels");
    const size_t bpp = h.depth < 8 ? 1 : size_t(ch) * size_t(h.depth / 8); // distance to the "previous pixel" of the filters
    static const uint8_t greyScale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    std::vector<uint8_t> zeros(rowBytes, 0);
    for (uint32_t j = 0; j < ht; j++) {
        uint8_t* row = raw + j * (rowBytes + 1);
        const int filter = row[0];
        uint8_t* cur = row + 1;
        const uint8_t* up = j ? cur - (rowBytes + 1) : zeros.data();
        if (filter > 4) fail("PNG: invalid filter");
        for (size_t k = 0; k < rowBytes; k++) {
            const int left = k >= bpp ? cur[k - bpp] : 0;
            const int above = up[k];
            const int aboveLeft = k >= bpp ? up[k - bpp] : 0;
            int predicted = 0;
            switch (filter) {
            case 1: predicted = left; break;
            case 2: predicted = above; break;
            case 3: predicted = (left + above) >> 1; break;
            case 4: predicted = paeth(left, above, aboveLeft); break;
            default: break;
            }
            cur[k] = uint8_t(cur[k] + predicted);
        }
        uint8_t* dst = out + (size_t(y0 + j * dy) * h.width + x0) * 3;
        for (uint32_t i = 0; i < w; i++, dst += size_t(dx) * 3) {
            uint8_t s[4] = { 0, 0, 0, 0 };
            for (int c = 0; c < ch; c++) {
                const size_t sample = size_t(i) * size_t(ch) + size_t(c);
                if (h.depth == 8) {
                    s[c] = cur[sample];
                } else if (h.depth == 16) {
                    s[c] = cur[sample * 2]; // the more significant byte
                } else {
                    const size_t bit = sample * size_t(h.depth);
                    const unsigned v = (cur[bit >> 3] >> (8 - h.depth - int(bit & 7))) & ((1u << h.depth) - 1u);
                    s[c] = uint8_t(h.colour == 0 ? v * greyScale[h.depth] : v);
                }
            }
            if (h.colour == 3) {
                if (s[0] < paletteEntries) {
                    dst[0] = palette[s[0] * 3];
                    dst[1] = palette[s[0] * 3 + 1];
                    dst[2] = palette[s[0] * 3 + 2];
                } else {
                    dst[0] = dst[1] = dst[2] = 0;
                }
            } else if (h.colour & 2) {
                dst[0] = s[0];
                dst[1] = s[1];
                dst[2] = s[2];
            } else {
                dst[0] = dst[1] = dst[2] = s[0];
            }
        }
    }
    return need;
}

Image decode_png(const std::vector<uint8_t>& file)
{
    PngHeader h;
    bool haveHeader = false, done = false;
    uint8_t palette[256 * 3] = {};
    uint32_t paletteEntries = 0;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    while (!done) {
        if (pos + 8 > file.size()) fail("PNG: file ends before IEND");
        const uint32_t len = be32(&file[pos]);
        const uint32_t type = be32(&file[pos + 4]);
        pos += 8;
        if (len > file.size() - pos) fail("PNG: chunk runs past the end of the file");
        const uint8_t* body = file.data() + pos;
        auto is = [type](const char* t) { return type == be32(reinterpret_cast<const uint8_t*>(t)); };
        if (!haveHeader && !is("IHDR") && !is("CgBI")) fail("PNG: first chunk is not IHDR");
        if (is("CgBI")) {
            fail("PNG: CgBI variant not supported");
        } else if (is("IHDR")) {
            if (haveHeader) fail("PNG: multiple IHDR");
            if (len != 13) fail("PNG: bad IHDR length");
            haveHeader = true;
            h.width = be32(body);
            h.height = be32(body + 4);
            h.depth = body[8];
            h.colour = body[9];
            h.interlace = body[12];
            if (h.width > (1u << 24) || h.height > (1u << 24)) fail("PNG: too large");
            if (h.depth != 1 && h.depth != 2 && h.depth != 4 && h.depth != 8 && h.depth != 16) fail("PNG: 1/2/4/8/16-bit only");
            if (h.colour > 6 || (h.colour == 3 && h.depth == 16) || (h.colour != 3 && (h.colour & 1))) fail("PNG: bad colour type");
            if (body[10] != 0) fail("PNG: bad compression method");
            if (body[11] != 0) fail("PNG: bad filter method");
            if (h.interlace > 1) fail("PNG: bad interlace method");
            if (h.width == 0 || h.height == 0) fail("PNG: 0-pixel image");
            if (h.depth < 8 && h.colour != 0 && h.colour != 3) fail("PNG: colour type needs at least 8 bits per sample");
            const uint32_t perPixel = h.colour == 3 ? 4u : uint32_t(h.channels());
            if ((1u << 30) / h.width / perPixel < h.height) fail("PNG: too large");
        } else if (is("PLTE")) {
            if (len > 256 * 3 || len % 3 != 0) fail("PNG: invalid PLTE");
            paletteEntries = len / 3;
            std::memcpy(palette, body, len);
        } else if (is("tRNS")) {
            // transparency never reaches a three-channel result; only the checks stb_image makes remain
            if (!idat.empty()) fail("PNG: tRNS after IDAT");
            if (h.colour == 3) {
                if (paletteEntries == 0) fail("PNG: tRNS before PLTE");
                if (len > paletteEntries) fail("PNG: bad tRNS length");
            } else {
                if (h.colour & 4) fail("PNG: tRNS with alpha");
                if (len != uint32_t(h.channels()) * 2u) fail("PNG: bad tRNS length");
            }
        } else if (is("IDAT")) {
            if (h.colour == 3 && paletteEntries == 0) fail("PNG: no PLTE");
            idat.insert(idat.end(), body, body + len);
        } else if (is("IEND")) {
            done = true;
        } else if ((type & (1u << 29)) == 0) {
            fail("PNG: unknown critical chunk");
        }
        pos += len;
        pos += 4; // CRC, not checked
    }
    if (idat.empty()) fail("PNG: no IDAT");

    const int ch = h.channels();
    const size_t rowBytes = (size_t(ch) * h.width * size_t(h.depth) + 7) >> 3;
    std::vector<uint8_t> raw = Inflater(idat.data(), idat.size()).run((rowBytes + 1) * h.height);
    std::vector<uint8_t> rgb(size_t(h.width) * h.height * 3);
    if (!h.interlace) {
        png_pass(h, palette, paletteEntries, raw.data(), raw.size(), h.width, h.height, 0, 0, 1, 1, rgb.data());
    } else {
        static const uint32_t xo[7] = { 0, 4, 0, 2, 0, 1, 0 }, yo[7] = { 0, 0, 4, 0, 2, 0, 1 }, xs[7] = { 8, 8, 4, 4, 2, 2, 1 }, ys[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t used = 0;
        for (int p = 0; p < 7; p++) {
            const uint32_t w = (h.width - xo[p] + xs[p] - 1) / xs[p], ht = (h.height - yo[p] + ys[p] - 1) / ys[p];
            if (h.width <= xo[p] && xo[p] != 0) continue;
            if (h.height <= yo[p] && yo[p] != 0) continue;
            if (w == 0 || ht == 0) continue;
            used += png_pass(h, palette, paletteEntries, raw.data() + used, raw.size() - used, w, ht, xo[p], yo[p], xs[p], ys[p], rgb.data());
        }
    }
    return Image(int(h.width), int(h.height), std::move(rgb));
}

// --------------------------------------------------------------------------------------------------
// JPEG
// --------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64 + 15] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
    15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a corrupt run may point past the block: those land on the last coefficient
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };

struct JpegHuffman {
    bool defined = false;
    uint8_t values[256] = {};
    int32_t maxCode[18] = {}; // exclusive upper bound of the codes of each length, -1 when there are none
    int32_t firstIndex[17] = {}, firstCode[17] = {};
    uint16_t quick[512] = {}; // 9-bit prefix -> length << 8 | value, 0 = longer code

    void build(const int counts[16])
    {
        int code = 0, index = 0;
        for (int len = 1; len <= 16; len++) {
            firstIndex[len] = index;
            firstCode[len] = code;
            code += counts[len - 1];
            index += counts[len - 1];
            if (counts[len - 1] && code - 1 >= (1 << len)) fail("JPEG: bad code lengths");
            maxCode[len] = counts[len - 1] ? code : -1;
            code <<= 1;
        }
        std::memset(quick, 0, sizeof quick);
        for (int len = 1; len <= 9; len++)
            for (int k = 0; k < counts[len - 1]; k++) {
                const int c = (firstCode[len] + k) << (9 - len);
                for (int fillBits = 0; fillBits < (1 << (9 - len)); fillBits++) quick[c + fillBits] = uint16_t((len << 8) | values[firstIndex[len] + k]);
            }
        defined = true;
    }
};

struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int pred = 0;
    int x = 0, y = 0;   // samples that belong to the picture
    int w2 = 0, h2 = 0; // allocated plane (whole MCUs)
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff; // progressive only: blocksAcross x blocksDown x 64
    int blocksAcross = 0;
};

class JpegDecoder {
public:
    explicit JpegDecoder(const std::vector<uint8_t>& file) : m_file(file) {}

    Image run()
    {
        decode_frame();
        return to_rgb();
    }

private:
    const std::vector<uint8_t>& m_file;
    size_t m_pos = 0;
    // entropy-coded segment reader
    uint32_t m_bitBuffer = 0;
    int m_bitCount = 0;
    int m_marker = 0xff; // 0xff = none pending
    bool m_noMore = false;
    // tables and frame
    uint16_t m_quant[4][64] = {};
    JpegHuffman m_dc[4], m_ac[4];
    JpegComponent m_comp[4];
    int m_nComp = 0, m_width = 0, m_height = 0;
    int m_hMax = 1, m_vMax = 1, m_mcusAcross = 0, m_mcusDown = 0;
    bool m_progressive = false, m_jfif = false;
    int m_adobeTransform = -1, m_rgbIds = 0;
    int m_restartInterval = 0, m_todo = 0;
    // scan
    int m_scanComps = 0, m_order[4] = {};
    int m_specStart = 0, m_specEnd = 63, m_succHigh = 0, m_succLow = 0, m_eobRun = 0;

    int byte() { return m_pos < m_file.size() ? m_file[m_pos++] : 0; }
    bool at_end() const { return m_pos >= m_file.size(); }
    int word()
    {
        const int hi = byte();
        return (hi << 8) | byte();
    }
    void skip(int n) { m_pos = n < 0 ? m_file.size() : std::min(m_file.size(), m_pos + size_t(n)); }

    // next marker code, or 0xff when the next byte does not start one
    int next_marker()
    {
        if (m_marker != 0xff) {
            const int m = m_marker;
            m_marker = 0xff;
            return m;
        }
        int x = byte();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = byte();
        return x;
    }

    // ---- bits of the entropy-coded segment: bytes are stuffed (ff 00 = ff), a marker ends the data and zeros follow
    void refill()
    {
        do {
            const unsigned b = m_noMore ? 0u : unsigned(byte());
            if (b == 0xff) {
                int c = byte();
                while (c == 0xff) c = byte();
                if (c != 0) {
                    m_marker = c;
                    m_noMore = true;
                    return;
                }
            }
            m_bitBuffer |= b << (24 - m_bitCount);
            m_bitCount += 8;
        } while (m_bitCount <= 24);
    }
    int take(int n) // n unsigned bits
    {
        if (n == 0) return 0;
        if (m_bitCount < n) refill();
        // the refill that meets a marker adds nothing; the next one pads with the zero bits that follow the data, so
        // the count never goes negative (a negative count would make refill() shift by 32 or more)
        if (m_bitCount < n) refill();
        const uint32_t v = m_bitBuffer >> (32 - n);
        m_bitBuffer <<= n;
        m_bitCount -= n;
        return int(v);
    }
    int take_signed(int n) // the "extend" of the JPEG specification: n bits, the first one tells the sign
    {
        if (n == 0) return 0;
        const int v = take(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int symbol(const JpegHuffman& h)
    {
        if (m_bitCount < 16) refill();
        const uint16_t q = h.quick[m_bitBuffer >> 23];
        if (q) {
            const int len = q >> 8;
            if (len > m_bitCount) fail("JPEG: bad huffman code");
            m_bitBuffer <<= len;
            m_bitCount -= len;
            return q & 0xff;
        }
        const int32_t top = int32_t(m_bitBuffer >> 16);
        for (int len = 10; len <= 16; len++) {
            const int32_t code = top >> (16 - len);
            if (h.maxCode[len] >= 0 && code < h.maxCode[len] && code >= h.firstCode[len]) {
                if (len > m_bitCount) fail("JPEG: bad huffman code");
                m_bitBuffer <<= len;
                m_bitCount -= len;
                return h.values[h.firstIndex[len] + (code - h.firstCode[len])];
            }
        }
        fail("JPEG: bad huffman code");
    }

    void restart_state()
    {
        m_bitBuffer = 0;
        m_bitCount = 0;
        m_noMore = false;
        for (JpegComponent& c : m_comp) c.pred = 0;
        m_marker = 0xff;
        m_todo = m_restartInterval ? m_restartInterval : 0x7fffffff;
        m_eobRun = 0;
    }
    // counts an MCU; false when the data stop here (no restart marker where one is due): what was decoded so far is kept
    bool mcu_done()
    {
        if (--m_todo > 0) return true;
        if (m_bitCount < 24) refill();
        if (m_marker < 0xd0 || m_marker > 0xd7) return false;
        restart_state();
        return true;
    }

    // ---- tables and headers
    void tables_or_application_segment(int m)
    {
        if (m == 0xff) fail("JPEG: expected marker");
        if (m == 0xdd) {
            if (word() != 4) fail("JPEG: bad DRI length");
            m_restartInterval = word();
            return;
        }
        if (m == 0xdb) {
            int left = word() - 2;
            while (left > 0) {
                const int q = byte(), wide = q >> 4, t = q & 15;
                if (wide > 1) fail("JPEG: bad DQT type");
                if (t > 3) fail("JPEG: bad DQT table");
                for (int i = 0; i < 64; i++) m_quant[t][kZigzag[i]] = uint16_t(wide ? word() : byte());
                left -= wide ? 129 : 65;
            }
            if (left != 0) fail("JPEG: bad DQT length");
            return;
        }
        if (m == 0xc4) {
            int left = word() - 2;
            while (left > 0) {
                const int q = byte(), cls = q >> 4, slot = q & 15;
                if (cls > 1 || slot > 3) fail("JPEG: bad DHT header");
                int counts[16], total = 0;
                for (int& c : counts) {
                    c = byte();
                    total += c;
                }
                if (total > 256) fail("JPEG: bad DHT header");
                JpegHuffman& h = cls ? m_ac[slot] : m_dc[slot];
                for (int i = 0; i < total; i++) h.values[i] = uint8_t(byte());
                h.build(counts);
                left -= 17 + total;
            }
            if (left != 0) fail("JPEG: bad DHT length");
            return;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
            int left = word();
            if (left < 2) fail("JPEG: bad segment length");
            left -= 2;
            if (m == 0xe0 && left >= 5) {
                static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool same = true;
                for (char t : tag) same = (byte() == uint8_t(t)) && same;
                left -= 5;
                if (same) m_jfif = true;
            } else if (m == 0xee && left >= 12) {
                static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool same = true;
                for (char t : tag) same = (byte() == uint8_t(t)) && same;
                left -= 6;
                if (same) {
                    byte();
                    word();
                    word();
                    m_adobeTransform = byte();
                    left -= 6;
                }
            }
            skip(left);
            return;
        }
        fail("JPEG: unknown marker");
    }

    void frame_header()
    {
        const int len = word();
        if (len < 11) fail("JPEG: bad SOF length");
        if (byte() != 8) fail("JPEG: 8-bit only");
        m_height = word();
        m_width = word();
        if (m_height == 0) fail("JPEG: no header height");
        if (m_width == 0) fail("JPEG: 0 width");
        m_nComp = byte();
        if (m_nComp != 1 && m_nComp != 3 && m_nComp != 4) fail("JPEG: bad component count");
        if (len != 8 + 3 * m_nComp) fail("JPEG: bad SOF length");
        m_rgbIds = 0;
        for (int i = 0; i < m_nComp; i++) {
            JpegComponent& c = m_comp[i];
            c.id = byte();
            if (m_nComp == 3 && c.id == "RGB"[i]) m_rgbIds++;
            const int q = byte();
            c.h = q >> 4;
            c.v = q & 15;
            if (c.h < 1 || c.h > 4) fail("JPEG: bad H");
            if (c.v < 1 || c.v > 4) fail("JPEG: bad V");
            c.tq = byte();
            if (c.tq > 3) fail("JPEG: bad TQ");
        }
        if (size_t(m_width) * size_t(m_height) * size_t(m_nComp) > (size_t(1) << 31) - 1) fail("JPEG: too large");
        m_hMax = m_vMax = 1;
        for (int i = 0; i < m_nComp; i++) {
            m_hMax = std::max(m_hMax, m_comp[i].h);
            m_vMax = std::max(m_vMax, m_comp[i].v);
        }
        for (int i = 0; i < m_nComp; i++) {
            if (m_hMax % m_comp[i].h != 0) fail("JPEG: bad H");
            if (m_vMax % m_comp[i].v != 0) fail("JPEG: bad V");
        }
        m_mcusAcross = (m_width + m_hMax * 8 - 1) / (m_hMax * 8);
        m_mcusDown = (m_height + m_vMax * 8 - 1) / (m_vMax * 8);
        for (int i = 0; i < m_nComp; i++) {
            JpegComponent& c = m_comp[i];
            c.x = (m_width * c.h + m_hMax - 1) / m_hMax;
            c.y = (m_height * c.v + m_vMax - 1) / m_vMax;
            c.w2 = m_mcusAcross * c.h * 8;
            c.h2 = m_mcusDown * c.v * 8;
            c.plane.assign(size_t(c.w2) * size_t(c.h2), 0);
            if (m_progressive) {
                c.blocksAcross = c.w2 / 8;
                c.coeff.assign(size_t(c.w2) * size_t(c.h2), 0);
            }
        }
    }

    void scan_header()
    {
        const int len = word();
        m_scanComps = byte();
        if (m_scanComps < 1 || m_scanComps > 4 || m_scanComps > m_nComp) fail("JPEG: bad SOS component count");
        if (len != 6 + 2 * m_scanComps) fail("JPEG: bad SOS length");
        for (int i = 0; i < m_scanComps; i++) {
            const int id = byte(), q = byte();
            int which = 0;
            while (which < m_nComp && m_comp[which].id != id) which++;
            if (which == m_nComp) fail("JPEG: scan names an unknown component");
            m_comp[which].hd = q >> 4;
            m_comp[which].ha = q & 15;
            if (m_comp[which].hd > 3) fail("JPEG: bad DC huff");
            if (m_comp[which].ha > 3) fail("JPEG: bad AC huff");
            m_order[i] = which;
        }
        m_specStart = byte();
        m_specEnd = byte();
        const int aa = byte();
        m_succHigh = aa >> 4;
        m_succLow = aa & 15;
        if (m_progressive) {
            if (m_specStart > 63 || m_specEnd > 63 || m_specStart > m_specEnd || m_succHigh > 13 || m_succLow > 13) fail("JPEG: bad SOS");
        } else {
            if (m_specStart != 0 || m_succHigh != 0 || m_succLow != 0) fail("JPEG: bad SOS");
            m_specEnd = 63;
        }
    }

    // ---- coefficient blocks
    // sequential: one whole block, already multiplied by the quantiser and cut to 16 bits
    void block_sequential(JpegComponent& c, int16_t* data)
    {
        const JpegHuffman &dc = m_dc[c.hd], &ac = m_ac[c.ha];
        const uint16_t* quant = m_quant[c.tq];
        const int t = symbol(dc);
        if (t > 15) fail("JPEG: bad huffman code");
        std::memset(data, 0, 64 * sizeof(int16_t));
        c.pred += take_signed(t);
        data[0] = int16_t(c.pred * quant[0]);
        int k = 1;
        do {
            const int rs = symbol(ac);
            const int size = rs & 15, run = rs >> 4;
            if (size == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += run;
                const int at = kZigzag[k++];
                data[at] = int16_t(take_signed(size) * quant[at]);
            }
        } while (k < 64);
    }

    void block_progressive_dc(JpegComponent& c, int16_t* data)
    {
        if (m_specEnd != 0) fail("JPEG: can't merge dc and ac");
        if (m_succHigh == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = symbol(m_dc[c.hd]);
            if (t > 15) fail("JPEG: can't merge dc and ac");
            c.pred += take_signed(t);
            data[0] = int16_t(c.pred * (1 << m_succLow));
        } else if (take(1)) {
            data[0] = int16_t(data[0] + int16_t(1 << m_succLow));
        }
    }

    // one more bit of precision for a coefficient that is already non-zero
    void refine(int16_t& v, int16_t bit)
    {
        if (take(1) && (v & bit) == 0) v = int16_t(v > 0 ? v + bit : v - bit);
    }

    void block_progressive_ac(JpegComponent& c, int16_t* data)
    {
        if (m_specStart == 0) fail("JPEG: can't merge dc and ac");
        const JpegHuffman& ac = m_ac[c.ha];
        if (m_succHigh == 0) {
            if (m_eobRun) {
                m_eobRun--;
                return;
            }
            int k = m_specStart;
            do {
                const int rs = symbol(ac);
                const int size = rs & 15, run = rs >> 4;
                if (size == 0) {
                    if (run < 15) {
                        m_eobRun = (1 << run) + take(run) - 1;
                        break;
                    }
                    k += 16;
                } else {
                    k += run;
                    data[kZigzag[k++]] = int16_t(take_signed(size) * (1 << m_succLow));
                }
            } while (k <= m_specEnd);
            return;
        }
        const int16_t bit = int16_t(1 << m_succLow);
        if (m_eobRun) {
            m_eobRun--;
            for (int k = m_specStart; k <= m_specEnd; k++) {
                int16_t& v = data[kZigzag[k]];
                if (v != 0) refine(v, bit);
            }
            return;
        }
        int k = m_specStart;
        do {
            const int rs = symbol(ac);
            int size = rs & 15, run = rs >> 4;
            int fresh = 0;
            if (size == 0) {
                if (run < 15) {
                    m_eobRun = (1 << run) - 1 + take(run);
                    run = 64; // nothing new in this block: only refinements up to the end of the band
                }
            } else {
                if (size != 1) fail("JPEG: bad huffman code");
                fresh = take(1) ? bit : -bit;
            }
            while (k <= m_specEnd) {
                int16_t& v = data[kZigzag[k++]];
                if (v != 0) {
                    refine(v, bit);
                } else {
                    if (run == 0) {
                        v = int16_t(fresh);
                        break;
                    }
                    run--;
                }
            }
        } while (k <= m_specEnd);
    }

    // ---- inverse DCT (8x8, integer, after the "islow" factorisation): 12 fractional bits in the constants, two more
    // kept between the passes
    struct Butterfly {
        int e0, e1, e2, e3; // even part
        int o0, o1, o2, o3; // odd part
    };
    static constexpr int fixed12(float c) { return int(c * 4096 + 0.5); }
    static Butterfly pass(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        Butterfly r;
        const int z = (s2 + s6) * fixed12(0.5411961f);
        const int a = z + s6 * fixed12(-1.847759065f);
        const int b = z + s2 * fixed12(0.765366865f);
        const int c = (s0 + s4) * 4096;
        const int d = (s0 - s4) * 4096;
        r.e0 = c + b;
        r.e3 = c - b;
        r.e1 = d + a;
        r.e2 = d - a;
        const int q3 = s7 + s3, q4 = s5 + s1, q1 = s7 + s1, q2 = s5 + s3;
        const int q5 = (q3 + q4) * fixed12(1.175875602f);
        const int m1 = q5 + q1 * fixed12(-0.899976223f);
        const int m2 = q5 + q2 * fixed12(-2.562915447f);
        const int m3 = q3 * fixed12(-1.961570560f);
        const int m4 = q4 * fixed12(-0.390180644f);
        r.o3 = s1 * fixed12(1.501321110f) + (m1 + m4);
        r.o2 = s3 * fixed12(3.072711026f) + (m2 + m3);
        r.o1 = s5 * fixed12(2.053119869f) + (m2 + m4);
        r.o0 = s7 * fixed12(0.298631336f) + (m1 + m3);
        return r;
    }
    static uint8_t clamp8(int v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void inverse_dct(uint8_t* out, int stride, const int16_t* d)
    {
        int mid[64];
        for (int i = 0; i < 8; i++) {
            Butterfly r = pass(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
            r.e0 += 512;
            r.e1 += 512;
            r.e2 += 512;
            r.e3 += 512;
            mid[i] = (r.e0 + r.o3) >> 10;
            mid[56 + i] = (r.e0 - r.o3) >> 10;
            mid[8 + i] = (r.e1 + r.o2) >> 10;
            mid[48 + i] = (r.e1 - r.o2) >> 10;
            mid[16 + i] = (r.e2 + r.o1) >> 10;
            mid[40 + i] = (r.e2 - r.o1) >> 10;
            mid[24 + i] = (r.e3 + r.o0) >> 10;
            mid[32 + i] = (r.e3 - r.o0) >> 10;
        }
        for (int i = 0; i < 8; i++) {
            const int* v = mid + 8 * i;
            uint8_t* o = out + size_t(stride) * size_t(i);
            Butterfly r = pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const int bias = 65536 + (128 << 17); // rounding, and the level shift to 0..255
            r.e0 += bias;
            r.e1 += bias;
            r.e2 += bias;
            r.e3 += bias;
            o[0] = clamp8((r.e0 + r.o3) >> 17);
            o[7] = clamp8((r.e0 - r.o3) >> 17);
            o[1] = clamp8((r.e1 + r.o2) >> 17);
            o[6] = clamp8((r.e1 - r.o2) >> 17);
            o[2] = clamp8((r.e2 + r.o1) >> 17);
            o[5] = clamp8((r.e2 - r.o1) >> 17);
            o[3] = clamp8((r.e3 + r.o0) >> 17);
            o[4] = clamp8((r.e3 - r.o0) >> 17);
        }
    }

    // ---- one scan
    void scan_data()
    {
        restart_state();
        int16_t block[64];
        if (m_scanComps == 1) {
            // a scan of one component covers exactly the blocks that hold picture samples, row by row
            JpegComponent& c = m_comp[m_order[0]];
            const int across = (c.x + 7) >> 3, down = (c.y + 7) >> 3;
            for (int j = 0; j < down; j++)
                for (int i = 0; i < across; i++) {
                    if (!m_progressive) {
                        block_sequential(c, block);
                        inverse_dct(c.plane.data() + size_t(c.w2) * size_t(j) * 8 + size_t(i) * 8, c.w2, block);
                    } else {
                        int16_t* data = c.coeff.data() + 64 * (size_t(i) + size_t(j) * size_t(c.blocksAcross));
                        if (m_specStart == 0) block_progressive_dc(c, data);
                        else block_progressive_ac(c, data);
                    }
                    if (!mcu_done()) return;
                }
            return;
        }
        for (int j = 0; j < m_mcusDown; j++)
            for (int i = 0; i < m_mcusAcross; i++) {
                for (int k = 0; k < m_scanComps; k++) {
                    JpegComponent& c = m_comp[m_order[k]];
                    for (int y = 0; y < c.v; y++)
                        for (int x = 0; x < c.h; x++) {
                            const int bx = i * c.h + x, by = j * c.v + y;
                            if (!m_progressive) {
                                block_sequential(c, block);
                                inverse_dct(c.plane.data() + size_t(c.w2) * size_t(by) * 8 + size_t(bx) * 8, c.w2, block);
                            } else {
                                block_progressive_dc(c, c.coeff.data() + 64 * (size_t(bx) + size_t(by) * size_t(c.blocksAcross)));
                            }
                        }
                }
                if (!mcu_done()) return;
            }
    }

    void decode_frame()
    {
        if (next_marker() != 0xd8) fail("JPEG: no SOI");
        int m = next_marker();
        while (m != 0xc0 && m != 0xc1 && m != 0xc2) {
            tables_or_application_segment(m);
            m = next_marker();
            while (m == 0xff) {
                if (at_end()) fail("JPEG: no SOF");
                m = next_marker();
            }
        }
        m_progressive = m == 0xc2;
        frame_header();
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                scan_header();
                scan_data();
                if (m_marker == 0xff) {
                    // stray bytes after the entropy-coded data: look for the next marker
                    while (!at_end()) {
                        if (byte() == 0xff) {
                            m_marker = byte();
                            break;
                        }
                    }
                }
            } else if (m == 0xdc) {
                const int len = word(), lines = word();
                if (len != 4) fail("JPEG: bad DNL length");
                if (lines != m_height) fail("JPEG: bad DNL height");
            } else {
                tables_or_application_segment(m);
            }
            m = next_marker();
        }
        if (m_progressive) {
            // all scans are in: multiply by the quantisers in force now (16-bit products) and transform
            for (int n = 0; n < m_nComp; n++) {
                JpegComponent& c = m_comp[n];
                const int across = (c.x + 7) >> 3, down = (c.y + 7) >> 3;
                for (int j = 0; j < down; j++)
                    for (int i = 0; i < across; i++) {
                        int16_t* data = c.coeff.data() + 64 * (size_t(i) + size_t(j) * size_t(c.blocksAcross));
                        for (int k = 0; k < 64; k++) data[k] = int16_t(data[k] * m_quant[c.tq][k]);
                        inverse_dct(c.plane.data() + size_t(c.w2) * size_t(j) * 8 + size_t(i) * 8, c.w2, data);
                    }
            }
        }
    }

    // ---- upsampling: `near` is the source row closest to the output row, `far` the next closest
    static const uint8_t* upsample_row(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) return near;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; i++) out[i] = uint8_t((3 * near[i] + far[i] + 2) >> 2);
            return out;
        }
        if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = near[0];
                return out;
            }
            out[0] = near[0];
            out[1] = uint8_t((near[0] * 3 + near[1] + 2) >> 2);
            for (int i = 1; i < w - 1; i++) {
                const int n = 3 * near[i] + 2;
                out[i * 2] = uint8_t((n + near[i - 1]) >> 2);
                out[i * 2 + 1] = uint8_t((n + near[i + 1]) >> 2);
            }
            out[(w - 1) * 2] = uint8_t((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
            out[(w - 1) * 2 + 1] = near[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {
            // vertical 3:1 blend first, then the same horizontally on the unrounded sums
            int right = 3 * near[0] + far[0];
            if (w == 1) {
                out[0] = out[1] = uint8_t((right + 2) >> 2);
                return out;
            }
            out[0] = uint8_t((right + 2) >> 2);
            for (int i = 1; i < w; i++) {
                const int left = right;
                right = 3 * near[i] + far[i];
                out[i * 2 - 1] = uint8_t((3 * left + right + 8) >> 4);
                out[i * 2] = uint8_t((3 * right + left + 8) >> 4);
            }
            out[w * 2 - 1] = uint8_t((right + 2) >> 2);
            return out;
        }
        for (int i = 0; i < w; i++)
            for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
        return out;
    }

    static constexpr int fixed20(float c) { return int(c * 4096.0f + 0.5f) << 8; }
    static void ycc_to_rgb(uint8_t* out, int y, int cb, int cr)
    {
        const int yf = (y << 20) + (1 << 19);
        cr -= 128;
        cb -= 128;
        int r = yf + cr * fixed20(1.40200f);
        int g = int(uint32_t(yf + cr * -fixed20(0.71414f)) + (uint32_t(cb * -fixed20(0.34414f)) & 0xffff0000u));
        int b = yf + cb * fixed20(1.77200f);
        out[0] = clamp8(r >> 20);
        out[1] = clamp8(g >> 20);
        out[2] = clamp8(b >> 20);
    }
    static uint8_t scale8(unsigned x, unsigned y) // x * y / 255, rounded
    {
        const unsigned t = x * y + 128;
        return uint8_t((t + (t >> 8)) >> 8);
    }

    Image to_rgb()
    {
        const bool isRgb = m_nComp == 3 && (m_rgbIds == 3 || (m_adobeTransform == 0 && !m_jfif));
        struct Up {
            int hs, vs, wLow, step, rowNumber;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buffer;
        } up[4];
        for (int k = 0; k < m_nComp; k++) {
            Up& u = up[k];
            u.hs = m_hMax / m_comp[k].h;
            u.vs = m_vMax / m_comp[k].v;
            u.step = u.vs >> 1;
            u.wLow = (m_width + u.hs - 1) / u.hs;
            u.rowNumber = 0;
            u.line0 = u.line1 = m_comp[k].plane.data();
            u.buffer.assign(size_t(m_width) + 8, 0);
        }
        std::vector<uint8_t> rgb(size_t(m_width) * size_t(m_height) * 3);
        const uint8_t* rows[4] = {};
        for (int j = 0; j < m_height; j++) {
            uint8_t* out = rgb.data() + size_t(j) * size_t(m_width) * 3;
            for (int k = 0; k < m_nComp; k++) {
                Up& u = up[k];
                const bool lower = u.step >= (u.vs >> 1); // output row lies in the lower half of source row line0..line1
                rows[k] = upsample_row(u.buffer.data(), lower ? u.line1 : u.line0, lower ? u.line0 : u.line1, u.wLow, u.hs, u.vs);
                if (++u.step >= u.vs) {
                    u.step = 0;
                    u.line0 = u.line1;
                    if (++u.rowNumber < m_comp[k].y) u.line1 += m_comp[k].w2;
                }
            }
            for (int i = 0; i < m_width; i++, out += 3) {
                if (m_nComp == 1) {
                    out[0] = out[1] = out[2] = rows[0][i];
                } else if (m_nComp == 3) {
                    if (isRgb) {
                        out[0] = rows[0][i];
                        out[1] = rows[1][i];
                        out[2] = rows[2][i];
                    } else {
                        ycc_to_rgb(out, rows[0][i], rows[1][i], rows[2][i]);
                    }
                } else if (m_adobeTransform == 0) { // CMYK
                    const unsigned k4 = rows[3][i];
                    out[0] = scale8(rows[0][i], k4);
                    out[1] = scale8(rows[1][i], k4);
                    out[2] = scale8(rows[2][i], k4);
                } else if (m_adobeTransform == 2) { // YCCK
                    const unsigned k4 = rows[3][i];
                    ycc_to_rgb(out, rows[0][i], rows[1][i], rows[2][i]);
                    out[0] = scale8(255u - out[0], k4);
                    out[1] = scale8(255u - out[1], k4);
                    out[2] = scale8(255u - out[2], k4);
                } else { // fourth channel ignored
                    ycc_to_rgb(out, rows[0][i], rows[1][i], rows[2][i]);
                }
            }
        }
        return Image(m_width, m_height, std::move(rgb));
    }
};

std::vector<uint8_t> read_file(const std::filesystem::path& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path.string());
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

} // namespace

Image decodeImage(const uint8_t* data, size_t size)
{
    std::vector<uint8_t> file(data, data + size);
    static const uint8_t pngSignature[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (file.size() >= 8 && std::memcmp(file.data(), pngSignature, 8) == 0) return decode_png(file);
    if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) return JpegDecoder(file).run();
    fail("not a PNG or JPEG file");
}

Image::Image(const std::filesystem::path& filePath)
    : width(0), height(0)
{
    if (!std::filesystem::exists(filePath)) fail("texture file " + filePath.string() + " does not exist");
    const std::vector<uint8_t> file = read_file(filePath);
    *this = decodeImage(file.data(), file.size());
}

} // namespace rtamd
