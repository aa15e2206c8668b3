// sgh_inflate.cpp — a DEFLATE decoder that can start in the middle of a gzip stream (RFC 1951 / 1952).
//
// Why: a plain .gz file is ONE deflate stream; zlib inflates it on one core (~0.8 GB/s of FASTQ text) and everything behind it
// waits — the reference does exactly that on the sample's thread (flate2 inside fxread, call site src/count.rs:24).  To spread
// the work over cores the stream is cut into chunks of compressed bytes and every chunk is decoded speculatively, in parallel,
// before the data in front of it is known (the scheme of pugz / rapidgzip):
//   * find_block_start: from a bit position on, the first offset that parses as a deflate block header (dynamic-Huffman or
//     stored, not final), whose Huffman code sets are complete, whose whole block decodes without an error into text bytes, and
//     whose successor parses too;
//   * decode: blocks from a block start up to the first block boundary at or behind a stop position.  Without a window the
//     output is 16-bit SYMBOLS: a value < 256 is a byte, 0x8000 | i stands for byte i of the unknown 32 KiB in front of the chunk
//     (back-references copy symbols, so markers spread until the data stops pointing there).  With the window known (the
//     resolved tail of the chunk before) the same routine produces bytes only;
//   * resolve: symbols -> bytes once the window is known.
// gzip members (header, trailer with CRC-32 and ISIZE) are walked inside decode; the CRC of every member is checked by the caller
// over the resolved bytes.  This file is only the decoder; the threading lives in TextFeeder (sgh.cpp).
#include <algorithm>
#include <cstring>

#include "sgh.hpp"

namespace sgh {

namespace {

struct Bits {
    const uint8_t *p;
    size_t n;            // bytes available
    uint64_t pos;        // bit position of the next unread bit
    bool over = false;   // read past the end
    uint64_t peek(unsigned k) {           // k <= 32: the next k bits, LSB first, without consuming
        const size_t byte = (size_t)(pos >> 3);
        uint64_t w = 0;
        if (byte + 8 <= n) memcpy(&w, p + byte, 8);
        else for (size_t i = byte; i < n && i < byte + 8; i++) w |= (uint64_t)p[i] << (8 * (i - byte));
        return (w >> (pos & 7)) & ((1ull << k) - 1ull);
    }
    void skip(unsigned k) { pos += k; if (pos > 8ull * n) over = true; }
    uint32_t get(unsigned k) { const uint32_t v = (uint32_t)peek(k); skip(k); return v; }
    void align() { pos = (pos + 7) & ~7ull; }
};

// canonical Huffman code: a primary table for codes of <= PRIM bits (entry = symbol << 4 | length, 0 = longer code or invalid),
// longer codes by the classic count/first-code walk
template <unsigned PRIM>
struct Huff {
    uint16_t fast[1u << PRIM];
    uint16_t count[16], symbol[320];
    unsigned maxlen = 0;
    // returns 0 ok, 1 incomplete (under-subscribed), -1 over-subscribed
    int build(const uint8_t *len, unsigned n) {
        memset(count, 0, sizeof(count));
        for (unsigned i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        maxlen = 0;
        for (unsigned l = 1; l < 16; l++) if (count[l]) maxlen = l;
        int left = 1;
        for (unsigned l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return -1; }
        uint16_t offs[16];
        offs[1] = 0;
        for (unsigned l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (unsigned i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        memset(fast, 0, sizeof(fast));
        // fill the primary table: codes are assigned in order of (length, symbol); the bit-reversed code indexes the table
        unsigned code = 0, idx = 0;
        for (unsigned l = 1; l <= PRIM && l < 16; l++) {
            for (unsigned k = 0; k < count[l]; k++, idx++, code++) {
                unsigned rev = 0;
                for (unsigned b = 0; b < l; b++) rev |= ((code >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)((symbol[idx] << 4) | l);
                for (unsigned x = rev; x < (1u << PRIM); x += 1u << l) fast[x] = e;
            }
            code <<= 1;
        }
        return left > 0 ? 1 : 0;
    }
    // decodes one symbol; -1 = invalid code
    int decode(Bits &b) const {
        const uint32_t w = (uint32_t)b.peek(15);
        const uint16_t e = fast[w & ((1u << PRIM) - 1u)];
        if (e) { b.skip(e & 15u); return e >> 4; }
        int code = 0, first = 0, index = 0;
        for (unsigned l = 1; l <= maxlen; l++) {
            code |= (int)((w >> (l - 1)) & 1u);
            const int c = count[l];
            if (code - c < first) { b.skip(l); return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Codes { Huff<11> lit; Huff<8> dist; };

// the code sets of a dynamic block; false = not a valid header
bool read_dynamic(Bits &b, Codes &c) {
    const unsigned hlit = b.get(5) + 257, hdist = b.get(5) + 1, hclen = b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t pl[19] = {0};
    for (unsigned i = 0; i < hclen; i++) pl[order[i]] = (uint8_t)b.get(3);
    Huff<7> pre;
    if (pre.build(pl, 19) != 0) return false;                // zlib: the code-length code must be complete
    uint8_t len[320];
    unsigned i = 0;
    while (i < hlit + hdist) {
        const int s = pre.decode(b);
        if (s < 0 || b.over) return false;
        if (s < 16) len[i++] = (uint8_t)s;
        else {
            unsigned rep, val = 0;
            if (s == 16) { if (i == 0) return false; val = len[i - 1]; rep = 3 + b.get(2); }
            else if (s == 17) rep = 3 + b.get(3);
            else rep = 11 + b.get(7);
            if (i + rep > hlit + hdist) return false;
            while (rep--) len[i++] = (uint8_t)val;
        }
    }
    if (len[256] == 0) return false;                          // no end-of-block code
    if (c.lit.build(len, hlit) != 0) {                        // complete, or (zlib allows it) a single code of length 1
        if (!(c.lit.count[1] == 1 && c.lit.maxlen == 1)) return false;
    }
    const int d = c.dist.build(len + hlit, hdist);
    if (d < 0) return false;
    if (d > 0 && !(c.dist.maxlen <= 1)) return false;         // incomplete distance code: only the one-code case is legal
    return !b.over;
}

void fixed_codes(Codes &c) {
    uint8_t len[288];
    for (unsigned i = 0; i < 144; i++) len[i] = 8;
    for (unsigned i = 144; i < 256; i++) len[i] = 9;
    for (unsigned i = 256; i < 280; i++) len[i] = 7;
    for (unsigned i = 280; i < 288; i++) len[i] = 8;
    c.lit.build(len, 288);
    uint8_t dl[30];
    for (unsigned i = 0; i < 30; i++) dl[i] = 5;
    c.dist.build(dl, 30);
}

inline bool is_text(unsigned c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

}  // namespace

// One deflate block starting at b.pos (the 3 header bits included).  Symbols are appended to out; `window` (32 KiB, may be
// null) holds the bytes in front of out[0].  text_only: a literal outside printable ASCII / newline / tab fails the block (the
// validation of a speculative block start).  Returns 0 = block done, 1 = it was the final block of its member, < 0 = error.
// out is used as a buffer: out.size() is its capacity, n the symbols in it (the caller trims it in the end).
static int inflate_block(Bits &b, std::vector<uint16_t> &out, size_t &n_io, const uint8_t *window, bool text_only, size_t max_out) {
    size_t n = n_io;                  // (kept in a register: written back at the exits)
    const unsigned final_block = b.get(1), type = b.get(2);
    if (b.over) return -1;
    if (type == 3) return -1;
    if (type == 0) {
        b.align();
        const unsigned len = b.get(16), nlen = b.get(16);
        if (b.over || (len ^ nlen) != 0xFFFFu) return -1;
        const size_t byte = (size_t)(b.pos >> 3);
        if (byte + len > b.n) return -1;
        if (n + len > max_out) return -3;
        if (out.size() < n + len) out.resize(std::max<size_t>(2 * out.size(), n + len + (1u << 16)));
        for (unsigned i = 0; i < len; i++) {
            const uint8_t ch = b.p[byte + i];
            if (text_only && !is_text(ch)) return -2;
            out[n++] = ch;
        }
        b.pos += 8ull * len;
        n_io = n;
        return (int)final_block;
    }
    static thread_local Codes codes;
    if (type == 1) fixed_codes(codes);
    else if (!read_dynamic(b, codes)) return -1;
    // the symbol loop: a local 64-bit bit buffer (refilled with one unaligned load while at least 8 bytes of input remain),
    // output through a raw pointer into capacity reserved ahead
    const uint8_t *const in = b.p;
    const size_t in_n = b.n;
    size_t byte = (size_t)(b.pos >> 3);
    uint64_t bb = 0;
    unsigned bc = 0;                                     // valid bits in bb
    {   // prime: the bits of the current byte from b.pos on
        const unsigned skip = (unsigned)(b.pos & 7);
        if (byte < in_n) { bb = (uint64_t)in[byte] >> skip; bc = 8 - skip; byte++; }
    }
    if (out.size() < n + 65536) out.resize(std::max<size_t>(2 * out.size(), n + (1u << 20)));
    uint16_t *o = out.data();
    size_t cap = out.size();
    int rc = -1;
#define REFILL()                                                                                                   \
    do {                                                                                                           \
        if (bc <= 32) {                                                                                            \
            if (byte + 8 <= in_n) { uint64_t w; memcpy(&w, in + byte, 8); bb |= w << bc; const unsigned k = (63 - bc) >> 3; byte += k; bc += 8 * k; } \
            else while (bc <= 56 && byte < in_n) { bb |= (uint64_t)in[byte++] << bc; bc += 8; }                       \
        }                                                                                                          \
    } while (0)
    for (;;) {
        REFILL();
        if (n + 300 > cap) {
            if (n > max_out) { rc = -3; break; }
            out.resize(2 * cap); o = out.data(); cap = out.size();
        }
        // literal / length symbol
        int s;
        {
            const uint16_t e = codes.lit.fast[bb & 0x7FFu];
            if (e) { const unsigned l = e & 15u; if (l > bc) break; bb >>= l; bc -= l; s = e >> 4; }
            else {
                int code = 0, first = 0, index = 0; s = -1;
                for (unsigned l = 1; l <= codes.lit.maxlen; l++) {
                    code |= (int)((bb >> (l - 1)) & 1u);
                    const int c = codes.lit.count[l];
                    if (code - c < first) { if (l > bc) break; s = codes.lit.symbol[index + (code - first)]; bb >>= l; bc -= l; break; }
                    index += c; first += c; first <<= 1; code <<= 1;
                }
                if (s < 0) break;
            }
        }
        if (s < 256) {
            if (text_only && !is_text((unsigned)s)) { rc = -2; break; }
            o[n++] = (uint16_t)s;
            continue;
        }
        if (s == 256) { rc = (int)final_block; break; }
        if (s > 285) break;
        REFILL();
        unsigned len = LEN_BASE[s - 257];
        { const unsigned x = LEN_EXTRA[s - 257]; if (x > bc) break; len += (unsigned)(bb & ((1u << x) - 1u)); bb >>= x; bc -= x; }
        int ds;
        {
            const uint16_t e = codes.dist.fast[bb & 0xFFu];
            if (e) { const unsigned l = e & 15u; if (l > bc) break; bb >>= l; bc -= l; ds = e >> 4; }
            else {
                int code = 0, first = 0, index = 0; ds = -1;
                for (unsigned l = 1; l <= codes.dist.maxlen; l++) {
                    code |= (int)((bb >> (l - 1)) & 1u);
                    const int c = codes.dist.count[l];
                    if (code - c < first) { if (l > bc) break; ds = codes.dist.symbol[index + (code - first)]; bb >>= l; bc -= l; break; }
                    index += c; first += c; first <<= 1; code <<= 1;
                }
                if (ds < 0) break;
            }
        }
        if (ds > 29) break;
        REFILL();
        size_t dist = DIST_BASE[ds];
        { const unsigned x = DIST_EXTRA[ds]; if (x > bc) break; dist += (size_t)(bb & ((1u << x) - 1u)); bb >>= x; bc -= x; }
        if (dist > n + 32768) break;                          // points in front of any possible window
        uint16_t *dst = o + n;
        if (dist <= n) {
            const uint16_t *src = dst - dist;
            if (dist >= len) memcpy(dst, src, 2 * (size_t)len);
            else if (dist == 1) { const uint16_t v = src[0]; for (unsigned i = 0; i < len; i++) dst[i] = v; }      // a run of one symbol (quality strings)
            else {
                // an overlapping copy repeats the last `dist` symbols: lay one period down, then double what is there
                memcpy(dst, src, 2 * dist);
                size_t copied = dist;
                while (copied < len) { const size_t c = std::min<size_t>(copied, len - copied); memcpy(dst + copied, dst, 2 * c); copied += c; }
            }
        } else {
            for (unsigned i = 0; i < len; i++) {
                const size_t at = n + i;                      // position being written
                if (dist <= at) dst[i] = o[at - dist];
                else {
                    const size_t w = 32768 - (dist - at);     // index into the window in front of the chunk
                    dst[i] = window ? (uint16_t)window[w] : (uint16_t)(0x8000u | w);
                }
            }
        }
        n += len;
    }
#undef REFILL
    n_io = n;
    if (rc < 0) return rc;
    // hand the position back: `byte` bytes were loaded, `bc` bits of them are still unread
    b.pos = 8ull * byte - bc;
    if (b.pos > 8ull * in_n) { b.over = true; return -1; }
    return rc;
}

// gzip member header at byte position `byte`; returns the byte position of the deflate data, 0 on error
static size_t gzip_header(const uint8_t *p, size_t n, size_t byte) {
    if (byte + 10 > n || p[byte] != 0x1f || p[byte + 1] != 0x8b || p[byte + 2] != 8) return 0;
    const unsigned flg = p[byte + 3];
    size_t at = byte + 10;
    if (flg & 4) { if (at + 2 > n) return 0; const size_t xlen = p[at] + 256u * p[at + 1]; at += 2 + xlen; }
    if (flg & 8) { while (at < n && p[at]) at++; at++; }
    if (flg & 16) { while (at < n && p[at]) at++; at++; }
    if (flg & 2) at += 2;
    return at <= n ? at : 0;
}

// Decodes from block start `start_bit` until the first block boundary >= stop_bit (a member end + the next member's header
// count as part of the block before), or the end of the stream.
int inflate_span(const uint8_t *data, size_t size, uint64_t start_bit, uint64_t stop_bit, const uint8_t *window, InflateSpan &out,
                 size_t max_out) {
    Bits b{data, size, start_bit};
    out.start_bit = start_bit;
    out.members.clear();
    out.end_of_stream = false;
    size_t n = out.sym.size();
    struct Trim { std::vector<uint16_t> &v; size_t &n; ~Trim() { v.resize(n); } } trim{out.sym, n};
    for (;;) {
        const int rc = inflate_block(b, out.sym, n, window, false, max_out);
        if (rc < 0) return rc;
        if (rc == 1) {
            // member trailer: CRC-32, ISIZE
            b.align();
            const size_t byte = (size_t)(b.pos >> 3);
            if (byte + 8 > size) return -1;
            InflateSpan::MemberEnd me;
            me.at = n;
            me.crc = data[byte] | data[byte + 1] << 8 | data[byte + 2] << 16 | (uint32_t)data[byte + 3] << 24;
            me.isize = data[byte + 4] | data[byte + 5] << 8 | data[byte + 6] << 16 | (uint32_t)data[byte + 7] << 24;
            out.members.push_back(me);
            size_t next = byte + 8;
            while (next < size && data[next] == 0) next++;      // zero padding behind the last member
            if (next >= size) { b.pos = 8ull * size; out.end_of_stream = true; break; }
            const size_t d = gzip_header(data, size, next);
            if (!d) return -4;                                   // trailing garbage
            b.pos = 8ull * d;
            window = nullptr;                                    // a member never points in front of its own start...
            // ... but positions are still measured from the chunk's start: make "in front of this member" unreachable by
            // keeping the symbols (a valid stream never asks for them); the marker branch above would flag a broken one
        }
        if (b.pos >= stop_bit) break;
    }
    out.end_bit = b.pos;
    return 0;
}

// The first bit position in [from_bit, to_bit) that starts a plausible non-final dynamic or stored block whose whole block decodes
// to text and whose successor header parses as well.  Returns false if there is none.
bool find_block_start(const uint8_t *data, size_t size, uint64_t from_bit, uint64_t to_bit, uint64_t &found) {
    std::vector<uint16_t> scratch;
    for (uint64_t p = from_bit; p < to_bit; p++) {
        Bits b{data, size, p};
        const uint32_t hdr = (uint32_t)b.peek(3);
        if (hdr & 1u) continue;                                  // final blocks are not searched for (one per member)
        const unsigned type = hdr >> 1;
        if (type != 2) continue;                                 // only dynamic-Huffman blocks are searched for: a stored block's header is three zero
                                                                 // bits in front of padding (ambiguous), a fixed one has no structure to verify —
                                                                 // gzip writes neither into text worth compressing; a chunk that starts with one is decoded in order
        {
            Bits s = b;
            s.skip(3);
            // cheap rejections before the tables are built
            const unsigned hlit = (unsigned)s.peek(5), hdist = (unsigned)(s.peek(10) >> 5);
            if (hlit > 29 || hdist > 29) continue;
        }
        size_t produced = 0;
        const int rc = inflate_block(b, scratch, produced, nullptr, true, (size_t)1 << 26);
        if (rc != 0) continue;                                   // error, or a final block after all
        if (produced < 64) continue;                             // a block of a few bytes proves nothing
        // the successor must look like a block too
        const uint32_t nh = (uint32_t)b.peek(3);
        if ((nh >> 1) == 3) continue;
        if ((nh >> 1) == 2) {
            Bits s = b;
            s.skip(3);
            static thread_local Codes probe;
            if (!read_dynamic(s, probe)) continue;
        }
        found = p;
        return true;
    }
    return false;
}

}  // namespace sgh
