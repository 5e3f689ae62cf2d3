// Device helpers shared by the FASTQ text kernels (ppqc_fastq_gpu.hip, ppqc_fastq1.hip): byte masks on 4-byte words,
// the header automaton for the albacore fields ch= / start_time= (lib/seq_routines.py:51).
#pragma once
#include "ppqc_internal.hpp"

namespace ppqc {

// high bit of each byte set where the byte equals '\n' (exact zero-byte detector on v ^ 0x0A..)
__device__ __forceinline__ uint32_t nl_mask(uint32_t v) {
    const uint32_t x = v ^ 0x0A0A0A0Au;
    const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu;
    return ~t;
}

__device__ __forceinline__ uint32_t cr_mask(uint32_t v) {   // the same for '\r'
    const uint32_t x = v ^ 0x0D0D0D0Du;
    const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu;
    return ~t;
}

// 16-bit mask (bit k = byte k of the lane's 16-byte vector is a newline), bytes at or beyond `valid` cleared
__device__ __forceinline__ uint32_t vec_nl_bits(const uint4& v, int valid) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = nl_mask(w[k]);
        bits |= (((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u)) << (4 * k);
    }
    if (valid < 16) bits &= valid <= 0 ? 0u : ((1u << valid) - 1u);
    return bits;
}

// high bit of each byte set where the byte is ASCII white space (blank, \t \n \v \f \r): what rstrip() removes at a line end
__device__ __forceinline__ uint32_t ws_mask(uint32_t x) {
    const uint32_t H = 0x80808080u, hi = x & H, lo7 = x & 0x7F7F7F7Fu;
    const uint32_t ge9 = (lo7 + 0x77777777u) & H, ge14 = (lo7 + 0x72727272u) & H, ge32 = (lo7 + 0x60606060u) & H;
    const uint32_t ge33 = (lo7 + 0x5F5F5F5Fu) & H;
    return ((ge9 & ~ge14) | (ge32 & ~ge33)) & ~hi;
}
// number of white-space bytes among the first `valid` bytes of the vector
__device__ __forceinline__ uint32_t vec_ws_count(const uint4& v, int valid) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t m = ws_mask(w[k]);
        const int left = valid - 4 * k;   // bytes of this word that count
        if (left < 4) m &= left <= 0 ? 0u : ((1u << (8 * left)) - 1u);
        c += __popc(m);
    }
    return c;
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

__device__ inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

// m[w]: the bytes of word w of a 16-byte vector that lie in [x0, x1), 0 <= x0 < x1 <= 16 (32-bit shifts only)
__device__ __forceinline__ void range_masks(int x0, int x1, uint32_t m[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int l = x0 - 4 * w, h = x1 - 4 * w;
        const uint32_t below_h = h >= 4 ? 0xFFFFFFFFu : (h <= 0 ? 0u : (1u << (8 * h)) - 1u);
        const uint32_t below_l = l >= 4 ? 0xFFFFFFFFu : (l <= 0 ? 0u : (1u << (8 * l)) - 1u);
        m[w] = below_h & ~below_l;
    }
}

// Two small automata over the bytes of a title, one per key: a key starts behind the '@' or behind a blank or tab and only
// its first occurrence counts, "ch=" is followed by decimal digits, "start_time=" by the 19 characters
// YYYY-MM-DD(T or blank)HH:MM:SS (the rules of the host packer, ppqc_fastq.cpp).
struct HeaderScan {
    int cm = 0;         // "ch=": 0..2 characters matched (0: a key may start here), 3 digits follow, 4 done, -1 no key before the next blank
    int ts = 0;         // "start_time=": 0..10 matched, 11 the time follows, 12 parsed, 13 does not parse, -1 as above
    int ti = 0;         // characters of the time seen
    uint32_t ch = 0;
    uint64_t tacc = 0;  // YYYYMMDDHHMMSS as one decimal number

    __device__ __forceinline__ void step(uint32_t c) {
        const bool blank = c == ' ' || c == '\t';
        const uint32_t d = c - '0';
        const bool digit = d <= 9u;
        if (cm == 3) {
            if (digit) ch = ch * 10u + d; else cm = 4;
        } else if (cm < 3) {
            const uint32_t want = (0x3D6863u >> (8 * (cm < 0 ? 0 : cm))) & 0xffu;   // "ch="
            cm = (cm >= 0 && c == want) ? cm + 1 : (blank ? 0 : -1);
        }
        if (ts == 11) {
            const bool sep_pos = ti == 4 || ti == 7 || ti == 10 || ti == 13 || ti == 16;
            const bool sep_ok = ti == 10 ? (c == 'T' || c == ' ') : (ti < 10 ? c == '-' : c == ':');
            if (sep_pos ? sep_ok : digit) {
                if (!sep_pos) tacc = tacc * 10u + d;
                if (++ti == 19) ts = 12;
            } else {
                ts = 13;
            }
        } else if (ts < 11) {
            const int q = ts < 0 ? 0 : ts;
            const uint64_t K0 = 0x69745F7472617473ull, K1 = 0x3D656Dull;   // "start_ti", "me="
            const uint32_t want = (uint32_t)((q < 8 ? K0 >> (8 * q) : K1 >> (8 * (q - 8))) & 0xffu);
            ts = (ts >= 0 && c == want) ? ts + 1 : (blank ? 0 : -1);
        }
    }
    __device__ __forceinline__ bool done() const { return cm == 4 && ts >= 12; }
    __device__ __forceinline__ int64_t time() const {
        if (ts != 12) return INT64_MIN;
        uint64_t x = tacc;
        const int64_t ss = (int64_t)(x % 100); x /= 100;
        const int64_t mi = (int64_t)(x % 100); x /= 100;
        const int64_t hh = (int64_t)(x % 100); x /= 100;
        const unsigned dd = (unsigned)(x % 100); x /= 100;
        const unsigned mo = (unsigned)(x % 100); x /= 100;
        if (mo < 1 || mo > 12 || dd < 1 || dd > 31) return INT64_MIN;
        return days_from_civil((int64_t)x, mo, dd) * 86400 + hh * 3600 + mi * 60 + ss;
    }
};


// The title [a, e0) of a record streamed through registers in aligned 16-byte granules (a granule that holds a byte of
// the text lies in that byte's page; the next one is in flight while one is looked at): the two fields; BLANKS: the
// number of blanks, \t, \v and \f in it (what a title may hold anywhere); CR: whether a '\r' is followed by anything but
// white space (a bare CR line end, which neither SeqIO's universal newlines nor the host packer take as part of a title).
struct HeaderFields {
    uint32_t ch;
    int64_t time;
    uint32_t blanks;
    bool cr_inside;
};
template <bool BLANKS, bool CR>
__device__ __forceinline__ HeaderFields scan_header(const uint8_t* __restrict__ text, uint64_t a, uint64_t e0) {
    const int64_t len = (int64_t)(e0 - a);
    const uintptr_t base = (uintptr_t)(text + a), endp = (uintptr_t)(text + e0);
    uintptr_t ga = base & ~(uintptr_t)15;
    int64_t k = -(int64_t)(base & 15u);   // index in the title of the granule's first byte
    HeaderScan h;
    HeaderFields out{0, INT64_MIN, 0, false};
    bool seen_cr = false;
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (ga < endp) cur = *reinterpret_cast<const uint4*>(ga);
    while (ga < endp) {
        const uintptr_t gn = ga + 16;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (gn < endp) nxt = *reinterpret_cast<const uint4*>(gn);
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        if (!h.done()) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t kk = k + t;
                if (kk >= 1 && kk < len) h.step((w[t >> 2] >> (8 * (t & 3))) & 0xffu);   // byte 0 is the '@'
            }
        }
        if (BLANKS || CR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t k0 = k + 4 * q;   // index in the title of the word's first byte
                uint32_t inr = 0xFFFFFFFFu;     // the word's bytes that lie in the title
                if (k0 < 0) inr &= k0 <= -4 ? 0u : ~((1u << (8 * (int)(-k0))) - 1u);
                if (k0 + 4 > len) inr &= k0 >= len ? 0u : ((1u << (8 * (int)(len - k0))) - 1u);
                const uint32_t ws = ws_mask(w[q]), cr = cr_mask(w[q]) & inr;
                if (BLANKS) out.blanks += __popc(ws & ~nl_mask(w[q]) & ~cr & inr);
                if (CR) {
                    const uint32_t solid = ~ws & 0x80808080u & inr;   // bytes that are no white space
                    if (seen_cr && solid) out.cr_inside = true;
                    if (cr) {
                        const uint32_t first = cr & (0u - cr);
                        if (solid & ~((first << 1) - 1u)) out.cr_inside = true;
                        seen_cr = true;
                    }
                }
            }
        } else if (h.done()) {
            break;
        }
        cur = nxt; ga = gn; k += 16;
    }
    out.ch = h.ch;
    out.time = h.time();
    return out;
}

}  // namespace ppqc
