// FASTQ text of SHORT lines -> per-read statistics in ONE pass over the text (gfx950).
//
// Replaces, like ppqc_fastq_gpu.hip, the record parsing the reference leaves to Biopython (SeqIO.parse(handle, "fastq"),
// lib/seq_routines.py:47-53).  For reads of a few hundred bases the three passes of that file (count the line ends,
// place the line starts, sum every line with both classifiers) cost 2.8 ms for 2.2 GB of 500-base reads; here the text
// is read once and nothing the size of the text is written:
//
//   fq1_scan_kernel   One wave per REGION of the text (consecutive 8 KiB chunks, some 4000 regions).  Every byte goes
//                     through ONE table lookup that yields, whatever line the byte lies in, its fixed-point error (as a
//                     quality), whether it is G/C/S, whether it is N, and whether it lies outside '!'..'~' ("odd": line
//                     ends, blanks, invalid bytes); the byte sum comes from v_sad_u8.  The wave keeps RUNNING PREFIXES
//                     of these five quantities and of the newline count over its region and writes them down at every
//                     line end (32 bytes per line, "points").  A line's sums are differences of two points, whichever
//                     kind of line it is: no line numbering, and so no global scan, is needed while the text is read.
//                     A lane owns a contiguous 128-byte strip of each chunk (eight 16-byte loads at a stride of 128
//                     bytes between lanes stream as fast as coalesced ones, tools/probes/strided_probe.hip), so the
//                     prefix runs inside a lane and ONE set of wave scans per chunk joins the lanes.  The few vectors
//                     that hold a newline are noted in LDS (the lane's running sums in front of them) and worked off
//                     together at the end of the chunk, 64 at a time with the lanes packed: the vector is read again
//                     (a cache hit) and the partial sums in front of each newline come from a masked second lookup.
//   fq1_regions_kernel  exclusive prefix over the regions' totals (a single block), line count and flags to the host.
//   fq1_finalize_kernel one thread per record: its five points -> lengths, stripped line ends, validation, the header
//                     fields, the per-read arrays fastq_finalize_kernel writes.  It reads the text only for the title
//                     and where a line end is not plain "\n" or "\r\n".
//
// What the pass cannot do it leaves to ppqc_fastq_gpu.hip (the host looks at the flags before anything is committed): a
// region with more lines than its share of the point table (kFq1LinesPerChunk per chunk: reads of fewer than ~100 bases
// throughout a region).  Results are bit-identical to that path and to the host packer + CSR path
// (tests/test_gpu_fastq.py runs every text through all of them).
#include "ppqc_fastq_dev.hpp"
#include "ppqc_internal.hpp"

namespace ppqc {

constexpr int kFq1Vecs = 8;              // 16-byte vectors of half a strip: a lane's 256-byte strip is two runs of eight
constexpr int kFq1Cap = 512;             // newline vectors of one chunk the wave's buffer holds (of 1024 vectors: a chunk with more
                                         // has lines of 32 bytes, four times what the point table is made for -> declined)
constexpr int kFq1Keep = 64;             // ... of which the first few keep their 16 bytes in LDS as well (500-base reads: 56 per chunk)
constexpr int kF1Gcs = 55, kF1N = 58, kF1Odd = 61;   // three-bit counters above the 55 bits of an error sum of <= 7 bytes
constexpr uint64_t kF1Err = (1ull << 55) - 1;
constexpr uint64_t kM48 = (1ull << 48) - 1;

// lut2[b]: '!'..'~': error of the quality b - 33 (the tile kernels' table) | G/C/S flag | N flag; other bytes: odd flag
// only (their clamped error is added by fq1_finalize where such a byte lies inside a quality line)
__device__ __forceinline__ uint64_t lut2_entry(uint32_t b, uint64_t err) {
    if (b < 33 || b > 126) return 1ull << kF1Odd;
    const uint32_t u = b & 0x5Fu;   // case bit dropped
    uint64_t v = err;
    if (u == 'G' || u == 'C' || u == 'S') v |= 1ull << kF1Gcs;
    if (u == 'N') v |= 1ull << kF1N;
    return v;
}

// gno = G/C/S | N << 12 | odd << 24: the three counts of some bytes side by side (the top one has eight bits: 255 bytes)
struct F1Sums { uint64_t err; uint32_t gno; };
__device__ __forceinline__ uint32_t gno_g(uint32_t gno) { return gno & 0xFFFu; }
__device__ __forceinline__ uint32_t gno_n(uint32_t gno) { return (gno >> 12) & 0xFFFu; }
__device__ __forceinline__ uint32_t gno_o(uint32_t gno) { return gno >> 24; }
__device__ __forceinline__ F1Sums f1_sums(const uint4& x, const uint64_t* lut2) {
    // three accumulators (6, 5 and 5 bytes): their error sums stay below 2^55 and their counters below 8
    const uint64_t a0 = lut2[x.x & 0xffu] + lut2[(x.x >> 8) & 0xffu] + lut2[(x.x >> 16) & 0xffu] + lut2[x.x >> 24] +
                        lut2[x.y & 0xffu] + lut2[(x.y >> 8) & 0xffu];
    const uint64_t a1 = lut2[(x.y >> 16) & 0xffu] + lut2[x.y >> 24] + lut2[x.z & 0xffu] + lut2[(x.z >> 8) & 0xffu] +
                        lut2[(x.z >> 16) & 0xffu];
    const uint64_t a2 = lut2[x.z >> 24] + lut2[x.w & 0xffu] + lut2[(x.w >> 8) & 0xffu] + lut2[(x.w >> 16) & 0xffu] +
                        lut2[x.w >> 24];
    // the nine counter bits of each accumulator; their plain sum times 2^55 is what the three counter fields add to
    // a0 + a1 + a2 (mod 2^64), so taking it off leaves the error sum (< 2^57) without masking each accumulator first
    const uint32_t y0 = (uint32_t)(a0 >> 55), y1 = (uint32_t)(a1 >> 55), y2 = (uint32_t)(a2 >> 55);
    F1Sums s;
    s.err = a0 + a1 + a2 - ((uint64_t)(y0 + y1 + y2) << 55);
    // a 24-bit multiply by 1 + 2^9 + 2^18 puts three copies of the nine bits side by side; G/C/S of the first, N of the
    // second and odd of the third lie 12 bits apart, and the three accumulators' fields (<= 6 each) add without carries
    s.gno = (__umul24(y0, 0x40201u) & 0x07007007u) + (__umul24(y1, 0x40201u) & 0x07007007u) + (__umul24(y2, 0x40201u) & 0x07007007u);
    return s;
}
__device__ __forceinline__ uint32_t nl_nomatch(uint32_t v) {   // bit 7 of every byte that is NOT '\n' (other bits: anything)
    const uint32_t x = v ^ 0x0A0A0A0Au;
    return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
}
// bit k: byte k of the word is '\n'
__device__ __forceinline__ uint32_t nl_nibble(uint32_t v) {
    uint32_t t = (~nl_nomatch(v) & 0x80808080u) >> 7;   // bits 0, 8, 16, 24
    t |= t >> 7;                                        // ... and 1, 9, 17
    t |= t >> 14;                                       // ... and 2, 3
    return t & 15u;
}
__device__ __forceinline__ uint32_t f1_bytesum(const uint4& x, uint32_t acc = 0u) {
    uint32_t q = __builtin_amdgcn_sad_u8(x.x, 0u, acc);
    q = __builtin_amdgcn_sad_u8(x.y, 0u, q);
    q = __builtin_amdgcn_sad_u8(x.z, 0u, q);
    return __builtin_amdgcn_sad_u8(x.w, 0u, q);
}

// The lane's vector at `pos` (16-byte aligned).  Bytes at and beyond n read as zero, except that a text whose last
// byte is no '\n' gets one at position n: its last line then ends like every other (the chunk count covers position n).
__device__ __forceinline__ uint4 f1_load(const uint8_t* __restrict__ text, uint64_t pos, uint64_t n, bool ends_nl) {
    if (pos + 16 <= n) return *reinterpret_cast<const uint4*>(text + pos);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (pos < n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + pos);   // the granule of an in-bounds byte lies in its page
        const int keep = (int)(n - pos);
        uint32_t m[4];
        range_masks(0, keep, m);
        w0 = v.x & m[0]; w1 = v.y & m[1]; w2 = v.z & m[2]; w3 = v.w & m[3];
    }
    if (!ends_nl && pos <= n && n < pos + 16) {
        const int at = (int)(n - pos);
        const uint32_t nl = 0x0Au << (8 * (at & 3));
        w0 |= (at >> 2) == 0 ? nl : 0u; w1 |= (at >> 2) == 1 ? nl : 0u;
        w2 |= (at >> 2) == 2 ? nl : 0u; w3 |= (at >> 2) == 3 ? nl : 0u;
    }
    return make_uint4(w0, w1, w2, w3);
}

// The four bytes b - 2 .. b + 1 around byte b (0 .. 15) of the vector x, whose neighbours in the text are the words
// `pre` (the four bytes in front of it) and `post` (the four behind it): two of the six words picked by compares and
// one v_alignbyte, all in registers (indexing the vector as an array sends it through scratch memory).
__device__ __forceinline__ uint32_t f1_window(uint32_t pre, const uint4& x, uint32_t post, int b) {
    const int q = b + 2;            // offset of byte b - 2 in the 24-byte window
    const int k = q >> 2;           // 0 .. 4
    const uint32_t lo_a = (k & 1) ? x.x : pre, lo_b = (k & 1) ? x.z : x.y;
    const uint32_t hi_a = (k & 1) ? x.y : x.x, hi_b = (k & 1) ? x.w : x.z;
    uint32_t lo = (k & 2) ? lo_b : lo_a, hi = (k & 2) ? hi_b : hi_a;
    lo = (k & 4) ? x.w : lo;
    hi = (k & 4) ? post : hi;
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(q & 3));
}

// point = the running prefixes just in front of a line end, relative to the region's start (32 bytes):
//   q0: error sum, low 48 bits | class of the byte behind the newline << 48 (1 '@', 2 '+', 0 other) | what is in front
//       of it << 50 (0: no white space, 1: exactly one '\r', 3: look at the text)
//   q1: error sum >> 48 (32 bits) | byte sum << 32
//   q2: G/C/S count | N count << 24
//   q3: odd bytes | position << 24          (a region holds at most 2^24 bytes)
__global__ __launch_bounds__(256) void fq1_scan_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_chunks,
                                                       uint32_t cpr, const uint64_t* __restrict__ lut_g,
                                                       uint64_t* __restrict__ points, uint32_t cap_lines,
                                                       Fq1Region* __restrict__ regions) {
    __shared__ uint64_t lut2[256];
    // the newline vectors of the chunk, each with the lane's running sums in front of it:
    //   x, y: error sum (60 bits) | bits 6, 7 of the lane's newlines in front of it << 60;  z: gno
    //   w: byte sum (16 bits) | lane * 16 + vector << 16 | bits 0 .. 5 of the newlines << 26
    __shared__ uint4 br[4][kFq1Cap];
    __shared__ uint4 bx[4][kFq1Keep];   // the first kFq1Keep of those vectors themselves (the others are read again)
    __shared__ uint2 bn[4][kFq1Keep];   // ... with the word in front of each and the word behind it, where the lane holds them
    lut2[threadIdx.x] = lut2_entry(threadIdx.x, lut_g[threadIdx.x]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint64_t region = (uint64_t)blockIdx.x * 4 + wib;
    const uint64_t c0 = region * cpr;
    if (c0 >= n_chunks) return;
    const uint64_t c1 = c0 + cpr < n_chunks ? c0 + cpr : n_chunks;
    const bool ends_nl = text[n - 1] == '\n';
    uint64_t* const out = points + region * (uint64_t)cap_lines * 4;
    // the region's running totals (wave-uniform)
    uint32_t B_lines = 0, B_gcs = 0, B_n = 0, B_odd = 0;
    uint64_t B_qb = 0, B_a = 0, B_b = 0;   // error sum = B_a + B_b * 2^48, B_a < 2^48
    bool over = false;
    for (uint64_t c = c0; c < c1; ++c) {
        const uint64_t cb = c * kFq1Chunk;
        const bool inside = cb + kFq1Chunk <= n;   // wave-uniform: all of the chunk lies in the text
        uint64_t r_err = 0;                       // the lane's running sums over its strip, in front of the vector at hand
        uint32_t r_gno = 0, r_qb = 0, r_nl = 0;   // (r_gno: its last vector kept apart, so that no field of it passes 240)
        uint32_t last_gno = 0;
        uint32_t tailw = 0;                       // last word of the strip's first half
        uint32_t ne = 0;                          // newline vectors of the chunk so far (wave-uniform)
        for (int h = 0; h < 2; ++h) {
            const uint64_t sb = cb + (uint64_t)lane * 256 + (uint64_t)h * 128;
            uint4 vv[kFq1Vecs];
            if (inside) {
#pragma unroll
                for (int j = 0; j < kFq1Vecs; ++j) vv[j] = *reinterpret_cast<const uint4*>(text + sb + (uint64_t)j * 16);
            } else {
#pragma unroll
                for (int j = 0; j < kFq1Vecs; ++j) vv[j] = f1_load(text, sb + (uint64_t)j * 16, n, ends_nl);
            }
#pragma unroll
            for (int j = 0; j < kFq1Vecs; ++j) {
                const uint4& x = vv[j];
                const F1Sums s = f1_sums(x, lut2);
                r_gno += last_gno;
                // t: bit 7 of a byte is clear iff the byte is '\n'; the four words folded before the test
                const uint32_t t0 = nl_nomatch(x.x), t1 = nl_nomatch(x.y), t2 = nl_nomatch(x.z), t3 = nl_nomatch(x.w);
                const uint32_t any = ~(t0 & t1 & t2 & t3) & 0x80808080u;
                const uint64_t ball = __ballot(any != 0);
                if (ball) {
                    const uint32_t at = ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
                    if (any != 0) {
                        if (at < (uint32_t)kFq1Cap) {
                            br[wib][at] = make_uint4((uint32_t)r_err, (uint32_t)(r_err >> 32) | ((r_nl >> 6) << 28), r_gno,
                                                     r_qb | ((uint32_t)(lane * 16 + h * 8 + j) << 16) | ((r_nl & 63u) << 26));
                            if (at < (uint32_t)kFq1Keep) {
                                bx[wib][at] = x;
                                // the neighbours in the lane's registers (the strip's two outer ends and the first half's
                                // last vector have none here: the compact pass reads the text where it needs them)
                                bn[wib][at] = make_uint2(j ? vv[j ? j - 1 : 0].w : tailw, vv[j < kFq1Vecs - 1 ? j + 1 : j].x);
                            }
                        } else {
                            over = true;
                        }
                    }
                    ne += (uint32_t)__popcll(ball);
                    r_nl += __popc(((~t0 & 0x80808080u) >> 7) | ((~t1 & 0x80808080u) >> 6) | ((~t2 & 0x80808080u) >> 5) | ((~t3 & 0x80808080u) >> 4));
                }
                r_err += s.err; last_gno = s.gno; r_qb = f1_bytesum(x, r_qb);
            }
            tailw = vv[kFq1Vecs - 1].w;
        }
        if (ne > (uint32_t)kFq1Cap) ne = kFq1Cap;   // (the region is flagged: its points will not be used)
        // join the lanes: six inclusive scans of the lane totals (every partial sum fits 32 bits: 256 bytes per lane)
        const uint32_t o1 = (gno_g(r_gno) + gno_g(last_gno)) | ((gno_n(r_gno) + gno_n(last_gno)) << 16);   // G/C/S | N << 16
        const uint32_t o2 = r_qb;                                                                        // byte sum
        const uint32_t o3 = (gno_o(r_gno) + gno_o(last_gno)) | (r_nl << 16);                              // odd | newlines << 16
        const uint32_t o4 = (uint32_t)r_err & 0x3FFFFFFu, o5 = (uint32_t)(r_err >> 26) & 0x3FFFFFFu, o6 = (uint32_t)(r_err >> 52);
        const uint32_t i1 = wave_scan_incl_u32(o1), i2 = wave_scan_incl_u32(o2), i3 = wave_scan_incl_u32(o3);
        const uint32_t i4 = wave_scan_incl_u32(o4), i5 = wave_scan_incl_u32(o5), i6 = wave_scan_incl_u32(o6);
        for (uint32_t eb = 0; eb < ne; eb += 64) {
            // the newline vectors of the chunk, one per lane, 64 at a time
            const bool live = eb + (uint32_t)lane < ne;
            uint4 x = make_uint4(0, 0, 0, 0), rr = make_uint4(0, 0, 0, 0);
            if (live) rr = br[wib][eb + lane];
            const uint32_t vi = (rr.w >> 16) & 0x3FFu;
            const uint64_t gpos0 = cb + (uint64_t)vi * 16;
            uint32_t pre = 0x78787878u, post = 0;   // the text around the vector ('x': no white space in front of the text)
            const bool kept = eb + (uint32_t)lane < (uint32_t)kFq1Keep;
            if (live) {
                if (kept) {
                    x = bx[wib][eb + lane];
                    const uint2 nb = bn[wib][eb + lane];
                    pre = nb.x; post = nb.y;
                } else {
                    x = f1_load(text, gpos0, n, ends_nl);
                }
            }
            const int src = (int)(vi >> 4);   // the lane that owns the vector
            const uint32_t e1 = (uint32_t)__shfl((int)(i1 - o1), src), e2 = (uint32_t)__shfl((int)(i2 - o2), src);
            const uint32_t e3 = (uint32_t)__shfl((int)(i3 - o3), src), e4 = (uint32_t)__shfl((int)(i4 - o4), src);
            const uint32_t e5 = (uint32_t)__shfl((int)(i5 - o5), src), e6 = (uint32_t)__shfl((int)(i6 - o6), src);
            const uint32_t gcs0 = B_gcs + (e1 & 0xFFFFu) + gno_g(rr.z), n0 = B_n + (e1 >> 16) + gno_n(rr.z);
            const uint32_t odd0 = B_odd + (e3 & 0xFFFFu) + gno_o(rr.z);
            uint32_t line = B_lines + (e3 >> 16) + ((rr.w >> 26) | ((rr.y >> 28) << 6));
            const uint64_t qb0 = B_qb + e2 + (rr.w & 0xFFFFu);
            const uint64_t sum0 = B_a + e4 + ((uint64_t)e5 << 26) + (((uint64_t)(rr.y & 0x0FFFFFFFu) << 32) | rr.x);   // < 2^62
            const uint64_t bb0 = B_b + ((uint64_t)e6 << 4);
            const uint32_t rel0 = (uint32_t)(c - c0) * kFq1Chunk + vi * 16;
            uint32_t bits = nl_nibble(x.x) | (nl_nibble(x.y) << 4) | (nl_nibble(x.z) << 8) | (nl_nibble(x.w) << 12);
            if (!live) bits = 0;
            // the neighbouring words from the text where LDS does not hold them and a newline stands where they matter (the
            // vector's first two bytes, its last): a few per cent of the vectors -- reading them for every vector made
            // the kernel fetch a fifth more than the text, the lines having left the L2 by then
            if ((bits & 3u) != 0 && !(kept && (vi & 15u) != 0) && gpos0 >= 4)
                pre = *reinterpret_cast<const uint32_t*>(text + gpos0 - 4);
            if ((bits >> 15) != 0 && !(kept && (vi & 7u) != 7u)) {
                post = 0;
                if (gpos0 + 20 <= n) post = *reinterpret_cast<const uint32_t*>(text + gpos0 + 16);
                else if (gpos0 + 16 < n) post = (uint32_t)text[gpos0 + 16];   // its first byte is all that is looked at
            }
            // sums of the vector's bytes in front of the newline at hand: from a masked lookup of the whole vector for
            // its first newline, for a further one up to three bytes on ("\n+\n") by adding the bytes between
            int pb = -1;
            uint64_t perr = 0;
            uint32_t pgno = 0, pqb = 0;
            while (__ballot(bits != 0) != 0) {
                if (bits != 0) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    const uint32_t win = f1_window(pre, x, post, b);   // bytes b - 2, b - 1, b ('\n'), b + 1
                    const uint32_t prev2 = win & 0xFFu, prev = (win >> 8) & 0xFFu, next = win >> 24;
                    if (pb < 0 || b - pb > 3) {
                        uint32_t m[4];
                        range_masks(0, b, m);
                        const uint4 xm = make_uint4(x.x & m[0], x.y & m[1], x.z & m[2], x.w & m[3]);
                        const F1Sums p = f1_sums(xm, lut2);   // the 16 - b zeroed bytes count as odd
                        perr = p.err; pgno = p.gno - ((uint32_t)(16 - b) << 24);
                        pqb = f1_bytesum(xm);
                    } else {
                        pgno += 1u << 24; pqb += '\n';   // the newline at pb
                        if (b - pb == 3) {
                            const uint64_t v = lut2[prev2];
                            const uint32_t y = (uint32_t)(v >> 55);
                            perr += v & kF1Err; pgno += __umul24(y, 0x40201u) & 0x07007007u; pqb += prev2;
                        }
                        if (b - pb >= 2) {
                            const uint64_t v = lut2[prev];
                            const uint32_t y = (uint32_t)(v >> 55);
                            perr += v & kF1Err; pgno += __umul24(y, 0x40201u) & 0x07007007u; pqb += prev;
                        }
                    }
                    pb = b;
                    // the bytes around the line end
                    uint32_t tw = 0;
                    if (is_space((uint8_t)prev)) tw = (prev == '\r' && b >= 2 && !is_space((uint8_t)prev2)) ? 1u : 3u;
                    const uint32_t cls = next == '@' ? 1u : (next == '+' ? 2u : 0u);
                    const uint64_t S = sum0 + perr;
                    const uint64_t q0 = (S & kM48) | ((uint64_t)cls << 48) | ((uint64_t)tw << 50);
                    const uint64_t q1 = ((bb0 + (S >> 48)) & 0xFFFFFFFFull) | ((qb0 + pqb) << 32);
                    const uint64_t q2 = (uint64_t)(gcs0 + gno_g(pgno)) | ((uint64_t)(n0 + gno_n(pgno)) << 24);
                    const uint64_t q3 = (uint64_t)(odd0 + gno_o(pgno)) | ((uint64_t)(rel0 + (uint32_t)b) << 24);
                    if (line < cap_lines) {
                        ulonglong2* o = reinterpret_cast<ulonglong2*>(out + (uint64_t)line * 4);
                        o[0] = make_ulonglong2(q0, q1);
                        o[1] = make_ulonglong2(q2, q3);
                    } else {
                        over = true;
                    }
                    ++line;
                }
            }
        }
        // the chunk's totals onto the region's
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63), t2 = (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
        const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)i3, 63), t4 = (uint32_t)__builtin_amdgcn_readlane((int)i4, 63);
        const uint32_t t5 = (uint32_t)__builtin_amdgcn_readlane((int)i5, 63), t6 = (uint32_t)__builtin_amdgcn_readlane((int)i6, 63);
        B_gcs += t1 & 0xFFFFu; B_n += t1 >> 16; B_odd += t3 & 0xFFFFu; B_lines += t3 >> 16; B_qb += t2;
        const uint64_t S = B_a + t4 + ((uint64_t)t5 << 26);
        B_a = S & kM48;
        B_b += (S >> 48) + ((uint64_t)t6 << 4);
    }
    const bool any_over = __ballot(over) != 0;
    if (lane == 0) {
        Fq1Region r;
        r.lines = B_lines; r.gcs = B_gcs; r.n = B_n; r.odd = B_odd; r.qb = B_qb; r.err_a = B_a; r.err_b = B_b;
        r.flags = any_over || B_lines > cap_lines ? 2u : 0u;
        regions[region] = r;
    }
}

// Single block: regions[k] becomes the exclusive prefix of the regions' totals; head = {lines of the text, OR of the
// flags}, also to the pinned words the host waits for.  Every thread adds up a run of regions; each wave scans its 64
// runs with shuffles, field by field, and one barrier joins the eight waves (the error sum's low part as a plain sum:
// 2^15 regions of 2^48 at most).
constexpr int kRegThreads = 512;   // eight waves: room for 256 registers each (eight 64-bit fields, their loads in flight)
__global__ __launch_bounds__(kRegThreads) void fq1_regions_kernel(Fq1Region* __restrict__ regions, uint64_t n_regions,
                                                                  uint64_t* __restrict__ head, uint64_t* __restrict__ host_out,
                                                                  uint64_t ticket) {
    __shared__ uint64_t wsum[8][kRegThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const uint64_t seg = (n_regions + kRegThreads - 1) / kRegThreads;
    const uint64_t b = t * seg < n_regions ? t * seg : n_regions, e = b + seg < n_regions ? b + seg : n_regions;
    uint64_t tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // lines, gcs, n, odd, qb, err_a, err_b, flags
    for (uint64_t i = b; i < e; ++i) {
        const Fq1Region v = regions[i];
        tot[0] += v.lines; tot[1] += v.gcs; tot[2] += v.n; tot[3] += v.odd; tot[4] += v.qb; tot[5] += v.err_a;
        tot[6] += v.err_b; tot[7] |= v.flags;
    }
    uint64_t run[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) run[f] = tot[f];
#pragma unroll
    for (uint32_t o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const uint64_t x = __shfl_up((unsigned long long)run[f], o);
            if (lane >= o) run[f] = f == 7 ? (run[f] | x) : run[f] + x;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int f = 0; f < 8; ++f) wsum[f][w] = run[f];
    }
    __syncthreads();
    uint64_t all_flags = 0;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        uint64_t before = 0;
        for (uint32_t k = 0; k < (uint32_t)(kRegThreads / 64); ++k) {
            const uint64_t x = wsum[f][k];
            if (f == 7) all_flags |= x;
            else if (k < w) before += x;
        }
        if (f < 7) run[f] += before - tot[f];   // exclusive
    }
    if (t == kRegThreads - 1) {
        head[0] = run[0] + tot[0]; head[1] = all_flags;
        if (host_out) {   // pinned words the host polls: the two results, then (behind a fence) this launch's ticket
            host_out[0] = run[0] + tot[0]; host_out[1] = all_flags;
            __threadfence_system();
            __hip_atomic_store(&host_out[4], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (uint64_t i = b; i < e; ++i) {
        const Fq1Region v = regions[i];
        Fq1Region o;
        o.lines = run[0]; o.gcs = run[1]; o.n = run[2]; o.odd = run[3]; o.qb = run[4];
        o.err_a = run[5] & kM48; o.err_b = run[6] + (run[5] >> 48); o.flags = 0;
        regions[i] = o;
        run[0] += v.lines; run[1] += v.gcs; run[2] += v.n; run[3] += v.odd; run[4] += v.qb; run[5] += v.err_a; run[6] += v.err_b;
    }
}

// ---- the title's fields --------------------------------------------------------------------------------------------
// ch= and start_time= of the title [a, e) ('@' at a, line end stripped or not), by the rules of the host packer
// (ppqc_fastq.cpp: a key counts where it stands at the start of the title or behind a blank or tab, and only its first
// occurrence; "ch=" is followed by decimal digits, "start_time=" by YYYY-MM-DD(T or blank)HH:MM:SS) -- the same
// fields HeaderScan finds byte by byte.  Here the title is searched by words for '=' (a handful per title), and only
// the bytes around each are looked at: some 500 instructions per title instead of 3000.  n_cr: its '\r' bytes.
__device__ __forceinline__ uint32_t byte_eq_mask(uint32_t v, uint32_t k4) {   // high bit of each byte that equals the byte of k4
    const uint32_t x = v ^ k4;
    const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu;
    return ~t;
}
__device__ __forceinline__ uint32_t eq_mask(uint32_t v) { return byte_eq_mask(v, 0x3D3D3D3Du); }
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {   // high bit of each byte of x that is zero (exact)
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
struct TitleFields {
    uint32_t ch;
    int64_t time;
    uint32_t n_cr;
};
__device__ __forceinline__ bool f1_blank(uint32_t c) { return c == ' ' || c == '\t'; }
// An '=' at position P of the text whose left neighbour is 'h' or 'e': the 32 bytes text[P - 12 .. P + 20) in two
// unaligned loads (cache hits: the title has just been streamed), every byte of them at a fixed place.
struct KeyWindow {
    uint32_t w[8];
    __device__ __forceinline__ uint32_t at(int i) const { return (w[(i + 12) >> 2] >> (8 * ((i + 12) & 3))) & 0xffu; }   // text[P + i]
    __device__ __forceinline__ void load(const uint8_t* __restrict__ text, uint64_t P) { __builtin_memcpy(w, text + P - 12, 32); }
};
// "ch=" with its '=' at P: true if it is the key (at the title's start t0 or behind a blank); *ch = its digits
__device__ __forceinline__ bool title_ch(const uint8_t* __restrict__ text, const KeyWindow& k, uint64_t t0, uint64_t e,
                                         uint64_t P, uint32_t* ch) {
    if (!(k.at(-2) == 'c' && P >= t0 + 2 && (P - 2 == t0 || f1_blank(k.at(-3))))) return false;
    uint32_t c = 0;
    bool open = true;     // digits so far
#pragma unroll
    for (int i = 1; i < 20; ++i) {
        const uint32_t d = k.at(i) - '0';
        open = open && d <= 9u && P + (uint64_t)i < e;
        if (open) c = c * 10u + d;
    }
    if (open)             // twenty digits and more: the host packer goes on (and wraps around)
        for (uint64_t p = P + 20; p < e; ++p) {
            const uint32_t d = (uint32_t)text[p] - '0';
            if (d > 9u) break;
            c = c * 10u + d;
        }
    *ch = c;
    return true;
}
// the same for "start_time="; *time stays INT64_MIN if what follows is no time
__device__ __forceinline__ bool title_time(const KeyWindow& k, uint64_t t0, uint64_t e, uint64_t P, int64_t* time) {
    if (!(P >= t0 + 10 && (P - 10 == t0 || f1_blank(k.at(-11))))) return false;
    const char key[] = "start_tim";
    bool same = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) same = same && k.at(i - 10) == (uint32_t)(uint8_t)key[i];
    if (!same) return false;
    if (P + 20 > e) return true;
    uint32_t f[6] = {0, 0, 0, 0, 0, 0};   // year, month, day, hour, minute, second
    bool good = true;
#pragma unroll
    for (int i = 0; i < 19; ++i) {
        const uint32_t c = k.at(1 + i);
        const bool sep = i == 4 || i == 7 || i == 10 || i == 13 || i == 16;
        if (sep) good = good && (i == 10 ? (c == 'T' || c == ' ') : (i < 10 ? c == '-' : c == ':'));
        else {
            const uint32_t d = c - '0';
            good = good && d <= 9u;
            const int fi = i < 4 ? 0 : (i - 2) / 3;   // 5,6 -> 1; 8,9 -> 2; 11,12 -> 3; 14,15 -> 4; 17,18 -> 5
            f[fi] = f[fi] * 10u + d;
        }
    }
    if (good && f[1] >= 1 && f[1] <= 12 && f[2] >= 1 && f[2] <= 31)
        *time = days_from_civil((int64_t)f[0], f[1], f[2]) * 86400 + (int64_t)f[3] * 3600 + (int64_t)f[4] * 60 + (int64_t)f[5];
    return true;
}
// Two steps, so that the lanes of a wave stay together: first the whole title by words, noting where "h=" and "e="
// stand (up to four each, in order); then the candidates, every lane its first, its second, ...  The loads come in
// batches -- kTG granules of the title at a time, one batch ahead (two: with four the kernel held 119 registers and four
// waves per SIMD, with two 96 and five: 271 -> 261 us at 2 * 10^6 records; one spills); the windows of the first two
// candidates together --
// because the kernel is a chain of memory latencies and little else.
constexpr int kTG = 2;
__device__ __forceinline__ TitleFields parse_title(const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t e) {
    TitleFields out{0, INT64_MIN, 0};
    if (e <= a + 1) return out;
    bool slow = a < 12 || e + 20 > n || e - a > 60000;   // the key windows would leave the text (the first and the last record at most)
    uint64_t hc = 0, ec = 0;     // title indices of the candidates, 16 bits each
    uint32_t nh = 0, ne = 0;
    bool many = false;           // two candidates in one word, or more than four in the title
    if (!slow) {
        const int len = (int)(e - a);
        const uintptr_t base = (uintptr_t)(text + a), endp = (uintptr_t)(text + e);
        uintptr_t ga = base & ~(uintptr_t)15;
        int k = -(int)(base & 15u);           // index in the title of the granule's first byte
        uint32_t last_w = 0, any_cr = 0;      // the word in front of the one looked at; '\r' bytes seen
        uint4 cur[kTG], nxt[kTG];
#pragma unroll
        for (int i = 0; i < kTG; ++i) {
            cur[i] = make_uint4(0, 0, 0, 0);
            if (ga + 16 * i < endp) cur[i] = *reinterpret_cast<const uint4*>(ga + 16 * i);
        }
        while (ga < endp) {
#pragma unroll
            for (int i = 0; i < kTG; ++i) {
                nxt[i] = make_uint4(0, 0, 0, 0);
                if (ga + 16 * kTG + 16 * i < endp) nxt[i] = *reinterpret_cast<const uint4*>(ga + 16 * kTG + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < kTG; ++i) {
                // the granule's bytes outside the title (the '@' at index 0 too) become zero: they match nothing
                const int kg = k + 16 * i;
                const int x0 = 1 - kg < 0 ? 0 : (1 - kg > 16 ? 16 : 1 - kg), x1 = len - kg < 0 ? 0 : (len - kg > 16 ? 16 : len - kg);
                uint32_t m[4];
                range_masks(x0, x1 < x0 ? x0 : x1, m);
                const uint32_t w[4] = {cur[i].x & m[0], cur[i].y & m[1], cur[i].z & m[2], cur[i].w & m[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // byte b is a candidate when it is '=' and the byte in front of it 'h' (or 'e'): one zero-byte test
                    // of (w ^ "====") | (the stream shifted by a byte ^ "hhhh")
                    const uint32_t before = __builtin_amdgcn_alignbit(w[q], last_w, 24);   // byte b: the byte in front of byte b of w
                    last_w = w[q];
                    const uint32_t xe = w[q] ^ 0x3D3D3D3Du;
                    const uint32_t c1 = zero_bytes(xe | (before ^ 0x68686868u)), c2 = zero_bytes(xe | (before ^ 0x65656565u));
                    any_cr |= zero_bytes(w[q] ^ 0x0D0D0D0Du);
                    if (c1) {
                        if (nh < 4) hc |= (uint64_t)(uint32_t)(kg + 4 * q + ((__ffs((int)c1) - 1) >> 3)) << (16 * nh);
                        ++nh;
                        many = many || (c1 & (c1 - 1)) != 0;
                    }
                    if (c2) {
                        if (ne < 4) ec |= (uint64_t)(uint32_t)(kg + 4 * q + ((__ffs((int)c2) - 1) >> 3)) << (16 * ne);
                        ++ne;
                        many = many || (c2 & (c2 - 1)) != 0;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kTG; ++i) cur[i] = nxt[i];
            ga += 16 * kTG; k += 16 * kTG;
        }
        if (any_cr)
            for (uint64_t p = a + 1; p < e; ++p) out.n_cr += text[p] == '\r';
        bool ch_done = false, st_done = false;
        KeyWindow kh, ke;   // the first candidate of each key, loaded together
        if (nh) kh.load(text, a + (hc & 0xFFFFu));
        if (ne) ke.load(text, a + (ec & 0xFFFFu));
        if (nh) ch_done = title_ch(text, kh, a + 1, e, a + (hc & 0xFFFFu), &out.ch);
        if (ne) st_done = title_time(ke, a + 1, e, a + (ec & 0xFFFFu), &out.time);
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if ((uint32_t)i < nh && !ch_done) {
                kh.load(text, a + ((hc >> (16 * i)) & 0xFFFFu));
                ch_done = title_ch(text, kh, a + 1, e, a + ((hc >> (16 * i)) & 0xFFFFu), &out.ch);
            }
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if ((uint32_t)i < ne && !st_done) {
                ke.load(text, a + ((ec >> (16 * i)) & 0xFFFFu));
                st_done = title_time(ke, a + 1, e, a + ((ec >> (16 * i)) & 0xFFFFu), &out.time);
            }
        slow = many || (nh > 4 && !ch_done) || (ne > 4 && !st_done);   // candidates that were not noted
    }
    if (slow) {   // byte by byte
        const HeaderFields hf = scan_header<false, false>(text, a, e);
        out.ch = hf.ch; out.time = hf.time; out.n_cr = 0;
        for (uint64_t p = a; p < e; ++p) out.n_cr += text[p] == '\r';
    }
    return out;
}

// ---- records from points -------------------------------------------------------------------------------------------
struct F1Point {
    uint64_t pos, gcs, n, odd, qb, ea, eb;   // absolute position of the line end; prefixes of the text in front of it
    uint32_t cls, tw;
};
__device__ __forceinline__ F1Point f1_decode(const ulonglong2& a, const ulonglong2& b, const Fq1Region& R, uint64_t reg,
                                             uint64_t region_bytes);
// the end of line k of the text; reg: a region whose first line is at or before k (moved forward as needed), R: the
// prefix of region held (the record's five points lie in one region, seldom in two)
__device__ __forceinline__ F1Point f1_point(const uint64_t* __restrict__ points, const Fq1Region* __restrict__ rb,
                                            uint64_t n_regions, uint64_t& reg, Fq1Region& R, uint64_t& held, uint64_t k,
                                            uint32_t cap_lines, uint64_t region_bytes) {
    while (reg + 1 < n_regions && rb[reg + 1].lines <= k) ++reg;
    if (held != reg) { R = rb[reg]; held = reg; }
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(points + (reg * (uint64_t)cap_lines + (k - R.lines)) * 4);
    return f1_decode(q[0], q[1], R, reg, region_bytes);
}
__device__ __forceinline__ F1Point f1_decode(const ulonglong2& a, const ulonglong2& b, const Fq1Region& R, uint64_t reg,
                                             uint64_t region_bytes) {
    F1Point p;
    p.pos = reg * region_bytes + ((b.y >> 24) & 0xFFFFFFull);
    p.gcs = R.gcs + (b.x & 0xFFFFFFull); p.n = R.n + ((b.x >> 24) & 0xFFFFFFull); p.odd = R.odd + (b.y & 0xFFFFFFull);
    p.qb = R.qb + (a.y >> 32);
    const uint64_t S = R.err_a + (a.x & kM48);
    p.ea = S & kM48; p.eb = R.err_b + (a.y & 0xFFFFFFFFull) + (S >> 48);
    p.cls = (uint32_t)(a.x >> 48) & 3u; p.tw = (uint32_t)(a.x >> 50) & 3u;
    return p;
}

// What lies between two consecutive line ends P and Q: one line, its newline not counted.
struct F1Line {
    uint64_t a, e;            // [a, e): the line without its '\n'
    uint64_t gcs, n, odd, qb;
    unsigned __int128 err;
};
__device__ __forceinline__ F1Line f1_line(const F1Point& P, const F1Point& Q) {
    F1Line l;
    l.a = P.pos + 1; l.e = Q.pos;
    l.gcs = Q.gcs - P.gcs; l.n = Q.n - P.n;
    l.odd = Q.odd - P.odd - 1;     // the '\n' at P.pos is odd and lies between the two prefixes
    l.qb = Q.qb - P.qb - 10;
    l.err = (((unsigned __int128)Q.eb << 48) + Q.ea) - (((unsigned __int128)P.eb << 48) + P.ea);
    return l;
}

// Strips the line end of a sequence (QUAL = false) or quality line and settles what its odd bytes are.  Returns the
// stripped length; qb and err become those of the stripped line as the other paths compute them (invalid bytes inside a
// quality line: counted in *bad, error as clamped by the table); *misplaced counts white space inside the line.
template <bool QUAL>
__device__ __forceinline__ uint64_t f1_strip(const uint8_t* __restrict__ text, F1Line& l, uint32_t tw,
                                             const uint64_t* __restrict__ lut_g, uint32_t* misplaced, uint32_t* bad) {
    const uint64_t raw = l.e - l.a;
    if (l.odd == 0) return raw;
    if (l.odd == 1 && tw == 1) {   // "\r\n"
        l.qb -= '\r';
        return raw - 1;
    }
    uint64_t e = l.e;
    while (e > l.a && is_space(text[e - 1])) { --e; l.qb -= text[e]; }
    if (l.e - e != l.odd) {        // odd bytes inside the line: look at every byte
        for (uint64_t p = l.a; p < e; ++p) {
            const uint8_t ch = text[p];
            if (ch >= 33 && ch <= 126) continue;
            if (is_space(ch)) ++*misplaced;
            else if (QUAL) { ++*bad; l.err += lut_g[ch]; }
        }
    }
    return e - l.a;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void fq1_finalize_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_records,
                                                           const uint64_t* __restrict__ points,
                                                           const Fq1Region* __restrict__ rb, uint64_t n_regions,
                                                           uint32_t cap_lines, uint64_t region_bytes, ReadArrays ra,
                                                           uint64_t read_base, const uint64_t* __restrict__ lut_g,
                                                           QThr thr, uint8_t mux, uint64_t* __restrict__ errors,
                                                           uint32_t n_channels, uint64_t* __restrict__ gate,
                                                           float reg_per_line, Fq1Index ix,
                                                           const uint64_t* __restrict__ head, uint64_t max_records,
                                                           uint64_t* __restrict__ host_out, uint64_t ticket) {
    __shared__ unsigned int s_mis, s_bad, s_max;
    if (head != nullptr) {
        if (host_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
            // the pinned words the host polls (launch_fq1_scan): from here, so that the regions kernel in front ends
            // without stores to host memory behind it and this kernel starts at once
            host_out[0] = head[0]; host_out[1] = head[1];
            __threadfence_system();
            __hip_atomic_store(&host_out[4], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // launched before the host knew the text's lines (launch_fq1_finalize_early): the count comes from the regions
        // kernel's words, and a text the host will not commit as it stands (declined by the scan, lines that are no
        // records, more records than the arrays hold) is left alone -- the host sees the same words and does the rest
        const uint64_t lines = head[0];
        if (head[1] != 0 || (lines & 3u) != 0 || (lines >> 2) > max_records) return;
        n_records = lines >> 2;
        reg_per_line = (float)n_regions / (float)(lines ? lines : 1);
    }
    if (threadIdx.x == 0) { s_mis = 0; s_bad = 0; s_max = 0; }
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t misplaced = 0, bad = 0;
    if (r < n_records) {
        const uint64_t g = read_base + r;
        // the region of the record's first point: lines are spread evenly over a text, so the region in proportion is the
        // right one or next to it (a binary search is thirteen dependent loads)
        const uint64_t k0 = r ? 4 * r - 1 : 0;
        uint64_t reg = (uint64_t)((float)k0 * reg_per_line);
        if (reg >= n_regions) reg = n_regions - 1;
        Fq1Region R = rb[reg];
        uint64_t next_first = reg + 1 < n_regions ? rb[reg + 1].lines : ~0ull;
        if (R.lines > k0 || (next_first <= k0 && reg + 2 < n_regions && rb[reg + 2].lines <= k0)) {
            // not there and not next to it (a text whose line lengths drift): binary search over the regions' first lines
            uint64_t lo = 0, hi = n_regions;   // rb[lo].lines <= k0 < rb[hi].lines (rb[n_regions] = +inf)
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (rb[mid].lines <= k0) lo = mid; else hi = mid;
            }
            reg = lo;
            R = rb[reg];
            next_first = reg + 1 < n_regions ? rb[reg + 1].lines : ~0ull;
        }
        uint64_t held = reg;
        F1Point P[5];
        if (r != 0 && R.lines <= k0 && 4 * r + 3 < next_first) {
            // the five points lie in one region (nearly always): their ten loads go out together
            const ulonglong2* q = reinterpret_cast<const ulonglong2*>(points + (reg * (uint64_t)cap_lines + (k0 - R.lines)) * 4);
            ulonglong2 v[10];
#pragma unroll
            for (int t = 0; t < 10; ++t) v[t] = q[t];
#pragma unroll
            for (int t = 0; t < 5; ++t) P[t] = f1_decode(v[2 * t], v[2 * t + 1], R, reg, region_bytes);
        } else {
            if (r) P[0] = f1_point(points, rb, n_regions, reg, R, held, 4 * r - 1, cap_lines, region_bytes);
            else {   // in front of the text: as if a newline ended at position -1
                P[0].pos = ~0ull; P[0].gcs = 0; P[0].n = 0; P[0].odd = ~0ull; P[0].qb = (uint64_t)(-10ll); P[0].ea = 0; P[0].eb = 0;
                P[0].cls = text[0] == '@' ? 1u : 0u; P[0].tw = 0;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) P[t + 1] = f1_point(points, rb, n_regions, reg, R, held, 4 * r + t, cap_lines, region_bytes);
        }
        F1Line H = f1_line(P[0], P[1]), S = f1_line(P[1], P[2]), U = f1_line(P[2], P[3]), Q = f1_line(P[3], P[4]);
        // title: '@', and no '\r' but at its end (a bare CR is a line end to SeqIO); the fields
        const TitleFields hf = parse_title(text, n, H.a, H.e - (P[1].tw == 1 ? 1 : 0));   // without the '\r' of a plain "\r\n"
        bool ok = P[0].cls == 1 && P[2].cls == 2;
        if (hf.n_cr != 0) {   // is any of them inside?
            uint64_t he = H.e;
            while (he > H.a && is_space(text[he - 1])) --he;
            bool inside = false;
            for (uint64_t p = H.a; p < he; ++p) inside = inside || text[p] == '\r';
            if (inside) ++misplaced;
        }
        // '+' line: alone, or followed by the title again
        {
            uint64_t e = U.e;
            if (U.odd != 0 && !(U.odd == 1 && P[3].tw == 1 && (--e, true))) {
                e = U.e;
                while (e > U.a && is_space(text[e - 1])) --e;
            }
            if (ok && e - U.a > 1) {
                uint64_t he = H.e;
                while (he > H.a && is_space(text[he - 1])) --he;
                ok = e - U.a == he - H.a;
                for (uint64_t i = 1; ok && i < e - U.a; ++i) ok = text[U.a + i] == text[H.a + i];
            }
        }
        const uint64_t L = f1_strip<false>(text, S, P[2].tw, lut_g, &misplaced, &bad);
        const uint64_t Lq = f1_strip<true>(text, Q, P[4].tw, lut_g, &misplaced, &bad);
        ok = ok && L == Lq && L < 0xFFFFFFFFull;
        uint64_t hi32 = 0, lo32 = 0;
        if (!ok) {
            atomicAdd((unsigned long long*)&errors[1], 1ull);
            atomicMin((unsigned long long*)&errors[2], (unsigned long long)r);
        } else if (L != 0) {
            hi32 = (uint64_t)(Q.err >> 32); lo32 = (uint64_t)Q.err & 0xffffffffull;
        }
        const uint64_t Lr = ok ? L : 0;
        uint8_t flags = 0;
        double mq = 0.0;
        if (Lr != 0) flags = quality_flags(Lr, hi32, lo32, thr, &mq);
        else flags = 0.0 < thr.fail_q ? 1 : 0;
        ra.len[g] = (uint32_t)Lr; ra.gc[g] = ok ? (uint32_t)S.gcs : 0u; ra.nn[g] = ok ? (uint32_t)S.n : 0u;
        ra.qsum[g] = Lr ? (int64_t)Q.qb - 33 * (int64_t)Lr : 0;
        ra.err_hi[g] = hi32; ra.err_lo[g] = lo32;
        ra.mean_q[g] = mq;
        ra.flags[g] = flags;
        const uint32_t ch = ok ? hf.ch : 0u;
        ra.channel[g] = ch;
        ra.mux[g] = mux;
        ra.time[g] = ok ? hf.time : INT64_MIN;
        // the sort key and the sort gate, as finalize_reads leaves them: the sort + N50 chain may start behind this kernel
        const uint32_t key = (ch >= 1 && ch <= n_channels) ? (uint32_t)Lr : 0u;
        ra.lens32[g] = key;
        if (key) atomicMax(&s_max, key);
        if (ix.on) ra.hour[g] = index_cell(ch, (uint32_t)Lr, ok ? hf.time : INT64_MIN, mux, ix.tp, ix.cells);   // time_mode 0
    }
    misplaced = wave_sum_u32(misplaced);
    bad = wave_sum_u32(bad);
    if ((threadIdx.x & 63) == 0) {
        if (misplaced) atomicAdd(&s_mis, misplaced);
        if (bad) atomicAdd(&s_bad, bad);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_mis) atomicAdd((unsigned long long*)&errors[6], (unsigned long long)s_mis);
        if (s_bad) atomicAdd((unsigned long long*)&errors[3], (unsigned long long)s_bad);
        // by atomicMax: the words still bound every key by their OR, and their largest is the largest length of the texts
        // (tables_kernel takes it from here when no extremes kernel has run in front of it)
        if (s_max) atomicMax((unsigned long long*)&gate[blockIdx.x & (kGateWords - 1)], (unsigned long long)s_max);
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
uint64_t fq1_chunks(uint64_t n) { return n / kFq1Chunk + 1; }   // position n lies in the last chunk
void fq1_shape(uint64_t n, int want_regions, uint32_t* cpr, uint64_t* n_regions) {
    const uint64_t n_chunks = fq1_chunks(n);
    uint64_t want = want_regions > 0 ? (uint64_t)want_regions : 4096;
    uint64_t c = (n_chunks + want - 1) / want;
    if (c < 1) c = 1;
    if (c > kFq1MaxRegionChunks) c = kFq1MaxRegionChunks;
    *cpr = (uint32_t)c;
    *n_regions = (n_chunks + c - 1) / c;
}

hipError_t launch_fq1_scan(const uint8_t* text, uint64_t n, const Fq1Scratch& f, uint32_t cpr, uint64_t n_regions,
                           const uint64_t* lut_dev, uint64_t* host_out, uint64_t ticket, hipStream_t s) {
    const uint64_t n_chunks = fq1_chunks(n);
    hipLaunchKernelGGL(fq1_scan_kernel, dim3((unsigned)((n_regions + 3) / 4)), dim3(256), 0, s, text, n, n_chunks, cpr, lut_dev,
                       f.points, cpr * kFq1LinesPerChunk, f.regions);
    hipLaunchKernelGGL(fq1_regions_kernel, dim3(1), dim3(kRegThreads), 0, s, f.regions, n_regions, f.head, host_out, ticket);
    return hipGetLastError();
}

hipError_t launch_fq1_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const Fq1Scratch& f, uint32_t cpr,
                               uint64_t n_regions, ReadArrays ra, uint64_t read_base, const uint64_t* lut_dev,
                               const QThr& thr, uint8_t mux, uint64_t* errors, uint32_t n_channels, uint64_t* gate,
                               const Fq1Index& ix, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    hipLaunchKernelGGL(fq1_finalize_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n,
                       n_records, f.points, f.regions, n_regions, cpr * kFq1LinesPerChunk, (uint64_t)cpr * kFq1Chunk, ra,
                       read_base, lut_dev, thr, mux, errors, n_channels, gate,
                       (float)((double)n_regions / (4.0 * (double)n_records)), ix, (const uint64_t*)nullptr, (uint64_t)0,
                       (uint64_t*)nullptr, (uint64_t)0);
    return hipGetLastError();
}

// The same, enqueued straight behind the regions kernel, before the host has seen the line count: the grid is sized for
// max_records records (a thread each), f.head tells the kernel how many there are, and it does nothing unless the text is
// one the host will commit with at most max_records records (the host applies the same test to the same words).
hipError_t launch_fq1_finalize_early(const uint8_t* text, uint64_t n, uint64_t max_records, const Fq1Scratch& f, uint32_t cpr, uint64_t n_regions, ReadArrays ra,
                                     uint64_t read_base, const uint64_t* lut_dev, const QThr& thr, uint8_t mux,
                                     uint64_t* errors, uint32_t n_channels, uint64_t* gate, const Fq1Index& ix,
                                     uint64_t* host_out, uint64_t ticket, hipStream_t s) {
    if (max_records == 0) max_records = 1;   // (the kernel must run: it hands the line count to the host)
    hipLaunchKernelGGL(fq1_finalize_kernel, dim3((unsigned)((max_records + 255) / 256)), dim3(256), 0, s, text, n,
                       (uint64_t)0, f.points, f.regions, n_regions, cpr * kFq1LinesPerChunk, (uint64_t)cpr * kFq1Chunk, ra,
                       read_base, lut_dev, thr, mux, errors, n_channels, gate, 0.0f, ix, (const uint64_t*)f.head, max_records,
                       host_out, ticket);
    return hipGetLastError();
}

}  // namespace ppqc
