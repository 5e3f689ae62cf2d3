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
//                     that hold a newline are set aside in LDS and worked off together at the end of the chunk (the
//                     partial sums in front of each newline: a masked second lookup), with the lanes packed.
//   fq1_regions_kernel  exclusive prefix over the regions' totals (a single block), line count and flags to the host.
//   fq1_finalize_kernel one thread per record: its five points -> lengths, stripped line ends, validation, the header
//                     fields, the per-read arrays fastq_finalize_kernel writes.  It reads the text only for the title
//                     and where a line end is not plain "\n" or "\r\n".
//
// What the pass cannot do it leaves to ppqc_fastq_gpu.hip (the host looks at the flags before anything is committed): a
// chunk with more than kFq1Cap newline vectors or a region with more lines than its share of the point table (reads of
// fewer than ~100 bases).  Results are bit-identical to that path and to the host packer + CSR path
// (tests/test_gpu_fastq.py runs every text through all of them).
#include "ppqc_fastq_dev.hpp"
#include "ppqc_internal.hpp"

namespace ppqc {

constexpr int kFq1Vecs = 8;              // 16-byte vectors of a lane's strip
constexpr int kFq1Cap = 64;              // newline vectors of one chunk the wave's buffer holds
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

struct F1Sums { uint64_t err; uint32_t gno; };   // gno = G/C/S | N << 8 | odd << 16
__device__ __forceinline__ F1Sums f1_sums(const uint4& x, const uint64_t* lut2) {
    // three accumulators (6, 5 and 5 bytes): their error sums stay below 2^55 and their counters below 8
    const uint64_t a0 = lut2[x.x & 0xffu] + lut2[(x.x >> 8) & 0xffu] + lut2[(x.x >> 16) & 0xffu] + lut2[x.x >> 24] +
                        lut2[x.y & 0xffu] + lut2[(x.y >> 8) & 0xffu];
    const uint64_t a1 = lut2[(x.y >> 16) & 0xffu] + lut2[x.y >> 24] + lut2[x.z & 0xffu] + lut2[(x.z >> 8) & 0xffu] +
                        lut2[(x.z >> 16) & 0xffu];
    const uint64_t a2 = lut2[x.z >> 24] + lut2[x.w & 0xffu] + lut2[(x.w >> 8) & 0xffu] + lut2[(x.w >> 16) & 0xffu] +
                        lut2[x.w >> 24];
    F1Sums s;
    s.err = (a0 & kF1Err) + (a1 & kF1Err) + (a2 & kF1Err);
    auto widen = [](uint64_t a) {
        const uint32_t c = (uint32_t)(a >> 55);
        return (c & 7u) | ((c & 0x38u) << 5) | ((c & 0x1C0u) << 10);
    };
    s.gno = widen(a0) + widen(a1) + widen(a2);
    return s;
}
__device__ __forceinline__ uint32_t f1_bytesum(const uint4& x) {
    uint32_t q = __builtin_amdgcn_sad_u8(x.x, 0u, 0u);
    q = __builtin_amdgcn_sad_u8(x.y, 0u, q);
    q = __builtin_amdgcn_sad_u8(x.z, 0u, q);
    return __builtin_amdgcn_sad_u8(x.w, 0u, q);
}
__device__ __forceinline__ uint32_t f1_byte(const uint4& x, int i) {
    const uint32_t w = i < 8 ? (i < 4 ? x.x : x.y) : (i < 12 ? x.z : x.w);
    return (w >> (8 * (i & 3))) & 0xffu;
}

// The lane's vector at `pos` (16-byte aligned).  Bytes at and beyond n read as zero, except that a text whose last
// byte is no '\n' gets one at position n: its last line then ends like every other (the chunk count covers position n).
__device__ __forceinline__ uint4 f1_load(const uint8_t* __restrict__ text, uint64_t pos, uint64_t n, bool ends_nl) {
    if (pos + 16 <= n) return *reinterpret_cast<const uint4*>(text + pos);
    uint32_t w[4] = {0, 0, 0, 0};
    if (pos < n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + pos);   // the granule of an in-bounds byte lies in its page
        const int keep = (int)(n - pos);
        uint32_t m[4];
        range_masks(0, keep, m);
        w[0] = v.x & m[0]; w[1] = v.y & m[1]; w[2] = v.z & m[2]; w[3] = v.w & m[3];
    }
    if (!ends_nl && pos <= n && n < pos + 16) {
        const int at = (int)(n - pos);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((at >> 2) == k) w[k] |= 0x0Au << (8 * (at & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
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
    __shared__ uint4 bx[4][kFq1Cap];      // the newline vectors of the chunk ...
    __shared__ uint4 br[4][kFq1Cap];      // ... the lane's running sums in front of each: error (2 words), gno, byte sum
    __shared__ uint32_t bm[4][kFq1Cap];   // ... lane * 8 + vector | the lane's newlines in front of it << 16
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
    bool dense = false, over = false;
    for (uint64_t c = c0; c < c1; ++c) {
        const uint64_t cb = c * kNlChunk;
        uint4 vv[kFq1Vecs];
#pragma unroll
        for (int j = 0; j < kFq1Vecs; ++j) vv[j] = f1_load(text, cb + (uint64_t)lane * 128 + (uint64_t)j * 16, n, ends_nl);
        uint64_t r_err = 0;                       // the lane's running sums over its strip
        uint32_t r_gno = 0, r_qb = 0, r_nl = 0;
        uint32_t ne = 0;                          // newline vectors of the chunk so far (wave-uniform)
#pragma unroll
        for (int j = 0; j < kFq1Vecs; ++j) {
            const uint4& x = vv[j];
            const F1Sums s = f1_sums(x, lut2);
            const uint32_t m0 = nl_mask(x.x), m1 = nl_mask(x.y), m2 = nl_mask(x.z), m3 = nl_mask(x.w);
            const uint32_t any = m0 | m1 | m2 | m3;
            const uint64_t ball = __ballot(any != 0);
            if (ball) {
                const uint32_t at = ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
                if (any != 0 && at < (uint32_t)kFq1Cap) {
                    bx[wib][at] = x;
                    br[wib][at] = make_uint4((uint32_t)r_err, (uint32_t)(r_err >> 32), r_gno, r_qb);
                    bm[wib][at] = (uint32_t)(lane * 8 + j) | (r_nl << 16);
                }
                ne += (uint32_t)__popcll(ball);
                r_nl += __popc((m0 >> 7) | (m1 >> 6) | (m2 >> 5) | (m3 >> 4));
            }
            r_err += s.err; r_gno += s.gno; r_qb += f1_bytesum(x);
        }
        // join the lanes: six inclusive scans of the lane totals (every partial sum fits 32 bits: 128 bytes per lane)
        const uint32_t o1 = (r_gno & 0xFFu) | (((r_gno >> 8) & 0xFFu) << 16);                 // G/C/S | N << 16
        const uint32_t o2 = r_qb;                                                             // byte sum
        const uint32_t o3 = ((r_gno >> 16) & 0xFFu) | (r_nl << 16);                           // odd | newlines << 16
        const uint32_t o4 = (uint32_t)r_err & 0x3FFFFFFu, o5 = (uint32_t)(r_err >> 26) & 0x3FFFFFFu, o6 = (uint32_t)(r_err >> 52);
        const uint32_t i1 = wave_scan_incl_u32(o1), i2 = wave_scan_incl_u32(o2), i3 = wave_scan_incl_u32(o3);
        const uint32_t i4 = wave_scan_incl_u32(o4), i5 = wave_scan_incl_u32(o5), i6 = wave_scan_incl_u32(o6);
        if (ne > (uint32_t)kFq1Cap) dense = true;
        const uint32_t nq = ne < (uint32_t)kFq1Cap ? ne : (uint32_t)kFq1Cap;
        if (nq) {
            // the newline vectors of the chunk, one per lane: lane q takes entry q
            const bool live = (uint32_t)lane < nq;
            uint4 x = make_uint4(0, 0, 0, 0), rr = make_uint4(0, 0, 0, 0);
            uint32_t meta = 0;
            if (live) { x = bx[wib][lane]; rr = br[wib][lane]; meta = bm[wib][lane]; }
            const int src = (int)((meta & 0xFFFFu) >> 3);   // the lane that owns the vector
            const uint32_t e1 = (uint32_t)__shfl((int)(i1 - o1), src), e2 = (uint32_t)__shfl((int)(i2 - o2), src);
            const uint32_t e3 = (uint32_t)__shfl((int)(i3 - o3), src), e4 = (uint32_t)__shfl((int)(i4 - o4), src);
            const uint32_t e5 = (uint32_t)__shfl((int)(i5 - o5), src), e6 = (uint32_t)__shfl((int)(i6 - o6), src);
            const uint32_t vi = meta & 0xFFFFu;
            const uint32_t gcs0 = B_gcs + (e1 & 0xFFFFu) + (rr.z & 0xFFu), n0 = B_n + (e1 >> 16) + ((rr.z >> 8) & 0xFFu);
            const uint32_t odd0 = B_odd + (e3 & 0xFFFFu) + ((rr.z >> 16) & 0xFFu);
            uint32_t line = B_lines + (e3 >> 16) + (meta >> 16);
            const uint64_t qb0 = B_qb + e2 + rr.w;
            const uint64_t sum0 = B_a + e4 + ((uint64_t)e5 << 26) + (((uint64_t)rr.y << 32) | rr.x);   // < 2^61
            const uint64_t bb0 = B_b + ((uint64_t)e6 << 4);
            const uint32_t rel0 = (uint32_t)(c - c0) * kNlChunk + vi * 16;
            const uint64_t gpos0 = cb + (uint64_t)vi * 16;
            uint32_t bits = live ? vec_nl_bits(x, 16) : 0u;
            while (__ballot(bits != 0) != 0) {
                if (bits != 0) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    uint32_t m[4];
                    range_masks(0, b, m);
                    const uint4 xm = make_uint4(x.x & m[0], x.y & m[1], x.z & m[2], x.w & m[3]);
                    const F1Sums p = f1_sums(xm, lut2);   // the 16 - b zeroed bytes count as odd
                    const uint32_t podd = ((p.gno >> 16) & 0xFFu) - (uint32_t)(16 - b);
                    const uint64_t gpos = gpos0 + (uint64_t)b;
                    // the bytes around the line end
                    uint32_t prev = 'x', next = 0;
                    if (b >= 1) prev = f1_byte(x, b - 1);
                    else if (gpos > 0) prev = text[gpos - 1];
                    if (gpos + 1 < n) next = b < 15 ? f1_byte(x, b + 1) : (uint32_t)text[gpos + 1];
                    uint32_t tw = 0;
                    if (is_space((uint8_t)prev)) tw = (prev == '\r' && b >= 2 && !is_space((uint8_t)f1_byte(x, b - 2))) ? 1u : 3u;
                    const uint32_t cls = next == '@' ? 1u : (next == '+' ? 2u : 0u);
                    const uint64_t S = sum0 + p.err;
                    const uint64_t q0 = (S & kM48) | ((uint64_t)cls << 48) | ((uint64_t)tw << 50);
                    const uint64_t q1 = ((bb0 + (S >> 48)) & 0xFFFFFFFFull) | ((qb0 + f1_bytesum(xm)) << 32);
                    const uint64_t q2 = (uint64_t)(gcs0 + (p.gno & 0xFFu)) | ((uint64_t)(n0 + ((p.gno >> 8) & 0xFFu)) << 24);
                    const uint64_t q3 = (uint64_t)(odd0 + podd) | ((uint64_t)(rel0 + (uint32_t)b) << 24);
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
        r.flags = (dense ? 1u : 0u) | (any_over || B_lines > cap_lines ? 2u : 0u);
        regions[region] = r;
    }
}

// Single block: regions[k] becomes the exclusive prefix of the regions' totals; head = {lines of the text, OR of the
// flags}, also to the pinned words the host waits for.
__global__ __launch_bounds__(256) void fq1_regions_kernel(Fq1Region* __restrict__ regions, uint64_t n_regions,
                                                          uint64_t* __restrict__ head, uint64_t* __restrict__ host_out) {
    __shared__ Fq1Region part[256];
    const uint32_t t = threadIdx.x;
    const uint64_t seg = (n_regions + 255) / 256;
    const uint64_t b = t * seg, e = b + seg < n_regions ? b + seg : n_regions;
    Fq1Region s = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [](Fq1Region& a, const Fq1Region& v) {
        a.lines += v.lines; a.gcs += v.gcs; a.n += v.n; a.odd += v.odd; a.qb += v.qb;
        const uint64_t S = a.err_a + v.err_a;
        a.err_a = S & kM48; a.err_b += v.err_b + (S >> 48);
        a.flags |= v.flags;
    };
    for (uint64_t i = b; i < e; ++i) add(s, regions[i]);
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        Fq1Region run = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i) { const Fq1Region v = part[i]; part[i] = run; add(run, v); }
        head[0] = run.lines; head[1] = run.flags;
        if (host_out) {
            host_out[0] = run.lines; host_out[1] = run.flags;
            __threadfence_system();
        }
    }
    __syncthreads();
    Fq1Region run = part[t];
    run.flags = 0;
    for (uint64_t i = b; i < e; ++i) { const Fq1Region v = regions[i]; regions[i] = run; add(run, v); }
}

// ---- records from points -------------------------------------------------------------------------------------------
struct F1Point {
    uint64_t pos, gcs, n, odd, qb, ea, eb;   // absolute position of the line end; prefixes of the text in front of it
    uint32_t cls, tw;
};
// the end of line k of the text; reg: a region whose first line is at or before k (moved forward as needed)
__device__ __forceinline__ F1Point f1_point(const uint64_t* __restrict__ points, const Fq1Region* __restrict__ rb,
                                            uint64_t n_regions, uint64_t& reg, uint64_t k, uint32_t cap_lines,
                                            uint64_t region_bytes) {
    while (reg + 1 < n_regions && rb[reg + 1].lines <= k) ++reg;
    const Fq1Region R = rb[reg];
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(points + (reg * (uint64_t)cap_lines + (k - R.lines)) * 4);
    const ulonglong2 a = q[0], b = q[1];
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

__global__ __launch_bounds__(256) void fq1_finalize_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_records,
                                                           const uint64_t* __restrict__ points,
                                                           const Fq1Region* __restrict__ rb, uint64_t n_regions,
                                                           uint32_t cap_lines, uint64_t region_bytes, ReadArrays ra,
                                                           uint64_t read_base, const uint64_t* __restrict__ lut_g,
                                                           QThr thr, uint8_t mux, uint64_t* __restrict__ errors) {
    __shared__ unsigned int s_mis, s_bad;
    if (threadIdx.x == 0) { s_mis = 0; s_bad = 0; }
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t misplaced = 0, bad = 0;
    if (r < n_records) {
        const uint64_t g = read_base + r;
        // the region of the record's first point: binary search over the regions' first lines
        const uint64_t k0 = r ? 4 * r - 1 : 0;
        uint64_t lo = 0, hi = n_regions;   // rb[lo].lines <= k0 < rb[hi].lines (rb[n_regions] = +inf)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rb[mid].lines <= k0) lo = mid; else hi = mid;
        }
        uint64_t reg = lo;
        F1Point P[5];
        if (r) P[0] = f1_point(points, rb, n_regions, reg, 4 * r - 1, cap_lines, region_bytes);
        else {   // in front of the text: as if a newline ended at position -1
            P[0].pos = ~0ull; P[0].gcs = 0; P[0].n = 0; P[0].odd = ~0ull; P[0].qb = (uint64_t)(-10ll); P[0].ea = 0; P[0].eb = 0;
            P[0].cls = text[0] == '@' ? 1u : 0u; P[0].tw = 0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) P[t + 1] = f1_point(points, rb, n_regions, reg, 4 * r + t, cap_lines, region_bytes);
        F1Line H = f1_line(P[0], P[1]), S = f1_line(P[1], P[2]), U = f1_line(P[2], P[3]), Q = f1_line(P[3], P[4]);
        // title: '@', and no '\r' but at its end (a bare CR is a line end to SeqIO); the fields
        const HeaderFields hf = scan_header<false, true>(text, H.a, H.e + (H.e < n ? 1 : 0));
        bool ok = P[0].cls == 1 && P[2].cls == 2;
        if (hf.cr_inside) ++misplaced;
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
        ra.channel[g] = ok ? hf.ch : 0u;
        ra.mux[g] = mux;
        ra.time[g] = ok ? hf.time : INT64_MIN;
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
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
uint64_t fq1_chunks(uint64_t n) { return n / kNlChunk + 1; }   // position n lies in the last chunk
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
                           const uint64_t* lut_dev, uint64_t* host_out, hipStream_t s) {
    const uint64_t n_chunks = fq1_chunks(n);
    hipLaunchKernelGGL(fq1_scan_kernel, dim3((unsigned)((n_regions + 3) / 4)), dim3(256), 0, s, text, n, n_chunks, cpr, lut_dev,
                       f.points, cpr * kFq1LinesPerChunk, f.regions);
    hipLaunchKernelGGL(fq1_regions_kernel, dim3(1), dim3(256), 0, s, f.regions, n_regions, f.head, host_out);
    return hipGetLastError();
}

hipError_t launch_fq1_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const Fq1Scratch& f, uint32_t cpr,
                               uint64_t n_regions, ReadArrays ra, uint64_t read_base, const uint64_t* lut_dev,
                               const QThr& thr, uint8_t mux, uint64_t* errors, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    hipLaunchKernelGGL(fq1_finalize_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n, n_records,
                       f.points, f.regions, n_regions, cpr * kFq1LinesPerChunk, (uint64_t)cpr * kNlChunk, ra, read_base,
                       lut_dev, thr, mux, errors);
    return hipGetLastError();
}

}  // namespace ppqc
