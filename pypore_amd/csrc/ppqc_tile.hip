// Per-read statistics: wave-level segmented reduction over the flat CSR byte streams (gfx950).
//
// Replaces the per-read arithmetic of the reference: GC() at lib/fast5_reader.py:58 (six str.count passes over the
// read), the read length / mean_qscore attributes read at :55-56 (here recomputed from the FASTQ record itself, the
// ONT definition -10 log10(mean 10^(-q/10))), plus the N count and integer Phred sum of SURVEY.md 8a-11.
//
// Mapping.  The two byte streams (bases, quals) are cut into tiles of 2^tile_shift bytes (<= 64 KiB); one wavefront owns one tile at a time and
// sweeps it in 1 KiB spans: lane l loads the 16 bytes at span + 16*l of each stream (global_load_dwordx4, 1 KiB
// contiguous per wave instruction, always 16-byte aligned whatever the read boundaries are).  Read boundaries are
// wave-uniform scalars walked along the offsets array.  A span that lies inside one read takes the fast path (SWAR byte
// classification, v_sad_u8 byte sums, a 256-entry fixed-point LUT in LDS for 10^(-q/10)).  A span that contains read
// boundaries evaluates, per boundary, one prefix-masked pass F(x) = sums over the first x bytes of the lane's vector;
// segment sums are differences of prefixes (all quantities are integers, so this is exact).  Ultra-long reads are
// thereby split over as many wavefronts as they have tiles; each wave writes a head/tail partial per tile and
// finalize_reads merges them in tile order.  Everything is integer, so results do not depend on the decomposition.
#include <stdlib.h>

#include "ppqc_internal.hpp"

namespace ppqc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Acc {
    uint32_t gcs, n, qb;
    uint64_t err;
};

// G/C/S (either case) and N (either case) byte counts of a 4-byte word, exact for arbitrary bytes.
// lo drops the case bit and the high bit, so every byte is <= 0x5F; (lo ^ K) + 0x7F has its byte's high bit clear
// iff the byte equals K (no inter-byte carries since lo ^ K <= 0x7F).  Bytes with the high bit set never match (h).
__device__ __forceinline__ void bases_word(uint32_t v, uint32_t& gcs, uint32_t& n) {
    const uint32_t h = v & 0x80808080u;
    const uint32_t lo = v & 0x5F5F5F5Fu;
    const uint32_t tGC = ((lo & 0x5B5B5B5Bu) ^ 0x43434343u) + 0x7F7F7F7Fu;   // C = 0x43, G = 0x47 differ in bit 2
    const uint32_t tS = (lo ^ 0x53535353u) + 0x7F7F7F7Fu;
    const uint32_t tN = (lo ^ 0x4E4E4E4Eu) + 0x7F7F7F7Fu;
    gcs += __popc(~((tGC & tS) | h) & 0x80808080u);
    n += __popc(~(tN | h) & 0x80808080u);
}

__device__ __forceinline__ void quals_word(uint32_t w, const uint64_t* lut, uint32_t& qb, uint64_t& err) {
    qb = __builtin_amdgcn_sad_u8(w, 0u, qb);
    err += lut[w & 0xffu];
    err += lut[(w >> 8) & 0xffu];
    err += lut[(w >> 16) & 0xffu];
    err += lut[w >> 24];
}

__device__ __forceinline__ Acc vec_sums(const uint4& vb, const uint4& vq, const uint64_t* lut) {
    Acc a{0, 0, 0, 0};
    bases_word(vb.x, a.gcs, a.n);
    bases_word(vb.y, a.gcs, a.n);
    bases_word(vb.z, a.gcs, a.n);
    bases_word(vb.w, a.gcs, a.n);
    uint64_t e0 = 0, e1 = 0;
    quals_word(vq.x, lut, a.qb, e0);
    quals_word(vq.y, lut, a.qb, e1);
    quals_word(vq.z, lut, a.qb, e0);
    quals_word(vq.w, lut, a.qb, e1);
    a.err = e0 + e1;
    return a;
}

// Line mode (FASTQ text, bases == quals == text): a span inside one line needs only that line's classifier.
__device__ __forceinline__ Acc vec_sums_bases(const uint4& vb) {
    Acc a{0, 0, 0, 0};
    bases_word(vb.x, a.gcs, a.n);
    bases_word(vb.y, a.gcs, a.n);
    bases_word(vb.z, a.gcs, a.n);
    bases_word(vb.w, a.gcs, a.n);
    return a;
}
__device__ __forceinline__ Acc vec_sums_quals(const uint4& vq, const uint64_t* lut) {
    Acc a{0, 0, 0, 0};
    uint64_t e0 = 0, e1 = 0;
    quals_word(vq.x, lut, a.qb, e0);
    quals_word(vq.y, lut, a.qb, e1);
    quals_word(vq.z, lut, a.qb, e0);
    quals_word(vq.w, lut, a.qb, e1);
    a.err = e0 + e1;
    return a;
}

// Sums over the first x (0..16) bytes of the lane's vectors.
__device__ __noinline__ Acc vec_prefix(uint4 vb, uint4 vq, int x, const uint64_t* lut) {
    Acc a{0, 0, 0, 0};
    const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w};
    const uint32_t wq[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int k = x - 4 * w;
        k = k < 0 ? 0 : (k > 4 ? 4 : k);
        const uint32_t mask = (uint32_t)((1ull << (8 * k)) - 1ull);
        bases_word(wb[w] & mask, a.gcs, a.n);
        const uint32_t q = wq[w] & mask;
        a.qb = __builtin_amdgcn_sad_u8(q, 0u, a.qb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t e = lut[(q >> (8 * j)) & 0xffu];
            a.err += (j < k) ? e : 0ull;
        }
    }
    return a;
}

__device__ __forceinline__ void acc_add(Acc& a, const Acc& b) {
    a.gcs += b.gcs; a.n += b.n; a.qb += b.qb; a.err += b.err;
}
__device__ __forceinline__ Acc acc_sub(const Acc& a, const Acc& b) {
    return Acc{a.gcs - b.gcs, a.n - b.n, a.qb - b.qb, a.err - b.err};
}

// Wave-uniform walk state over the reads of the batch.
struct Walk {
    uint64_t r;    // current read
    uint64_t rs;   // its first byte
    uint64_t re;   // one past its last byte
    bool done;     // no read left
};

__device__ __forceinline__ void walk_advance(Walk& w, const uint64_t* __restrict__ off, uint64_t R) {
    w.rs = w.re;
    uint64_t r = w.r + 1;
    while (r < R) {
        const uint64_t e = uniform_u64(off[r + 1]);
        if (e != w.rs) { w.re = e; break; }   // skip empty reads
        ++r;
    }
    w.r = r;
    w.done = r >= R;
}

// Largest r in [0, R) with off[r] <= pos (pos < off[R]); 64-ary search, one probe per lane per round.
__device__ __forceinline__ uint64_t find_read(const uint64_t* __restrict__ off, uint64_t R, uint64_t pos, int lane) {
    uint64_t lo = 0, hi = R;
    while (hi - lo > 1) {
        const uint64_t step = (hi - lo + 63) >> 6;
        const uint64_t idx = lo + (uint64_t)lane * step;
        const bool ok = idx < hi && off[idx] <= pos;
        const int c = __popcll(__ballot(ok));   // >= 1: lane 0 probes lo
        lo = uniform_u64(lo + (uint64_t)(c - 1) * step);
        const uint64_t nh = lo + step;
        hi = nh < hi ? nh : hi;
    }
    return lo;
}

__device__ __forceinline__ void flush(Acc& acc, int lane, uint64_t* cnt, uint64_t* hi, uint64_t* lo) {
    const uint64_t pk = wave_sum_u64(pack_cnt(acc.gcs, acc.n, acc.qb));
    const uint64_t h = wave_sum_u64(acc.err >> 32);
    const uint64_t l = wave_sum_u64(acc.err & 0xffffffffull);
    if (lane == 0) { *cnt = pk; *hi = h; *lo = l; }
    acc = Acc{0, 0, 0, 0};
}

// Per-tile constants shared by the span routine.
struct TileCtx {
    uint64_t t, tile_start, read_base, R;
    const uint64_t* __restrict__ off;
    ReadArrays ra;
    Partial* __restrict__ head;
    const uint64_t* lut;
    int lane;
    uint32_t lane_off;
};

__device__ __forceinline__ void flush_read(Acc& acc, const Walk& w, const TileCtx& c) {
    // a read that ends here is complete if it also started in this tile, else this is the head partial of the tile
    const bool whole = w.rs >= c.tile_start;
    Partial* p = &c.head[c.t];
    const uint64_t ri = c.read_base + w.r;
    flush(acc, c.lane, whole ? &c.ra.rec_cnt[ri] : &p->cnt, whole ? &c.ra.rec_hi[ri] : &p->hi,
          whole ? &c.ra.rec_lo[ri] : &p->lo);
}

// One 1 KiB span (both streams) of the tile: fast path if it lies inside the current read, else one prefix-masked
// pass per read boundary inside it.
template <bool LINE>
__device__ __forceinline__ void process_span(const uint4& vb, const uint4& vq, uint32_t sp, Walk& w, Acc& acc,
                                             const TileCtx& c) {
    const uint32_t span_rel = sp * kSpan;
    const uint64_t span_end = c.tile_start + span_rel + kSpan;   // end of the FULL span
    Acc tot;
    if (LINE && w.re >= span_end) {
        // segments are text lines, line number mod 4 = header / sequence / '+' / quality (wave-uniform): only the
        // sums fastq_finalize reads are kept exact; header and '+' lines cost no arithmetic at all
        const uint32_t ty = (uint32_t)w.r & 3u;
        if (ty == 1u) tot = vec_sums_bases(vb);
        else if (ty == 3u) tot = vec_sums_quals(vq, c.lut);
        else tot = Acc{0, 0, 0, 0};
    } else {
        tot = vec_sums(vb, vq, c.lut);
    }
    if (w.re >= span_end) {
        acc_add(acc, tot);
        if (w.re == span_end) {
            flush_read(acc, w, c);
            walk_advance(w, c.off, c.R);
        }
    } else {
        // one or more reads end strictly inside this span (or the data ends inside it)
        Acc prev{0, 0, 0, 0};
        const int vrel = (int)(span_rel + c.lane_off);
        do {
            int x = (int)(uint32_t)(w.re - c.tile_start) - vrel;
            x = x < 0 ? 0 : (x > kVec ? kVec : x);
            const Acc f = vec_prefix(vb, vq, x, c.lut);
            acc_add(acc, acc_sub(f, prev));
            prev = f;
            flush_read(acc, w, c);
            walk_advance(w, c.off, c.R);
        } while (!w.done && w.re < span_end);
        if (!w.done) {
            acc = acc_sub(tot, prev);   // the read that starts in this span owns the rest of it
            if (w.re == span_end) {
                flush_read(acc, w, c);   // starts and ends in this tile
                walk_advance(w, c.off, c.R);
            }
        }
    }
}

template <int NB, bool NT, bool LINE>
__device__ __forceinline__ void load_batch(uint4 (&vb)[NB], uint4 (&vq)[NB], const uint8_t* __restrict__ pb,
                                           const uint8_t* __restrict__ pq, uint32_t s0, uint32_t lane_off,
                                           uint32_t last_vec) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        uint32_t o = (s0 + i) * kSpan + lane_off;
        o = o < last_vec ? o : last_vec;   // clamp instead of predicating: out-of-range lanes are masked by the walk
        if (LINE) {   // one stream: the text is both
            const u32x4 tb = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pb + o));
            vb[i] = make_uint4(tb[0], tb[1], tb[2], tb[3]);
        } else if (NT) {   // read-once streams: nontemporal loads (measured +16 % on MI355X)
            const u32x4 tb = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pb + o));
            const u32x4 tq = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pq + o));
            vb[i] = make_uint4(tb[0], tb[1], tb[2], tb[3]);
            vq[i] = make_uint4(tq[0], tq[1], tq[2], tq[3]);
        } else {
            vb[i] = *reinterpret_cast<const uint4*>(pb + o);
            vq[i] = *reinterpret_cast<const uint4*>(pq + o);
        }
    }
}

template <int NB, bool LINE>
__device__ __forceinline__ void process_batch(const uint4 (&vb)[NB], const uint4 (&vq)[NB], uint32_t s0,
                                              uint32_t n_spans, Walk& w, Acc& acc, const TileCtx& c) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (s0 + i >= n_spans || w.done) break;
        process_span<LINE>(vb[i], LINE ? vb[i] : vq[i], s0 + i, w, acc, c);
    }
}

// NB spans per batch; NT nontemporal loads; PIPE: two register sets, the next batch is in flight while one is processed.
template <int NB, bool NT, bool PIPE, bool LINE>
__device__ __forceinline__ void tile_body(const BatchView& b, const uint64_t* __restrict__ lut_g, const ReadArrays& ra,
                                          uint64_t read_base, Partial* __restrict__ head, Partial* __restrict__ tail,
                                          uint64_t n_tiles) {
    __shared__ uint64_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    const uint64_t kTile = 1ull << b.tile_shift;
    TileCtx c;
    c.read_base = read_base; c.R = b.n_reads; c.off = b.offsets; c.ra = ra; c.head = head; c.lut = lut;
    c.lane = lane; c.lane_off = (uint32_t)lane * kVec;

    for (uint64_t t = (uint64_t)blockIdx.x * 4 + wib; t < n_tiles; t += n_waves) {
        const uint64_t tile_start = t << b.tile_shift;
        const uint64_t remain = b.n_bytes - tile_start;
        const uint32_t tile_valid = remain < kTile ? (uint32_t)remain : (uint32_t)kTile;   // > 0
        const uint32_t n_spans = (tile_valid + kSpan - 1) / kSpan;
        const uint32_t last_vec = (tile_valid - 1) & ~(uint32_t)(kVec - 1);   // last in-bounds 16-byte granule
        const uint8_t* __restrict__ pb = b.bases + tile_start;
        const uint8_t* __restrict__ pq = b.quals + tile_start;
        c.t = t; c.tile_start = tile_start;

        uint4 vb[NB], vq[NB];
        load_batch<NB, NT, LINE>(vb, vq, pb, pq, 0, c.lane_off, last_vec);   // in flight during the read search

        Walk w;
        w.r = find_read(c.off, c.R, tile_start, lane);
        w.rs = uniform_u64(c.off[w.r]);
        w.re = uniform_u64(c.off[w.r + 1]);
        w.done = false;
        Acc acc{0, 0, 0, 0};

        if (PIPE) {
            uint4 ub[NB], uq[NB];
            for (uint32_t s0 = 0; s0 < n_spans && !w.done; s0 += 2 * NB) {
                if (s0 + NB < n_spans) load_batch<NB, NT, LINE>(ub, uq, pb, pq, s0 + NB, c.lane_off, last_vec);
                process_batch<NB, LINE>(vb, vq, s0, n_spans, w, acc, c);
                if (s0 + 2 * NB < n_spans) load_batch<NB, NT, LINE>(vb, vq, pb, pq, s0 + 2 * NB, c.lane_off, last_vec);
                if (s0 + NB < n_spans) process_batch<NB, LINE>(ub, uq, s0 + NB, n_spans, w, acc, c);
            }
        } else {
            for (uint32_t s0 = 0; s0 < n_spans && !w.done; s0 += NB) {
                if (s0) load_batch<NB, NT, LINE>(vb, vq, pb, pq, s0, c.lane_off, last_vec);
                process_batch<NB, LINE>(vb, vq, s0, n_spans, w, acc, c);
            }
        }
        // a read that continues into the next tile leaves a partial: tail if it started here, else head
        if (!w.done && w.rs < tile_start + kTile && w.re > tile_start + kTile) {
            Partial* p = w.rs >= tile_start ? &tail[t] : &head[t];
            flush(acc, lane, &p->cnt, &p->hi, &p->lo);
        }
    }
}

template <int NB, bool NT, bool PIPE>
__global__ __launch_bounds__(256) void tile_kernel(BatchView b, const uint64_t* __restrict__ lut_g, ReadArrays ra,
                                                   uint64_t read_base, Partial* __restrict__ head,
                                                   Partial* __restrict__ tail, uint64_t n_tiles) {
    tile_body<NB, NT, PIPE, false>(b, lut_g, ra, read_base, head, tail, n_tiles);
}

// Line mode (FASTQ text): one stream.  8 spans per batch at 5 waves per SIMD (95 VGPRs); 16 spans spill (measured
// 1.0 TB/s of text against 1.27 TB/s for the whole tokenizer path).
template <int NB>
__global__ __launch_bounds__(256)
void tile_kernel_line(BatchView b, const uint64_t* __restrict__ lut_g, ReadArrays ra, uint64_t read_base,
                      Partial* __restrict__ head, Partial* __restrict__ tail, uint64_t n_tiles) {
    tile_body<NB, true, false, true>(b, lut_g, ra, read_base, head, tail, n_tiles);
}

// One thread per read: merge the partials of reads that span tiles, define empty reads, then decode the raw sums.
__global__ __launch_bounds__(256) void finalize_reads_kernel(BatchView b, ReadArrays ra, uint64_t read_base,
                                                             const Partial* __restrict__ head,
                                                             const Partial* __restrict__ tail, uint64_t thr_fail,
                                                             double fail_q, uint64_t* __restrict__ csr_errors) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_reads) return;
    uint64_t a = b.offsets[i], e = b.offsets[i + 1];
    // A device batch cannot be validated on the host: a malformed CSR must neither fault (the partial arrays are
    // indexed by tile) nor pass silently -- the read is zeroed and counted; ppqc_fetch_tables / ppqc_read_stats fail.
    if (e < a || e > b.n_bytes || e - a >= 0xFFFFFFFFull || (i == 0 && a != 0) || (i + 1 == b.n_reads && e != b.n_bytes)) {
        atomicAdd((unsigned long long*)csr_errors, 1ull);
        a = e = 0;
    }
    const uint64_t L = e - a, g = read_base + i;
    uint64_t gcs = 0, nn = 0, qb = 0, hi = 0, lo = 0;
    if (L != 0) {
        const SegSums m = merged_sums(a, e, g, ra, head, tail, b.tile_shift);
        gcs = m.gcs; nn = m.nn; qb = m.qb; hi = m.hi; lo = m.lo;
    }
    // canonical split S = H * 2^32 + l32, then one correctly rounded conversion to fp64
    const uint64_t H = hi + (lo >> 32), l32 = lo & 0xffffffffull;
    const bool flagged = b.fail_flag != nullptr && b.fail_flag[i] != 0;
    uint8_t flags = 0;
    double mq = 0.0;
    if (flagged) {
        // lib/fast5_reader.py:37-40: a 'fail' path keeps ReadLength = 0, QScore = 6.99, GCcont = 0
        flags = 2 | (6.99 < fail_q ? 1 : 0);
        ra.len[g] = 0; ra.gc[g] = 0; ra.nn[g] = 0; ra.qsum[g] = 0; ra.err_hi[g] = 0; ra.err_lo[g] = 0;
        mq = 6.99;
    } else {
        if (L != 0) {
            const double S = (double)H * 4294967296.0 + (double)l32;
            mq = -10.0 * log10(S / ((double)L * 4503599627370496.0));
            const unsigned __int128 S128 = ((unsigned __int128)H << 32) + l32;
            flags = S128 > (unsigned __int128)L * thr_fail ? 1 : 0;
        } else {
            flags = 0.0 < fail_q ? 1 : 0;
        }
        ra.len[g] = (uint32_t)L; ra.gc[g] = (uint32_t)gcs; ra.nn[g] = (uint32_t)nn;
        ra.qsum[g] = (int64_t)qb - 33 * (int64_t)L;
        ra.err_hi[g] = H; ra.err_lo[g] = l32;
    }
    ra.mean_q[g] = mq;
    ra.flags[g] = flags;
    ra.channel[g] = b.channel[i];
    ra.mux[g] = b.mux[i];
    ra.time[g] = b.time[i];
}

template <int NB, bool NT, bool PIPE, bool LINE = false>
static hipError_t launch_tile_variant(const BatchView& b, const uint64_t* lut_dev, ReadArrays ra, uint64_t read_base,
                                      Partial* head, Partial* tail, int n_cu, int blocks_per_cu, hipStream_t s) {
    const uint64_t n_tiles = (b.n_bytes + (1ull << b.tile_shift) - 1) >> b.tile_shift;
    // Default: one tile per wavefront, 4 waves per block.  The hardware dispatcher then acts as the dynamic tile
    // scheduler; measured on MI355X (config 2) this beats every persistent grid: 3.07 ms vs 3.13 (64 blocks/CU),
    // 3.37 (8/CU), 3.6 (resident blocks only).
    uint64_t blocks = (n_tiles + 3) / 4;
    if (blocks_per_cu > 0) {   // tuning knob: persistent grid of blocks_per_cu blocks per CU, grid-stride over tiles
        const uint64_t cap = (uint64_t)n_cu * blocks_per_cu;
        if (blocks > cap) blocks = cap;
    }
    if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
    if (LINE)
        hipLaunchKernelGGL((tile_kernel_line<NB>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra, read_base,
                           head, tail, n_tiles);
    else
        hipLaunchKernelGGL((tile_kernel<NB, NT, PIPE>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra,
                           read_base, head, tail, n_tiles);
    return hipGetLastError();
}

// PPQC_TILE_VARIANT / PPQC_TILE_BLOCKS_PER_CU are tuning knobs for A/B measurements; the default is the shipped one.
hipError_t launch_tile_kernel(const BatchView& b, const uint64_t* lut_dev, ReadArrays ra, uint64_t read_base,
                              Partial* head, Partial* tail, int n_cu, hipStream_t s, bool line_mode) {
    if (b.n_bytes == 0 || b.n_reads == 0) return hipSuccess;
    if (line_mode) return launch_tile_variant<8, true, false, true>(b, lut_dev, ra, read_base, head, tail, n_cu, 0, s);
    const char* v = getenv("PPQC_TILE_VARIANT");
    const char* bp = getenv("PPQC_TILE_BLOCKS_PER_CU");
    const int bpc = bp ? atoi(bp) : 0;
    const int variant = v ? atoi(v) : 0;
    switch (variant) {
        case 1: return launch_tile_variant<4, true, false>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        case 2: return launch_tile_variant<8, false, false>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        case 3: return launch_tile_variant<8, true, false>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        case 4: return launch_tile_variant<4, true, true>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        case 5: return launch_tile_variant<2, true, true>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        case 6: return launch_tile_variant<16, true, false>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
        default: return launch_tile_variant<8, true, false>(b, lut_dev, ra, read_base, head, tail, n_cu, bpc, s);
    }
}

hipError_t launch_finalize_reads(const BatchView& b, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, uint64_t thr_fail, double fail_q, uint64_t* csr_errors,
                                 hipStream_t s) {
    if (b.n_reads == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((b.n_reads + 255) / 256);
    hipLaunchKernelGGL(finalize_reads_kernel, dim3(blocks), dim3(256), 0, s, b, ra, read_base, head, tail, thr_fail,
                       fail_q, csr_errors);
    return hipGetLastError();
}

}  // namespace ppqc
