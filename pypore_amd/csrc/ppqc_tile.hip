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
// classification, v_sad_u8 byte sums, a 256-entry fixed-point LUT in LDS for 10^(-q/10)).  A read boundary cuts exactly
// one lane's vectors: the other lanes add their fast-path sums under a lane mask and the cut lane's 32 bytes are
// classified one byte per lane (boundaries_bytepar); tiles in which many reads end are left to dense_kernel, which
// works in entry space (below).  All quantities are integers, so every
// decomposition gives the same sums.  Ultra-long reads are
// thereby split over as many wavefronts as they have tiles; each wave writes a head/tail partial per tile and
// finalize_reads merges them in tile order.  Everything is integer, so results do not depend on the decomposition.
#include "ppqc_internal.hpp"

namespace ppqc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Sums over the first x (0..16) bytes of the lane's vectors (dense_kernel): the bytes from x on are zeroed (a zero byte
// matches no base class and adds nothing to the byte sum) and their LUT entries, lut[0] = 2^52 each, are taken off again.
__device__ __noinline__ Acc vec_prefix(uint4 vb, uint4 vq, int x, const uint64_t* lut) {
    uint32_t m[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int k = x - 4 * w;
        k = k < 0 ? 0 : (k > 4 ? 4 : k);
        m[w] = (uint32_t)((1ull << (8 * k)) - 1ull);
    }
    const uint4 mb = make_uint4(vb.x & m[0], vb.y & m[1], vb.z & m[2], vb.w & m[3]);
    const uint4 mq = make_uint4(vq.x & m[0], vq.y & m[1], vq.z & m[2], vq.w & m[3]);
    Acc a = vec_sums(mb, mq, lut);
    a.err -= (uint64_t)(kVec - x) << PPQC_ERR_SHIFT;
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
    uint32_t re_rel;   // min(re - tile_start, 2^31 - 1): what the per-span tests compare (32-bit scalar compares)
    uint64_t re2;  // SOFF: offsets[min(r + 2, R)], requested one read ahead through the scalar cache
    bool done;     // no read left
    // window of upcoming read ends, one per lane: lane i holds offsets[win_r0 + 1 + i] (the end of read win_r0 + i)
    uint64_t win, win_r0;
    bool win_ok;
};

// Wave-uniform read of offsets[idx] through the scalar data cache (s_load_dwordx2: its own counter and path, so a read
// boundary neither queues behind the wave's streaming loads nor waits for its record stores).  Read-only: the
// offsets are never written by this kernel.
typedef const __attribute__((address_space(4))) uint64_t* const_u64_ptr;
__device__ __forceinline__ uint64_t sload_u64(const uint64_t* p, uint64_t idx) {
    return ((const_u64_ptr)(uintptr_t)p)[idx];
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

// Wave-reduce the per-lane sums of a finished read (or tile partial) and store them.  Everything is reduced as u32:
// per lane and tile gcs, n <= 2^10 and qb <= 2^18, and the error sum (< 2^62 per lane) goes in three 21-bit pieces
// whose wave sums stay below 2^27.  S = E0 + E1 * 2^21 + E2 * 2^42 is stored as lo = E0 + (E1 << 21), hi = E2 << 10
// (S = hi * 2^32 + lo; finalize_reads normalises the split).
// kind (wave-uniform; line mode): 1 = a sequence line, only G and N are ever read; 2 = a quality line, only Q and the
// error sum are -- the other reductions are left out and their fields stored as zero.
__device__ __forceinline__ void flush(Acc& acc, int lane, uint64_t* cnt, uint64_t* hi, uint64_t* lo, int kind = 0) {
    uint32_t G = 0, N = 0, Q = 0, E0 = 0, E1 = 0, E2 = 0;
    if (kind != 2) { G = wave_sum_u32(acc.gcs); N = wave_sum_u32(acc.n); }
    if (kind != 1) {
        Q = wave_sum_u32(acc.qb);
        E0 = wave_sum_u32((uint32_t)acc.err & 0x1FFFFFu);
        E1 = wave_sum_u32((uint32_t)(acc.err >> 21) & 0x1FFFFFu);
        E2 = wave_sum_u32((uint32_t)(acc.err >> 42));
    }
    if (lane == 0) {
        *cnt = pack_cnt(G, N, Q);
        *hi = (uint64_t)E2 << 10;
        *lo = (uint64_t)E0 + ((uint64_t)E1 << 21);
    }
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
    uint32_t* xs;   // 64 words of LDS per wave (cut_lane_bytes, span_scan)
    bool line;      // line mode (FASTQ text): the sums of header and '+' lines (even line numbers) are never read
};

// offsets[r + 1] (the end of read r < R) out of the register window; one coalesced load refills it every 64 reads, so
// a boundary does not wait for a dependent global load
__device__ __forceinline__ uint64_t read_end(Walk& w, const TileCtx& c, uint64_t r) {
    if (!w.win_ok || r < w.win_r0 || r - w.win_r0 >= 64) {
        uint64_t idx = r + 1 + (uint64_t)c.lane;
        idx = idx < c.R ? idx : c.R;
        w.win = c.off[idx];
        w.win_r0 = r;
        w.win_ok = true;
    }
    return readlane_u64(w.win, (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r - w.win_r0)));
}

__device__ __forceinline__ uint32_t rel_end(uint64_t e, uint64_t tile_start) {
    const uint64_t d = e - tile_start;   // e > tile_start for the read the walk is in
    return d > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)d;
}

template <bool WIN, bool SOFF = false>
__device__ __forceinline__ void walk_advance(Walk& w, const TileCtx& c) {
    w.rs = w.re;
    uint64_t r = w.r + 1;
    if (SOFF) {
        uint64_t e = w.re2;   // offsets[r + 1] of the new r, requested when the previous read began
        while (r < c.R && e == w.rs) {   // skip empty reads
            ++r;
            if (r < c.R) e = sload_u64(c.off, r + 1);
        }
        w.r = r;
        w.done = r >= c.R;
        if (!w.done) {
            w.re = e;
            w.re_rel = rel_end(e, c.tile_start);
            w.re2 = sload_u64(c.off, r + 2 < c.R ? r + 2 : c.R);
        }
        return;
    }
    while (r < c.R) {
        const uint64_t e = WIN ? read_end(w, c, r) : uniform_u64(c.off[r + 1]);
        if (e != w.rs) { w.re = e; w.re_rel = rel_end(e, c.tile_start); break; }   // skip empty reads
        ++r;
    }
    w.r = r;
    w.done = r >= c.R;
}

__device__ __forceinline__ void flush_read(Acc& acc, const Walk& w, const TileCtx& c) {
    if (c.line && !(w.r & 1)) {   // a header or '+' line: nothing of it is wanted, so no reduction and no store
        acc = Acc{0, 0, 0, 0};
        return;
    }
    // a read that ends here is complete if it also started in this tile, else this is the head partial of the tile
    const bool whole = w.rs >= c.tile_start;
    Partial* p = &c.head[c.t];
    const uint64_t ri = c.read_base + w.r;
    flush(acc, c.lane, whole ? &c.ra.rec_cnt[ri] : &p->cnt, whole ? &c.ra.rec_hi[ri] : &p->hi,
          whole ? &c.ra.rec_lo[ri] : &p->lo, c.line ? ((w.r & 3) == 1 ? 1 : 2) : 0);
}

// The boundaries of a span one after the other, with the byte-masked part done byte-parallel (the default kernel).
// A read boundary cuts exactly ONE lane's 16-byte vectors (lane lb, at byte o); every other lane of the span belongs
// to one read entirely and contributes its fast-path sums `tot` under a lane mask.  Lane lb's 32 bytes go through LDS
// to lanes 0..31, one byte each (0..15: bases, 16..31: quality bytes); each of them classifies its byte with the same
// formulas as the fast path (so the results are identical by construction) and adds it to its own running sums when
// the byte lies on the ending read's side of the cut -- the wave reduction that a finished read needs anyway then
// brings everything together.  The bytes behind the cut start the next read's sums the same way.  It replaced a
// prefix-masked pass over all 64 lanes per boundary (375 vector instructions per boundary by SQ_INSTS_VALU, 22 % of
// config 2's vector instructions); together with the scalar look-ahead of the offsets (walk_advance) the tile kernel
// went from 5.6 to 6.3 TB/s at a mean read length of 5 kb and from 4.4 to 5.4 TB/s at 2.5 kb (tools/tile_ab.py).
struct CutBytes { uint32_t gcs, n, qb; uint64_t err; };

__device__ __noinline__ CutBytes cut_lane_bytes(uint4 vb, uint4 vq, int lb, int lane, uint32_t* xs, const uint64_t* lut) {
    __builtin_amdgcn_wave_barrier();
    if (lane == lb) {
        *reinterpret_cast<uint4*>(xs) = vb;
        *reinterpret_cast<uint4*>(xs + 4) = vq;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t bv = reinterpret_cast<const uint8_t*>(xs)[lane & 31];
    __builtin_amdgcn_wave_barrier();
    CutBytes r{0, 0, 0, 0};
    bases_word(lane < 16 ? bv : 0u, r.gcs, r.n);   // a zero byte matches no class
    const bool isq = lane >= 16 && lane < 32;
    const uint64_t e = lut[bv];
    r.qb = isq ? bv : 0u;
    r.err = isq ? e : 0ull;
    return r;
}

__device__ __forceinline__ void boundaries_bytepar(const uint4& vb, const uint4& vq, const Acc& tot, uint32_t span_rel,
                                                   Walk& w, Acc& acc, const TileCtx& c) {
    const uint32_t span_end_rel = span_rel + kSpan;
    int prev_lb = -1, o_prev = 0;   // lane and byte offset of the last cut handled in this span
    const int jj = c.lane & 15;
    const bool low32 = c.lane < 32;
    CutBytes cb{0, 0, 0, 0};
    do {
        const int x = (int)(w.re_rel - span_rel);   // 0 .. kSpan-1, wave-uniform
        const int lb = x >> 4, o = x & 15;
        if (lb == prev_lb) {
            // the read began in this very vector: it is bytes [o_prev, o) of lane lb, nothing else (cb is still valid)
            const bool in = low32 && jj >= o_prev && jj < o;
            acc.gcs = in ? cb.gcs : 0u; acc.n = in ? cb.n : 0u; acc.qb = in ? cb.qb : 0u; acc.err = in ? cb.err : 0ull;
        } else {
            if (c.lane > prev_lb && c.lane < lb) acc_add(acc, tot);
            if (o) {
                cb = cut_lane_bytes(vb, vq, lb, c.lane, c.xs, c.lut);
                if (low32 && jj < o) { acc.gcs += cb.gcs; acc.n += cb.n; acc.qb += cb.qb; acc.err += cb.err; }
            }
        }
        flush_read(acc, w, c);   // acc = 0
        walk_advance<false, true>(w, c);
        if (o) {   // the next read owns the bytes of lane lb behind the cut
            if (low32 && jj >= o) { acc.gcs = cb.gcs; acc.n = cb.n; acc.qb = cb.qb; acc.err = cb.err; }
            prev_lb = lb; o_prev = o;
        } else {   // cut on a vector boundary: lane lb goes in whole with the lanes behind it
            prev_lb = lb - 1; o_prev = 0;
        }
    } while (!w.done && w.re_rel < span_end_rel);
    if (!w.done) {
        if (c.lane > prev_lb) acc_add(acc, tot);   // the rest of the span belongs to the open read
        if (w.re_rel == span_end_rel) {
            flush_read(acc, w, c);
            walk_advance<false, true>(w, c);
        }
    }
}

// One 1 KiB span (both streams) of the tile: fast path if it lies inside the current read, else the boundaries one
// after the other (boundaries_bytepar).
template <bool LINE>
__device__ __forceinline__ void process_span(const uint4& vb, const uint4& vq, uint32_t sp, Walk& w, Acc& acc,
                                             const TileCtx& c) {
    const uint32_t span_rel = sp * kSpan;
    const uint32_t span_end_rel = span_rel + kSpan;
    Acc tot;
    if (LINE && w.re_rel >= span_end_rel) {
        // segments are text lines, line number mod 4 = header / sequence / '+' / quality (wave-uniform): only the
        // sums fastq_finalize reads are kept exact; header and '+' lines cost no arithmetic at all
        const uint32_t ty = (uint32_t)w.r & 3u;
        if (ty == 1u) tot = vec_sums_bases(vb);
        else if (ty == 3u) tot = vec_sums_quals(vq, c.lut);
        else tot = Acc{0, 0, 0, 0};
    } else {
        tot = vec_sums(vb, vq, c.lut);
    }
    if (w.re_rel >= span_end_rel) {
        acc_add(acc, tot);
        if (w.re_rel == span_end_rel) {
            flush_read(acc, w, c);
            walk_advance<false, true>(w, c);
        }
    } else {
        boundaries_bytepar(vb, vq, tot, span_rel, w, acc, c);
    }
}

// The streams are read once: nontemporal loads (measured +16 % on MI355X against temporal ones).
template <int NB, bool LINE>
__device__ __forceinline__ void load_batch(uint4 (&vb)[NB], uint4 (&vq)[NB], const uint8_t* __restrict__ pb,
                                           const uint8_t* __restrict__ pq, uint32_t s0, uint32_t lane_off,
                                           uint32_t last_vec) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        uint32_t o = (s0 + i) * kSpan + lane_off;
        o = o < last_vec ? o : last_vec;   // clamp instead of predicating: out-of-range lanes are masked by the walk
        const u32x4 tb = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pb + o));
        vb[i] = make_uint4(tb[0], tb[1], tb[2], tb[3]);
        if (!LINE) {   // line mode has one stream: the text is both
            const u32x4 tq = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pq + o));
            vq[i] = make_uint4(tq[0], tq[1], tq[2], tq[3]);
        }
    }
}

constexpr int kLineGroup = 4;

template <int NB, bool LINE>
__device__ __forceinline__ void process_batch(const uint4 (&vb)[NB], const uint4 (&vq)[NB], uint32_t s0,
                                              uint32_t n_spans, Walk& w, Acc& acc, const TileCtx& c) {
    if (LINE && NB % kLineGroup == 0) {
        // Line mode: kLineGroup spans at once when the current line covers all of them -- straight-line code whose LUT
        // reads and sums the compiler interleaves across the spans, and one set of scalar tests instead of four
#pragma unroll
        for (int g = 0; g < NB; g += kLineGroup) {
            if (s0 + g >= n_spans || w.done) break;
            const uint32_t gend_rel = (s0 + g + kLineGroup) * kSpan;
            if (s0 + g + kLineGroup <= n_spans && w.re_rel >= gend_rel) {
                const uint32_t ty = (uint32_t)w.r & 3u;
                if (ty == 1u) {
#pragma unroll
                    for (int k = 0; k < kLineGroup; ++k) acc_add(acc, vec_sums_bases(vb[g + k]));
                } else if (ty == 3u) {
#pragma unroll
                    for (int k = 0; k < kLineGroup; ++k) acc_add(acc, vec_sums_quals(vb[g + k], c.lut));
                }
                if (w.re_rel == gend_rel) {
                    flush_read(acc, w, c);
                    walk_advance<false, true>(w, c);
                }
            } else {
#pragma unroll
                for (int k = 0; k < kLineGroup; ++k) {
                    if (s0 + g + k >= n_spans || w.done) break;
                    process_span<LINE>(vb[g + k], vb[g + k], s0 + g + k, w, acc, c);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (s0 + i >= n_spans || w.done) break;
        process_span<LINE>(vb[i], LINE ? vb[i] : vq[i], s0 + i, w, acc, c);
    }
}

// What tile_kernel needs to hand dense tiles over to dense_kernel (see "the boundary strategy is chosen per tile" below).
// The dense list.  64 sub-lists, each with a counter of its own (one hot counter took 380 us for 30 000 appends): a
// tile goes to sub-list (tile / 4) % 64, which holds at most dense_sub_cap(n_tiles) entries.  Layout in u64 words:
//   [0, 64)    counters of the launches with parity 0        [64, 128)  counters of the launches with parity 1
//   [128 + j * cap, ...)  entries of sub-list j: tile | first read << 32
// ONE 64-bit kernel argument, because the kernel has no scalar registers to spare (two more and it spills, and any
// scratch costs it a fifth of its speed):
//   bits [0,48)  DenseList::base
//   bit  48      the parity of this launch
//   bits [49,51) 0 = probe every tile, 1 = no tile is dense, 2 = every tile is
//   bits [51,64) T = tile_bytes / dense_len: a tile is dense when at least T reads end inside it
constexpr uint32_t kDenseSubs = 64, kDenseHeader = 2 * kDenseSubs;
__host__ __device__ inline uint64_t dense_sub_cap(uint64_t n_tiles) { return 4 * (((n_tiles + 3) / 4 + kDenseSubs - 1) / kDenseSubs); }
typedef uint64_t DenseOut;
__host__ __device__ inline DenseOut make_dense_out(const uint64_t* base, int parity, uint32_t mode, uint32_t T) {
    return ((uint64_t)(uintptr_t)base & 0xFFFFFFFFFFFFull) | ((uint64_t)(parity & 1) << 48) | ((uint64_t)(mode & 3u) << 49) |
           ((uint64_t)(T & 8191u) << 51);
}

// NB spans per batch (all of their loads in flight before the first wait), WPB waves per block.
template <int NB, bool LINE, int WPB = 4>
__device__ __forceinline__ void tile_body(const BatchView& b, const uint64_t* __restrict__ lut_g, const ReadArrays& ra,
                                          uint64_t read_base, Partial* __restrict__ head, Partial* __restrict__ tail,
                                          uint64_t n_tiles, const DenseOut dz = make_dense_out(nullptr, 0, 1, 64)) {
    __shared__ uint64_t lut[256];
    __shared__ uint32_t xs[WPB][64];
#pragma unroll
    for (int i = 0; i < 4 / WPB; ++i) lut[threadIdx.x + i * 64 * WPB] = lut_g[threadIdx.x + i * 64 * WPB];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * WPB;
    const uint64_t kTile = 1ull << b.tile_shift;
    TileCtx c;
    c.read_base = read_base; c.R = b.n_reads; c.off = b.offsets; c.ra = ra; c.head = head; c.lut = lut;
    c.lane = lane; c.lane_off = (uint32_t)lane * kVec; c.xs = &xs[wib][0]; c.line = LINE;

    for (uint64_t t = (uint64_t)blockIdx.x * WPB + wib; t < n_tiles; t += n_waves) {
        const uint64_t tile_start = t << b.tile_shift;
        const uint64_t remain = b.n_bytes - tile_start;
        const uint32_t tile_valid = remain < kTile ? (uint32_t)remain : (uint32_t)kTile;   // > 0
        const uint32_t n_spans = (tile_valid + kSpan - 1) / kSpan;
        const uint32_t last_vec = (tile_valid - 1) & ~(uint32_t)(kVec - 1);   // last in-bounds 16-byte granule
        const uint8_t* __restrict__ pb = b.bases + tile_start;
        const uint8_t* __restrict__ pq = b.quals + tile_start;
        c.t = t; c.tile_start = tile_start;

        uint4 vb[NB], vq[NB];
        load_batch<NB, LINE>(vb, vq, pb, pq, 0, c.lane_off, last_vec);   // in flight during the read search

        Walk w;
        w.r = find_read(c.off, c.R, tile_start, lane);
        // density probe: one scalar load that depends on the search like the loads of the read's start and end; it is
        // requested in front of them and looked at behind them, so it adds no latency of its own
        const uint32_t dmode = (uint32_t)(dz >> 49) & 3u;
        uint64_t probe = ~0ull;
        if (dmode == 0) {
            const uint64_t T = dz >> 51;   // >= 1
            // (fewer than T reads left: not dense -- one stray dense tile, the batch's last, would cost dense_kernel a
            // whole tile's latency, 40 us, behind this kernel)
            if (w.r + T <= c.R) probe = sload_u64(c.off, w.r + T);
        }
        w.rs = uniform_u64(c.off[w.r]);
        w.done = false;
        w.win = 0; w.win_r0 = 0; w.win_ok = false;
        w.re = uniform_u64(c.off[w.r + 1]);
        if (dmode != 1 && (dmode == 2 || probe <= tile_start + tile_valid)) {   // left to dense_kernel
            if (lane == 0) {
                uint64_t* base = (uint64_t*)(uintptr_t)(dz & 0xFFFFFFFFFFFFull);
                const uint64_t j = (t >> 2) & (kDenseSubs - 1), cap = dense_sub_cap(n_tiles);
                const unsigned long long k = atomicAdd((unsigned long long*)(base + ((dz >> 48) & 1ull) * kDenseSubs + j), 1ull);
                if (k < cap) base[kDenseHeader + j * cap + k] = t | (w.r << 32);   // a count that starts at zero stays
                                                                                    // below cap: the tiles of sub-list j
            }
            continue;
        }
        w.re_rel = rel_end(w.re, tile_start);
        w.re2 = sload_u64(c.off, w.r + 2 < c.R ? w.r + 2 : c.R);
        Acc acc{0, 0, 0, 0};

        for (uint32_t s0 = 0; s0 < n_spans && !w.done; s0 += NB) {
            if (s0) load_batch<NB, LINE>(vb, vq, pb, pq, s0, c.lane_off, last_vec);
            process_batch<NB, LINE>(vb, vq, s0, n_spans, w, acc, c);
        }
        // a read that continues into the next tile leaves a partial: tail if it started here, else head
        if (!w.done && w.rs < tile_start + kTile && w.re > tile_start + kTile && !(LINE && !(w.r & 1))) {
            Partial* p = w.rs >= tile_start ? &tail[t] : &head[t];
            flush(acc, lane, &p->cnt, &p->hi, &p->lo, LINE ? ((w.r & 3) == 1 ? 1 : 2) : 0);
        }
    }
}

// ---- the boundary strategy is chosen PER TILE -------------------------------------------------------------------------
// A batch's mean read length says little about a tile: the adversarial mix of BASELINE config 4 (99.95 % reads of 200 b
// + 100 reads of 2 Mb) has a mean of 1.2 kb, yet every tile is either one long read or wall-to-wall 200-byte reads.  So
// each wave of tile_kernel classifies its own tile -- one scalar probe of the offsets behind the read search: do at least
// tile_bytes / dense_len reads end inside this tile? -- and
//   sparse tiles: stay in tile_kernel (boundaries_bytepar, one boundary after the other, above);
//   dense tiles : are appended to a device list (tile, first read) and left to dense_kernel, launched behind it, whose
//                 span_scan handles every boundary of a span at once at a cost that does not depend on their number.
// Two kernels, not two branches of one: the fast path of tile_kernel sits at exactly 128 VGPRs with no scratch, and
// either an inlined dense branch (spills) or a called one (the callee's register saves) gives the kernel a private
// segment -- measured, that alone costs 22 % on config 2 with instruction-for-instruction the same hot loop.
//
// span_scan.  Lane v of the data owns 16 bytes of the span.  A read end inside the span cuts one lane's vector at byte x
// (x = 0: the cut is in front of the vector).  With tot_v the lane's full sums and pre_v the sums of its first x bytes
// (ONE masked pass for all cuts of the span; uncut lanes take x = 16, pre = tot), the lanes' contributions are
// A_v = pre_v to the read that ends at or behind v and B_v = tot_v - pre_v (cut lanes only) to the read that begins in v.
// P = inclusive wave scan of A (u32 pieces, six v_add_u32_dpp each: four row_shr steps, row_bcast:15, row_bcast:31).
// The read that ends in cut lane c, the previous cut being lane p, owns P_c - P_p + B_p = P_c - G_p with G = P - B
// (wrapping u32 arithmetic on packed pieces is exact because every true field value is in range); lane c fetches G_p
// with one ds_bpermute per piece and STORES ITS OWN RECORD at read index r + (number of cut lanes below c).  The first
// cut of the span has no p: its read also owns the wave-uniform carry S (what earlier spans left), which lane c adds.
// The bytes behind the last cut become the new carry, P_63 - G_last, read out with v_readlane.  No wave reduction per
// read, no loop over the reads: about 3.3 fast-path spans of vector work per span however many reads end in it.
// Reads shorter than a vector (two cuts in one lane, or empty reads inside the span) fall back to one masked pass per
// boundary for that span.
struct Carry {   // wave-uniform: sums of the open read that are not in the per-lane accumulators
    uint32_t gcs, n, qb;
    uint64_t hi, lo;   // S = hi * 2^32 + lo
};

__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) { return wave_scan_incl_u32(v); }   // ppqc_internal.hpp

// Span-level sums (one 16-byte vector per stream and lane, at most 64 lanes added up) in four u32 pieces:
//   c1 = gcs | n << 16          (each <= 1024)
//   c2 = qb | (err >> 52) << 18 (qb <= 261120 < 2^18; err <= 2^56 per lane, so the top piece is <= 16, <= 1024 summed)
//   e0, e1 = bits [0,26) and [26,52) of err (64 * (2^26 - 1) < 2^32)
struct Pieces { uint32_t c1, c2, e0, e1; };
__device__ __forceinline__ Pieces to_pieces(const Acc& a) {
    return Pieces{a.gcs | (a.n << 16), a.qb | ((uint32_t)(a.err >> 52) << 18), (uint32_t)a.err & 0x3FFFFFFu,
                  (uint32_t)(a.err >> 26) & 0x3FFFFFFu};
}
__device__ __forceinline__ uint64_t pieces_err(const Pieces& p) {
    return (uint64_t)p.e0 + ((uint64_t)p.e1 << 26) + ((uint64_t)(p.c2 >> 18) << 52);   // < 2^63 for one span
}

// Store the open read's sums (carry + per-lane accumulators if any were used since the last boundary) and clear both.
__device__ __forceinline__ void flush_c(Acc& acc, bool& dirty, Carry& S, int lane, uint64_t* cnt, uint64_t* hi,
                                        uint64_t* lo) {
    uint32_t G = S.gcs, N = S.n, Q = S.qb;
    uint64_t H = S.hi, L = S.lo;
    if (dirty) {
        G += wave_sum_u32(acc.gcs); N += wave_sum_u32(acc.n); Q += wave_sum_u32(acc.qb);
        const uint32_t E0 = wave_sum_u32((uint32_t)acc.err & 0x1FFFFFu);
        const uint32_t E1 = wave_sum_u32((uint32_t)(acc.err >> 21) & 0x1FFFFFu);
        const uint32_t E2 = wave_sum_u32((uint32_t)(acc.err >> 42));
        L += (uint64_t)E0 + ((uint64_t)E1 << 21);
        H += (uint64_t)E2 << 10;
        acc = Acc{0, 0, 0, 0};
        dirty = false;
    }
    if (lane == 0) {
        *cnt = pack_cnt(G, N, Q);
        *hi = H;
        *lo = L;
    }
    S = Carry{0, 0, 0, 0, 0};
}

__device__ __forceinline__ void flush_read_c(Acc& acc, bool& dirty, Carry& S, const Walk& w, const TileCtx& c) {
    const bool whole = w.rs >= c.tile_start;
    Partial* p = &c.head[c.t];
    const uint64_t ri = c.read_base + w.r;
    flush_c(acc, dirty, S, c.lane, whole ? &c.ra.rec_cnt[ri] : &p->cnt, whole ? &c.ra.rec_hi[ri] : &p->hi,
            whole ? &c.ra.rec_lo[ri] : &p->lo);
}

constexpr uint32_t kNoCut = 0xFFFFFFFFu;

__device__ __forceinline__ void span_scan(const uint4& vb, const uint4& vq, uint32_t span_rel, Walk& w, Acc& acc,
                                       bool& dirty, Carry& S, const TileCtx& c) {
    const int lane = c.lane;
    const uint64_t span_start = c.tile_start + span_rel, span_end = span_start + kSpan;
    const uint64_t r_save = w.r, rs_save = w.rs, re_save = w.re;
    // phase 1: tell every data lane where its vector is cut (kNoCut = not at all)
    c.xs[lane] = kNoCut;
    __builtin_amdgcn_wave_barrier();
    bool conflict = false;
    for (int round = 0;; ++round) {
        if (!w.win_ok || w.r < w.win_r0 || w.r - w.win_r0 >= 64) {
            uint64_t idx = w.r + 1 + (uint64_t)lane;
            idx = idx < c.R ? idx : c.R;
            w.win = c.off[idx];
            w.win_r0 = w.r;
            w.win_ok = true;
        }
        const int pos = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w.r - w.win_r0));
        // (a malformed device CSR may hold ends in front of the span: they are no cuts, the index stays in range, and
        // finalize_reads reports the batch)
        const bool isb = lane >= pos && w.win_r0 + (uint64_t)lane < c.R && w.win < span_end && w.win >= span_start;
        const uint64_t nbm = __ballot(isb);
        if (!nbm) break;
        const uint32_t rel = isb ? (uint32_t)(w.win - span_start) : 0u;
        const int vl = (int)(rel >> 4);
        const int vlp = __shfl_up(vl, 1);
        uint32_t prior = kNoCut;
        if (round && isb) prior = c.xs[vl];   // a cut left by the previous window of this span
        if (__ballot(isb && ((lane > pos && vlp == vl) || prior != kNoCut))) { conflict = true; break; }
        __builtin_amdgcn_wave_barrier();
        if (isb) c.xs[vl] = rel & 15u;
        __builtin_amdgcn_wave_barrier();
        const int hb = 63 - __clzll((long long)nbm);
        w.rs = readlane_u64(w.win, hb);
        w.r = w.win_r0 + (uint64_t)hb + 1;
        if (hb != 63 || w.r >= c.R) break;   // else the window is used up and more reads may end in this span
    }
    const Acc tot = vec_sums(vb, vq, c.lut);
    if (conflict) {
        // reads shorter than a vector: one masked pass per boundary, from the state the span began with
        w.r = r_save; w.rs = rs_save; w.re = re_save;
        const int vrel = (int)(span_rel + c.lane_off);
        Acc prev{0, 0, 0, 0};
        do {
            int x = (int)(uint32_t)(w.re - c.tile_start) - vrel;
            x = x < 0 ? 0 : (x > kVec ? kVec : x);
            const Acc f = vec_prefix(vb, vq, x, c.lut);
            acc_add(acc, acc_sub(f, prev));
            dirty = true;
            prev = f;
            flush_read_c(acc, dirty, S, w, c);
            walk_advance<true>(w, c);
        } while (!w.done && w.re < span_end);
        if (!w.done) {
            acc = acc_sub(tot, prev);
            dirty = true;
            if (w.re == span_end) {
                flush_read_c(acc, dirty, S, w, c);
                walk_advance<true>(w, c);
            }
        }
        return;
    }
    const uint32_t myx = c.xs[lane];
    __builtin_amdgcn_wave_barrier();
    const bool cut = myx != kNoCut;
    const uint64_t B = __ballot(cut);   // != 0: the span has a boundary
    if (dirty) {   // fast-path spans since the last boundary: fold the per-lane sums into the carry, once per read
        Carry Z = S;
        uint32_t G = wave_sum_u32(acc.gcs), N = wave_sum_u32(acc.n), Q = wave_sum_u32(acc.qb);
        const uint32_t E0 = wave_sum_u32((uint32_t)acc.err & 0x1FFFFFu);
        const uint32_t E1 = wave_sum_u32((uint32_t)(acc.err >> 21) & 0x1FFFFFu);
        const uint32_t E2 = wave_sum_u32((uint32_t)(acc.err >> 42));
        Z.gcs += G; Z.n += N; Z.qb += Q;
        Z.lo += (uint64_t)E0 + ((uint64_t)E1 << 21);
        Z.hi += (uint64_t)E2 << 10;
        S = Z;
        acc = Acc{0, 0, 0, 0};
        dirty = false;
    }
    const Acc pre = vec_prefix(vb, vq, cut ? (int)myx : kVec, c.lut);   // uncut lanes: the whole vector
    const Pieces a = to_pieces(pre);
    Pieces P{wave_scan_u32(a.c1), wave_scan_u32(a.c2), wave_scan_u32(a.e0), wave_scan_u32(a.e1)};
    // G = P - B, B = tot - pre on cut lanes.  B is split into pieces AFTER the 64-bit subtraction: the pieces of a
    // difference of error sums are not field by field non-negative, the pieces of a non-negative value are.
    const Pieces bp = to_pieces(cut ? acc_sub(tot, pre) : Acc{0, 0, 0, 0});
    Pieces Gq;
    Gq.c1 = P.c1 - bp.c1;
    Gq.c2 = P.c2 - bp.c2;
    Gq.e0 = P.e0 - bp.e0;
    Gq.e1 = P.e1 - bp.e1;
    // previous cut lane
    const uint64_t below = B & (((uint64_t)1 << lane) - 1ull);
    const bool has_prev = below != 0;
    const int pl = has_prev ? 63 - __clzll((long long)below) : 0;
    Pieces X;
    {
        const uint32_t g1 = (uint32_t)__shfl((int)Gq.c1, pl), g2 = (uint32_t)__shfl((int)Gq.c2, pl);
        const uint32_t g3 = (uint32_t)__shfl((int)Gq.e0, pl), g4 = (uint32_t)__shfl((int)Gq.e1, pl);
        X.c1 = P.c1 - (has_prev ? g1 : 0u);
        X.c2 = P.c2 - (has_prev ? g2 : 0u);
        X.e0 = P.e0 - (has_prev ? g3 : 0u);
        X.e1 = P.e1 - (has_prev ? g4 : 0u);
    }
    if (cut) {
        uint32_t gcs = X.c1 & 0xFFFFu, nn = X.c1 >> 16, qb = X.c2 & 0x3FFFFu;
        uint64_t lo = pieces_err(X), hi = 0;
        const bool first = !has_prev;
        if (first) { gcs += S.gcs; nn += S.n; qb += S.qb; lo += S.lo; hi = S.hi; }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(B >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)B, 0u));
        const uint64_t ri = c.read_base + r_save + rank;
        const bool to_head = first && rs_save < c.tile_start;   // the read began in an earlier tile
        Partial* p = &c.head[c.t];
        *(to_head ? &p->cnt : &c.ra.rec_cnt[ri]) = pack_cnt(gcs, nn, qb);
        *(to_head ? &p->hi : &c.ra.rec_hi[ri]) = hi;
        *(to_head ? &p->lo : &c.ra.rec_lo[ri]) = lo;
    }
    {   // what lies behind the last cut opens the next read
        const int cl = 63 - __clzll((long long)B);
        Pieces L;
        L.c1 = (uint32_t)__builtin_amdgcn_readlane((int)P.c1, 63) - (uint32_t)__builtin_amdgcn_readlane((int)Gq.c1, cl);
        L.c2 = (uint32_t)__builtin_amdgcn_readlane((int)P.c2, 63) - (uint32_t)__builtin_amdgcn_readlane((int)Gq.c2, cl);
        L.e0 = (uint32_t)__builtin_amdgcn_readlane((int)P.e0, 63) - (uint32_t)__builtin_amdgcn_readlane((int)Gq.e0, cl);
        L.e1 = (uint32_t)__builtin_amdgcn_readlane((int)P.e1, 63) - (uint32_t)__builtin_amdgcn_readlane((int)Gq.e1, cl);
        S = Carry{L.c1 & 0xFFFFu, L.c1 >> 16, L.c2 & 0x3FFFFu, 0, pieces_err(L)};
    }
    w.done = w.r >= c.R;
    if (!w.done) {
        w.re = read_end(w, c, w.r);   // >= span_end: every end in front of it has been consumed
        w.re_rel = rel_end(w.re, c.tile_start);
        if (w.re == span_end) {
            flush_read_c(acc, dirty, S, w, c);
            walk_advance<true>(w, c);
        }
    }
}

#ifndef PPQC_DENSE_SPANS
#define PPQC_DENSE_SPANS 4
#endif
#ifndef PPQC_DENSE_OCC
#define PPQC_DENSE_OCC 4
#endif
constexpr int kDenseSpans = PPQC_DENSE_SPANS;   // spans per batch of a dense tile (32 VGPRs of data; the boundary pass needs the rest)

// Two registers of upcoming read ends: w0 lane l = offsets[r0 + 1 + l], w1 lane l = offsets[r0 + 65 + l] (clamped to
// offsets[R]).  Any 64 consecutive reads from r on, r - r0 < 64, are then at hand without a load; when r passes r0 + 64
// the registers shift and the load of the new w1 has 64 reads' worth of work to arrive.
struct Win2 {
    uint64_t w0, w1, r0;
    bool ok;
};
__device__ __forceinline__ void win_ensure(Win2& W, const TileCtx& c, uint64_t r) {
    if (!W.ok || r < W.r0 || r - W.r0 >= 128) {
        uint64_t i0 = r + 1 + (uint64_t)c.lane, i1 = r + 65 + (uint64_t)c.lane;
        W.w0 = c.off[i0 < c.R ? i0 : c.R];
        W.w1 = c.off[i1 < c.R ? i1 : c.R];
        W.r0 = r;
        W.ok = true;
    } else if (r - W.r0 >= 64) {
        W.w0 = W.w1;
        W.r0 += 64;
        const uint64_t i1 = W.r0 + 65 + (uint64_t)c.lane;
        W.w1 = c.off[i1 < c.R ? i1 : c.R];
    }
}
// lane j: the end of read r + j (r - W.r0 < 64)
__device__ __forceinline__ uint64_t win_ends(const Win2& W, uint64_t r, int lane) {
    const int idx = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r - W.r0)) + lane;   // 0 .. 126
    const uint64_t a = __shfl((unsigned long long)W.w0, idx & 63), b = __shfl((unsigned long long)W.w1, idx & 63);
    return idx < 64 ? a : b;
}

// The dense tiles of a batch, from the list tile_kernel left: entry = tile | first read << 32.  One tile per wave, like
// tile_kernel (persistent grids lose to the hardware dispatcher here as there); the host does not know the count, but a
// dense tile holds at least T = tile_bytes / dense_len read ends and every read ends in one tile, so the list has at most
// n_reads / T + 1 entries and the grid is sized by that bound: waves beyond the count leave at once (config 2: 15 000
// waves, 4 us).  counters[parity] is this launch's count; the other counter, last used by the previous launch (complete:
// stream order) and next by the following one, is put back to zero here (64 counters per parity: see DenseOut).
// ALL: a batch that is dense on average is launched as dense_kernel<.., true> alone: one wave per tile of the batch, no
// list (tile_kernel's pass over it would do nothing but probe and append: 0.21 of 1.40 ms at a mean of 500 B), the first
// read by the same 64-ary search.
//
// ENTRY SPACE.  The read ends inside a span are taken in read order, one lane each: lane j holds the j-th end p_j of the
// span (from the window registers), the vector it cuts v_j = (p_j - span_start) / 16 and the offset x_j inside it, and
// fetches the 32 bytes of data lane v_j (eight ds_bpermute).  The span's data lanes scan their full sums (u32 pieces,
// six v_add_u32_dpp each); with E_j the sums in front of the entry's vector (the exclusive scan at lane v_j) and pre_j the
// sums of the first x_j bytes of that vector, the read that ends at entry j owns
//     own_j = B_j + pre_j - pre_(j-1),   B_j = E_j - E_(j-1)       (j - 1 in the same span)
//                                        B_j = E_j + S              (first entry of its span; S = all whole vectors
//                                                                    since the vector of entry j - 1, that vector included)
// for ANY two consecutive entries of the tile: everything that needs a masked pass over cut vectors is pre_j - pre_(j-1).
// So an entry is QUEUED: a span leaves (cut vector, x_j, B_j) in a ring in LDS and moves on; when 64 entries are waiting,
// ONE masked pass computes their pre at full width, a shift brings pre_(j-1), and 64 consecutive lanes store 64
// consecutive records.  (Round 2's first version finished the entries of four spans at a time: at a mean read length of
// 500 B that is eight entries, so the masked pass and the record stores ran at an eighth of the wave's width and were a
// third of the kernel's vector work.  Same box, that version -> this one, GB/s of the per-read kernels: mean 150 B
// 3024 -> 3570, 250 B 3180 -> 3765, 500 B 3276 -> 3974, 1 kb 3609 -> 4416, adversarial mix 3765 -> 4132.)
// The carry S is kept without the cut part of the newest entry's vector; it is put right (minus that entry's pre) when the
// tile ends or a span has to go the slow way.  Ends that share a vector, empty reads, ends on a span's edge are all just
// entries; only a span with 64 or more ends (reads of fewer than 16 bytes on average) goes through span_scan, from a
// settled state.  The error sums are 64-bit (128-bit where the carry's high part comes in); the counters wrap in u32
// and are exact because every true value is in range.
constexpr int kQN = 128;   // ring slots per wave: a span adds at most 63 entries to fewer than 64 waiting ones

struct PreSums { uint32_t gcs, n, qb; uint64_t err; };

// SLOW.  A span in which 64 or more reads end (reads of fewer than 16 bytes on average) needs span_scan, whose registers
// would cost the kernel a wave per SIMD although hardly any input ever takes it (152 instead of 122 VGPRs).  So the
// kernel proper (SLOW = false, four waves per SIMD) does not contain it: a wave that meets such a span -- or a malformed
// index -- appends its tile to a second list, `slow`, of the same layout and goes on with the next tile, and the SLOW
// instantiation (three waves per SIMD, list-driven, launched behind it) does those tiles from their start; whatever
// the first wave had stored for the tile is stored again, with the same values.
template <bool LINE, bool ALL, bool SLOW>
__global__ __launch_bounds__(256, (SLOW ? 3 : PPQC_DENSE_OCC))
void dense_kernel(BatchView b, const uint64_t* __restrict__ lut_g, ReadArrays ra, uint64_t read_base,
                  Partial* __restrict__ head, Partial* __restrict__ tail, uint64_t* __restrict__ base, int parity,
                  DenseOut slow) {
    constexpr int ND = kDenseSpans;
    __shared__ uint64_t lut[256];
    __shared__ uint32_t xs[4][64];
    __shared__ uint4 qvb[4][kQN];      // cut vectors (bases)
    __shared__ uint4 qvq[4][kQN];      // (qualities)
    __shared__ uint4 qB[4][kQN];       // B: gcs, n, qb, hi >> 10
    __shared__ uint64_t qE[4][kQN];    // B: low part of the error sum
    __shared__ uint32_t qx[4][kQN];    // cut offset (bits 0..4), bit 8: the record is the tile's head partial
    lut[threadIdx.x] = lut_g[threadIdx.x];
    if (!ALL && blockIdx.x == 0 && threadIdx.x < kDenseSubs) base[(uint64_t)(parity ^ 1) * kDenseSubs + threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    const uint64_t kTile = 1ull << b.tile_shift;
    const uint64_t n_tiles = (b.n_bytes + kTile - 1) >> b.tile_shift;
    const uint64_t cap = dense_sub_cap(n_tiles);
    uint32_t incl = 0;
    uint64_t count = n_tiles;
    if (!ALL) {
        const uint64_t cj = base[(uint64_t)parity * kDenseSubs + lane];
        const uint32_t mine = (uint32_t)(cj < cap ? cj : cap);
        incl = wave_scan_u32(mine);
        count = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if ((uint64_t)blockIdx.x * 4 >= count) return;   // whole block: nothing below needs a block-level barrier
    TileCtx c;
    c.read_base = read_base; c.R = b.n_reads; c.off = b.offsets; c.ra = ra; c.head = head; c.lut = lut;
    c.lane = lane; c.lane_off = (uint32_t)lane * kVec; c.xs = &xs[wib][0]; c.line = false;
    uint4* __restrict__ Qb = &qvb[wib][0];
    uint4* __restrict__ Qq = &qvq[wib][0];
    uint4* __restrict__ QB = &qB[wib][0];
    uint64_t* __restrict__ QE = &qE[wib][0];
    uint32_t* __restrict__ Qx = &qx[wib][0];
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + wib; i < count; i += n_waves) {
        uint64_t e = i;
        if (!ALL) {
            const int j = __popcll(__ballot((uint64_t)incl <= i));
            const uint64_t before = j ? (uint32_t)__builtin_amdgcn_readlane((int)incl, j - 1) : 0u;
            e = sload_u64(base, kDenseHeader + (uint64_t)j * cap + (i - before));
        }
        const uint64_t t = e & 0xFFFFFFFFull;
        if (t >= n_tiles || (!ALL && (e >> 32) >= b.n_reads)) continue;
        const uint64_t tile_start = t << b.tile_shift;
        const uint64_t remain = b.n_bytes - tile_start;
        const uint32_t tile_valid = remain < kTile ? (uint32_t)remain : (uint32_t)kTile;
        const uint32_t n_spans = (tile_valid + kSpan - 1) / kSpan;
        const uint32_t last_vec = (tile_valid - 1) & ~(uint32_t)(kVec - 1);
        const uint8_t* __restrict__ pb = b.bases + tile_start;
        const uint8_t* __restrict__ pq = b.quals + tile_start;
        c.t = t; c.tile_start = tile_start;
        uint4 vb[ND], vq[ND];
        load_batch<ND, LINE>(vb, vq, pb, pq, 0, c.lane_off, last_vec);
        Walk w;
        w.r = ALL ? find_read(c.off, c.R, tile_start, lane) : e >> 32;
        const uint64_t r_first = w.r;
        bool hard = false;   // !SLOW: the tile has been left to the SLOW instantiation
        w.rs = uniform_u64(c.off[w.r]);
        w.done = false;
        w.win = 0; w.win_r0 = 0; w.win_ok = false;
        w.re2 = 0;
        Win2 W;
        W.ok = false;
        win_ensure(W, c, w.r);
        w.re = readlane_u64(win_ends(W, w.r, lane), 0);
        w.re_rel = rel_end(w.re, tile_start);
        Acc acc{0, 0, 0, 0};
        bool dirty = false;
        Carry S{0, 0, 0, 0, 0};          // without the cut part of the newest entry's vector: the true carry is S - Pp
        PreSums Pp{0, 0, 0, 0};          // pre of the newest FINISHED entry (zero in front of the tile's first one)
        uint32_t q_head = 0, q_tail = 0; // entries finished / queued so far (ring slot = index & (kQN - 1))
        uint64_t q_r0 = w.r;             // read of entry 0

        // finish up to 64 waiting entries: the masked pass, the records
        auto drain = [&](uint32_t m) {
            const uint32_t slot = (q_head + ((uint32_t)lane < m ? (uint32_t)lane : 0u)) & (kQN - 1);
            const uint32_t xf = Qx[slot];
            const Acc pre = vec_prefix(Qb[slot], LINE ? Qb[slot] : Qq[slot], (int)(xf & 31u), c.lut);
            uint32_t pg = (uint32_t)__shfl_up((int)pre.gcs, 1), pn = (uint32_t)__shfl_up((int)pre.n, 1);
            uint32_t pqb = (uint32_t)__shfl_up((int)pre.qb, 1);
            uint64_t pe = __shfl_up((unsigned long long)pre.err, 1);
            if (lane == 0) { pg = Pp.gcs; pn = Pp.n; pqb = Pp.qb; pe = Pp.err; }   // zero in front of the tile's first entry
            if ((uint32_t)lane < m) {
                const uint4 B = QB[slot];
                const uint32_t gcs = B.x + pre.gcs - pg, nn = B.y + pre.n - pn, qb = B.z + pre.qb - pqb;
                const unsigned __int128 X = ((unsigned __int128)B.w << 42) + QE[slot] + pre.err - pe;
                const uint64_t ri = c.read_base + q_r0 + q_head + (uint32_t)lane;
                const bool to_head = (xf & 256u) != 0;
                Partial* p = &c.head[c.t];
                *(to_head ? &p->cnt : &c.ra.rec_cnt[ri]) = pack_cnt(gcs, nn, qb);
                *(to_head ? &p->hi : &c.ra.rec_hi[ri]) = (uint64_t)(X >> 32);
                *(to_head ? &p->lo : &c.ra.rec_lo[ri]) = (uint64_t)X & 0xffffffffull;
            }
            Pp.gcs = (uint32_t)__builtin_amdgcn_readlane((int)pre.gcs, (int)m - 1);
            Pp.n = (uint32_t)__builtin_amdgcn_readlane((int)pre.n, (int)m - 1);
            Pp.qb = (uint32_t)__builtin_amdgcn_readlane((int)pre.qb, (int)m - 1);
            Pp.err = readlane_u64(pre.err, (int)m - 1);
            q_head += m;
        };
        // everything finished and the carry put right: S becomes the true carry, the ring is empty
        auto settle = [&]() {
            while (q_tail != q_head) {
                const uint32_t pend = q_tail - q_head;
                drain(pend < 64u ? pend : 64u);
            }
            S.gcs -= Pp.gcs; S.n -= Pp.n; S.qb -= Pp.qb;
            const unsigned __int128 X = ((unsigned __int128)S.hi << 32) + S.lo - Pp.err;
            S.hi = ((uint64_t)(X >> 42)) << 10;
            S.lo = (uint64_t)(X - ((unsigned __int128)S.hi << 32));
            Pp = PreSums{0, 0, 0, 0};
        };

        for (uint32_t s0 = 0; s0 < n_spans && !w.done && !hard; s0 += ND) {
            if (s0) load_batch<ND, LINE>(vb, vq, pb, pq, s0, c.lane_off, last_vec);
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                if (s0 + k >= n_spans || w.done || hard) break;
                const uint32_t span_rel = (s0 + k) * kSpan, span_end_rel = span_rel + kSpan;
                const uint64_t span_start = tile_start + span_rel, span_end = span_start + kSpan;
                if (w.re >= span_end) {   // no end inside: a fast span
                    Acc tot;
                    if (LINE) {
                        const uint32_t ty = (uint32_t)w.r & 3u;
                        if (ty == 1u) tot = vec_sums_bases(vb[k]);
                        else if (ty == 3u) tot = vec_sums_quals(vb[k], c.lut);
                        else tot = Acc{0, 0, 0, 0};
                    } else {
                        tot = vec_sums(vb[k], vq[k], c.lut);
                    }
                    acc_add(acc, tot);
                    dirty = true;
                    continue;
                }
                win_ensure(W, c, w.r);
                const uint64_t ends = win_ends(W, w.r, lane);
                const bool in = w.r + (uint64_t)lane < c.R && ends < span_end && ends >= span_start;
                const uint64_t nin = ~__ballot(in);
                const uint32_t n = nin ? (uint32_t)__builtin_ctzll(nin) : 64u;
                if (!SLOW && (n == 0 || n >= 64)) {   // left to the SLOW instantiation, from the tile's start
                    if (lane == 0) {
                        uint64_t* sb = (uint64_t*)(uintptr_t)(slow & 0xFFFFFFFFFFFFull);
                        const uint64_t j = (t >> 2) & (kDenseSubs - 1);
                        const unsigned long long kk = atomicAdd((unsigned long long*)(sb + ((slow >> 48) & 1ull) * kDenseSubs + j), 1ull);
                        if (kk < cap) sb[kDenseHeader + j * cap + kk] = t | (r_first << 32);
                    }
                    hard = true;
                    break;
                }
                if (SLOW && (n == 0 || n >= 64)) {
                    // 64 ends or more in one span (or a malformed index): the span goes through span_scan, from a settled state
                    settle();
                    w.re_rel = rel_end(w.re, tile_start);
                    w.win_ok = false;
                    if (w.re_rel <= span_rel && span_rel != 0) {
                        flush_read_c(acc, dirty, S, w, c);
                        walk_advance<true>(w, c);
                    }
                    if (!w.done) {
                        if (w.re_rel >= span_end_rel) {
                            const Acc tot = vec_sums(vb[k], LINE ? vb[k] : vq[k], c.lut);
                            acc_add(acc, tot);
                            dirty = true;
                            if (w.re_rel == span_end_rel) {
                                flush_read_c(acc, dirty, S, w, c);
                                walk_advance<true>(w, c);
                            }
                        } else {
                            span_scan(vb[k], LINE ? vb[k] : vq[k], span_rel, w, acc, dirty, S, c);
                        }
                    }
                    W.ok = false;
                    q_head = q_tail = 0;
                    q_r0 = w.r;
                    continue;
                }
                // ---- the entries of this span: their vectors into the ring
                const uint32_t rel = (uint32_t)(ends - span_start) & (kSpan - 1);
                const int v = (int)(rel >> 4);
                const uint4 eb = make_uint4((uint32_t)__shfl((int)vb[k].x, v), (uint32_t)__shfl((int)vb[k].y, v),
                                            (uint32_t)__shfl((int)vb[k].z, v), (uint32_t)__shfl((int)vb[k].w, v));
                uint4 eq = eb;
                if (!LINE)
                    eq = make_uint4((uint32_t)__shfl((int)vq[k].x, v), (uint32_t)__shfl((int)vq[k].y, v),
                                    (uint32_t)__shfl((int)vq[k].z, v), (uint32_t)__shfl((int)vq[k].w, v));
                const Acc tot = vec_sums(vb[k], LINE ? vb[k] : vq[k], c.lut);
                if (dirty) {   // fast spans since the last end: fold the per-lane sums into the carry
                    Carry Z = S;
                    const uint32_t G = wave_sum_u32(acc.gcs), N = wave_sum_u32(acc.n), Q = wave_sum_u32(acc.qb);
                    const uint32_t E0 = wave_sum_u32((uint32_t)acc.err & 0x1FFFFFu);
                    const uint32_t E1 = wave_sum_u32((uint32_t)(acc.err >> 21) & 0x1FFFFFu);
                    const uint32_t E2 = wave_sum_u32((uint32_t)(acc.err >> 42));
                    Z.gcs += G; Z.n += N; Z.qb += Q;
                    Z.lo += (uint64_t)E0 + ((uint64_t)E1 << 21);
                    Z.hi += (uint64_t)E2 << 10;
                    S = Z;
                    acc = Acc{0, 0, 0, 0};
                    dirty = false;
                }
                // data lanes: inclusive scan of the full sums, then exclusive; the error sum back in 64 bits
                const Pieces a = to_pieces(tot);
                const Pieces P{wave_scan_u32(a.c1), wave_scan_u32(a.c2), wave_scan_u32(a.e0), wave_scan_u32(a.e1)};
                const uint32_t Ec1 = P.c1 - a.c1, Eqb = (P.c2 - a.c2) & 0x3FFFFu;
                const uint64_t Perr = pieces_err(P);
                const uint64_t Eerr = Perr - tot.err;
                const uint32_t Tc1 = (uint32_t)__builtin_amdgcn_readlane((int)P.c1, 63);
                const uint32_t Tqb = (uint32_t)__builtin_amdgcn_readlane((int)P.c2, 63) & 0x3FFFFu;
                const uint64_t Terr = readlane_u64(Perr, 63);
                // entry lanes: what lies in front of the entry's vector, then B
                const uint32_t Jc1 = (uint32_t)__shfl((int)Ec1, v), Jqb = (uint32_t)__shfl((int)Eqb, v);
                const uint64_t Jerr = __shfl((unsigned long long)Eerr, v);
                const uint32_t pc1 = (uint32_t)__shfl_up((int)Jc1, 1), pqb = (uint32_t)__shfl_up((int)Jqb, 1);
                const uint64_t perr = __shfl_up((unsigned long long)Jerr, 1);
                if ((uint32_t)lane < n) {
                    const bool first = lane == 0;
                    const uint32_t dc1 = Jc1 - (first ? 0u : pc1);   // both fields of an exclusive prefix only grow
                    uint4 B = make_uint4(dc1 & 0xFFFFu, dc1 >> 16, Jqb - (first ? 0u : pqb), 0u);
                    uint64_t be = Jerr - (first ? 0ull : perr);
                    uint32_t flag = 0;
                    if (first) {
                        B.x += S.gcs; B.y += S.n; B.z += S.qb; B.w = (uint32_t)(S.hi >> 10);
                        be += S.lo;
                        flag = w.rs < tile_start ? 256u : 0u;   // the open read began in an earlier tile: the head partial
                    }
                    const uint32_t slot = (q_tail + (uint32_t)lane) & (kQN - 1);
                    Qb[slot] = eb;
                    if (!LINE) Qq[slot] = eq;
                    QB[slot] = B;
                    QE[slot] = be;
                    Qx[slot] = (rel & 15u) | flag;
                }
                // the whole vectors from the last entry's on open the next read
                {
                    const int last = (int)n - 1;
                    const uint32_t Lc1 = Tc1 - (uint32_t)__builtin_amdgcn_readlane((int)Jc1, last);
                    const uint32_t Lqb = Tqb - (uint32_t)__builtin_amdgcn_readlane((int)Jqb, last);
                    const uint64_t Lerr = Terr - readlane_u64(Jerr, last);
                    S = Carry{Lc1 & 0xFFFFu, Lc1 >> 16, Lqb, 0, Lerr};
                }
                q_tail += n;
                w.rs = readlane_u64(ends, (int)n - 1);
                w.r += n;
                w.done = w.r >= c.R;
                if (!w.done) {
                    w.re = readlane_u64(ends, (int)n);   // n < 64: the end of read w.r is in the same register
                    w.re_rel = rel_end(w.re, tile_start);
                }
                __builtin_amdgcn_wave_barrier();
                if (q_tail - q_head >= 64u) drain(64u);
            }
        }
        if (hard) continue;
        settle();
        if (!w.done) {
            if (w.re == tile_start + tile_valid) {
                flush_read_c(acc, dirty, S, w, c);     // the read ends with the tile: complete, or the head partial
            } else if (w.rs < tile_start + kTile && w.re > tile_start + kTile) {
                Partial* p = w.rs >= tile_start ? &tail[t] : &head[t];
                flush_c(acc, dirty, S, lane, &p->cnt, &p->hi, &p->lo);
            }
        }
    }
}

// The two-stream kernel: 8 spans of both streams per batch (16 loads, 16 KiB in flight per wave), 4 waves per SIMD at
// exactly 128 VGPRs.
__global__ __launch_bounds__(256, 4) void tile_kernel(BatchView b, const uint64_t* __restrict__ lut_g, ReadArrays ra,
                                                      uint64_t read_base, Partial* __restrict__ head,
                                                      Partial* __restrict__ tail, uint64_t n_tiles, DenseOut dz) {
    tile_body<8, false>(b, lut_g, ra, read_base, head, tail, n_tiles, dz);
}

// Line mode (FASTQ text, one stream): WPB waves per block, OCC waves per SIMD (launch_tile_kernel says which and why).
template <int NB, int WPB, int OCC>
__global__ __launch_bounds__(64 * WPB, OCC)
void tile_kernel_line_w(BatchView b, const uint64_t* __restrict__ lut_g, ReadArrays ra, uint64_t read_base,
                        Partial* __restrict__ head, Partial* __restrict__ tail, uint64_t n_tiles, DenseOut dz) {
    tile_body<NB, true, WPB>(b, lut_g, ra, read_base, head, tail, n_tiles, dz);
}

// One thread per read: merge the partials of reads that span tiles, define empty reads, then decode the raw sums.
// write_lens: also copies the batch's channel / mux / start time into the run's arrays, leaves the sort input of the run
// (lens32: the length, 0 for reads without a valid channel) and raises `gate`, the device words whose OR bounds every sort
// key, so that the sort + N50 chain can start right behind this kernel -- unless the batch's index kernel has done all
// that beside the per-read kernel (extremes_kernel<true>; the host then knows the largest key).
__global__ __launch_bounds__(256) void finalize_reads_kernel(BatchView b, ReadArrays ra, uint64_t read_base,
                                                             const Partial* __restrict__ head,
                                                             const Partial* __restrict__ tail, QThr thr,
                                                             uint64_t* __restrict__ csr_errors,
                                                             uint32_t n_channels, uint64_t* __restrict__ gate,
                                                             int write_lens) {
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.n_reads) {
        uint64_t a = b.offsets[i], e = b.offsets[i + 1];
        // A device batch cannot be validated on the host: a malformed CSR must neither fault (the partial arrays are
        // indexed by tile) nor pass silently -- the read is zeroed and counted; ppqc_fetch_tables / ppqc_read_stats fail.
        if (e < a || e > b.n_bytes || e - a >= 0xFFFFFFFFull || (i == 0 && a != 0) ||
            (i + 1 == b.n_reads && e != b.n_bytes)) {
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
        uint32_t len_out = 0;
        if (flagged) {
            // lib/fast5_reader.py:37-40: a 'fail' path keeps ReadLength = 0, QScore = 6.99, GCcont = 0
            flags = 2 | (6.99 < thr.fail_q ? 1 : 0) | (1 << 4);   // QScore 6.99: quality bin 0
            ra.len[g] = 0; ra.gc[g] = 0; ra.nn[g] = 0; ra.qsum[g] = 0; ra.err_hi[g] = 0; ra.err_lo[g] = 0;
            mq = 6.99;
        } else {
            if (L != 0) flags = quality_flags(L, H, l32, thr, &mq);
            else flags = 0.0 < thr.fail_q ? 1 : 0;
            len_out = (uint32_t)L;
            ra.len[g] = (uint32_t)L; ra.gc[g] = (uint32_t)gcs; ra.nn[g] = (uint32_t)nn;
            ra.qsum[g] = (int64_t)qb - 33 * (int64_t)L;
            ra.err_hi[g] = H; ra.err_lo[g] = l32;
        }
        ra.mean_q[g] = mq;
        ra.flags[g] = flags;
        if (write_lens) {   // else the batch's index kernel did all of this at submit time (extremes_kernel<true>)
            const uint32_t ch = b.channel[i];
            const uint32_t key = (ch >= 1 && ch <= n_channels) ? len_out : 0u;   // what extremes_kernel would write
            ra.lens32[g] = key;
            if (key) atomicMax(&s_max, key);
            ra.channel[g] = ch;
            ra.mux[g] = b.mux[i];
            ra.time[g] = b.time[i];
        }
    }
    __syncthreads();
    // 64 gate words, picked by block: thousands of atomics on ONE address would serialise (measured: +40 us)
    if (threadIdx.x == 0 && s_max) atomicOr((unsigned long long*)&gate[blockIdx.x & (kGateWords - 1)], (unsigned long long)s_max);
}

constexpr uint32_t kDenseLen = 768;    // a tile is dense when its reads average fewer bytes than this
// dense_kernel alone (no tile_kernel pass, no list) for a batch whose reads average fewer than kDenseAllLen bytes.  Same
// box, alone against the pair (GB/s): mean 1 kb 4458 / 3652, 1.5 kb 4675 / 4438, 2 kb 4706 / 4890, 2.5 kb 4886 / 5457,
// 4 kb 5248 / 5910 (its sparse spans run at 3 waves per SIMD with 8 loads in flight).  The per-tile threshold of the pair
// stays at 768 B (1280, 1792 and 2560 are slower at every mean tried).  A small batch pays the pair's second launch and the
// wait between the two (about 15 us) whatever it holds: below kSmallBatchBytes the bound is kDenseAllLenSmall (the
// adversarial mix of BASELINE config 4, 240 MB, mean 1.2 kb: 0.119 against 0.131 ms).
constexpr uint32_t kDenseAllLen = 1792, kDenseAllLenSmall = 2560;
constexpr uint64_t kSmallBatchBytes = 256ull << 20;

// The per-read kernels of one batch: tile_kernel over every tile (one tile per wavefront, 4 waves per block: the
// hardware dispatcher is the dynamic tile scheduler; on config 2 that beat every persistent grid, 3.07 ms against 3.13
// with 64 blocks per CU, 3.37 with 8, 3.6 with resident blocks only), then dense_kernel over the tiles it left on the
// list -- or, for a batch that is dense on average, dense_kernel alone over every tile.  `tn` (ppqc_set_tuning) can
// force either kernel on every tile and move the thresholds: the results are the same, the tests hold them to each other.
// Behind every dense_kernel: its SLOW instantiation over the tiles it left on the slow list (a bounded grid with a stride
// loop; the list is empty for all but pathological inputs, and the launch then costs a few microseconds).
template <bool LINE>
static hipError_t launch_dense_slow(const BatchView& b, const uint64_t* lut_dev, ReadArrays ra, uint64_t read_base,
                                    Partial* head, Partial* tail, DenseList& sl, uint64_t n_tiles, hipStream_t s) {
    uint64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((dense_kernel<LINE, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra, read_base,
                       head, tail, sl.base, sl.parity, (DenseOut)0);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) sl.parity ^= 1;   // it leaves the other set of counters at zero for the next launch
    return e;
}

hipError_t launch_tile_kernel(const BatchView& b, const uint64_t* lut_dev, ReadArrays ra, uint64_t read_base,
                              Partial* head, Partial* tail, DenseList& dl, DenseList& sl, const Tuning& tn, hipStream_t s,
                              bool line_mode, bool* dense_alone) {
    if (dense_alone) *dense_alone = false;
    if (b.n_bytes == 0 || b.n_reads == 0) return hipSuccess;
    uint32_t mode = tn.tile_dense < 0 ? 0u : (tn.tile_dense ? 2u : 1u);   // 0 probe every tile, 1 none dense, 2 all dense
    if (!dl.base || !sl.base) mode = 1;
    uint32_t dense_len = tn.dense_len > 0 ? (uint32_t)tn.dense_len : kDenseLen;
    if (dense_len < 16) dense_len = 16;
    uint32_t T = (uint32_t)((1ull << b.tile_shift) / dense_len);
    T = T < 1 ? 1 : (T > 8191 ? 8191 : T);
    const DenseOut dz = make_dense_out(dl.base, dl.parity, mode, T);
    const DenseOut sz = make_dense_out(sl.base, sl.parity, 0, 1);
    const uint64_t n_tiles = (b.n_bytes + (1ull << b.tile_shift) - 1) >> b.tile_shift;
    const bool all = sl.base != nullptr && (tn.tile_all >= 0
                         ? tn.tile_all != 0
                         : (mode == 0 && b.n_bytes / b.n_reads < (b.n_bytes < kSmallBatchBytes ? kDenseAllLenSmall : kDenseAllLen)));
    if (all) {
        if (dense_alone) *dense_alone = true;
        uint64_t blocks = (n_tiles + 3) / 4;
        if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
        if (line_mode)
            hipLaunchKernelGGL((dense_kernel<true, true, false>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra,
                               read_base, head, tail, dl.base, dl.parity, sz);
        else
            hipLaunchKernelGGL((dense_kernel<false, true, false>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra,
                               read_base, head, tail, dl.base, dl.parity, sz);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return line_mode ? launch_dense_slow<true>(b, lut_dev, ra, read_base, head, tail, sl, n_tiles, s)
                         : launch_dense_slow<false>(b, lut_dev, ra, read_base, head, tail, sl, n_tiles, s);
    }
    if (line_mode) {
        // 4 spans per batch, taken as one group when the line covers them (process_batch), 79 VGPRs = 6 waves per
        // SIMD, 2 waves per block: 0.70 ms for 3.97 GB of text (5.6 TB/s) against 0.76 for 8 spans, span by span,
        // at 5 waves.  Tried on the same text, whole path: 1, 2 or 4 waves per block 2.14 / 2.15 / 2.11 TB/s;
        // two register sets of 8 or 4 spans (the next batch in flight while one is processed) 2.07 / 2.12; groups
        // of 4 with 8 spans per batch spill at 5 waves (1.85) and give 2.13 at 4; 12 or 16 spans per batch at 4 waves
        // 1.95 / 1.92 (before the groups).  None of this moves the kernel by more than 2 %: what did was taking the
        // header kernel from beside it (ppqc_fastq_gpu.hip).
        uint64_t blocks = (n_tiles + 1) / 2;
        if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
        hipLaunchKernelGGL((tile_kernel_line_w<kLineGroup, 2, 6>), dim3((unsigned)blocks), dim3(128), 0, s, b, lut_dev, ra,
                           read_base, head, tail, n_tiles, dz);
    } else {
        uint64_t blocks = (n_tiles + 3) / 4;
        if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
        hipLaunchKernelGGL(tile_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra, read_base, head, tail,
                           n_tiles, dz);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || mode == 1) return e;
    // dense_kernel: one tile per wave over the list; the grid covers the most entries the list can have (see there)
    uint64_t bound = n_tiles;
    if (mode == 0) {
        const uint64_t most = b.n_reads / T + 1;
        if (most < bound) bound = most;
    }
    uint64_t blocks = (bound + 3) / 4;
    if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
    if (line_mode)
        hipLaunchKernelGGL((dense_kernel<true, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra,
                           read_base, head, tail, dl.base, dl.parity, sz);
    else
        hipLaunchKernelGGL((dense_kernel<false, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, b, lut_dev, ra,
                           read_base, head, tail, dl.base, dl.parity, sz);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the counters change roles only when a dense_kernel has run: it is what leaves the other counter at zero
    dl.parity ^= 1;
    return line_mode ? launch_dense_slow<true>(b, lut_dev, ra, read_base, head, tail, sl, n_tiles, s)
                     : launch_dense_slow<false>(b, lut_dev, ra, read_base, head, tail, sl, n_tiles, s);
}

uint64_t dense_list_words(uint64_t n_tiles) { return kDenseHeader + kDenseSubs * dense_sub_cap(n_tiles); }

hipError_t launch_finalize_reads(const BatchView& b, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, const QThr& thr, uint64_t* csr_errors,
                                 uint32_t n_channels, uint64_t* gate, bool write_lens, hipStream_t s) {
    if (b.n_reads == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((b.n_reads + 255) / 256);
    hipLaunchKernelGGL(finalize_reads_kernel, dim3(blocks), dim3(256), 0, s, b, ra, read_base, head, tail, thr,
                       csr_errors, n_channels, gate, write_lens ? 1 : 0);
    return hipGetLastError();
}

}  // namespace ppqc
