// FASTQ text -> per-read statistics without a host tokenizer (gfx950).
//
// Replaces, for text that is already in (or on its way to) HBM, the record parsing the reference leaves to Biopython
// (SeqIO.parse(handle, "fastq"), lib/seq_routines.py:50; record layout lib/fast5_reader.py:57, lib/fastqparser.py:7-16).
// Strict 4-line records only; anything else is counted as malformed and the run fails loudly (the host packer
// ppqc_fastq_scan/pack accepts blank lines between records and reports the record number).
//
// The text is never compacted into separate base / quality streams.  Instead:
//   1. ONE streaming pass over the text (nl_scan_kernel) counts newlines, notes the few vectors that hold a line end and,
//      for texts of long lines, sums every 1 KiB span that lies inside one line with the classifier its bytes suggest
//      (span sums); a scan numbers the lines, and nl_place_kernel places the line starts from the notes (chunks whose
//      notes overflowed are read again, a text of short lines as a whole: nl_fill_kernel);
//   2. the raw sums of every sequence and quality line: for long lines from the span sums, with only the two end spans
//      of a line read from the text (line_sum_kernel); for short lines by the per-read tile kernel in "line mode"
//      (bases = quals = text, offsets = line starts: a second pass over the text);
//   3. fastq_header_kernel parses ch= and start_time= out of the headers; fastq_finalize_kernel (one thread per record)
//      takes the base sums of line 2 and the quality sums of line 4, removes the contribution of the stripped line end
//      from the quality sums, validates the record and writes the same per-read arrays finalize_reads writes.
// Traffic: 1.19 reads of the text for long reads (profiles/r02*_bench_fastq_pmc_hbm.json), two to three for short ones,
// and no write of the sequence bytes.
#include "ppqc_internal.hpp"
#include "ppqc_fastq_dev.hpp"

namespace ppqc {

// high bit of each byte set where a QUALITY byte is invalid: outside '!'..'~' and not ASCII white space (line ends and
// trailing blanks are stripped, not judged)
__device__ __forceinline__ uint32_t badq_mask(uint32_t x, uint32_t* ws_out = nullptr) {
    const uint32_t H = 0x80808080u, hi = x & H, lo7 = x & 0x7F7F7F7Fu;
    const uint32_t ge9 = (lo7 + 0x77777777u) & H, ge14 = (lo7 + 0x72727272u) & H, ge32 = (lo7 + 0x60606060u) & H;
    const uint32_t ge33 = (lo7 + 0x5F5F5F5Fu) & H, ge127 = (lo7 + 0x01010101u) & H;
    const uint32_t invalid = hi | (~ge33 & H) | ge127;
    const uint32_t ws = ((ge9 & ~ge14) | (ge32 & ~ge33)) & ~hi;
    if (ws_out) *ws_out = ws;
    return invalid & ~ws;
}
// *n_ws (optional): the white-space bytes among the first `valid`, a by-product of the same masks
__device__ __forceinline__ uint32_t vec_badq_bits(const uint4& v, int valid, uint32_t* n_ws = nullptr) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t bits = 0, wsb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t ws = 0;
        const uint32_t m = badq_mask(w[k], &ws);
        bits |= (((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u)) << (4 * k);
        wsb |= (ws >> 7) << k;   // byte b of word k at bit 8 b + k: only their number matters
    }
    if (valid < 16) {
        bits &= valid <= 0 ? 0u : ((1u << valid) - 1u);
        if (n_ws) *n_ws = vec_ws_count(v, valid);   // the end of the text: once per text
    } else if (n_ws) {
        *n_ws = __popc(wsb);
    }
    return bits;
}

// true iff some byte of the 16-byte vector lies outside '!'..'~' (newlines, other white space and invalid quality bytes);
// the four words are folded before the test
__device__ __forceinline__ bool vec_is_odd(const uint4& v) {
    const uint32_t M = 0x7F7F7F7Fu, one = 0x01010101u, lowest = 0x5F5F5F5Fu;
    const uint32_t a0 = v.x & M, a1 = v.y & M, a2 = v.z & M, a3 = v.w & M;
    const uint32_t high = (v.x | v.y | v.z | v.w) | (a0 + one) | (a1 + one) | (a2 + one) | (a3 + one);   // a byte >= 0x7F
    const uint32_t ge = (a0 + lowest) & (a1 + lowest) & (a2 + lowest) & (a3 + lowest);                    // every byte >= '!'
    return ((high | ~ge) & 0x80808080u) != 0;
}

__device__ __forceinline__ uint4 load_vec_clamped(const uint8_t* __restrict__ text, uint64_t pos, uint64_t n) {
    // 16-byte aligned load; a vector that starts beyond the data is replaced by the last in-bounds one (masked later)
    const uint64_t last = (n - 1) & ~15ull;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // each pass streams the text once: nontemporal, like the tile kernel's loads
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(text + (pos < last ? pos : last)));
    return make_uint4(t[0], t[1], t[2], t[3]);
}

// ---- the table of line starts ------------------------------------------------------------------------------------------
// Pass 1 (nl_scan_kernel, one wave per chunk of kNlChunk bytes, the only full read of the text when lines are long):
// counts the chunk's newlines and writes down its ODD VECTORS -- the 16-byte vectors that hold a newline or a byte that
// is invalid as a quality -- as (vector, newline bits, invalid bits) entries, at most kNlEntries per chunk.  Reads of
// 500 bases and more give a chunk a handful (a line end every 1 KiB at a mean of 500: 8 of 512 vectors).  A scan of the
// counts numbers the lines; the host, which has to wait for the count anyway to size the tables, also learns whether any
// chunk had more odd vectors than its list holds.  If none: nl_place_kernel, half a wave per chunk, places the line
// starts from the entries (a few MB) without touching the text again.  Else (short reads: most vectors hold a line end)
// nl_fill_kernel reads the text a second time, as it always did.
// No wave waits for another one.  (Tried: one pass by a chained scan with decoupled look-back over two levels; correct, but
// some 4000 resident waves polling status words across the XCDs made it 1.94 ms for 3.97 GB, against 0.60 + 0.62 ms for
// the two passes.)
constexpr int kNlEntries = (int)kNlEntriesPerChunk;

// Span sums (SUMS): while the chunk's vectors are in registers anyway, every CLEAN 1 KiB span -- all of it inside the text
// and every byte in '!'..'~', so it lies inside one line -- is also summed by ONE classifier, guessed from the bytes: the
// base classifier when every byte is at or above '@' (a stretch of sequence), else the quality classifier.  The sums go
// to span_sum[span] = {w0, w1}:
//   w0 bit 63: the span has sums; bit 62: they are quality sums;
//   bases:   w0 bits [0,16) G/C/S count, [16,32) N count
//   quality: w0 bits [0,20) byte sum, w1 = fixed-point error sum (< 2^62)
// line_sum_kernel then adds up the span sums inside each sequence and quality line and reads from the text only the spans
// at a line's two ends and the (rare) spans whose guess was not the line's classifier: for long reads the text is read
// ONCE plus a fifth.  A wrong guess costs time, never correctness.
constexpr uint64_t kSpanHas = 1ull << 63, kSpanQual = 1ull << 62;

// entry = vector index in the chunk (bits 32..41) | newline bits (0..15) | invalid-quality bits (16..31)
// NOTES = false (a text of short lines, whose notes would overflow in most chunks and be thrown away): newlines are
// counted, nothing else.
template <bool SUMS, bool NOTES = true>
__global__ __launch_bounds__(256) void nl_scan_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ n_entries,
                                                      uint64_t* __restrict__ entries, uint64_t n_chunks,
                                                      uint64_t* __restrict__ overflow, uint64_t* __restrict__ span_sum,
                                                      const uint64_t* __restrict__ lut_g) {
    __shared__ uint64_t lut[SUMS ? 256 : 1];
    if (SUMS) {
        lut[threadIdx.x] = lut_g[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    constexpr int kIt = (int)(kNlChunk / kSpan);
    uint4 vv[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {   // all loads of the chunk in flight before the first use
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        vv[it] = load_vec_clamped(text, pos < n ? pos : 0, n);
    }
    uint32_t cnt = 0, ne = 0;   // ne: wave-uniform
    uint32_t wsn = 0;           // white-space bytes, newlines included (see fastq_finalize_kernel: the white-space balance)
    uint64_t s0 = 0, s1 = 0;    // SUMS: lane `it` keeps the words of span `it`
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        const uint4& v = vv[it];
        uint32_t bits = 0, bad = 0;
        // nearly every vector holds only '!'..'~' bytes: one cheap test, the exact masks only where it fires
        const bool oddv = pos < n && vec_is_odd(v);
        if (oddv) {
            const uint64_t rem = n - pos;
            bits = vec_nl_bits(v, rem < 16 ? (int)rem : 16);
            uint32_t nws = 0;
            if (NOTES) bad = vec_badq_bits(v, rem < 16 ? (int)rem : 16, &nws);
            else nws = vec_ws_count(v, rem < 16 ? (int)rem : 16);
            wsn += nws;
        }
        cnt += __popc(bits);
        const uint64_t odd = NOTES ? __ballot((bits | bad) != 0) : 0ull;
        if (NOTES && odd) {
            const uint32_t at = ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(odd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)odd, 0u));
            if ((bits | bad) != 0 && at < (uint32_t)kNlEntries)
                entries[chunk * kNlEntries + at] = ((uint64_t)(it * 64 + lane) << 32) | ((uint64_t)bad << 16) | bits;
            ne += (uint32_t)__popcll(odd);
        }
        if (SUMS) {
            const bool whole = chunk * kNlChunk + (uint64_t)(it + 1) * kSpan <= n;   // wave-uniform
            uint64_t w0 = 0, w1 = 0;
            if (whole && __ballot(oddv) == 0) {
                const bool low = ((v.x & v.y & v.z & v.w) & 0x40404040u) != 0x40404040u;   // a byte below '@'
                if (__ballot(low) == 0) {
                    const Acc a = vec_sums_bases(v);
                    const uint32_t P = wave_sum_u32(a.gcs | (a.n << 16));   // at most 1024 each
                    w0 = kSpanHas | P;
                } else {
                    const Acc a = vec_sums_quals(v, lut);
                    // per lane: byte sum <= 4080, error sum < 2^56, in pieces whose wave sums fit 32 bits
                    const uint32_t P0 = wave_sum_u32((uint32_t)a.err & 0x3FFFFFFu);
                    const uint32_t P1 = wave_sum_u32((uint32_t)(a.err >> 26) & 0x3FFFFFFu);
                    const uint32_t P2 = wave_sum_u32((uint32_t)(a.err >> 52) | (a.qb << 10));
                    w0 = kSpanHas | kSpanQual | (uint64_t)(P2 >> 10);
                    w1 = (uint64_t)P0 + ((uint64_t)P1 << 26) + ((uint64_t)(P2 & 0x3FFu) << 52);
                }
            }
            if (lane == it) { s0 = w0; s1 = w1; }
        }
    }
    if (SUMS && lane < kIt) {
        uint64_t* o = span_sum + 2 * (chunk * kIt + (uint64_t)lane);
        o[0] = s0; o[1] = s1;
    }
    cnt = wave_sum_u32(cnt);
    wsn = wave_sum_u32(wsn) - cnt;   // at most 8192 each
    if (lane == 0) {
        counts[chunk] = cnt | (wsn << 16);
        n_entries[chunk] = NOTES ? ne : (uint32_t)kNlEntries + 1u;   // no notes: as if the list had overflowed
        if (NOTES && ne > (uint32_t)kNlEntries) atomicAdd((unsigned long long*)overflow, 1ull);
    }
}

constexpr int kScanBlock = 2048;   // entries per block of the two-level scan
uint64_t nl_scan_block_words(uint64_t n_chunks) { return (n_chunks + kScanBlock - 1) / kScanBlock + 2; }

// counts[i] = newlines of chunk i | its other white-space bytes << 16; the latter are only added up (*ws_total)
__global__ __launch_bounds__(256) void nl_blocksum_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                          uint64_t* __restrict__ bsum, uint64_t* __restrict__ ws_total) {
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    uint64_t t = 0;   // both sums in one word: a block's chunks hold at most 2^24 bytes
    for (int k = 0; k < kScanBlock / 256; ++k) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) { const uint32_t c = counts[i]; t += (uint64_t)(c & 0xFFFFu) | ((uint64_t)(c >> 16) << 32); }
    }
    t = wave_sum_u64(t);
    if ((threadIdx.x & 63) == 0) atomicAdd(&acc, (unsigned long long)t);
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = acc & 0xFFFFFFFFull;
        if (acc >> 32) atomicAdd((unsigned long long*)ws_total, acc >> 32);
    }
}

// single block: exclusive scan of the block sums in place; grand total -> *total
// Also hands the host what it waits for (pinned words host_out[0..2]: newlines, chunks whose entry list overflowed, the
// last byte of the text), so that no copy command stands between this kernel and the host.
__global__ __launch_bounds__(256) void nl_scan_top_kernel(uint64_t* __restrict__ bsum, uint64_t nblocks,
                                                          uint64_t* __restrict__ total, uint64_t* __restrict__ host_out,
                                                          const uint8_t* __restrict__ last_byte) {
    __shared__ uint64_t part[256];
    const uint32_t t = threadIdx.x;
    const uint64_t seg = (nblocks + 255) / 256;
    const uint64_t b = t * seg, e = b + seg < nblocks ? b + seg : nblocks;
    uint64_t s = 0;
    for (uint64_t i = b; i < e; ++i) s += bsum[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 256; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
        *total = run;
        if (host_out) {
            host_out[0] = run;
            host_out[1] = total[1];
            host_out[2] = *last_byte;
            __threadfence_system();
        }
    }
    __syncthreads();
    uint64_t run = part[t];
    for (uint64_t i = b; i < e; ++i) { const uint64_t v = bsum[i]; bsum[i] = run; run += v; }
}

// per block of kScanBlock entries: thread owns 8 consecutive entries
__global__ __launch_bounds__(256) void nl_scan_apply_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                            const uint64_t* __restrict__ bsum,
                                                            uint64_t* __restrict__ prefix) {
    __shared__ uint64_t part[256];
    const uint32_t t = threadIdx.x;
    const uint64_t b0 = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)t * 8;
    uint32_t v[8];
    uint64_t loc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = b0 + k < n ? counts[b0 + k] & 0xFFFFu : 0; loc += v[k]; }
    part[t] = loc;
    __syncthreads();
    if (t == 0) {
        uint64_t run = bsum[blockIdx.x];
        for (int i = 0; i < 256; ++i) { const uint64_t x = part[i]; part[i] = run; run += x; }
    }
    __syncthreads();
    uint64_t run = part[t];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (b0 + k < n) prefix[b0 + k] = run;
        run += v[k];
    }
}


// half a wave per kPlaceIlp consecutive chunks: lane j of the half owns the j-th odd vector of each of them.  The loads
// of all its chunks are requested before the first is looked at: the kernel is a chain of three dependent loads per
// chunk (count, entry, then the stores), and one chunk per half-wave left it bound by that latency (51 us for 485 000
// chunks).
constexpr int kPlaceIlp = 4;
__global__ __launch_bounds__(256) void nl_place_kernel(const uint64_t* __restrict__ prefix, const uint32_t* __restrict__ n_entries,
                                                       const uint64_t* __restrict__ entries, uint64_t n_chunks,
                                                       uint64_t* __restrict__ line_off, uint64_t n, uint64_t n_lines,
                                                       uint64_t* __restrict__ errors) {
    const int lane = threadIdx.x & 63, j = lane & 31;
    const uint64_t chunk0 = (((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (uint64_t)(lane >> 5)) * kPlaceIlp;
    if (blockIdx.x == 0 && threadIdx.x == 0) { line_off[0] = 0; line_off[n_lines] = n; }
    uint32_t ne[kPlaceIlp];
    uint64_t pf[kPlaceIlp], en[kPlaceIlp];
#pragma unroll
    for (int q = 0; q < kPlaceIlp; ++q) {
        const bool live = chunk0 + q < n_chunks;
        ne[q] = live ? n_entries[chunk0 + q] : 0;
        if (ne[q] > (uint32_t)kNlEntries) ne[q] = 0;   // an overflowed list is incomplete: nl_fill_kernel takes the chunk
        pf[q] = live ? prefix[chunk0 + q] : 0;
    }
#pragma unroll
    for (int q = 0; q < kPlaceIlp; ++q) en[q] = (uint32_t)j < ne[q] ? entries[(chunk0 + q) * kNlEntries + j] : 0;
    uint32_t nbad = 0;
#pragma unroll
    for (int q = 0; q < kPlaceIlp; ++q) {
        const uint64_t e = en[q];
        uint32_t bits = (uint32_t)e & 0xFFFFu;
        const uint32_t bad = (uint32_t)(e >> 16) & 0xFFFFu;
        const uint32_t c = __popc(bits);
        // exclusive prefix of the newline counts inside each half: the full-wave scan, minus what the lower half added
        const uint32_t inc = wave_scan_incl_u32(c);
        const uint32_t low = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
        uint64_t k = pf[q] + (inc - c) - (lane >= 32 ? low : 0u);   // the line the vector starts in
        const uint64_t pos = (chunk0 + q) * kNlChunk + (uint64_t)(uint32_t)(e >> 32) * kVec;
        // quality lines are the lines 3 (mod 4): count their invalid bytes (what SeqIO rejects, lib/seq_routines.py:50)
        for (uint32_t m = bad; m; m &= m - 1) {
            const int b = __ffs((int)m) - 1;
            nbad += ((k + __popc(bits & ((1u << b) - 1u))) & 3) == 3;
        }
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            if (k + 1 < n_lines) line_off[k + 1] = pos + (uint64_t)b + 1;   // a final newline at n - 1 is line_off[n_lines] = n
            ++k;
        }
    }
    nbad = wave_sum_u32(nbad);
    if (lane == 0 && nbad) atomicAdd((unsigned long long*)&errors[3], (unsigned long long)nbad);
}

// one wave per chunk again: the k-th newline of the text (0-based) at byte p sets line_off[k + 1] = p + 1
__global__ __launch_bounds__(256) void nl_fill_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                      const uint64_t* __restrict__ prefix, uint64_t n_chunks,
                                                      uint64_t* __restrict__ line_off, uint64_t n_lines,
                                                      uint64_t* __restrict__ errors, const uint32_t* __restrict__ only_over) {
    const int lane = threadIdx.x & 63;
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) { line_off[0] = 0; line_off[n_lines] = n; }
    if (chunk >= n_chunks) return;
    // only_over: behind nl_place_kernel, for the chunks whose list of odd vectors overflowed (the others are placed)
    if (only_over && only_over[chunk] <= (uint32_t)kNlEntries) return;
    uint64_t base = uniform_u64(prefix[chunk]);
    uint32_t nbad = 0;
    constexpr int kIt = (int)(kNlChunk / kSpan);
    uint4 vv[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {   // all loads of the chunk in flight before the first use
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        vv[it] = load_vec_clamped(text, pos < n ? pos : 0, n);
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        uint32_t bits = 0, bad = 0;
        // nearly every vector holds only '!'..'~' bytes: one cheap test, the exact masks only where it fires
        const uint4& v = vv[it];
        if (pos < n && vec_is_odd(v)) {
            const uint64_t rem = n - pos;
            bits = vec_nl_bits(v, rem < 16 ? (int)rem : 16);
            bad = vec_badq_bits(v, rem < 16 ? (int)rem : 16);
        }
        const uint32_t cnt = __popc(bits);
        uint32_t inc = cnt;   // inclusive wave prefix of the lane counts
        if (__ballot(bits != 0) != 0) {   // most 1 KiB spans lie inside one line: no newline, nothing to scan
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
        }
        uint64_t k = base + (inc - cnt);   // number of the line the vector starts in
        // quality lines are the lines 3 (mod 4): count their invalid bytes (what SeqIO rejects, lib/seq_routines.py:50)
        if (bad) {
            if (!bits) nbad += (k & 3) == 3 ? __popc(bad) : 0;
            else
                for (uint32_t m = bad; m; m &= m - 1) {
                    const int b = __ffs((int)m) - 1;
                    nbad += ((k + __popc(bits & ((1u << b) - 1u))) & 3) == 3;
                }
        }
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            if (k + 1 < n_lines) line_off[k + 1] = pos + (uint64_t)b + 1;   // a final newline at n - 1 is line_off[n_lines] = n
            ++k;
        }
        base += (uint32_t)__shfl((int)inc, 63);
    }
    nbad = (uint32_t)wave_sum_u64(nbad);
    if (lane == 0 && nbad) atomicAdd((unsigned long long*)&errors[3], (unsigned long long)nbad);
}


// Pass 1 and the scan: counts, entries and the chunks' line numbers; fs.prefix[n_chunks] = newlines of the text,
// fs.prefix[n_chunks + 1] = chunks whose entry list overflowed, fs.prefix[n_chunks + 2] = white-space bytes of the text
// other than newlines.
hipError_t launch_nl_scan(const uint8_t* text, uint64_t n, const FastqScratch& fs, uint64_t* host_out, hipEvent_t counted,
                          hipStream_t s, const uint64_t* lut_dev, bool notes) {
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    hipError_t e = hipMemsetAsync(fs.prefix + n_chunks, 0, 24, s);
    if (e != hipSuccess || n_chunks == 0) return e;
    if (lut_dev)   // with span sums (fs.span_sum holds two words per 1 KiB span of every chunk)
        hipLaunchKernelGGL(nl_scan_kernel<true>, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, text, n, fs.counts,
                           fs.n_entries, fs.entries, n_chunks, fs.prefix + n_chunks + 1, fs.span_sum, lut_dev);
    else if (notes)
        hipLaunchKernelGGL(nl_scan_kernel<false>, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, text, n, fs.counts,
                           fs.n_entries, fs.entries, n_chunks, fs.prefix + n_chunks + 1, (uint64_t*)nullptr,
                           (const uint64_t*)nullptr);
    else
        hipLaunchKernelGGL((nl_scan_kernel<false, false>), dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, text, n,
                           fs.counts, fs.n_entries, fs.entries, n_chunks, fs.prefix + n_chunks + 1, (uint64_t*)nullptr,
                           (const uint64_t*)nullptr);
    const uint64_t nblocks = (n_chunks + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(nl_blocksum_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, fs.counts, n_chunks, fs.block_sums,
                       fs.prefix + n_chunks + 2);
    hipLaunchKernelGGL(nl_scan_top_kernel, dim3(1), dim3(256), 0, s, fs.block_sums, nblocks, fs.prefix + n_chunks, host_out,
                       text + n - 1);
    if (counted) {   // the host may go on from here while the chunks' line numbers are written
        e = hipEventRecord(counted, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(nl_scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, fs.counts, n_chunks,
                       fs.block_sums, fs.prefix);
    return hipGetLastError();
}

// line_off[0] = 0, line_off[k + 1] = position after the k-th newline, line_off[n_lines] = n; errors[3] += bytes of quality
// lines outside '!'..'~' (white space excepted).  `sparse`: from the entries of pass 1; else by a second pass over the text.
// n_over: chunks whose list of odd vectors overflowed.  None: everything from the notes.  A few (a run of short reads
// somewhere in a text of long ones): those chunks are read again, the rest comes from the notes.  Many: one pass over
// the whole text, no notes.
hipError_t launch_nl_place(const uint8_t* text, uint64_t n, uint64_t n_lines, uint64_t n_over, const FastqScratch& fs,
                           uint64_t* errors, hipStream_t s) {
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    const bool all = n_over * 8 >= n_chunks && n_over != 0;
    if (!all) {
        const uint64_t blocks = n_chunks ? (n_chunks + 8 * kPlaceIlp - 1) / (8 * kPlaceIlp) : 1;
        hipLaunchKernelGGL(nl_place_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fs.prefix, fs.n_entries, fs.entries,
                           n_chunks, fs.line_off, n, n_lines, errors);
    }
    if (n_over) {
        const uint64_t blocks = n_chunks ? (n_chunks + 3) / 4 : 1;
        hipLaunchKernelGGL(nl_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, text, n, fs.prefix, n_chunks, fs.line_off,
                           n_lines, errors, all ? (const uint32_t*)nullptr : fs.n_entries);
    }
    return hipGetLastError();
}

// The albacore header fields (the reference reads ch= at lib/seq_routines.py:51) need the line table only, not the sums: a
// kernel of its own in front of the line-mode tile kernel.  One thread per record streams its header through registers
// in aligned 16-byte granules (a granule that holds a byte of the text lies in that byte's page; the next one is in
// flight while one is looked at) and feeds every byte to two small automata, one per key: a key starts behind the '@'
// or behind a blank or tab and only its first occurrence counts, "ch=" is followed by decimal digits, "start_time=" by
// the 19 characters YYYY-MM-DD(T or blank)HH:MM:SS (the rules of the host packer, ppqc_fastq.cpp).  White space at the end of the line takes part in no match, so
// the stripped end need not be known.  0.02 ms for 200 000 records; walking the header through about 150 dependent byte
// loads took 0.16 ms, sixteen lanes per record over an LDS window 0.08 ms; beside the tile kernel on the auxiliary
// stream either one cost the tile kernel more than it took alone.  Tried on this version and slower for the whole path
// (tools/ab_fastq.sh, one box): four lanes per record, each looking for the keys that start in its quarter of the header
// (+26 us: three more loads in front of the first granule and four times the threads), and blocks of four granules with
// one block in flight (+10 us).  fastq_finalize overrides both fields for a malformed record.
// The kernel also keeps the WHITE-SPACE BALANCE of the text (errors[6], see fastq_finalize_kernel): it adds the count
// of the scan pass (*ws_seen: every blank, \t, \v, \f and \r of the text) and takes off the blanks, tabs, \v and \f of the
// header lines, which may hold them anywhere (the title is free text to SeqIO; a \r inside it is a bare CR line end, which
// the host packer refuses).  The keys end the search early, the count goes on to the end of the line by words.
__global__ __launch_bounds__(256) void fastq_header_kernel(const uint8_t* __restrict__ text, uint64_t n_records,
                                                           const uint64_t* __restrict__ lo, ReadArrays ra,
                                                           uint64_t read_base, uint8_t mux,
                                                           const uint64_t* __restrict__ ws_seen,
                                                           uint64_t* __restrict__ errors) {
    __shared__ unsigned int blanks_acc;
    if (threadIdx.x == 0) blanks_acc = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t blanks = 0;
    if (r < n_records) {
        const uint64_t g = read_base + r;
        const HeaderFields hf = scan_header<true, false>(text, lo[4 * r], lo[4 * r + 1]);
        blanks = hf.blanks;
        ra.channel[g] = hf.ch;
        ra.mux[g] = mux;
        ra.time[g] = hf.time;
    }
    blanks = wave_sum_u32(blanks);
    if ((threadIdx.x & 63) == 0 && blanks) atomicAdd(&blanks_acc, blanks);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t delta = (blockIdx.x == 0 ? *ws_seen : 0ull) - (uint64_t)blanks_acc;
        if (delta) atomicAdd((unsigned long long*)&errors[6], (unsigned long long)delta);
    }
}

// ---- line sums from the span sums of the scan pass ------------------------------------------------------------------------
// Per-lane sums of the bytes of span k (1 KiB at text + k * kSpan) that lie in [a, e), by one classifier.  Bytes outside
// are zeroed: a zero byte matches no base class and adds nothing to the byte sum, and its LUT entry (2^52) is taken off.
template <bool QUAL>
__device__ __forceinline__ void direct_span(const uint8_t* __restrict__ text, uint64_t n, uint64_t k, uint64_t a, uint64_t e,
                                            const uint64_t* lut, int lane, uint64_t& c0, uint64_t& c1, uint64_t& hi,
                                            uint64_t& lw) {
    const uint64_t pos = k * kSpan + (uint64_t)lane * kVec;
    const uint64_t b0 = a > pos ? a : pos, b1 = e < pos + kVec ? e : pos + kVec;
    if (b1 <= b0) return;
    const uint4 v = load_vec_clamped(text, pos, n);   // pos < e <= n
    const int x0 = (int)(b0 - pos), x1 = (int)(b1 - pos);
    uint32_t m[4];
    range_masks(x0, x1, m);
    const uint4 mv = make_uint4(v.x & m[0], v.y & m[1], v.z & m[2], v.w & m[3]);
    if (QUAL) {
        const Acc q = vec_sums_quals(mv, lut);
        const uint64_t err = q.err - ((uint64_t)(kVec - (x1 - x0)) << PPQC_ERR_SHIFT);
        c0 += q.qb; hi += err >> 32; lw += err & 0xffffffffull;
    } else {
        const Acc b = vec_sums_bases(mv);
        c0 += b.gcs; c1 += b.n;
    }
}

// One line [a, e) (its line end included, as in the line-mode tile kernel: fastq_finalize takes off what the stripped end
// added): the spans at its two ends from the text, the spans in between from their span sums -- or from the text where a
// span has none or the other classifier's.  Sequence line: rec_cnt = G/C/S count, rec_hi = N count; quality line:
// rec_cnt = byte sum, rec_hi / rec_lo = the halves of the error sum (S = hi * 2^32 + lo, as everywhere).
// In two steps, so that a wave has the loads of both its lines in flight before it looks at any of them (the kernel is a
// chain of memory latencies: line table -> end spans and span sums -> record; 215 us for 200 000 records when the lines
// were taken one after the other, load by load).
struct LineJob {
    uint64_t a, e, k0, k1;   // k0, k1: first and last span of the line
    uint4 v0, v1;            // the lane's vectors of those two spans
    uint64_t w0, w1;         // span sums of span k0 + 1 + lane
    bool in;
};
__device__ __forceinline__ void line_begin(LineJob& j, const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t e,
                                           const uint64_t* __restrict__ span_sum, int lane) {
    j.a = a; j.e = e; j.k0 = 0; j.k1 = 0; j.in = false; j.w0 = j.w1 = 0;
    j.v0 = j.v1 = make_uint4(0, 0, 0, 0);
    if (e <= a) return;   // wave-uniform
    j.k0 = a / kSpan; j.k1 = (e - 1) / kSpan;
    const uint64_t p0 = j.k0 * kSpan + (uint64_t)lane * kVec, p1 = j.k1 * kSpan + (uint64_t)lane * kVec;
    if (p0 + kVec > a && p0 < e) j.v0 = load_vec_clamped(text, p0, n);
    if (j.k1 != j.k0 && p1 < e) j.v1 = load_vec_clamped(text, p1, n);
    const uint64_t k = j.k0 + 1 + (uint64_t)lane;
    j.in = k < j.k1;
    if (j.in) {
        const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(span_sum + 2 * k);
        j.w0 = w.x; j.w1 = w.y;
    }
}

// the lane's part of span k of the line, from the vector it has loaded
template <bool QUAL>
__device__ __forceinline__ void edge_sum(const uint4& v, uint64_t k, uint64_t a, uint64_t e, const uint64_t* lut, int lane,
                                         uint64_t& c0, uint64_t& c1, uint64_t& hi, uint64_t& lw) {
    const uint64_t pos = k * kSpan + (uint64_t)lane * kVec;
    const uint64_t b0 = a > pos ? a : pos, b1 = e < pos + kVec ? e : pos + kVec;
    if (b1 <= b0) return;
    const int x0 = (int)(b0 - pos), x1 = (int)(b1 - pos);
    uint32_t m[4];
    range_masks(x0, x1, m);
    const uint4 mv = make_uint4(v.x & m[0], v.y & m[1], v.z & m[2], v.w & m[3]);
    if (QUAL) {
        const Acc q = vec_sums_quals(mv, lut);
        const uint64_t err = q.err - ((uint64_t)(kVec - (x1 - x0)) << PPQC_ERR_SHIFT);
        c0 += q.qb; hi += err >> 32; lw += err & 0xffffffffull;
    } else {
        const Acc b = vec_sums_bases(mv);
        c0 += b.gcs; c1 += b.n;
    }
}

template <bool QUAL>
__device__ __forceinline__ void line_finish(LineJob& j, const uint8_t* __restrict__ text, uint64_t n, uint64_t line,
                                            const uint64_t* __restrict__ span_sum, const uint64_t* lut, int lane,
                                            const ReadArrays& lra) {
    uint64_t c0 = 0, c1 = 0, hi = 0, lw = 0;
    if (j.e > j.a) {
        // The first span of the line uses the lanes from la on, the last one the lanes up to lb: when these do not meet
        // (every other line), ONE pass takes both, each lane with the vector of its span.
        const int la = (int)((j.a - j.k0 * kSpan) / kVec), lb = (int)((j.e - 1 - j.k1 * kSpan) / kVec);
        if (j.k1 != j.k0 && lb < la) {
            const bool first = lane >= la;
            const uint4 v = make_uint4(first ? j.v0.x : j.v1.x, first ? j.v0.y : j.v1.y, first ? j.v0.z : j.v1.z,
                                       first ? j.v0.w : j.v1.w);
            edge_sum<QUAL>(v, first ? j.k0 : j.k1, j.a, j.e, lut, lane, c0, c1, hi, lw);
        } else {
            edge_sum<QUAL>(j.v0, j.k0, j.a, j.e, lut, lane, c0, c1, hi, lw);
            if (j.k1 != j.k0) edge_sum<QUAL>(j.v1, j.k1, j.a, j.e, lut, lane, c0, c1, hi, lw);
        }
        for (uint64_t kb = j.k0 + 1; kb < j.k1; kb += 64) {   // wave-uniform; the first round's span sums are loaded
            if (kb != j.k0 + 1) {
                const uint64_t k = kb + (uint64_t)lane;
                j.in = k < j.k1;
                j.w0 = j.w1 = 0;
                if (j.in) {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(span_sum + 2 * k);
                    j.w0 = w.x; j.w1 = w.y;
                }
            }
            const bool ok = j.in && (j.w0 & kSpanHas) && (((j.w0 & kSpanQual) != 0) == QUAL);
            if (ok) {
                if (QUAL) { c0 += j.w0 & 0xFFFFFull; hi += j.w1 >> 32; lw += j.w1 & 0xffffffffull; }
                else { c0 += j.w0 & 0xFFFFull; c1 += (j.w0 >> 16) & 0xFFFFull; }
            }
            uint64_t bad = __ballot(j.in && !ok);
            while (bad) {
                const int b = __builtin_ctzll(bad);
                bad &= bad - 1;
                direct_span<QUAL>(text, n, kb + (uint64_t)b, j.a, j.e, lut, lane, c0, c1, hi, lw);
            }
        }
    }
    // u32 wave sums (seven instructions each, four times that for a u64).  A line is shorter than 2^32 bytes, so its
    // counts fit; per lane the byte sum stays below 2^35 and the halves of the error sum below 2^48 (2^16 span sums of
    // 2^18, 2^30 and 2^32 at most), which go in pieces of 24 bits whose wave sums fit.
    if (QUAL) {
        const uint64_t C = (uint64_t)wave_sum_u32((uint32_t)c0 & 0xFFFFFFu) + ((uint64_t)wave_sum_u32((uint32_t)(c0 >> 24)) << 24);
        const uint64_t H = (uint64_t)wave_sum_u32((uint32_t)hi & 0xFFFFFFu) + ((uint64_t)wave_sum_u32((uint32_t)(hi >> 24)) << 24);
        const uint64_t L = (uint64_t)wave_sum_u32((uint32_t)lw & 0xFFFFFFu) + ((uint64_t)wave_sum_u32((uint32_t)(lw >> 24)) << 24);
        if (lane == 0) { lra.rec_cnt[line] = C; lra.rec_hi[line] = H; lra.rec_lo[line] = L; }
    } else {
        const uint32_t G = wave_sum_u32((uint32_t)c0), N = wave_sum_u32((uint32_t)c1);
        if (lane == 0) { lra.rec_cnt[line] = G; lra.rec_hi[line] = N; lra.rec_lo[line] = 0; }
    }
}

// one wave per record: its sequence line and its quality line
__global__ __launch_bounds__(256) void line_sum_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_records,
                                                       const uint64_t* __restrict__ lo, const uint64_t* __restrict__ span_sum,
                                                       const uint64_t* __restrict__ lut_g, ReadArrays lra) {
    __shared__ uint64_t lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_records) return;
    // lines 4r + 1 .. 4r + 4 start at lo[4r + 1 ..]: one load, lanes 0..3
    const uint64_t mine = lo[4 * r + 1 + (uint64_t)(lane & 3)];
    const uint64_t a1 = readlane_u64(mine, 0), e1 = readlane_u64(mine, 1), a3 = readlane_u64(mine, 2), e3 = readlane_u64(mine, 3);
    LineJob s, q;
    line_begin(s, text, n, a1, e1, span_sum, lane);
    line_begin(q, text, n, a3, e3, span_sum, lane);
    line_finish<false>(s, text, n, 4 * r + 1, span_sum, lut, lane, lra);
    line_finish<true>(q, text, n, 4 * r + 3, span_sum, lut, lane, lra);
}

// record r of the text; returns the white-space bytes it has placed (see below)
template <bool WHOLE>
__device__ __forceinline__ uint32_t fastq_finalize_record(const uint8_t* __restrict__ text, uint64_t r,
                                                          const uint64_t* __restrict__ lo, const ReadArrays& lra,
                                                          const ReadArrays& ra, uint64_t read_base,
                                                          const Partial* __restrict__ head,
                                                          const Partial* __restrict__ tail, uint32_t tile_shift,
                                                          const uint64_t* __restrict__ lut, const QThr& thr,
                                                          uint64_t* __restrict__ errors, uint8_t mux, uint32_t n_channels,
                                                          const Fq1Index& ix, uint32_t* key_out) {
    const uint64_t g = read_base + r;
    uint64_t a[4], e[4];
    uint64_t strip_qb = 0, strip_err = 0;   // what the stripped end of the quality line added to its raw sums
    // The last kTailBytes bytes of all four lines and the first of lines 1 and 3 are requested before any is looked at
    // (a line end is one or two bytes; walking back byte by byte, line by line, was a chain of eight dependent loads).
    constexpr int kTailBytes = 3;
    uint8_t tb[4][kTailBytes];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = lo[4 * r + k];
        e[k] = lo[4 * r + k + 1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < kTailBytes; ++j) tb[k][j] = e[k] >= a[k] + (uint64_t)j + 1 ? text[e[k] - 1 - j] : (uint8_t)'x';
    const uint8_t first0 = e[0] > a[0] ? text[a[0]] : (uint8_t)0, first2 = e[2] > a[2] ? text[a[2]] : (uint8_t)0;
    // The white-space balance (errors[6]): the scan pass has counted every white-space byte of the text, line ends
    // included as far as they are not '\n'; what is stripped here and what a title may hold is taken off again (the
    // title line's own blanks by fastq_header_kernel).  Whatever remains lies INSIDE a sequence or quality line, or is
    // a '\r' inside a title: SeqIO refuses blanks and tabs in a sequence and white space as a quality, and a bare CR
    // is a line end to it (lib/seq_routines.py:49-50; ppqc_fastq_scan refuses the same).
    uint32_t placed = 0;
    auto place = [&](int k, uint8_t ch) { placed += ch != '\n' && (k != 0 || ch == '\r'); };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int j = 0;
        while (j < kTailBytes && is_space(tb[k][j])) {   // '\n', "\r\n", trailing blanks: what rstrip() removes
            if (k == 3) { strip_qb += tb[k][j]; strip_err += lut[tb[k][j]]; }
            place(k, tb[k][j]);
            ++j;
        }
        e[k] -= (uint64_t)j;
        if (j == kTailBytes)
            while (e[k] > a[k] && is_space(text[e[k] - 1])) {
                --e[k];
                if (k == 3) { strip_qb += text[e[k]]; strip_err += lut[text[e[k]]]; }
                place(k, text[e[k]]);
            }
    }
    const uint64_t L = e[1] - a[1];
    bool ok = e[0] > a[0] && first0 == '@' && e[2] > a[2] && first2 == '+' && L == e[3] - a[3] && L < 0xFFFFFFFFull;
    if (ok && e[2] - a[2] > 1) {   // an optional repeated title must match (SeqIO: "captions differ")
        ok = e[2] - a[2] == e[0] - a[0];
        for (uint64_t i = 1; ok && i < e[2] - a[2]; ++i) {
            const uint8_t ch = text[a[2] + i];
            ok = ch == text[a[0] + i];
            placed += is_space(ch) && ch != '\r';   // as in the title
        }
    }

    uint64_t gcs = 0, nn = 0, qb = 0, hi = 0, lw = 0;
    if (!ok) {
        atomicAdd((unsigned long long*)&errors[1], 1ull);
        atomicMin((unsigned long long*)&errors[2], (unsigned long long)r);
    } else if (L != 0) {
        const uint64_t s1 = lo[4 * r + 1], f1 = lo[4 * r + 2], s3 = lo[4 * r + 3], f3 = lo[4 * r + 4];
        SegSums b, q;
        if (WHOLE) {   // line_sum_kernel has left whole lines
            b.gcs = lra.rec_cnt[4 * r + 1]; b.nn = lra.rec_hi[4 * r + 1];
            q.qb = lra.rec_cnt[4 * r + 3]; q.hi = lra.rec_hi[4 * r + 3]; q.lo = lra.rec_lo[4 * r + 3];
        } else {
            b = merged_sums(s1, f1, 4 * r + 1, lra, head, tail, tile_shift);
            q = merged_sums(s3, f3, 4 * r + 3, lra, head, tail, tile_shift);
        }
        gcs = b.gcs; nn = b.nn;
        qb = q.qb - strip_qb;
        // S = hi * 2^32 + lo with halves that are sums; subtract the stripped bytes from the combined value
        const unsigned __int128 S = (((unsigned __int128)q.hi) << 32) + q.lo - strip_err;
        hi = (uint64_t)(S >> 32); lw = (uint64_t)S & 0xffffffffull;
    }
    const uint64_t Lr = ok ? L : 0;
    uint8_t flags = 0;
    double mq = 0.0;
    if (Lr != 0) flags = quality_flags(Lr, hi, lw, thr, &mq);
    else flags = 0.0 < thr.fail_q ? 1 : 0;
    ra.len[g] = (uint32_t)Lr; ra.gc[g] = (uint32_t)gcs; ra.nn[g] = (uint32_t)nn;
    ra.qsum[g] = (int64_t)qb - 33 * (int64_t)Lr;
    ra.err_hi[g] = hi; ra.err_lo[g] = lw;
    ra.mean_q[g] = mq;
    ra.flags[g] = flags;
    uint32_t ch = 0;
    int64_t t = INT64_MIN;
    if (!ok) {   // fastq_header_kernel has parsed the header of what is no record
        ra.channel[g] = 0;
        ra.time[g] = INT64_MIN;
    } else {
        ch = ra.channel[g]; t = ra.time[g];
    }
    // the sort key and (ix.on) hour bin and cell of the read, as fq1_finalize_kernel leaves them: the sort + N50 chain may
    // start behind this kernel and the finalize of the run needs no extremes kernel over these reads' hours
    const uint32_t key = (ch >= 1 && ch <= n_channels) ? (uint32_t)Lr : 0u;
    ra.lens32[g] = key;
    *key_out = key;
    if (ix.on) ra.hour[g] = index_cell(ch, (uint32_t)Lr, t, mux, ix.tp, ix.cells);
    return placed;
}

template <bool WHOLE>
__global__ __launch_bounds__(256) void fastq_finalize_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                             uint64_t n_records, const uint64_t* __restrict__ lo,
                                                             ReadArrays lra, ReadArrays ra, uint64_t read_base,
                                                             const Partial* __restrict__ head,
                                                             const Partial* __restrict__ tail, uint32_t tile_shift,
                                                             const uint64_t* __restrict__ lut, QThr thr,
                                                             uint8_t mux, uint64_t* __restrict__ errors,
                                                             uint32_t n_channels, uint64_t* __restrict__ gate, Fq1Index ix) {
    __shared__ unsigned int placed_acc, s_max;
    if (threadIdx.x == 0) { placed_acc = 0; s_max = 0; }
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t placed = 0, key = 0;
    if (r < n_records)
        placed = fastq_finalize_record<WHOLE>(text, r, lo, lra, ra, read_base, head, tail, tile_shift, lut, thr, errors, mux,
                                              n_channels, ix, &key);
    if (key) atomicMax(&s_max, key);
    placed = wave_sum_u32(placed);
    if ((threadIdx.x & 63) == 0 && placed) atomicAdd(&placed_acc, placed);
    __syncthreads();   // one atomic per block: the balance is one word
    if (threadIdx.x == 0) {
        if (placed_acc) atomicAdd((unsigned long long*)&errors[6], 0ull - (unsigned long long)placed_acc);
        // the sort gate, by atomicMax as in fq1_finalize_kernel: the largest word is the largest length of the texts
        if (s_max) atomicMax((unsigned long long*)&gate[blockIdx.x & (kGateWords - 1)], (unsigned long long)s_max);
    }
}

hipError_t launch_fastq_header(const uint8_t* text, uint64_t n_records, const uint64_t* line_off, ReadArrays ra,
                               uint64_t read_base, uint8_t mux, const uint64_t* ws_seen, uint64_t* errors, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    hipLaunchKernelGGL(fastq_header_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n_records,
                       line_off, ra, read_base, mux, ws_seen, errors);
    return hipGetLastError();
}

hipError_t launch_fastq_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const uint64_t* line_off,
                                 ReadArrays line_ra, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, uint32_t tile_shift, const uint64_t* lut, const QThr& thr,
                                 uint8_t mux, uint64_t* errors, uint32_t n_channels, uint64_t* gate, const Fq1Index& ix,
                                 hipStream_t s, bool whole_lines) {
    if (n_records == 0) return hipSuccess;
    if (whole_lines)
        hipLaunchKernelGGL(fastq_finalize_kernel<true>, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n,
                           n_records, line_off, line_ra, ra, read_base, head, tail, tile_shift, lut, thr, mux,
                           errors, n_channels, gate, ix);
    else
        hipLaunchKernelGGL(fastq_finalize_kernel<false>, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n,
                           n_records, line_off, line_ra, ra, read_base, head, tail, tile_shift, lut, thr, mux,
                           errors, n_channels, gate, ix);
    return hipGetLastError();
}

hipError_t launch_line_sums(const uint8_t* text, uint64_t n, uint64_t n_records, const uint64_t* line_off,
                            const uint64_t* span_sum, const uint64_t* lut_dev, ReadArrays line_ra, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    hipLaunchKernelGGL(line_sum_kernel, dim3((unsigned)((n_records + 3) / 4)), dim3(256), 0, s, text, n, n_records, line_off,
                       span_sum, lut_dev, line_ra);
    return hipGetLastError();
}

}  // namespace ppqc
