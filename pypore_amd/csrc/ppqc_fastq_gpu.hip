// FASTQ text -> per-read statistics without a host tokenizer (gfx950).
//
// Replaces, for text that is already in (or on its way to) HBM, the record parsing the reference leaves to Biopython
// (SeqIO.parse(handle, "fastq"), lib/seq_routines.py:50; record layout lib/fast5_reader.py:57, lib/fastqparser.py:7-16).
// Strict 4-line records only; anything else is counted as malformed and the run fails loudly (the host packer
// ppqc_fastq_scan/pack accepts blank lines between records and reports the record number).
//
// The text is never compacted into separate base / quality streams.  Instead:
//   1. the table of line starts: one streaming pass over the text that counts newlines and notes the few vectors that hold
//      one, a scan, and a placement from those notes (a second pass over the text only when lines are short);
//   2. the per-read tile kernel runs in "line mode": bases = quals = text, offsets = line starts, so every LINE gets
//      the raw sums of both classifiers (G/C/S and N counts, quality byte sum, fixed-point error sum);
//   3. fastq_finalize (one thread per record) takes the base sums of line 2 and the quality sums of line 4, removes
//      the contribution of the stripped line end from the quality sums, validates the record and writes the same
//      per-read arrays finalize_reads writes.
// Traffic: two reads of the text (2.05 bytes per base each; three when lines are short) and no write of the sequence bytes.
#include "ppqc_internal.hpp"

namespace ppqc {

// high bit of each byte set where the byte equals '\n' (exact zero-byte detector on v ^ 0x0A..)
__device__ __forceinline__ uint32_t nl_mask(uint32_t v) {
    const uint32_t x = v ^ 0x0A0A0A0Au;
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

// high bit of each byte set where a QUALITY byte is invalid: outside '!'..'~' and not ASCII white space (line ends and
// trailing blanks are stripped, not judged)
__device__ __forceinline__ uint32_t badq_mask(uint32_t x) {
    const uint32_t H = 0x80808080u, hi = x & H, lo7 = x & 0x7F7F7F7Fu;
    const uint32_t ge9 = (lo7 + 0x77777777u) & H, ge14 = (lo7 + 0x72727272u) & H, ge32 = (lo7 + 0x60606060u) & H;
    const uint32_t ge33 = (lo7 + 0x5F5F5F5Fu) & H, ge127 = (lo7 + 0x01010101u) & H;
    const uint32_t invalid = hi | (~ge33 & H) | ge127;
    const uint32_t ws = ((ge9 & ~ge14) | (ge32 & ~ge33)) & ~hi;
    return invalid & ~ws;
}
__device__ __forceinline__ uint32_t vec_badq_bits(const uint4& v, int valid) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = badq_mask(w[k]);
        bits |= (((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u)) << (4 * k);
    }
    if (valid < 16) bits &= valid <= 0 ? 0u : ((1u << valid) - 1u);
    return bits;
}

// nonzero iff some byte of the word lies outside '!'..'~' (newlines, other white space and invalid quality bytes)
__device__ __forceinline__ uint32_t odd_byte_mask(uint32_t x) {
    const uint32_t H = 0x80808080u, lo7 = x & 0x7F7F7F7Fu;
    return ((x | (lo7 + 0x01010101u)) & H) | (~(lo7 + 0x5F5F5F5Fu) & H);
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

// entry = vector index in the chunk (bits 32..41) | newline bits (0..15) | invalid-quality bits (16..31)
__global__ __launch_bounds__(256) void nl_scan_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ n_entries,
                                                      uint64_t* __restrict__ entries, uint64_t n_chunks,
                                                      uint64_t* __restrict__ overflow) {
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
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        const uint4& v = vv[it];
        uint32_t bits = 0, bad = 0;
        // nearly every vector holds only '!'..'~' bytes: one cheap test, the exact masks only where it fires
        if (pos < n && (odd_byte_mask(v.x) | odd_byte_mask(v.y) | odd_byte_mask(v.z) | odd_byte_mask(v.w))) {
            const uint64_t rem = n - pos;
            bits = vec_nl_bits(v, rem < 16 ? (int)rem : 16);
            bad = vec_badq_bits(v, rem < 16 ? (int)rem : 16);
        }
        cnt += __popc(bits);
        const uint64_t odd = __ballot((bits | bad) != 0);
        if (odd) {
            const uint32_t at = ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(odd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)odd, 0u));
            if ((bits | bad) != 0 && at < (uint32_t)kNlEntries)
                entries[chunk * kNlEntries + at] = ((uint64_t)(it * 64 + lane) << 32) | ((uint64_t)bad << 16) | bits;
            ne += (uint32_t)__popcll(odd);
        }
    }
    cnt = wave_sum_u32(cnt);
    if (lane == 0) {
        counts[chunk] = cnt;
        n_entries[chunk] = ne;
        if (ne > (uint32_t)kNlEntries) atomicAdd((unsigned long long*)overflow, 1ull);
    }
}

constexpr int kScanBlock = 2048;   // entries per block of the two-level scan
uint64_t nl_scan_block_words(uint64_t n_chunks) { return (n_chunks + kScanBlock - 1) / kScanBlock + 2; }

__global__ __launch_bounds__(256) void nl_blocksum_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                          uint64_t* __restrict__ bsum) {
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    uint64_t t = 0;
    for (int k = 0; k < kScanBlock / 256; ++k) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) t += counts[i];
    }
    t = wave_sum_u64(t);
    if ((threadIdx.x & 63) == 0) atomicAdd(&acc, (unsigned long long)t);
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = acc;
}

// single block: exclusive scan of the block sums in place; grand total -> *total
__global__ __launch_bounds__(256) void nl_scan_top_kernel(uint64_t* __restrict__ bsum, uint64_t nblocks,
                                                          uint64_t* __restrict__ total) {
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
    for (int k = 0; k < 8; ++k) { v[k] = b0 + k < n ? counts[b0 + k] : 0; loc += v[k]; }
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


// half a wave per chunk: lane j of the half owns the chunk's j-th odd vector
__global__ __launch_bounds__(256) void nl_place_kernel(const uint64_t* __restrict__ prefix, const uint32_t* __restrict__ n_entries,
                                                       const uint64_t* __restrict__ entries, uint64_t n_chunks,
                                                       uint64_t* __restrict__ line_off, uint64_t n, uint64_t n_lines,
                                                       uint64_t* __restrict__ errors) {
    const int lane = threadIdx.x & 63, j = lane & 31;
    const uint64_t chunk = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (uint64_t)(lane >> 5);
    if (blockIdx.x == 0 && threadIdx.x == 0) { line_off[0] = 0; line_off[n_lines] = n; }
    const bool live = chunk < n_chunks;
    const uint32_t ne = live ? n_entries[chunk] : 0;
    const uint64_t e = (live && (uint32_t)j < ne) ? entries[chunk * kNlEntries + j] : 0;
    uint32_t bits = (uint32_t)e & 0xFFFFu;
    const uint32_t bad = (uint32_t)(e >> 16) & 0xFFFFu;
    const uint32_t c = __popc(bits);
    // exclusive prefix of the newline counts inside each half: the full-wave scan, minus what the lower half added
    const uint32_t inc = wave_scan_incl_u32(c);
    const uint32_t low = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
    uint64_t k = (live ? prefix[chunk] : 0) + (inc - c) - (lane >= 32 ? low : 0u);   // the line the vector starts in
    const uint64_t pos = chunk * kNlChunk + (uint64_t)(uint32_t)(e >> 32) * kVec;
    uint32_t nbad = 0;
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
    nbad = wave_sum_u32(nbad);
    if (lane == 0 && nbad) atomicAdd((unsigned long long*)&errors[3], (unsigned long long)nbad);
}

// one wave per chunk again: the k-th newline of the text (0-based) at byte p sets line_off[k + 1] = p + 1
__global__ __launch_bounds__(256) void nl_fill_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                      const uint64_t* __restrict__ prefix, uint64_t n_chunks,
                                                      uint64_t* __restrict__ line_off, uint64_t n_lines,
                                                      uint64_t* __restrict__ errors) {
    const int lane = threadIdx.x & 63;
    const uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) { line_off[0] = 0; line_off[n_lines] = n; }
    if (chunk >= n_chunks) return;
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
        if (pos < n && (odd_byte_mask(v.x) | odd_byte_mask(v.y) | odd_byte_mask(v.z) | odd_byte_mask(v.w))) {
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
// fs.prefix[n_chunks + 1] = chunks whose entry list overflowed.
hipError_t launch_nl_scan(const uint8_t* text, uint64_t n, const FastqScratch& fs, hipStream_t s) {
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    hipError_t e = hipMemsetAsync(fs.prefix + n_chunks, 0, 16, s);
    if (e != hipSuccess || n_chunks == 0) return e;
    hipLaunchKernelGGL(nl_scan_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, text, n, fs.counts, fs.n_entries,
                       fs.entries, n_chunks, fs.prefix + n_chunks + 1);
    const uint64_t nblocks = (n_chunks + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(nl_blocksum_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, fs.counts, n_chunks, fs.block_sums);
    hipLaunchKernelGGL(nl_scan_top_kernel, dim3(1), dim3(256), 0, s, fs.block_sums, nblocks, fs.prefix + n_chunks);
    hipLaunchKernelGGL(nl_scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, fs.counts, n_chunks,
                       fs.block_sums, fs.prefix);
    return hipGetLastError();
}

// line_off[0] = 0, line_off[k + 1] = position after the k-th newline, line_off[n_lines] = n; errors[3] += bytes of quality
// lines outside '!'..'~' (white space excepted).  `sparse`: from the entries of pass 1; else by a second pass over the text.
hipError_t launch_nl_place(const uint8_t* text, uint64_t n, uint64_t n_lines, bool sparse, const FastqScratch& fs,
                           uint64_t* errors, hipStream_t s) {
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    if (sparse) {
        const uint64_t blocks = n_chunks ? (n_chunks + 7) / 8 : 1;
        hipLaunchKernelGGL(nl_place_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fs.prefix, fs.n_entries, fs.entries,
                           n_chunks, fs.line_off, n, n_lines, errors);
    } else {
        const uint64_t blocks = n_chunks ? (n_chunks + 3) / 4 : 1;
        hipLaunchKernelGGL(nl_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, text, n, fs.prefix, n_chunks, fs.line_off,
                           n_lines, errors);
    }
    return hipGetLastError();
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// " key=" at the start of the header or after a blank: returns the position after the key, or e if absent
__device__ inline uint64_t find_field(const uint8_t* __restrict__ t, uint64_t a, uint64_t e, const char* key, int klen) {
    for (uint64_t i = a; i + klen <= e; ++i) {
        if (i != a && t[i - 1] != ' ' && t[i - 1] != '\t') continue;
        bool ok = true;
        for (int k = 0; k < klen && ok; ++k) ok = t[i + k] == (uint8_t)key[k];
        if (ok) return i + klen;
    }
    return e;
}

__device__ inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

// "YYYY-MM-DDTHH:MM:SS" at t[p..e): seconds since the epoch (UTC), INT64_MIN if it does not parse
__device__ inline int64_t parse_iso8601(const uint8_t* __restrict__ t, uint64_t p, uint64_t e) {
    if (e - p < 19) return INT64_MIN;
    int64_t f[6];
    const int beg[6] = {0, 5, 8, 11, 14, 17}, end[6] = {4, 7, 10, 13, 16, 19};
    for (int k = 0; k < 6; ++k) {
        int64_t x = 0;
        for (int i = beg[k]; i < end[k]; ++i) {
            const uint8_t c = t[p + i];
            if (c < '0' || c > '9') return INT64_MIN;
            x = x * 10 + (c - '0');
        }
        f[k] = x;
    }
    if (t[p + 4] != '-' || t[p + 7] != '-' || (t[p + 10] != 'T' && t[p + 10] != ' ') || t[p + 13] != ':' || t[p + 16] != ':')
        return INT64_MIN;
    if (f[1] < 1 || f[1] > 12 || f[2] < 1 || f[2] > 31) return INT64_MIN;
    return days_from_civil(f[0], (unsigned)f[1], (unsigned)f[2]) * 86400 + f[3] * 3600 + f[4] * 60 + f[5];
}

// The albacore header fields (the reference reads ch= at lib/seq_routines.py:51) need the line table only, not the sums: a
// kernel of its own that runs beside the line-mode tile kernel on the auxiliary stream.  Sixteen lanes per record: the
// header goes to LDS in 16-byte granules (one load per lane, aligned, so a granule that holds a byte of the text lies in
// that byte's page), every lane looks at 16 positions for the end of the stripped line and for the two keys, and two
// lanes parse the digits out of LDS.  (One thread per record walking its header through about 150 dependent global
// loads took 0.16 ms for 200 000 records alone and 0.35 ms beside the tile kernel, which it slowed from 0.70 to
// 0.89 ms.)  A header of more than kHdrFast bytes is walked by one lane as before.  fastq_finalize overrides both fields
// for a malformed record.
constexpr int kHdrLanes = 16, kHdrBytes = 16 * kHdrLanes, kHdrFast = kHdrBytes - 16;

struct HeaderFields { uint32_t ch; int64_t tm; };

__device__ inline HeaderFields header_fields_serial(const uint8_t* __restrict__ text, uint64_t a, uint64_t e) {
    while (e > a && is_space(text[e - 1])) --e;
    HeaderFields h{0, INT64_MIN};
    if (e > a) {
        uint64_t p = find_field(text, a + 1, e, "ch=", 3);
        for (; p < e && text[p] >= '0' && text[p] <= '9'; ++p) h.ch = h.ch * 10 + (uint32_t)(text[p] - '0');
        p = find_field(text, a + 1, e, "start_time=", 11);
        if (p < e) h.tm = parse_iso8601(text, p, e);
    }
    return h;
}

__device__ __forceinline__ int group_min(int v) {   // over the 16 lanes of a record
#pragma unroll
    for (int o = 1; o < kHdrLanes; o <<= 1) {
        const int u = __shfl_xor(v, o, kHdrLanes);
        v = u < v ? u : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void fastq_header_kernel(const uint8_t* __restrict__ text, uint64_t n_records,
                                                           const uint64_t* __restrict__ lo, ReadArrays ra,
                                                           uint64_t read_base, uint8_t mux) {
    __shared__ __align__(16) uint8_t hb[256 / kHdrLanes][kHdrBytes + 16];
    const int grp = threadIdx.x / kHdrLanes, j = threadIdx.x % kHdrLanes;
    const uint64_t r = (uint64_t)blockIdx.x * (256 / kHdrLanes) + grp;
    const bool live = r < n_records;
    const uint64_t g = read_base + r;
    uint64_t a = 0, e0 = 0;
    if (live) { a = lo[4 * r]; e0 = lo[4 * r + 1]; }
    const uint64_t len = e0 - a;
    const bool fast = live && len <= (uint64_t)kHdrFast;
    uint8_t* b = &hb[grp][0];
    // granule j of the 16-byte aligned window that starts at or below text + a; sh = where the header starts in it
    const uintptr_t base = (uintptr_t)(text + a);
    const int sh = (int)(base & 15u);
    if (fast) {
        const uintptr_t ga = (base & ~(uintptr_t)15) + 16u * (unsigned)j;
        if (ga < (uintptr_t)(text + e0)) *reinterpret_cast<uint4*>(b + 16 * j) = *reinterpret_cast<const uint4*>(ga);
    }
    __syncthreads();
    if (!live) return;
    if (!fast) {
        if (j == 0) {
            const HeaderFields h = header_fields_serial(text, a, e0);
            ra.channel[g] = h.ch; ra.mux[g] = mux; ra.time[g] = h.tm;
        }
        return;
    }
    const uint8_t* hdr = b + sh;   // hdr[k] = text[a + k], k < len
    const int L = (int)len;
    // the stripped end: one past the last byte that is no white space
    int last = -1;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = 16 * j + t;
        if (k < L && !is_space(hdr[k])) last = k;
    }
    const int e = -group_min(-(last + 1));
    // first position of each key at the start of the header (k = 1, behind the '@') or behind a blank
    int p_ch = e, p_tm = e;
    for (int t = 0; t < 16; ++t) {
        const int k = 16 * j + t;
        if (k < 1 || k >= e) continue;
        const uint8_t c = hdr[k];
        if (c != 'c' && c != 's') continue;
        if (k != 1 && hdr[k - 1] != ' ' && hdr[k - 1] != '\t') continue;
        if (c == 'c') {
            if (p_ch == e && k + 3 <= e && hdr[k + 1] == 'h' && hdr[k + 2] == '=') p_ch = k + 3;
        } else if (p_tm == e && k + 11 <= e) {
            const char* key = "start_time=";
            bool ok = true;
            for (int q = 1; q < 11 && ok; ++q) ok = hdr[k + q] == (uint8_t)key[q];
            if (ok) p_tm = k + 11;
        }
    }
    p_ch = group_min(p_ch);
    p_tm = group_min(p_tm);
    if (j == 0) {
        uint32_t ch = 0;
        for (int p = p_ch; p < e && hdr[p] >= '0' && hdr[p] <= '9'; ++p) ch = ch * 10 + (uint32_t)(hdr[p] - '0');
        ra.channel[g] = ch;
        ra.mux[g] = mux;
    } else if (j == 1) {
        ra.time[g] = (e > 0 && p_tm < e) ? parse_iso8601(hdr, (uint64_t)p_tm, (uint64_t)e) : INT64_MIN;
    }
}

__global__ __launch_bounds__(256) void fastq_finalize_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                             uint64_t n_records, const uint64_t* __restrict__ lo,
                                                             ReadArrays lra, ReadArrays ra, uint64_t read_base,
                                                             const Partial* __restrict__ head,
                                                             const Partial* __restrict__ tail, uint32_t tile_shift,
                                                             const uint64_t* __restrict__ lut, uint64_t thr_fail,
                                                             double fail_q, uint8_t mux, uint64_t* __restrict__ errors) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const uint64_t g = read_base + r;
    uint64_t a[4], e[4];
    uint64_t strip_qb = 0, strip_err = 0;   // what the stripped end of the quality line added to its raw sums
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = lo[4 * r + k];
        e[k] = lo[4 * r + k + 1];
        while (e[k] > a[k] && is_space(text[e[k] - 1])) {   // '\n', "\r\n", trailing blanks: what rstrip() removes
            --e[k];
            if (k == 3) { strip_qb += text[e[k]]; strip_err += lut[text[e[k]]]; }
        }
    }
    const uint64_t L = e[1] - a[1];
    bool ok = e[0] > a[0] && text[a[0]] == '@' && e[2] > a[2] && text[a[2]] == '+' && L == e[3] - a[3] && L < 0xFFFFFFFFull;
    if (ok && e[2] - a[2] > 1) {   // an optional repeated title must match (SeqIO: "captions differ")
        ok = e[2] - a[2] == e[0] - a[0];
        for (uint64_t i = 1; ok && i < e[2] - a[2]; ++i) ok = text[a[2] + i] == text[a[0] + i];
    }
    uint64_t gcs = 0, nn = 0, qb = 0, hi = 0, lw = 0;
    if (!ok) {
        atomicAdd((unsigned long long*)&errors[1], 1ull);
        atomicMin((unsigned long long*)&errors[2], (unsigned long long)r);
    } else if (L != 0) {
        const uint64_t s1 = lo[4 * r + 1], f1 = lo[4 * r + 2], s3 = lo[4 * r + 3], f3 = lo[4 * r + 4];
        const SegSums b = merged_sums(s1, f1, 4 * r + 1, lra, head, tail, tile_shift);
        const SegSums q = merged_sums(s3, f3, 4 * r + 3, lra, head, tail, tile_shift);
        gcs = b.gcs; nn = b.nn;
        qb = q.qb - strip_qb;
        // S = hi * 2^32 + lo with halves that are sums; subtract the stripped bytes from the combined value
        const unsigned __int128 S = (((unsigned __int128)q.hi) << 32) + q.lo - strip_err;
        hi = (uint64_t)(S >> 32); lw = (uint64_t)S & 0xffffffffull;
    }
    const uint64_t Lr = ok ? L : 0;
    uint8_t flags = 0;
    double mq = 0.0;
    if (Lr != 0) {
        const double S = (double)hi * 4294967296.0 + (double)lw;
        mq = -10.0 * log10(S / ((double)Lr * 4503599627370496.0));
        const unsigned __int128 S128 = ((unsigned __int128)hi << 32) + lw;
        flags = S128 > (unsigned __int128)Lr * thr_fail ? 1 : 0;
    } else {
        flags = 0.0 < fail_q ? 1 : 0;
    }
    ra.len[g] = (uint32_t)Lr; ra.gc[g] = (uint32_t)gcs; ra.nn[g] = (uint32_t)nn;
    ra.qsum[g] = (int64_t)qb - 33 * (int64_t)Lr;
    ra.err_hi[g] = hi; ra.err_lo[g] = lw;
    ra.mean_q[g] = mq;
    ra.flags[g] = flags;
    if (!ok) {   // fastq_header_kernel has parsed the header of what is no record
        ra.channel[g] = 0;
        ra.time[g] = INT64_MIN;
    }
}

hipError_t launch_fastq_header(const uint8_t* text, uint64_t n_records, const uint64_t* line_off, ReadArrays ra,
                               uint64_t read_base, uint8_t mux, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    const uint64_t per_block = 256 / kHdrLanes;
    hipLaunchKernelGGL(fastq_header_kernel, dim3((unsigned)((n_records + per_block - 1) / per_block)), dim3(256), 0, s, text,
                       n_records, line_off, ra, read_base, mux);
    return hipGetLastError();
}

hipError_t launch_fastq_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const uint64_t* line_off,
                                 ReadArrays line_ra, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, uint32_t tile_shift, const uint64_t* lut, uint64_t thr_fail,
                                 double fail_q, uint8_t mux, uint64_t* errors, hipStream_t s) {
    if (n_records == 0) return hipSuccess;
    hipLaunchKernelGGL(fastq_finalize_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, s, text, n,
                       n_records, line_off, line_ra, ra, read_base, head, tail, tile_shift, lut, thr_fail, fail_q, mux,
                       errors);
    return hipGetLastError();
}

}  // namespace ppqc
