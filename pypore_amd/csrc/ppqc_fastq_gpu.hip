// FASTQ text -> per-read statistics without a host tokenizer (gfx950).
//
// Replaces, for text that is already in (or on its way to) HBM, the record parsing the reference leaves to Biopython
// (SeqIO.parse(handle, "fastq"), lib/seq_routines.py:50; record layout lib/fast5_reader.py:57, lib/fastqparser.py:7-16).
// Strict 4-line records only; anything else is counted as malformed and the run fails loudly (the host packer
// ppqc_fastq_scan/pack accepts blank lines between records and reports the record number).
//
// The text is never compacted into separate base / quality streams.  Instead:
//   1. nl_index builds the table of line starts in one streaming pass over the text (chained scan, decoupled look-back);
//   2. the per-read tile kernel runs in "line mode": bases = quals = text, offsets = line starts, so every LINE gets
//      the raw sums of both classifiers (G/C/S and N counts, quality byte sum, fixed-point error sum);
//   3. fastq_finalize (one thread per record) takes the base sums of line 2 and the quality sums of line 4, removes
//      the contribution of the stripped line end from the quality sums, validates the record and writes the same
//      per-read arrays finalize_reads writes.
// Traffic: two reads of the text (2.05 bytes per base each) and no write of the sequence bytes.
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

// ---- the table of line starts in ONE pass over the text -----------------------------------------------------------------
// A chained scan: one wave per chunk of kNlChunk bytes counts its newlines from registers, publishes the count, learns
// the number of newlines in front of its chunk from what the chunks before it have published, and then places its own
// line starts -- the text is read once, where counting, scanning and filling took two passes (0.60 + 0.62 ms for 3.97 GB)
// and four launches in between.  A plain decoupled look-back is no good here: some 4000 waves are resident at once, all
// of them publish their counts at about the same time and every one would walk back over all the others, 64 at a time
// (measured: 58 ms).  So the walk has two levels.  Chunks form groups of 64; the wave of a group's last chunk publishes
// the group's total as soon as its 63 predecessors have published their counts, and the group's inclusive prefix once it
// knows what lies in front of the group.  A wave then needs two steps: the counts of the chunks of its own group in front
// of it (one load per lane), and a look-back over GROUP words -- the 4000 resident waves are 64 groups, one load per lane.
//   word = flag << 62 | value.  chunk words: 1 = the chunk's count.  group words: 1 = the group's total, 2 = the inclusive
//   prefix up to and including the group.  One 64-bit word, so a reader never sees a flag without its value -- and since
//   nothing else is handed over, the accesses are RELAXED device-scope atomics: a release / acquire pair at device scope
//   writes back and invalidates the XCD's whole L2 on this part (measured: 0.75 ms for an 80 MB text).
// Chunks are handed out by a ticket counter (one ticket per block of 16 waves), in the order blocks START: a wave only ever waits for chunks whose waves are
// already running, and those wait for nothing behind them -- every wave's walk ends.
// status = [chunk words: n_chunks | ticket | newlines of the whole text | group words: ceil(n_chunks / 64)]
constexpr uint64_t kStatusCount = 1ull << 62, kStatusPrefix = 2ull << 62, kStatusValue = (1ull << 62) - 1;

constexpr int kNlWaves = 16;   // waves per block: one ticket per block (a ticket per wave: 484 000 atomics on one word, 6 ms)

__global__ __launch_bounds__(64 * kNlWaves) void nl_index_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t* status,
                                                       uint64_t n_chunks, uint64_t* __restrict__ line_off,
                                                       uint64_t line_cap, uint64_t* __restrict__ errors) {
    const int lane = threadIdx.x & 63;
    __shared__ uint64_t s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd((unsigned long long*)&status[n_chunks], 1ull);
    __syncthreads();   // the only block-level barrier: nothing below may wait for a wave of this block to reach one
    const uint64_t chunk = uniform_u64(s_ticket * kNlWaves + (threadIdx.x >> 6));
    if (chunk >= n_chunks) return;
    if (chunk == 0 && lane == 0) line_off[0] = 0;
    constexpr int kIt = (int)(kNlChunk / kSpan);
    uint4 vv[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {   // all loads of the chunk in flight before the first use
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        vv[it] = load_vec_clamped(text, pos < n ? pos : 0, n);
    }
    uint32_t nlb[kIt], cnt = 0;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        const uint4& v = vv[it];
        nlb[it] = 0;
        // nearly every vector holds only '!'..'~' bytes: one cheap test, the exact masks only where it fires
        if (pos < n && (odd_byte_mask(v.x) | odd_byte_mask(v.y) | odd_byte_mask(v.z) | odd_byte_mask(v.w))) {
            const uint64_t rem = n - pos;
            nlb[it] = vec_nl_bits(v, rem < 16 ? (int)rem : 16);
        }
        cnt += __popc(nlb[it]);
    }
    const uint64_t mine = wave_sum_u32(cnt);   // <= 8192
    uint64_t* __restrict__ gs = status + n_chunks + 2;
    const uint64_t g = chunk >> 6;
    const int m = (int)(chunk & 63);
    if (lane == 0) __hip_atomic_store(&status[chunk], kStatusCount | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t within = 0;   // newlines of the chunks of this group in front of this one
    if (m) {
        for (;;) {
            uint64_t st = kStatusCount;
            if (lane < m) st = __hip_atomic_load(&status[(g << 6) + (uint64_t)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot((st >> 62) == 0)) { __builtin_amdgcn_s_sleep(1); continue; }
            within = wave_sum_u64(lane < m ? (st & kStatusValue) : 0);
            break;
        }
    }
    const bool closes = m == 63;   // this wave speaks for its group
    if (closes && lane == 0)
        __hip_atomic_store(&gs[g], kStatusCount | (within + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t front = 0;    // newlines in front of this group
    if (g) {
        int64_t pos = (int64_t)g - 1;
        for (;;) {   // 64 groups at a time, nearest in lane 0
            const int64_t idx = pos - lane;
            uint64_t st = kStatusPrefix;   // in front of the text: prefix 0
            if (idx >= 0) st = __hip_atomic_load(&gs[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t isP = __ballot((st >> 62) == 2), isX = __ballot((st >> 62) == 0);
            const int firstP = isP ? __builtin_ctzll(isP) : 64;
            const uint64_t upto = firstP >= 63 ? ~0ull : ((2ull << firstP) - 1ull);
            if (isX & upto) { __builtin_amdgcn_s_sleep(1); continue; }   // a group in between has not closed yet
            front += wave_sum_u64(lane <= firstP ? (st & kStatusValue) : 0);
            if (firstP != 64) break;
            pos -= 64;
        }
    }
    const uint64_t excl = front + within;
    if (closes && lane == 0)
        __hip_atomic_store(&gs[g], kStatusPrefix | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (chunk == n_chunks - 1 && lane == 0) status[n_chunks + 1] = excl + mine;
    // place the line starts: the k-th newline of the text (0-based) at byte p sets line_off[k + 1] = p + 1
    uint64_t base = excl;
    uint32_t nbad = 0;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const uint64_t pos = chunk * kNlChunk + (uint64_t)it * kSpan + (uint64_t)lane * kVec;
        uint32_t bits = nlb[it], bad = 0;
        const uint4& v = vv[it];
        if (pos < n && (odd_byte_mask(v.x) | odd_byte_mask(v.y) | odd_byte_mask(v.z) | odd_byte_mask(v.w))) {
            const uint64_t rem = n - pos;
            bad = vec_badq_bits(v, rem < 16 ? (int)rem : 16);
        }
        const uint32_t c = __popc(bits);
        uint32_t inc = c;   // inclusive wave prefix of the lane counts
        if (__ballot(bits != 0) != 0) {   // most 1 KiB spans lie inside one line: no newline, nothing to scan
            inc = wave_scan_incl_u32(c);
        }
        uint64_t k = base + (inc - c);   // number of the line the vector starts in
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
            if (k + 1 <= line_cap) line_off[k + 1] = pos + (uint64_t)b + 1;   // beyond the table's capacity: counted only
            ++k;
        }
        base += (uint32_t)__shfl((int)inc, 63);
    }
    nbad = wave_sum_u32(nbad);
    if (lane == 0 && nbad) atomicAdd((unsigned long long*)&errors[3], (unsigned long long)nbad);
}

uint64_t nl_status_words(uint64_t n_chunks) { return n_chunks + 2 + (n_chunks + 63) / 64; }

// status: nl_status_words(n_chunks) u64.  Leaves line_off[0 .. min(lines, line_cap)] and the number of newlines in
// status[n_chunks + 1]; the caller sets line_off[n_lines] = n for a text without a final newline.
hipError_t launch_nl_index(const uint8_t* text, uint64_t n, uint64_t* status, uint64_t* line_off, uint64_t line_cap,
                           uint64_t* errors, hipStream_t s) {
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    hipError_t e = hipMemsetAsync(status, 0, nl_status_words(n_chunks) * 8, s);
    if (e != hipSuccess || n_chunks == 0) return e;
    hipLaunchKernelGGL(nl_index_kernel, dim3((unsigned)((n_chunks + kNlWaves - 1) / kNlWaves)), dim3(64 * kNlWaves), 0, s, text, n, status, n_chunks,
                       line_off, line_cap, errors);
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
    // albacore header fields (the reference reads ch= at lib/seq_routines.py:51)
    uint32_t ch = 0;
    int64_t tm = INT64_MIN;
    if (ok) {
        uint64_t p = find_field(text, a[0] + 1, e[0], "ch=", 3);
        for (; p < e[0] && text[p] >= '0' && text[p] <= '9'; ++p) ch = ch * 10 + (uint32_t)(text[p] - '0');
        p = find_field(text, a[0] + 1, e[0], "start_time=", 11);
        if (p < e[0]) tm = parse_iso8601(text, p, e[0]);
    }
    ra.channel[g] = ch;
    ra.mux[g] = mux;
    ra.time[g] = tm;
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
