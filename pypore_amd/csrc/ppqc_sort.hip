// Hand-written LSD radix sort (8-bit digits, stable) and the exact N50 reduction over the sorted lengths (gfx950).
//
// The reference has no N50 (SURVEY.md 8a-11 vi defines it: sort lengths descending, first length whose running sum c
// has 2c >= total); the sort also orders reads by (channel, start time) for the hour-wrap quirk of
// lib/fast5_reader.py:117-126.  One wavefront owns one contiguous chunk of keys in all three phases of a pass, so
// stability needs no cross-wave ordering inside a chunk:
//   hist    : per-wave 256-bin digit histogram in LDS            -> counts[digit][wave]
//   scan    : one block per digit scans its row, then the scatter waves prefix the 256 digit totals themselves
//   scatter : per 64 keys, ballot-based match-any gives each lane its rank among equal digits (wave64: 64-bit masks)
#include "ppqc_internal.hpp"

namespace ppqc {

constexpr uint32_t kMaxSortWaves = 2048;
constexpr int kSortUnroll = 8;

static inline void sort_geometry(uint64_t n, uint32_t* nw, uint32_t* chunk) {
    uint64_t c = (n + kMaxSortWaves - 1) / kMaxSortWaves;
    const uint64_t cmin = 512;   // measured on MI355X, 10^6 keys: whole step 3.51 ms with 2048, 3.47 with 1024, 3.42 with 512
    if (c < cmin) c = cmin;
    c = (c + 63) & ~63ull;
    *chunk = (uint32_t)c;
    *nw = (uint32_t)((n + c - 1) / c);
}

// Counter words that suffice for EVERY key count up to n.  The number of waves is not monotone in the key count (2^21
// keys: 2048 chunks of 1024; 2.6 M keys: 1952 chunks of 1344), and callers size the buffer once for a capacity and then
// sort any smaller count with it, so this is the bound, not the geometry of n itself: a chunk has at least 64 keys and
// there are at most kMaxSortWaves chunks.
uint64_t sort_count_words(uint64_t n) {
    uint64_t nw = ((n ? n : 1) + 63) / 64;
    if (nw > kMaxSortWaves) nw = kMaxSortWaves;
    return 256ull * nw + 256;
}

template <typename K>
__global__ __launch_bounds__(256) void rs_hist_kernel(const K* __restrict__ keys, uint64_t n, int shift,
                                                      uint32_t* __restrict__ counts, uint32_t NW, uint32_t chunk,
                                                      const uint64_t* __restrict__ gate, int gate_words) {
    __shared__ uint32_t h[4][256];
    if (gate && shift && (gate_bound(gate, gate_words) >> shift) == 0) return;   // device-gated pass: no key has a bit at or above `shift`
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4 + wib;
    for (int d = lane; d < 256; d += 64) h[wib][d] = 0;
    __syncthreads();
    if (w < NW) {
        const uint64_t base = (uint64_t)w * chunk;
        const uint64_t end = base + chunk < n ? base + chunk : n;
        // eight independent loads in flight per lane: the chunk is walked by one wave, so without them every 64 keys
        // would pay a full memory latency
        for (uint64_t i0 = base + lane; i0 < end; i0 += 64 * kSortUnroll) {
            K k[kSortUnroll];
#pragma unroll
            for (int j = 0; j < kSortUnroll; ++j) {
                const uint64_t i = i0 + 64ull * j;
                k[j] = i < end ? keys[i] : (K)0;
            }
#pragma unroll
            for (int j = 0; j < kSortUnroll; ++j)
                if (i0 + 64ull * j < end) atomicAdd(&h[wib][(uint32_t)(k[j] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (w < NW)
        for (int d = lane; d < 256; d += 64) counts[(uint64_t)d * NW + w] = h[wib][d];
}

// Block d: exclusive scan of row d (NW entries) in place; row total -> counts[256*NW + d].
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ counts, uint32_t NW, int shift,
                                                      const uint64_t* __restrict__ gate, int gate_words) {
    __shared__ uint32_t part[256];
    if (gate && shift && (gate_bound(gate, gate_words) >> shift) == 0) return;
    const uint32_t d = blockIdx.x, t = threadIdx.x;
    uint32_t* row = counts + (uint64_t)d * NW;
    const uint32_t seg = (NW + 255) / 256;
    const uint32_t b = t * seg, e = b + seg < NW ? b + seg : NW;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += row[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
        counts[256ull * NW + d] = run;
    }
    __syncthreads();
    uint32_t run = part[t];
    for (uint32_t i = b; i < e; ++i) { const uint32_t v = row[i]; row[i] = run; run += v; }
}

template <typename K, bool HAS_VAL>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         K* __restrict__ okeys, uint32_t* __restrict__ ovals,
                                                         uint64_t n, int shift, const uint32_t* __restrict__ counts,
                                                         uint32_t NW, uint32_t chunk,
                                                         const uint64_t* __restrict__ gate, int gate_words) {
    __shared__ uint32_t base[4][256];
    if (gate && shift && (gate_bound(gate, gate_words) >> shift) == 0) return;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4 + wib;
    if (w >= NW) return;   // whole waves leave; no block-level barrier below
    // exclusive prefix of the 256 digit totals: lane owns digits 4*lane .. 4*lane+3
    const uint32_t* tot = counts + 256ull * NW;
    uint32_t t4[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { t4[j] = tot[lane * 4 + j]; s += t4[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    uint32_t run = inc - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = lane * 4 + j;
        base[wib][d] = run + counts[(uint64_t)d * NW + w];
        run += t4[j];
    }
    const uint64_t b0 = (uint64_t)w * chunk;
    const uint64_t end = b0 + chunk < n ? b0 + chunk : n;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint64_t q0 = b0; q0 < end; q0 += 64 * kSortUnroll) {
        K kk[kSortUnroll];
        uint32_t vv[kSortUnroll];
#pragma unroll
        for (int j = 0; j < kSortUnroll; ++j) {   // loads of the next 512 keys in flight together
            const uint64_t i = q0 + 64ull * j + lane;
            kk[j] = i < end ? keys[i] : (K)0;
            if (HAS_VAL) vv[j] = i < end ? vals[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kSortUnroll; ++j) {
            const uint64_t i = q0 + 64ull * j + lane;
            if (q0 + 64ull * j >= end) break;   // wave-uniform
            const bool valid = i < end;
            const K key = kk[j];
            const uint32_t d = (uint32_t)(key >> shift) & 255u;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bm = __ballot(valid && one);
                peers &= one ? bm : ~bm;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
            if (valid) {
                const uint32_t pos = base[wib][d] + rank;
                okeys[pos] = key;
                if (HAS_VAL) ovals[pos] = vv[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) base[wib][d] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename K, bool HAS_VAL>
static hipError_t radix_sort(const K* in, K* buf_a, K* buf_b, const uint32_t* vin, uint32_t* va, uint32_t* vb,
                             uint64_t n, uint32_t* counts, uint32_t byte_mask, hipStream_t s, const K** out,
                             const uint32_t** vout, const uint64_t* gate = nullptr, int gate_words = 1) {
    *out = in;
    if (vout) *vout = vin;
    if (n < 2) return hipSuccess;
    if (n >= (1ull << 32)) return hipErrorInvalidValue;
    uint32_t NW, chunk;
    sort_geometry(n, &NW, &chunk);
    if (256ull * NW + 256 > sort_count_words(n)) return hipErrorInvalidValue;   // the bound callers allocate by
    const unsigned blocks = (NW + 3) / 4;
    const K* src = in; K* dst = buf_a;
    const uint32_t* vsrc = vin; uint32_t* vdst = va;
    for (int byte = 0; byte < (int)sizeof(K); ++byte) {
        if (!((byte_mask >> byte) & 1u)) continue;
        const int shift = 8 * byte;
        hipLaunchKernelGGL((rs_hist_kernel<K>), dim3(blocks), dim3(256), 0, s, src, n, shift, counts, NW, chunk, gate, gate_words);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, s, counts, NW, shift, gate, gate_words);
        hipLaunchKernelGGL((rs_scatter_kernel<K, HAS_VAL>), dim3(blocks), dim3(256), 0, s, src, vsrc, dst, vdst, n,
                           shift, counts, NW, chunk, gate, gate_words);
        src = dst; vsrc = vdst;
        dst = dst == buf_a ? buf_b : buf_a;
        vdst = vdst == va ? vb : va;
    }
    *out = src;
    if (vout) *vout = vsrc;
    return hipGetLastError();
}

hipError_t sort_u32(const uint32_t* in, uint32_t* buf_a, uint32_t* buf_b, uint64_t n, uint32_t* counts,
                    uint32_t byte_mask, hipStream_t s, const uint32_t** out) {
    return radix_sort<uint32_t, false>(in, buf_a, buf_b, nullptr, nullptr, nullptr, n, counts, byte_mask & 0xFu, s, out,
                                       nullptr);
}
// All four byte passes are enqueued, and each decides on the device whether it runs: pass p > 0 is skipped when
// gate[0] (an upper bound of the keys, e.g. the max length left by the extremes kernel) has no bit at or above 8p.
// The passes that run are 0..top, so the buffer roles are static (in -> a -> b -> a -> b) and the sorted keys end in
// buf_a when top is even, else buf_b: launch_n50_gated makes the same choice from the same word.  No host round trip.
hipError_t sort_u32_gated(const uint32_t* in, uint32_t* buf_a, uint32_t* buf_b, uint64_t n, uint32_t* counts,
                          const uint64_t* gate, int gate_words, hipStream_t s) {
    const uint32_t* out = nullptr;
    return radix_sort<uint32_t, false>(in, buf_a, buf_b, nullptr, nullptr, nullptr, n, counts, 0xFu, s, &out, nullptr,
                                       gate, gate_words);
}
hipError_t sort_u64_pairs(uint64_t* keys, uint64_t* tmp_keys, uint32_t* vals, uint32_t* tmp_vals, uint64_t n,
                          uint32_t* counts, uint32_t byte_mask, hipStream_t s) {
    const uint64_t* ok = nullptr;
    const uint32_t* ov = nullptr;
    hipError_t e = radix_sort<uint64_t, true>(keys, tmp_keys, keys, vals, tmp_vals, vals, n, counts, byte_mask & 0xFFu,
                                              s, &ok, &ov);
    if (e != hipSuccess) return e;
    if (ok != keys) {   // odd number of passes: bring the result home
        e = hipMemcpyAsync(keys, ok, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(vals, ov, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
    }
    return e;
}

// ---- N50 over ascending-sorted lengths -----------------------------------------------------------------------------
// suffix(i) = sum_{j >= i} s[j];  N50 = s[i*] for the largest i* with 2 * suffix(i*) >= total.
constexpr int kN50Items = 8;
constexpr int kN50Block = 256 * kN50Items;
constexpr int kN50Groups = 8;   // key blocks per workgroup: one ticket (a device-scope atomic on ONE word, about 90 per us
                                // for the whole chip) per 16 384 keys instead of per 2048

uint64_t n50_block_words(uint64_t n) { return (n + kN50Block - 1) / kN50Block + 2; }   // [0] ticket counter, then the sums

// Which buffer holds the sorted keys after a gated sort (see sort_u32_gated); without a gate it is `a`.
__device__ __forceinline__ const uint32_t* gated_pick(const uint32_t* a, const uint32_t* b, const uint64_t* gate,
                                                      int gate_words, const uint32_t* in, uint64_t n) {
    if (!gate) return a;
    if (n < 2) return in;   // radix_sort enqueues nothing for fewer than two keys
    const uint64_t m = gate_bound(gate, gate_words);
    const int top = m < (1ull << 8) ? 0 : (m < (1ull << 16) ? 1 : (m < (1ull << 24) ? 2 : 3));
    return (top & 1) ? b : a;
}

// One kernel.  Every workgroup writes the sums of its kN50Groups blocks of 2048 keys and takes a ticket; the block that draws the last ticket has
// all sums behind its fence, scans them from the top (256 threads, a segment each), finds the block the N50 lies in and
// searches that block's keys.  ws[0] is the ticket counter (zero between launches: the last block puts it back),
// ws[1 + b] the sum of block b.  `host_out` (may be null): a pinned host word that receives the N50 as well, so that a
// caller who has everything else on the host already needs no copy behind this kernel.
__global__ __launch_bounds__(256) void n50_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb,
                                                  const uint64_t* __restrict__ gate, int gate_words,
                                                  const uint32_t* __restrict__ in, uint64_t n, uint64_t* ws,
                                                  uint32_t nblocks, uint64_t* __restrict__ out,
                                                  uint64_t* __restrict__ host_out) {
    const uint32_t* __restrict__ s = gated_pick(sa, sb, gate, gate_words, in, n);
    __shared__ uint64_t part[256];
    __shared__ unsigned long long acc;
    __shared__ uint64_t s_total, s_after;
    __shared__ uint32_t s_blk, s_last;
    const uint32_t t = threadIdx.x;
    for (uint32_t kb = blockIdx.x * kN50Groups; kb < (blockIdx.x + 1) * kN50Groups && kb < nblocks; ++kb) {
        if (t == 0) acc = 0;
        __syncthreads();
        const uint64_t b0 = (uint64_t)kb * kN50Block;
        uint32_t v[kN50Items];
#pragma unroll
        for (int k = 0; k < kN50Items; ++k) {
            const uint64_t i = b0 + (uint64_t)k * 256 + t;
            v[k] = i < n ? s[i] : 0u;
        }
        uint64_t sum = 0;
#pragma unroll
        for (int k = 0; k < kN50Items; ++k) sum += v[k];
        sum = wave_sum_u64(sum);
        if ((t & 63) == 0) atomicAdd(&acc, (unsigned long long)sum);
        __syncthreads();
        if (t == 0) __hip_atomic_store(&ws[1 + kb], (uint64_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if (t == 0) {
        __threadfence();
        const unsigned long long ticket = atomicAdd((unsigned long long*)&ws[0], 1ull);
        s_last = ticket == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const uint32_t seg = (nblocks + 255) / 256;
    const uint32_t b = t * seg < nblocks ? t * seg : nblocks, e = b + seg < nblocks ? b + seg : nblocks;
    uint64_t loc = 0;
    for (uint32_t i = b; i < e; ++i) loc += __hip_atomic_load(&ws[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part[t] = loc;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 255; i >= 0; --i) { const uint64_t x = part[i]; part[i] = run; run += x; }
        s_total = run;
        s_blk = 0;
        s_after = 0;
        out[1] = run;
        out[0] = 0;
        if (host_out) host_out[0] = 0;
        __hip_atomic_store(&ws[0], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    const uint64_t total = s_total;
    if (total == 0) return;
    {   // the block whose keys carry the running sum from the top across half the total
        uint64_t after = part[t];
        for (uint32_t i = e; i > b; --i) {
            const uint64_t x = __hip_atomic_load(&ws[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // block i - 1
            const uint64_t suf = after + x;
            if (2 * suf >= total && 2 * after < total) { s_blk = i - 1; s_after = after; }
            after = suf;
        }
    }
    __syncthreads();
    const uint64_t b0 = (uint64_t)s_blk * kN50Block + (uint64_t)t * kN50Items;   // thread owns 8 consecutive keys
    uint32_t v[kN50Items];
    loc = 0;
#pragma unroll
    for (int k = 0; k < kN50Items; ++k) { const uint64_t i = b0 + k; v[k] = i < n ? s[i] : 0; loc += v[k]; }
    const uint64_t blk_after = s_after;
    __syncthreads();
    part[t] = loc;
    __syncthreads();
    if (t == 0) {
        uint64_t run = blk_after;
        for (int i = 255; i >= 0; --i) { const uint64_t x = part[i]; part[i] = run; run += x; }
    }
    __syncthreads();
    uint64_t after = part[t];   // sum of everything after this thread's keys
#pragma unroll
    for (int k = kN50Items - 1; k >= 0; --k) {
        const uint64_t i = b0 + k;
        const uint64_t suf = after + v[k];
        // predicate true at i, false at i+1 (or i is the last key)
        if (i < n && 2 * suf >= total && (i + 1 == n || 2 * after < total)) {
            out[0] = v[k];
            if (host_out) host_out[0] = v[k];
        }
        after = suf;
    }
}

static hipError_t n50_launch(const uint32_t* sa, const uint32_t* sb, const uint64_t* gate, int gate_words,
                             const uint32_t* in, uint64_t n, uint64_t* block_sums, uint64_t* out, uint64_t* host_out,
                             hipStream_t s) {
    if (n == 0) {
        if (host_out) host_out[0] = 0;   // nothing in flight writes it
        return hipMemsetAsync(out, 0, 16, s);
    }
    const uint64_t nblocks = (n + kN50Block - 1) / kN50Block;
    hipLaunchKernelGGL(n50_kernel, dim3((unsigned)((nblocks + kN50Groups - 1) / kN50Groups)), dim3(256), 0, s, sa, sb, gate, gate_words, in, n, block_sums,
                       (uint32_t)nblocks, out, host_out);
    return hipGetLastError();
}
hipError_t launch_n50(const uint32_t* sorted, uint64_t n, uint64_t* block_sums, uint64_t* out, uint64_t* host_out,
                      hipStream_t s) {
    return n50_launch(sorted, nullptr, nullptr, 1, sorted, n, block_sums, out, host_out, s);
}
hipError_t launch_n50_gated(const uint32_t* in, const uint32_t* buf_a, const uint32_t* buf_b, const uint64_t* gate,
                            int gate_words, uint64_t n, uint64_t* block_sums, uint64_t* out, uint64_t* host_out,
                            hipStream_t s) {
    return n50_launch(buf_a, buf_b, gate, gate_words, in, n, block_sums, out, host_out, s);
}

// ---- N50 by weighted radix select (multi-GPU: only PPQC_N_SEL-word histograms cross xGMI) --------------------------
__global__ __launch_bounds__(256) void select_hist_kernel(const uint32_t* __restrict__ lens, uint64_t n, int shift,
                                                          int nbits, const uint64_t* __restrict__ state,
                                                          uint64_t* __restrict__ sel,
                                                          const uint64_t* __restrict__ gate, int gate_words) {
    __shared__ unsigned long long h[PPQC_N_SEL];
    if (gate && shift && (gate_bound(gate, gate_words) >> shift) == 0) return;   // no key has a bit in this round's digit or above
    for (int d = threadIdx.x; d < PPQC_N_SEL; d += blockDim.x) h[d] = 0;
    const uint32_t prefix = (uint32_t)state[0];
    const uint32_t mask_hi = shift + nbits >= 32 ? 0u : ~0u << (shift + nbits);
    const uint32_t want = prefix & mask_hi, dmask = (1u << nbits) - 1u;
    __syncthreads();
    // equal digits come in runs (most reads share their top bits): combine a run in registers, one LDS atomic per run
    int last_d = -1;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = lens[i];
        if (k == 0 || (k & mask_hi) != want) continue;
        const int d = (int)((k >> shift) & dmask);
        if (d != last_d) {
            if (last_d >= 0) atomicAdd(&h[last_d], acc);
            last_d = d;
            acc = 0;
        }
        acc += k;
    }
    if (last_d >= 0) atomicAdd(&h[last_d], acc);
    __syncthreads();
    for (int d = threadIdx.x; d < PPQC_N_SEL; d += blockDim.x)
        if (h[d]) atomicAdd((unsigned long long*)&sel[d], h[d]);
}

// One block.  pred(d) = 2 * (above + sum_{d' >= d} sel[d']) >= total is monotone; the chosen digit is the largest d
// with pred(d): the weight of strictly longer keys stays below half the total, with digit d included it reaches it.
// `total` == nullptr: the total weight is the sum of the round's histogram when no bits are chosen yet (`first`: every
// positive length falls into some digit of the top round), kept in state[2] for the later rounds.  That makes the
// whole select independent of the tables, so it can run as soon as the lengths are known.
__global__ __launch_bounds__(256) void select_pick_kernel(uint64_t* __restrict__ sel, const uint64_t* __restrict__ total,
                                                          uint64_t* __restrict__ state, int shift, int nbits, int last,
                                                          int first, uint64_t* __restrict__ out,
                                                          uint64_t* __restrict__ host_out,
                                                          const uint64_t* __restrict__ gate, int gate_words) {
    __shared__ uint64_t part[256];
    __shared__ uint64_t s_T;
    // gate: the rounds whose digits are zero in every key are skipped on the device (their histogram kernel returned at
    // once as well); the first round that runs sums its histogram for the total (state[3] says that it has been done)
    if (gate && shift && (gate_bound(gate, gate_words) >> shift) == 0) return;
    const uint32_t t = threadIdx.x;
    const int per = (1 << nbits) / 256;   // 8 or 4 digits per thread
    const uint64_t prefix = state[0], above = state[1];
    uint64_t w[8], loc = 0;
    for (int k = 0; k < per; ++k) { w[k] = sel[t * per + k]; loc += w[k]; }
    part[t] = loc;
    __syncthreads();
    if (t == 0) {
        uint64_t run = 0;
        for (int i = 255; i >= 0; --i) { const uint64_t x = part[i]; part[i] = run; run += x; }
        uint64_t T;
        if (total) T = total[0];
        else if (gate ? state[3] == 0 : first != 0) { T = run; state[2] = run; if (gate) state[3] = 1; }
        else T = state[2];
        s_T = T;
    }
    __syncthreads();
    const uint64_t T = s_T;
    uint64_t after = above + part[t];
    for (int k = per - 1; k >= 0; --k) {
        const uint64_t suf = after + w[k];
        if (T != 0 && 2 * suf >= T && 2 * after < T) {
            const uint64_t np = prefix | ((uint64_t)(t * per + k) << shift);
            state[0] = np;
            state[1] = after;
            if (last) {
                out[0] = np;
                if (host_out) host_out[0] = np;
            }
        }
        after = suf;
    }
    for (int d = t; d < PPQC_N_SEL; d += 256) sel[d] = 0;
    if (last && t == 0) {
        out[1] = T;
        if (T == 0) {
            out[0] = 0;
            if (host_out) host_out[0] = 0;
        }
    }
}

hipError_t launch_select_hist(const uint32_t* lens, uint64_t n, int shift, int nbits, const uint64_t* state,
                              uint64_t* sel, hipStream_t s, const uint64_t* gate, int gate_words) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(select_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lens, n, shift, nbits, state, sel, gate, gate_words);
    return hipGetLastError();
}

hipError_t launch_select_pick(uint64_t* sel, const uint64_t* total, uint64_t* state, int shift, int nbits, bool last,
                              bool first, uint64_t* out, uint64_t* host_out, hipStream_t s, const uint64_t* gate,
                              int gate_words) {
    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, s, sel, total, state, shift, nbits, last ? 1 : 0,
                       first ? 1 : 0, out, last ? host_out : nullptr, gate, gate_words);
    return hipGetLastError();
}

}  // namespace ppqc
