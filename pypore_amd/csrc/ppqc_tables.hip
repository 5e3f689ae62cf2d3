// Run-level tables from the per-read records (gfx950): LDS-privatised histograms and per-channel accumulators flushed
// with global atomics; per-(channel, mux, hour) cells by global atomics (too large for LDS at 3000 x 4 x 72).
//
// Replaces the aggregation of parsing_func (lib/fast5_reader.py:117-145 == lib/multi_read_fast5_reader.py:47-77), the
// sums of FastStats (lib/seq_routines.py:99-137) and the per-channel tables of result_plotting / ChannelPlotter_f
// (:378-394, :507-514, :673-680).  All cells are u64 so one RCCL dtype covers the merged buffer.
#include "ppqc_internal.hpp"

namespace ppqc {

// hour of day of (t - ref + tz): int(time.strftime('%H', time.localtime(t - ref)))  (lib/fast5_reader.py:118)
__device__ __forceinline__ int32_t hour_of_day(int64_t t, const TableParams& tp) {
    int64_t h = floordiv64(t - tp.ref_time_s + tp.tz_offset_s, (int64_t)tp.bin_seconds) % 24;
    return (int32_t)(h < 0 ? h + 24 : h);
}

// ---- pass 1: extremes, sort input, absolute hour bins --------------------------------------------------------------
// Per-channel min/max are privatised in LDS (u32 pairs, dynamic: 8 B per channel) when they fit, and flushed once per
// block; run extremes go through thread-local registers, one LDS atomic per thread, one global atomic per block.
// min_buf holds the bitwise complement of each minimum (signed: ~L = -L-1, unseen = ~INT64_MAX = INT64_MIN), so that a
// minimum is merged by MAX and [max_buf | min_buf] crosses xGMI in ONE all-reduce(MAX).
// BATCH = false: over the run's per-read arrays at finalize time (ppqc_finalize_extremes).
// BATCH = true: over ONE batch's CSR index when the batch is submitted, on the auxiliary stream beside the tile kernel:
// everything here depends on offsets, channel, start time and the fail flag only, none of it on the bases.  It
// accumulates into the buffers ppqc_reset cleared, writes lens32 / hour of the batch's reads (run index base + i), and
// the block that draws the last ticket (bound[1]) stores the run's largest key so far (max_buf[0]) into the pinned
// host word: the host then knows how many radix passes the N50 sort needs.
// `cells` (BATCH only): the (channel, mux, hour) cells of the batch's reads as well -- they, too, depend on the index
// alone (channel, mux, start time, length) -- one packed atomic per read, count above bit kCmhShift (cmh_unpack_kernel
// splits the cells; the host checks at finalize time that no field can have overflowed, cmh_packable).  These atomics
// execute at the memory side at about 20 per ns for the whole chip, whoever issues them: issued here they cost the
// per-read kernel beside them nothing, issued by tables_kernel they were two thirds of the step's tail at 4 * 10^6 reads.
template <bool BATCH>
__global__ __launch_bounds__(1024) void extremes_kernel(BatchView b, uint64_t base, ReadArrays ra, uint64_t n,
                                                       TableParams tp, uint64_t* max_buf, uint64_t* min_buf,
                                                       int64_t* tminmax, uint32_t nchL, int write_lens, uint64_t* bound,
                                                       uint64_t* __restrict__ bound_host, uint64_t* cells,
                                                       uint64_t* gate) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* s_cmax = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_cmin = s_cmax + nchL;
    __shared__ unsigned long long s_max, s_min;
    __shared__ long long s_tmin, s_tmax;
    if (threadIdx.x == 0) { s_max = 0; s_min = (unsigned long long)INT64_MAX; s_tmin = INT64_MAX; s_tmax = INT64_MIN; }
    for (uint32_t k = threadIdx.x; k < nchL; k += blockDim.x) { s_cmax[k] = 0; s_cmin[k] = 0xFFFFFFFFu; }
    __syncthreads();
    const uint32_t nch = tp.lay.n_channels;
    unsigned long long lmax = 0, lmin = (unsigned long long)INT64_MAX;
    long long tmin = INT64_MAX, tmax = INT64_MIN;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t ch, L;
        int64_t t;
        uint64_t g = i;
        const bool from_run = BATCH && b.offsets == nullptr;   // the reads of a FASTQ text: in the run's arrays already, with
                                                               // hours and cells (fq1_finalize_kernel)
        if (from_run) {
            g = base + i;
            ch = ra.channel[g]; L = ra.len[g]; t = ra.time[g];
        } else if (BATCH) {
            g = base + i;
            ch = b.channel[i];
            t = b.time[i];
            // the length finalize_reads_kernel will store: 0 for a malformed row and for a read flagged as failed
            uint64_t a = b.offsets[i], e = b.offsets[i + 1];
            if (e < a || e > b.n_bytes || e - a >= 0xFFFFFFFFull || (i == 0 && a != 0) || (i + 1 == n && e != b.n_bytes))
                a = e = 0;
            L = (b.fail_flag != nullptr && b.fail_flag[i] != 0) ? 0u : (uint32_t)(e - a);
        } else {
            ch = ra.channel[i];
            L = ra.len[i];
            t = ra.time[i];
        }
        const bool good = ch >= 1 && ch <= nch;
        if (write_lens && !from_run) ra.lens32[g] = good ? L : 0;   // else finalize_reads left it (and a sort may be reading it)
        int32_t hour = -1;
        if (good) {
            lmax = L > lmax ? L : lmax;
            if (L > 0 && L < lmin) lmin = L;
            if (nchL) {
                atomicMax(&s_cmax[ch - 1], L);
                atomicMin(&s_cmin[ch - 1], L);
            } else {
                atomicMax((unsigned long long*)&max_buf[ch], (unsigned long long)L);
                atomicMax((long long*)&min_buf[ch], ~(long long)L);
            }
            if (t != INT64_MIN) {   // INT64_MIN: the read has no start time (a FASTQ header without start_time=):
                                    // outside the time range, no hour bin, and no arithmetic that would overflow
                tmin = t < tmin ? t : tmin;
                tmax = t > tmax ? t : tmax;
                if (tp.time_mode == 0) {
                    const int64_t h = floordiv64(t - tp.ref_time_s + tp.tz_offset_s, (int64_t)tp.bin_seconds);
                    hour = (h >= 0 && h < (int64_t)tp.lay.n_time_bins) ? (int32_t)h : -1;
                }
            }
        }
        if (tp.time_mode == 0 && !from_run) ra.hour[g] = hour;
        uint32_t m = 0;
        if (BATCH && !from_run) {   // the batch's metadata into the run's arrays (finalize_reads leaves them to this kernel)
            m = b.mux[i];
            ra.channel[g] = ch; ra.mux[g] = (uint8_t)m; ra.time[g] = t;
        }
        if (BATCH && cells != nullptr && hour >= 0) {   // hour >= 0: channel and start time are good, time_mode is 0
            if (m >= 1 && m <= tp.lay.n_mux)
                atomicAdd((unsigned long long*)&cells[((uint64_t)(ch - 1) * tp.lay.n_mux + (m - 1)) * tp.lay.n_time_bins + (uint32_t)hour],
                          (1ull << kCmhShift) | (unsigned long long)L);
        }
    }
    if (lmax) atomicMax(&s_max, lmax);
    if (lmin != (unsigned long long)INT64_MAX) atomicMin(&s_min, lmin);
    if (tmin != INT64_MAX) { atomicMin(&s_tmin, tmin); atomicMax(&s_tmax, tmax); }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nchL; k += blockDim.x) {
        if (s_cmin[k] != 0xFFFFFFFFu) {   // channel seen by this block
            atomicMax((unsigned long long*)&max_buf[1 + k], (unsigned long long)s_cmax[k]);
            atomicMax((long long*)&min_buf[1 + k], ~(long long)s_cmin[k]);
        }
    }
    if (threadIdx.x == 0) {
        atomicMax((unsigned long long*)&max_buf[0], s_max);
        atomicMax((long long*)&min_buf[0], ~(long long)s_min);
        atomicMin((long long*)&tminmax[0], s_tmin);
        atomicMax((long long*)&tminmax[1], s_tmax);
        if (BATCH) {
            // the sort gate (words whose OR bounds every sort key of the run: finalize_reads_kernel raises it otherwise)
            // (not for the reads of a FASTQ text: their finalize kernel has raised it, and keeps the largest length in it)
            if (gate && s_max) atomicOr((unsigned long long*)&gate[blockIdx.x & (kGateWords - 1)], s_max);
            __threadfence();
            const unsigned long long ticket = atomicAdd((unsigned long long*)&bound[1], 1ull);
            if (ticket == gridDim.x - 1ull) {
                __threadfence();
                bound_host[0] = __hip_atomic_load(&max_buf[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&bound[1], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The (channel, mux, hour) cells take two device-scope atomics per read (reads += 1, bases += L).  When no cell can
// overflow either field -- fewer than 2^24 reads and fewer than 2^40 bases in the run, which the host knows -- both go into
// ONE atomic on the bases cell, the count above bit kCmhShift, and cmh_unpack_kernel splits the cells afterwards.  Exact
// by construction: a cell's count is at most the run's reads, its bases at most the run's bases.
bool cmh_packable(uint64_t n_reads, uint64_t n_bases) {
    return n_reads < (1ull << (64 - kCmhShift)) && n_bases < (1ull << kCmhShift);
}

__global__ __launch_bounds__(256) void cmh_unpack_kernel(uint64_t* sum_buf, TableLayout lay) {
    const uint64_t cells = (uint64_t)lay.n_channels * lay.n_mux * lay.n_time_bins;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = sum_buf[lay.off_cmh_bases + i];
        if (v) {
            sum_buf[lay.off_cmh_reads + i] = v >> kCmhShift;
            sum_buf[lay.off_cmh_bases + i] = v & ((1ull << kCmhShift) - 1);
        }
    }
}

// PPQC_COMPAT_MULTI_READ_SINGLE (include/pypore_qc.h): what channel_parser makes of a channel that holds exactly one
// read (lib/multi_read_fast5_reader.py:64-67: `if len(l) > 1` filter by mux, `else` take the read under EVERY mux).  A
// pass over the reads behind tables_kernel / cmh_unpack_kernel: a read whose channel counted one read is that channel's
// only contributor, so its thread rewrites the channel's cells with plain stores and tops up the hour totals (the sum
// over channels and muxes of lib/seq_routines.py:103-108) by what the normal rule did not give them.
//
// `job_count` (ppqc_compat_dist_count): the channel's read count over ALL ranks of a multi-GPU job.  A read whose channel
// counts one read in the whole job is on this rank alone, so the same stores apply, and the table SUM that follows adds
// the rewritten cells to the zeros of the other ranks.  nullptr: one rank, the local table is the job's.
__global__ __launch_bounds__(256) void compat_single_kernel(ReadArrays ra, uint64_t n, TableParams tp, uint64_t* sum_buf,
                                                            const uint64_t* __restrict__ job_count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t chn = ra.channel[i], nch = tp.lay.n_channels, nmux = tp.lay.n_mux, nb = tp.lay.n_time_bins;
    if (chn < 1 || chn > nch) return;
    const uint32_t c = chn - 1;
    if ((job_count ? job_count[c] : sum_buf[tp.lay.off_chan_reads + c]) != 1) return;
    const uint64_t L = ra.len[i], fail = ra.flags[i] & 1u;
    sum_buf[tp.lay.off_chan_fail + c] = (uint64_t)nmux * fail;          // `FailedReads += 1` once per mux (:69-70)
    const int32_t h = ra.hour[i];
    if (h < 0 || h >= (int32_t)nb) return;
    for (uint32_t k = 0; k < nmux; ++k) {
        const uint64_t cell = ((uint64_t)c * nmux + k) * nb + (uint32_t)h;
        sum_buf[tp.lay.off_cmh_reads + cell] = 1;
        sum_buf[tp.lay.off_cmh_bases + cell] = L;
    }
    const uint32_t m = ra.mux[i];
    const uint64_t extra = nmux - ((m >= 1 && m <= nmux) ? 1u : 0u);     // the normal rule counted it once, or not at all
    if (extra) {
        atomicAdd((unsigned long long*)&sum_buf[tp.lay.off_hour_reads + h], (unsigned long long)extra);
        atomicAdd((unsigned long long*)&sum_buf[tp.lay.off_hour_bases + h], (unsigned long long)(extra * L));
    }
}

// Reads per channel of this rank (ReadPerChannel = len(l), lib/multi_read_fast5_reader.py:50), before the tables exist:
// what the ranks of a job add up to apply the rule above (ppqc_compat_dist_count).  `out` is cleared by the launcher.
__global__ __launch_bounds__(256) void chan_count_kernel(const uint32_t* __restrict__ channel, uint64_t n, uint32_t nch,
                                                         uint64_t* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ch = channel[i];
        if (ch >= 1 && ch <= nch) atomicAdd((unsigned long long*)&out[ch - 1], 1ull);
    }
}

// ---- pass 2: all SUM tables ------------------------------------------------------------------------------------------
// Every block keeps private tables in LDS and leaves them, zeros included, in a slab of its own (plain coalesced stores);
// tables_reduce_kernel adds the slabs up.  (Flushing the private tables with global atomics cost one memory-side atomic
// per non-zero cell and block -- 2.4 * 10^6 of them at 3000 channels and 256 blocks, about 120 us at the chip's 20 per ns.)
// Dynamic LDS layout (u64 first for alignment):
//   u64 totals[16] | u64 hour_bases[nb] | u64 chan_bases[nchL] |
//   u32 len_log2[40 * kRep] | u32 len_lin[100 * kRep] | u32 q[8 * kRep] | u32 gc[20 * kRep] | u32 hour_reads[nb] |
//   u32 chan_reads[nchL] | u32 chan_fail[nchL]
// The four histograms have few, hot bins: kRep copies each (copy = lane mod kRep), so that the lanes of a wave that hit
// one bin do not all queue on one LDS word.
// nchL = n_channels when the per-channel accumulators fit in LDS, else 0 (then they go straight to global atomics).
// cell_mode: 0 the cells are there already (every batch left them at submit time), 1 one packed atomic per read, 2 two.
// EXT: the extremes of extremes_kernel as well, for a run whose reads came from FASTQ texts (their lengths are the last
// thing known, so an extremes kernel of their own would stand in the step's tail, in front of this kernel or -- fighting
// it for LDS -- beside it).  Behind the layout above: u32 chan_max[nchL] | u32 chan_min[nchL] | four u64 run words; the
// slab grows by [max_buf | min_buf | ~tmin | tmax] in the layout and encoding of the result buffer from max_buf on
// (minima complemented: everything merges by signed MAX), and tables_reduce_kernel merges them into that buffer.  The one
// thing this kernel needs of the extremes itself, the largest length, it finds in the sort gate (max_words words that
// the FASTQ finalize kernels raise by atomicMax).
constexpr uint32_t kRep = 8;
constexpr uint32_t kHistBins = PPQC_N_LEN_LOG2 + PPQC_N_LEN_LIN + PPQC_N_QHIST + PPQC_N_GCHIST;
constexpr unsigned kTabThreads = 512, kTabMaxBlocks = 512;

template <bool EXT>
__global__ __launch_bounds__(EXT ? 2 * kTabThreads : kTabThreads) void tables_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                                                      uint64_t* sum_buf, const uint64_t* max_buf,
                                                                                      int max_words, uint32_t nchL, int cell_mode,
                                                                                      uint64_t* __restrict__ slab, uint32_t slab_stride) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t nb = tp.lay.n_time_bins, nch = tp.lay.n_channels, nmux = tp.lay.n_mux;
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* s_hbases = s_tot + PPQC_N_TOTALS;
    unsigned long long* s_cbases = s_hbases + nb;
    uint32_t* s_log2 = reinterpret_cast<uint32_t*>(s_cbases + nchL);
    uint32_t* s_lin = s_log2 + PPQC_N_LEN_LOG2 * kRep;
    uint32_t* s_q = s_lin + PPQC_N_LEN_LIN * kRep;
    uint32_t* s_gc = s_q + PPQC_N_QHIST * kRep;
    uint32_t* s_hreads = s_gc + PPQC_N_GCHIST * kRep;
    uint32_t* s_creads = s_hreads + nb;
    uint32_t* s_cfail = s_creads + nchL;
    uint32_t* s_cmax = s_cfail + nchL;                      // EXT only (nchL is even or the words below are not used: see launch_tables)
    uint32_t* s_cmin = s_cmax + nchL;
    long long* s_run = reinterpret_cast<long long*>(   // lmax, ~lmin, ~tmin, tmax (at the next 8-byte boundary)
        smem + ((reinterpret_cast<unsigned char*>(s_cmin + nchL) - smem + 7) & ~(size_t)7));
    const uint32_t n64 = PPQC_N_TOTALS + nb + nchL;
    const uint32_t n32 = kHistBins * kRep + nb + 2 * nchL;
    for (uint32_t k = threadIdx.x; k < n64; k += blockDim.x) s_tot[k] = 0;
    for (uint32_t k = threadIdx.x; k < n32; k += blockDim.x) s_log2[k] = 0;
    if (EXT) {
        for (uint32_t k = threadIdx.x; k < nchL; k += blockDim.x) { s_cmax[k] = 0; s_cmin[k] = 0xFFFFFFFFu; }
        if (threadIdx.x < 4) s_run[threadIdx.x] = threadIdx.x == 0 ? 0 : INT64_MIN;
    }
    __syncthreads();

    unsigned long long* g = reinterpret_cast<unsigned long long*>(sum_buf);
    // the run's largest length: max_buf[0] behind an extremes kernel, or the largest of the sort gate's words when every
    // read of the run came from a FASTQ text (their finalize kernels raise each word by atomicMax)
    uint64_t run_max = 0;
    for (int k = 0; k < max_words; ++k) run_max = max_buf[k] > run_max ? max_buf[k] : run_max;
    const uint64_t lin_w = run_max / PPQC_N_LEN_LIN + 1;
    const bool chan_lds = nchL != 0;
    const uint32_t rep = threadIdx.x & (kRep - 1);

    uint64_t tot[PPQC_TOT_NO_TIME + 1] = {};   // thread-local run totals, added to LDS once at the end
    uint32_t lmax = 0, lmin = 0xFFFFFFFFu;     // EXT: the thread's extremes over reads with a good channel
    long long tmin = INT64_MAX, tmax = INT64_MIN;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        tot[PPQC_TOT_N_READS] += 1;
        const uint32_t chn = ra.channel[i];
        if (chn < 1 || chn > nch) { tot[PPQC_TOT_BAD_CHANNEL] += 1; continue; }
        const uint32_t c = chn - 1;
        const uint32_t L = ra.len[i];
        const uint32_t gc = ra.gc[i];
        const uint8_t fl = ra.flags[i];
        const uint32_t fail = fl & 1u;
        const uint32_t m = ra.mux[i];
        const int32_t h = ra.hour[i];
        tot[PPQC_TOT_N_BASES] += L;
        tot[PPQC_TOT_N_FAIL] += fail;
        tot[PPQC_TOT_N_GC] += gc;
        tot[PPQC_TOT_N_N] += ra.nn[i];
        tot[PPQC_TOT_QSUM] += (uint64_t)ra.qsum[i];
        tot[PPQC_TOT_ZERO_LEN] += (L == 0);
        const int64_t t = ra.time[i];
        tot[PPQC_TOT_NO_TIME] += (t == INT64_MIN);
        if (EXT) {
            lmax = L > lmax ? L : lmax;
            if (L > 0 && L < lmin) lmin = L;
            atomicMax(&s_cmax[c], L);
            atomicMin(&s_cmin[c], L);
            if (t != INT64_MIN) { tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax; }
        }
        atomicAdd(&s_log2[(L == 0 ? 0 : 32 - __clz(L)) * kRep + rep], 1u);
        atomicAdd(&s_lin[(uint32_t)(L / lin_w) * kRep + rep], 1u);
        const uint32_t qb = fl >> 4;   // 1 + the bin of 6+j <= mean_q < 7+j, left by whoever computed mean_q (quality_flags)
        if (qb) atomicAdd(&s_q[(qb - 1) * kRep + rep], 1u);
        if (gc > 0 && L != 0) {   // (a record may claim bases of an empty read: no bin, no division)
            uint32_t bin = (uint32_t)((20ull * gc) / L);
            atomicAdd(&s_gc[(bin > PPQC_N_GCHIST - 1 ? PPQC_N_GCHIST - 1 : bin) * kRep + rep], 1u);
        }
        if (chan_lds) { atomicAdd(&s_creads[c], 1u); atomicAdd(&s_cbases[c], (unsigned long long)L); }
        else { atomicAdd(&g[tp.lay.off_chan_reads + c], 1ull); atomicAdd(&g[tp.lay.off_chan_bases + c], (unsigned long long)L); }
        if (m < 1 || m > nmux) { tot[PPQC_TOT_BAD_MUX] += 1; continue; }
        if (fail) {
            if (chan_lds) atomicAdd(&s_cfail[c], 1u);
            else atomicAdd(&g[tp.lay.off_chan_fail + c], 1ull);
        }
        if (h < 0 || h >= (int32_t)nb) { tot[PPQC_TOT_TIME_OOB] += 1; continue; }
        tot[PPQC_TOT_N_BINNED] += 1;
        atomicAdd(&s_hreads[h], 1u);
        atomicAdd(&s_hbases[h], (unsigned long long)L);
        if (cell_mode == 0) continue;
        const uint64_t cell = ((uint64_t)c * nmux + (m - 1)) * nb + (uint32_t)h;
        if (cell_mode == 1) {   // one atomic per read: count in the bits above kCmhShift (cmh_unpack_kernel splits the cells)
            atomicAdd(&g[tp.lay.off_cmh_bases + cell], (1ull << kCmhShift) | (unsigned long long)L);
        } else {
            atomicAdd(&g[tp.lay.off_cmh_reads + cell], 1ull);
            atomicAdd(&g[tp.lay.off_cmh_bases + cell], (unsigned long long)L);
        }
    }
#pragma unroll
    for (int k = 0; k <= PPQC_TOT_NO_TIME; ++k)
        if (tot[k]) atomicAdd(&s_tot[k], (unsigned long long)tot[k]);
    if (EXT) {
        if (lmax) atomicMax(&s_run[0], (long long)lmax);
        if (lmin != 0xFFFFFFFFu) atomicMax(&s_run[1], ~(long long)lmin);
        if (tmin != INT64_MAX) { atomicMax(&s_run[2], ~tmin); atomicMax(&s_run[3], tmax); }
    }
    __syncthreads();
    // the block's tables into its slab, laid out like sum_buf[0, off_cmh_reads)
    uint64_t* out = slab + (uint64_t)blockIdx.x * slab_stride;
    if (EXT) {   // ... and behind them its extremes, laid out like the result buffer from max_buf on
        long long* ext = reinterpret_cast<long long*>(out + tp.lay.off_cmh_reads);
        const uint32_t nch1 = nch + 1;
        if (threadIdx.x == 0) { ext[0] = s_run[0]; ext[nch1] = s_run[1]; ext[2 * nch1] = s_run[2]; ext[2 * nch1 + 1] = s_run[3]; }
        for (uint32_t k = threadIdx.x; k < nch; k += blockDim.x) {
            const uint32_t mn = s_cmin[k];
            ext[1 + k] = mn != 0xFFFFFFFFu ? (long long)s_cmax[k] : INT64_MIN;   // unseen channel: nothing to merge
            ext[nch1 + 1 + k] = mn != 0xFFFFFFFFu ? ~(long long)mn : INT64_MIN;
        }
    }
    for (uint32_t k = threadIdx.x; k < PPQC_N_TOTALS; k += blockDim.x) out[tp.lay.off_totals + k] = s_tot[k];
    for (uint32_t k = threadIdx.x; k < kHistBins; k += blockDim.x) {   // the four histograms lie back to back in both layouts
        uint32_t v = 0;
#pragma unroll
        for (uint32_t r = 0; r < kRep; ++r) v += s_log2[k * kRep + r];
        out[tp.lay.off_len_log2 + k] = v;
    }
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) {
        out[tp.lay.off_hour_reads + k] = s_hreads[k];
        out[tp.lay.off_hour_bases + k] = s_hbases[k];
    }
    for (uint32_t k = threadIdx.x; k < nch; k += blockDim.x) {
        out[tp.lay.off_chan_reads + k] = chan_lds ? s_creads[k] : 0u;
        out[tp.lay.off_chan_bases + k] = chan_lds ? s_cbases[k] : 0ull;
        out[tp.lay.off_chan_fail + k] = chan_lds ? s_cfail[k] : 0u;
    }
}

// sum_buf[k] += sum over the blocks' slabs, k < n_small: 16 words per block of 256 threads, sixteen threads per word with
// four loads in flight each (the kernel is nothing but the latency of its loads).  Words n_small .. n_small + n_ext of a
// slab (tables_kernel<true>): merged by signed MAX into ext_dst[k - n_small] -- the result buffer from max_buf on --
// except word tmin_at, which holds a complemented minimum for a word that is kept plain.
__global__ __launch_bounds__(256) void tables_reduce_kernel(const uint64_t* __restrict__ slab, uint32_t n_slabs,
                                                            uint32_t stride, uint32_t n_small, uint64_t* sum_buf,
                                                            uint32_t n_ext, long long* ext_dst, uint32_t tmin_at) {
    __shared__ uint64_t part[16][16];
    const uint32_t w = threadIdx.x & 15, q = threadIdx.x >> 4;
    const uint32_t k = blockIdx.x * 16 + w;
    const bool is_ext = k >= n_small;   // block-uniform when n_small is a multiple of 16; correct either way
    uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (k < n_small) {
        uint32_t b = q;
        for (; b + 48 < n_slabs; b += 64) {
            v0 += slab[(uint64_t)b * stride + k];
            v1 += slab[(uint64_t)(b + 16) * stride + k];
            v2 += slab[(uint64_t)(b + 32) * stride + k];
            v3 += slab[(uint64_t)(b + 48) * stride + k];
        }
        for (; b < n_slabs; b += 16) v0 += slab[(uint64_t)b * stride + k];
        v0 += v1 + v2 + v3;
    } else if (k < n_small + n_ext) {
        long long m = INT64_MIN;
        for (uint32_t b = q; b < n_slabs; b += 16) {
            const long long x = (long long)slab[(uint64_t)b * stride + k];
            m = x > m ? x : m;
        }
        v0 = (uint64_t)m;
    }
    part[q][w] = v0;
    __syncthreads();
    if (q == 0 && k < n_small) {
        uint64_t t = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += part[j][w];
        sum_buf[k] += t;
    } else if (q == 0 && is_ext && k < n_small + n_ext) {
        long long m = INT64_MIN;
#pragma unroll
        for (int j = 0; j < 16; ++j) m = (long long)part[j][w] > m ? (long long)part[j][w] : m;
        const uint32_t e = k - n_small;
        if (m != INT64_MIN) {
            if (e == tmin_at) { if (~m < ext_dst[e]) ext_dst[e] = ~m; }
            else if (m > ext_dst[e]) ext_dst[e] = m;
        }
    }
}

// ---- time_mode 1: reference hour-of-day + wrap quirk (lib/fast5_reader.py:117-126) ---------------------------------
// Sequential rule: for index e > 1 in start-time order, hour[e] += 24 if hour[e] < hour[e-1] (already fixed value).
// Once a read is bumped (>= 24) every later read compares below it, so bumped(e) = OR_{2<=j<=e} (h[j] < h[j-1]) on the
// raw hours: all reads from the first descent at index >= 2 onwards get +24.
__global__ __launch_bounds__(256) void mode1_keys_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                         const int64_t* tminmax, uint64_t* keys, uint32_t* vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ch = ra.channel[i];
    const bool good = ch >= 1 && ch <= tp.lay.n_channels;
    // reads with a bad channel sort behind every real channel (sentinel channel n_channels + 1) and are skipped later
    // a read without a start time (INT64_MIN) sorts behind every timed read of its channel: it gets no hour bin and
    // takes no part in the descent test, so the timed reads keep the positions they would have without it
    const int64_t t = ra.time[i];
    const uint64_t tk = t == INT64_MIN ? 0xFFFFFFFFFFull : ((uint64_t)(t - tminmax[0]) & 0xFFFFFFFFFFull);
    keys[i] = good ? (((uint64_t)ch << 40) | tk) : ((uint64_t)(tp.lay.n_channels + 1) << 40);
    vals[i] = (uint32_t)i;
    ra.hour[i] = -1;
}

__global__ __launch_bounds__(256) void mode1_starts_kernel(const uint64_t* keys, uint64_t n, uint32_t nch,
                                                           uint32_t* chan_start) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t ch = (uint32_t)(keys[p] >> 40);
    if (ch > nch) return;
    if (p == 0 || (uint32_t)(keys[p - 1] >> 40) != ch) chan_start[ch - 1] = (uint32_t)p;
}

__global__ __launch_bounds__(256) void mode1_descent_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                            const uint64_t* keys, const uint32_t* vals,
                                                            const uint32_t* chan_start, uint32_t* first_desc) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t ch = (uint32_t)(keys[p] >> 40);
    if (ch > tp.lay.n_channels) return;
    const uint32_t pos = (uint32_t)p - chan_start[ch - 1];
    if (pos < 2) return;
    const int64_t t = ra.time[vals[p]], tprev = ra.time[vals[p - 1]];
    if (t == INT64_MIN || tprev == INT64_MIN) return;
    const int32_t h = hour_of_day(t, tp), hp = hour_of_day(tprev, tp);
    if (h < hp) atomicMin(&first_desc[ch - 1], pos);
}

__global__ __launch_bounds__(256) void mode1_assign_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                           const uint64_t* keys, const uint32_t* vals,
                                                           const uint32_t* chan_start, const uint32_t* first_desc) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t ch = (uint32_t)(keys[p] >> 40);
    if (ch > tp.lay.n_channels) return;
    const uint32_t pos = (uint32_t)p - chan_start[ch - 1];
    const uint32_t idx = vals[p];
    const int64_t t = ra.time[idx];
    if (t == INT64_MIN) return;   // hour stays -1 (mode1_keys_kernel)
    int32_t h = hour_of_day(t, tp) + (pos >= first_desc[ch - 1] ? 24 : 0);
    ra.hour[idx] = h < (int32_t)tp.lay.n_time_bins ? h : -1;
}

__global__ void fill_u64_kernel(uint64_t* p, uint64_t n, uint64_t v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill_u64(uint64_t* p, uint64_t n, uint64_t v, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

// Pre-computed reads: fill the per-read record arrays directly (no sequence bytes exist for them).
__global__ __launch_bounds__(256) void records_kernel(ReadArrays ra, uint64_t base, uint64_t n, const uint32_t* len,
                                                      const double* mean_q, const uint32_t* gc, const uint32_t* channel,
                                                      const uint8_t* mux, const int64_t* time, double fail_q) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t g = base + i;
    const double q = mean_q[i];
    ra.len[g] = len[i]; ra.gc[g] = gc ? gc[i] : 0; ra.nn[g] = 0; ra.qsum[g] = 0; ra.mean_q[g] = q;
    ra.err_hi[g] = 0; ra.err_lo[g] = 0;
    // float(q) < 7.00, lib/seq_routines.py:855; the reference only has the float, so the quality bin is floor(q - 6) in fp64
    const uint32_t qbin1 = (q >= 6.0 && q < 14.0) ? 1u + (uint32_t)floor(q - 6.0) : 0u;
    ra.flags[g] = (uint8_t)(4u | (q < fail_q ? 1u : 0u) | (qbin1 << 4));
    ra.channel[g] = channel[i]; ra.mux[g] = mux[i]; ra.time[g] = time[i];
}
hipError_t launch_records(ReadArrays ra, uint64_t read_base, uint64_t n, const uint32_t* len, const double* mean_q,
                          const uint32_t* gc, const uint32_t* channel, const uint8_t* mux, const int64_t* time,
                          double fail_q, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ra, read_base, n, len, mean_q,
                       gc, channel, mux, time, fail_q);
    return hipGetLastError();
}

// the input-error words behind the results: [0] bad CSR rows, [1] bad FASTQ records, [2] first of them, [3] bad
// quality bytes, [6] the white-space balance of the FASTQ texts ([4], [5]: length bound and ticket);
// [8 .. 8 + kGateWords) the sort gate (OR of the sort keys of the run, raised by finalize_reads)
// `tail` > 0: the words behind the results (input errors, length bound and ticket, sort gate) are reset as well
// (all zero but the number of the first bad FASTQ record): ppqc_reset clears everything with one launch.
__global__ void init_results_kernel(uint64_t* res, uint64_t n_sum, uint64_t nch1, uint32_t tail) {
    const uint64_t total = n_sum + 2 * nch1 + 8;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + tail; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        if (i >= n_sum + nch1 && i < n_sum + 2 * nch1) v = (uint64_t)INT64_MIN;          // min section: ~INT64_MAX
        else if (i == n_sum + 2 * nch1) v = (uint64_t)INT64_MAX;                         // tmin
        else if (i == n_sum + 2 * nch1 + 1) v = (uint64_t)INT64_MIN;                     // tmax
        else if (i == total + 2) v = ~0ull;                                              // first bad FASTQ record: none
        res[i] = v;
    }
}

hipError_t launch_init_results(uint64_t* res, uint64_t n_sum, uint64_t nch1, bool with_tail, hipStream_t s) {
    const uint64_t total = n_sum + 2 * nch1 + 8 + (with_tail ? 8 + kGateWords : 0);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(init_results_kernel, dim3((unsigned)blocks), dim3(256), 0, s, res, n_sum, nch1,
                       with_tail ? 8u + kGateWords : 0u);
    return hipGetLastError();
}

static inline unsigned grid_for(uint64_t n, unsigned cap) {
    uint64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// 8 B of dynamic LDS per channel beside the kernel's 32 B of static words: 8000 channels stay under 64 KiB

hipError_t launch_extremes(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* max_buf, uint64_t* min_buf,
                           int64_t* tminmax, bool write_lens, hipStream_t s) {
    if (n == 0) return hipSuccess;
    // a length of 2^32-1 cannot be told from "unseen" in the u32 LDS copy: lengths are < 2^32-1 by the API contract
    const uint32_t nchL = tp.lay.n_channels <= kExtremesLdsChannels ? tp.lay.n_channels : 0;
    hipLaunchKernelGGL(extremes_kernel<false>, dim3(grid_for(n, 256)),
                       dim3(512), (size_t)nchL * 8, s, BatchView{}, (uint64_t)0, ra, n, tp,
                       max_buf, min_buf, tminmax, nchL, write_lens ? 1 : 0, (uint64_t*)nullptr, (uint64_t*)nullptr,
                       (uint64_t*)nullptr, (uint64_t*)nullptr);
    return hipGetLastError();
}

// The same for one batch at submit time (see extremes_kernel<true>); bound[1] must be zero between launches.
hipError_t launch_batch_extremes(const BatchView& b, uint64_t base, ReadArrays ra, const TableParams& tp,
                                 uint64_t* max_buf, uint64_t* min_buf, int64_t* tminmax, uint64_t* bound,
                                 uint64_t* bound_host, uint64_t* cells, uint64_t* gate, hipStream_t s) {
    if (b.n_reads == 0) return hipSuccess;
    const uint32_t nchL = tp.lay.n_channels <= kExtremesLdsChannels ? tp.lay.n_channels : 0;
    // A small grid on purpose: the tile kernel fills every SIMD's registers (4 waves of 128), so each wave resident
    // beside it displaces one of its waves for as long as it lives, and under a saturated HBM a latency-bound wave lives
    // long.  64 x 256 threads finish well inside the tile kernel of any batch worth the trouble and cost it next to
    // nothing (2048 waves of 8 reads each cost config 2 about 15 us of tile-kernel time).
    // ... while a batch of many short reads needs a wider grid, or this kernel outlasts the per-read kernel (8 * 10^6 reads of
    // 150 bases: 1.4 ms on 64 workgroups beside 0.7 ms of dense_kernel): 16 384 reads per workgroup, 64 to 512 workgroups.
    // b.offsets == nullptr (the reads of a FASTQ text, from the run's arrays): nothing runs beside it, a wider grid
    uint64_t wg = b.n_reads / (b.offsets ? 16384 : 4096);
    wg = wg < 64 ? 64 : (wg > 512 ? 512 : wg);
    hipLaunchKernelGGL(extremes_kernel<true>, dim3(grid_for(b.n_reads, (unsigned)wg)),
                       dim3(256), (size_t)nchL * 8, s, b, base, ra, b.n_reads, tp, max_buf,
                       min_buf, tminmax, nchL, 1, bound, bound_host, cells, gate);
    return hipGetLastError();
}

hipError_t launch_chan_count(ReadArrays ra, uint64_t n, uint32_t n_channels, uint64_t* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n_channels * 8, s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(chan_count_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, ra.channel, n, n_channels, out);
    return hipGetLastError();
}

// The cells alone from the run's per-read arrays (two atomics per read): what ppqc_cells_early falls back to when a
// batch could not leave its cells at submit time.  The region must be zero.
__global__ __launch_bounds__(256) void cells_kernel(ReadArrays ra, uint64_t n, TableParams tp, uint64_t* sum_buf) {
    unsigned long long* g = reinterpret_cast<unsigned long long*>(sum_buf);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t chn = ra.channel[i], m = ra.mux[i];
        const int32_t h = ra.hour[i];
        if (chn < 1 || chn > tp.lay.n_channels || m < 1 || m > tp.lay.n_mux || h < 0 || h >= (int32_t)tp.lay.n_time_bins) continue;
        const uint64_t cell = ((uint64_t)(chn - 1) * tp.lay.n_mux + (m - 1)) * tp.lay.n_time_bins + (uint32_t)h;
        atomicAdd(&g[tp.lay.off_cmh_reads + cell], 1ull);
        atomicAdd(&g[tp.lay.off_cmh_bases + cell], (unsigned long long)ra.len[i]);
    }
}
hipError_t launch_cells(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* sum_buf, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cells_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, ra, n, tp, sum_buf);
    return hipGetLastError();
}

hipError_t launch_cmh_unpack(const TableParams& tp, uint64_t* sum_buf, hipStream_t s) {
    const uint64_t cells = (uint64_t)tp.lay.n_channels * tp.lay.n_mux * tp.lay.n_time_bins;
    hipLaunchKernelGGL(cmh_unpack_kernel, dim3(grid_for(cells, 512)), dim3(256), 0, s, sum_buf, tp.lay);
    return hipGetLastError();
}

// a slab holds a block's sums and, behind them, its extremes (tables_kernel<true>)
static inline uint32_t tables_ext_words(const TableLayout& lay) { return 2 * (lay.n_channels + 1) + 2; }
uint64_t tables_slab_words(const TableLayout& lay) { return (uint64_t)kTabMaxBlocks * (lay.off_cmh_reads + tables_ext_words(lay)); }

static inline uint64_t tables_lds(const TableLayout& lay, uint32_t k, bool ext) {
    const uint64_t nb = lay.n_time_bins;
    return 8ull * (PPQC_N_TOTALS + nb + k) + 4ull * (kHistBins * kRep + nb + 2ull * k) + (ext ? 8ull * k + 8 + 32 : 0);
}
// tables_kernel<true> keeps six accumulators per channel in LDS: one block of 1024 threads per CU, up to 128 KiB
bool tables_ext_fits(const TableLayout& lay) { return lay.n_channels <= 3584 && tables_lds(lay, lay.n_channels, true) <= 128 * 1024; }

// ext_dst != nullptr: the extremes as well (tables_kernel<true>), merged into ext_dst = the result buffer from max_buf on;
// max_buf[0 .. max_words) are then the words of the sort gate.  The caller has asked tables_ext_fits.
hipError_t launch_tables(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* sum_buf, const uint64_t* max_buf,
                         int max_words, uint64_t* slab, hipStream_t s, int cell_mode, const uint64_t* job_chan_count,
                         long long* ext_dst) {
    if (n == 0) return hipSuccess;
    const bool ext = ext_dst != nullptr;
    // per-channel accumulators in LDS need 16 B per channel; a block stays under 64 KiB, so two fit a CU.  When
    // channels and time bins together do not fit (3000 channels x 2048 bins), the per-channel accumulators go straight
    // to global atomics (nchL = 0); the rest fits for every n_time_bins ppqc_create accepts (4096: 55 KiB).
    uint32_t nchL = tp.lay.n_channels <= 3584 ? tp.lay.n_channels : 0;
    if (ext) {
        if (!tables_ext_fits(tp.lay)) return hipErrorInvalidValue;
    } else if (tables_lds(tp.lay, nchL, false) > 64 * 1024) {
        nchL = 0;
    }
    const size_t lds = tables_lds(tp.lay, nchL, ext);
    if (!ext && lds > 64 * 1024) return hipErrorInvalidValue;
    const uint32_t n_small = (uint32_t)tp.lay.off_cmh_reads, n_ext = ext ? tables_ext_words(tp.lay) : 0u;
    const uint32_t stride = n_small + n_ext;
    unsigned blocks;
    if (ext) {
        // one block of 1024 threads per CU: the same four waves per SIMD, one set of tables per CU
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tables_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const uint64_t b = (n + 2 * kTabThreads - 1) / (2 * kTabThreads);
        blocks = (unsigned)(b > kTabMaxBlocks / 2 ? kTabMaxBlocks / 2 : b);
        hipLaunchKernelGGL(tables_kernel<true>, dim3(blocks), dim3(2 * kTabThreads), lds, s, ra, n, tp, sum_buf, max_buf,
                           max_words, nchL, cell_mode, slab, stride);
    } else {
        // two blocks of 512 threads per CU (4 waves per SIMD): the kernel is bound by the latency of its eleven loads per read
        blocks = grid_for(n, kTabMaxBlocks);
        hipLaunchKernelGGL(tables_kernel<false>, dim3(blocks), dim3(kTabThreads), lds, s, ra, n, tp, sum_buf, max_buf,
                           max_words, nchL, cell_mode, slab, stride);
    }
    hipLaunchKernelGGL(tables_reduce_kernel, dim3((unsigned)((stride + 15) / 16)), dim3(256), 0, s, slab, blocks, stride,
                       n_small, sum_buf, n_ext, ext_dst, 2 * (tp.lay.n_channels + 1));
    if (cell_mode == 1) {
        const hipError_t e = launch_cmh_unpack(tp, sum_buf, s);
        if (e != hipSuccess) return e;
    }
    if (tp.compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE)
        hipLaunchKernelGGL(compat_single_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ra, n, tp, sum_buf,
                           job_chan_count);
    return hipGetLastError();
}

hipError_t launch_mode1_keys(ReadArrays ra, uint64_t n, const TableParams& tp, const int64_t* tminmax, uint64_t* keys,
                             uint32_t* vals, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mode1_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ra, n, tp, tminmax, keys,
                       vals);
    return hipGetLastError();
}

hipError_t launch_mode1_hours(ReadArrays ra, uint64_t n, const TableParams& tp, const uint64_t* keys,
                              const uint32_t* vals, uint32_t* chan_start, uint32_t* first_desc, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipError_t e = hipMemsetAsync(first_desc, 0xFF, sizeof(uint32_t) * tp.lay.n_channels, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mode1_starts_kernel, dim3(blocks), dim3(256), 0, s, keys, n, tp.lay.n_channels, chan_start);
    hipLaunchKernelGGL(mode1_descent_kernel, dim3(blocks), dim3(256), 0, s, ra, n, tp, keys, vals, chan_start,
                       first_desc);
    hipLaunchKernelGGL(mode1_assign_kernel, dim3(blocks), dim3(256), 0, s, ra, n, tp, keys, vals, chan_start,
                       first_desc);
    return hipGetLastError();
}


// ---- time_mode 1 across ranks ----------------------------------------------------------------------------------------
// The sort above needs all reads of a channel in one place.  The rule itself does not: with the reads of a channel in
// start-time order, S = all of them but the first, a read is bumped iff it lies at or behind the first element of S whose
// hour of day is below the hour of some EARLIER element of S (a sequence has a descent between neighbours iff it has an
// element below its prefix maximum).  The hour of day rises with the time inside one day, so such an element can only be
// undercut by elements of earlier days, and within a day the earliest element is the first to be: with, per channel and
// day d, tmin[d] / tmax[d] = the earliest / latest start time of S in that day,
//     T* = tmin[D] of the first day D with  hour(tmin[D]) < max over d < D of hour(tmax[d]),   bumped(read) = t >= T*.
// (Reads tied in time have one hour: ties change nothing.)  So the ranks need to agree on three small tables only:
//   phase 0: first[ch] = the earliest start of the channel                      (stored complemented -> all-reduce MAX)
//   phase 1: ties[ch]  = reads starting at first[ch] (one of them is not in S)   (all-reduce SUM; ties[n_channels] counts
//            days[ch][d] = tmax, ~tmin over the reads behind first[ch]           reads outside the day window)  (MAX)
//   phase 2: T* per channel, then every read's hour from its own time.
// Days are counted from the reference time, day -1 .. kMode1Days - 2 (day -1: a negative zone offset in front of it).
constexpr int kMode1Days = PPQC_MODE1_DAYS;

__device__ __forceinline__ int64_t mode1_day(int64_t t, const TableParams& tp) {
    return floordiv64(t - tp.ref_time_s + tp.tz_offset_s, 24 * (int64_t)tp.bin_seconds) + 1;
}

__global__ __launch_bounds__(256) void mode1d_first_kernel(ReadArrays ra, uint64_t n, uint32_t nch, int64_t* first) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ch = ra.channel[i];
        const int64_t t = ra.time[i];
        if (ch < 1 || ch > nch || t == INT64_MIN) continue;
        if (~t > __hip_atomic_load(&first[ch - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax((long long*)&first[ch - 1], (long long)~t);
    }
}

__global__ __launch_bounds__(256) void mode1d_days_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                          const int64_t* __restrict__ first, uint64_t* ties,
                                                          int64_t* dmax, int64_t* dmin) {
    const uint32_t nch = tp.lay.n_channels;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ch = ra.channel[i];
        const int64_t t = ra.time[i];
        if (ch < 1 || ch > nch || t == INT64_MIN) continue;
        if (t == ~first[ch - 1]) { atomicAdd((unsigned long long*)&ties[ch - 1], 1ull); continue; }
        const int64_t d = mode1_day(t, tp);
        if (d < 0 || d >= kMode1Days) { atomicAdd((unsigned long long*)&ties[nch], 1ull); continue; }
        const uint64_t k = (uint64_t)(ch - 1) * kMode1Days + (uint64_t)d;
        if (t > __hip_atomic_load(&dmax[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((long long*)&dmax[k], (long long)t);
        if (~t > __hip_atomic_load(&dmin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((long long*)&dmin[k], (long long)~t);
    }
}

// one thread per channel: first[ch] <- T* (INT64_MAX: no read of the channel is bumped)
__global__ __launch_bounds__(256) void mode1d_pick_kernel(TableParams tp, int64_t* first, uint64_t* ties,
                                                          const int64_t* __restrict__ dmax, const int64_t* __restrict__ dmin) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= tp.lay.n_channels) return;
    const int64_t f = first[c];
    int64_t tstar = INT64_MAX;
    if (f != INT64_MIN) {   // the channel has a timed read
        const int64_t t0 = ~f;
        const bool t0_in_s = ties[c] >= 2;
        const int64_t d0 = mode1_day(t0, tp);
        // Several reads tied at t0: all but one belong to S, and S's day tables have no entry outside the window -- the
        // run fails like for any other read of S out there (every rank sees the same merged tables and counts the same).
        // A lone first read outside the window is in no table and needs none.
        if (t0_in_s && (d0 < 0 || d0 >= kMode1Days)) atomicAdd((unsigned long long*)&ties[tp.lay.n_channels], 1ull);
        int32_t m = -1;
        for (int d = 0; d < kMode1Days; ++d) {
            int64_t mx = dmax[(uint64_t)c * kMode1Days + d], mn = ~dmin[(uint64_t)c * kMode1Days + d];
            bool any = mx != INT64_MIN;
            if (t0_in_s && d == d0) { mn = t0; if (!any) mx = t0; any = true; }
            if (!any) continue;
            if (hour_of_day(mn, tp) < m) { tstar = mn; break; }
            const int32_t h = hour_of_day(mx, tp);
            m = h > m ? h : m;
        }
    }
    first[c] = tstar;
}

__global__ __launch_bounds__(256) void mode1d_assign_kernel(ReadArrays ra, uint64_t n, TableParams tp,
                                                            const int64_t* __restrict__ tstar) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ch = ra.channel[i];
    const int64_t t = ra.time[i];
    int32_t hour = -1;
    if (ch >= 1 && ch <= tp.lay.n_channels && t != INT64_MIN) {
        const int32_t h = hour_of_day(t, tp) + (t >= tstar[ch - 1] ? 24 : 0);
        hour = h < (int32_t)tp.lay.n_time_bins ? h : -1;
    }
    ra.hour[i] = hour;
}

uint64_t mode1_dist_words(uint32_t n_channels) { return 2ull * n_channels + 1 + 2ull * n_channels * kMode1Days; }

// buf = [first: nch | ties: nch + 1 | dmax: nch * D | dmin: nch * D]
hipError_t launch_mode1_dist(int phase, ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* buf, hipStream_t s) {
    const uint32_t nch = tp.lay.n_channels;
    int64_t* first = (int64_t*)buf;
    uint64_t* ties = buf + nch;
    int64_t* dmax = (int64_t*)(ties + nch + 1);
    int64_t* dmin = dmax + (uint64_t)nch * kMode1Days;
    const unsigned grid = grid_for(n ? n : 1, 2048);
    if (phase == 0) {
        hipError_t e = launch_fill_u64(buf, nch, (uint64_t)INT64_MIN, s);                        // ~INT64_MAX: unseen
        if (e == hipSuccess) e = hipMemsetAsync(ties, 0, (nch + 1) * 8, s);
        if (e == hipSuccess) e = launch_fill_u64((uint64_t*)dmax, 2ull * nch * kMode1Days, (uint64_t)INT64_MIN, s);
        if (e != hipSuccess) return e;
        if (n) hipLaunchKernelGGL(mode1d_first_kernel, dim3(grid), dim3(256), 0, s, ra, n, nch, first);
    } else if (phase == 1) {
        if (n) hipLaunchKernelGGL(mode1d_days_kernel, dim3(grid), dim3(256), 0, s, ra, n, tp, first, ties, dmax, dmin);
    } else {
        hipLaunchKernelGGL(mode1d_pick_kernel, dim3((nch + 255) / 256), dim3(256), 0, s, tp, first, ties, dmax, dmin);
        if (n) hipLaunchKernelGGL(mode1d_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ra, n, tp, first);
    }
    return hipGetLastError();
}

}  // namespace ppqc
