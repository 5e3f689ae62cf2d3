// Internal declarations shared by the HIP translation units of libpypore_qc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pypore_qc.h"

namespace ppqc {

constexpr int kWave = 64;               // CDNA wavefront
constexpr int kVec = 16;                // bytes per lane per global_load_dwordx4
constexpr int kSpan = kWave * kVec;     // 1 KiB: one wave-wide load of one stream
constexpr int kMaxTileShift = 16;        // tiles are 2^shift bytes of each stream, 4 KiB .. 64 KiB (packed counters
constexpr int kMinTileShift = 12;        // and per-lane u64 error sums are sized for <= 64 KiB)
constexpr int kDefaultTileShift = 16;
constexpr int kGateWords = 64;           // the sort gate is spread over 64 words (see finalize_reads_kernel)

// A partial (or complete) per-read sum produced by one wave for one tile.
//   cnt : packed counters  [0,26) sum of quality bytes, [26,44) N count, [44,62) G/C/S count
//   hi  : sum over lanes of (lane error sum >> 32)
//   lo  : sum over lanes of (lane error sum & 0xffffffff)      S = hi * 2^32 + lo
struct __attribute__((aligned(32))) Partial {
    uint64_t cnt, hi, lo, pad;
};
constexpr int kQbBits = 26, kNBits = 18;
__host__ __device__ inline uint64_t pack_cnt(uint32_t gcs, uint32_t n, uint32_t qb) {
    return (uint64_t)qb | ((uint64_t)n << kQbBits) | ((uint64_t)gcs << (kQbBits + kNBits));
}
__host__ __device__ inline uint32_t cnt_qb(uint64_t c) { return (uint32_t)(c & ((1ull << kQbBits) - 1)); }
__host__ __device__ inline uint32_t cnt_n(uint64_t c) { return (uint32_t)((c >> kQbBits) & ((1ull << kNBits) - 1)); }
__host__ __device__ inline uint32_t cnt_gcs(uint64_t c) { return (uint32_t)(c >> (kQbBits + kNBits)); }

// Everything the per-read kernels need about one submitted batch (device pointers).
struct BatchView {
    const uint8_t* bases;
    const uint8_t* quals;
    const uint64_t* offsets;   // n_reads + 1, offsets[0] == 0
    uint64_t n_reads, n_bytes;
    const uint8_t* fail_flag;  // may be null
    uint32_t tile_shift;       // log2 of the tile size in bytes
    const uint32_t* channel;   // per-read metadata of the batch, copied into the run arrays by finalize_reads
    const uint8_t* mux;
    const int64_t* time;
};

// Per-read record arrays of the context (device, capacity >= n_reads of the whole run).
struct ReadArrays {
    uint64_t *rec_cnt, *rec_hi, *rec_lo;   // raw sums written by the tile kernel / merged by finalize_reads
    uint32_t *len, *gc, *nn;
    int64_t* qsum;
    double* mean_q;
    uint64_t *err_hi, *err_lo;
    uint8_t* flags;
    uint32_t* channel;
    uint8_t* mux;
    int64_t* time;
    int32_t* hour;
    uint32_t* lens32;
};

struct TableLayout {
    uint32_t n_channels, n_mux, n_time_bins;
    uint64_t off_totals, off_len_log2, off_len_lin, off_q_hist, off_gc_hist, off_hour_reads, off_hour_bases,
        off_chan_reads, off_chan_bases, off_chan_fail, off_cmh_reads, off_cmh_bases, n_sum;
};

struct TableParams {
    TableLayout lay;
    uint32_t bin_seconds, time_mode;
    uint32_t compat_flags;
    int64_t tz_offset_s;
    int64_t ref_time_s;
    double fail_q;
};

// Fixed-point images of the quality thresholds (finalize_reads_kernel, fastq_finalize_kernel)
struct QThr {
    uint64_t fail;                    // image of fail_q
    uint64_t q[PPQC_N_QHIST + 1];     // images of 6, 7, ..., 14
    double fail_q;
};
// per-read flags on the device: bits 0..2 as in ppqc_per_read.flags; bits 4..7 = 1 + the read's bin of the quality
// histogram (0: none), left by whoever computes the read's mean quality so that tables_kernel needs neither the error
// sum nor any 128-bit arithmetic.  ppqc_read_stats hands out bits 0..3 only.
constexpr uint8_t kFlagsPublic = 0x0F;

// ---- launchers (each enqueues on `s`, returns the hipError of the launch) -----------------------------------------
// Device list of the dense tiles of one launch with two sets of counters used alternately: launch k
// counts in set k & 1 and its dense kernel puts counters[(k + 1) & 1] back to zero.  launch_tile_kernel flips
// `parity` itself, and only when it has launched a dense kernel (a launch without one -- an A/B knob -- counts nothing).
struct DenseList {
    uint64_t* base;   // dense_list_words(tiles) u64: 2 x 64 counters, then 64 sub-lists (layout: ppqc_tile.hip, DenseOut)
    int parity;
};
uint64_t dense_list_words(uint64_t n_tiles);
// ppqc_set_tuning: which of several code paths that give the same results a context takes (-1 = the library's own
// choice).  Tests hold the paths to each other, tools/tile_ab.py times them; nothing in the library reads the environment.
struct Tuning {
    int tile_dense = -1;    // 0: every tile stays in tile_kernel; 1: every tile goes onto the dense list; -1: per-tile probe
    int dense_len = -1;     // a tile is dense when its reads average fewer bytes than this (kDenseLen)
    int tile_all = -1;      // 1: dense_kernel alone over every tile; 0: always the pair; -1: by the batch's mean read length
    int tile_shift = -1;    // log2 of the tile size (12 .. 16); -1: by the batch's size
    int pre = -1;           // 0: no index kernel at submit time (extremes at finalize time); -1 / 1: whenever possible
    int fastq_sums = -1;    // FASTQ text: 0 line-mode second pass, 2 span sums of the scan pass, -1 / 1 by the last text's lines
    int sort_beside = -1;   // 1: the sort + N50 chain starts behind the batches' index kernels, beside the per-read kernel; 0: at
                            // that kernel's end; -1: beside it when it is dense_kernel alone (ppqc_finalize_n50)
    int fastq_hdr_aux = -1; // FASTQ text: 0 header kernel in front of the line pass, 2 beside it, -1 / 1 beside line_sum only
    int fastq_one_pass = -1;  // FASTQ text: 1 the one-pass kernels of ppqc_fastq1.hip for every text, 0 never, -1 by the last text's lines
    int fq1_regions = -1;     // ... the number of regions (waves) they split a text into (4096)
    int fq1_early = -1;       // ... 0: their finalize kernel is launched when the host knows the line count; -1 / 1: straight behind
                              // the scan, the count read on the device, when the last text's line length says the arrays suffice
    int n50_select = -1;      // local N50: -1 / 1 by radix select (three histograms of the lengths) whenever the run left its index at
                              // submit time, 0 by the sort
    int fastq_extremes = -1;  // FASTQ text: 0 an extremes kernel over its reads in line behind the finalize kernel; -1 / 1: left to
                              // the tables kernel of ppqc_finalize when every read of the run came from a text (that kernel
                              // takes the largest length from the gate; whoever needs the extremes earlier gets the kernel then)
};
hipError_t launch_tile_kernel(const BatchView& b, const uint64_t* lut_dev, ReadArrays ra, uint64_t read_base,
                              Partial* head, Partial* tail, DenseList& dl, DenseList& slow, const Tuning& tn, hipStream_t s,
                              bool line_mode = false, bool* dense_alone = nullptr);   // *dense_alone: dense_kernel took every tile
hipError_t launch_finalize_reads(const BatchView& b, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, const QThr& thr, uint64_t* csr_errors,
                                 uint32_t n_channels, uint64_t* gate, bool write_lens, hipStream_t s);

// Pre-computed reads (sequencing_summary.txt rows): device arrays of n entries appended at read_base.
hipError_t launch_records(ReadArrays ra, uint64_t read_base, uint64_t n, const uint32_t* len, const double* mean_q,
                          const uint32_t* gc, const uint32_t* channel, const uint8_t* mux, const int64_t* time,
                          double fail_q, hipStream_t s);

hipError_t launch_extremes(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* max_buf, uint64_t* min_buf,
                           int64_t* tminmax, bool write_lens, hipStream_t s);
hipError_t launch_batch_extremes(const BatchView& b, uint64_t base, ReadArrays ra, const TableParams& tp,
                                 uint64_t* max_buf, uint64_t* min_buf, int64_t* tminmax, uint64_t* bound,
                                 uint64_t* bound_host, uint64_t* cells, uint64_t* gate, hipStream_t s);   // at submit time, beside the tile
                                 // kernel; cells != nullptr: the batch's packed (channel, mux, hour) cells too (cmh_bases region)
// (channel, mux, hour) cells: one packed atomic per read when the run's reads and bases (host-known) fit the two fields
bool cmh_packable(uint64_t n_reads, uint64_t n_bases);
hipError_t launch_cmh_unpack(const TableParams& tp, uint64_t* sum_buf, hipStream_t s);
hipError_t launch_cells(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* sum_buf, hipStream_t s);   // unpacked, region zero
// job_chan_count: per-channel read count of the whole multi-GPU job for PPQC_COMPAT_MULTI_READ_SINGLE (nullptr: this
// rank's own table is the job's)
// cell_mode 0: the cells are in sum_buf already (left at submit time; the caller unpacks them), 1: one packed atomic
// per read + cmh_unpack, 2: two atomics per read
// slab: tables_slab_words(layout) u64 of scratch (every block leaves its private tables there; a second kernel adds them up)
uint64_t tables_slab_words(const TableLayout& lay);
// max_buf[0 .. max_words): words whose largest is the run's largest length (max_buf of the extremes kernels, 1 word; or the
// sort gate, kGateWords words, when the FASTQ finalize kernels alone have raised it)
hipError_t launch_tables(ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* sum_buf, const uint64_t* max_buf,
                         int max_words, uint64_t* slab, hipStream_t s, int cell_mode, const uint64_t* job_chan_count = nullptr,
                         long long* ext_dst = nullptr);   // ext_dst: the run's extremes as well, merged into the result buffer from
                                                          // max_buf on (only where tables_ext_fits; max_buf = the sort gate then)
bool tables_ext_fits(const TableLayout& lay);
hipError_t launch_chan_count(ReadArrays ra, uint64_t n, uint32_t n_channels, uint64_t* out, hipStream_t s);

hipError_t launch_fill_u64(uint64_t* p, uint64_t n, uint64_t v, hipStream_t s);
// res = [n_zero sums (tables + select histogram) = 0 | nch1 max = 0 | nch1 ~min = ~INT64_MAX | tmin = INT64_MAX,
//        tmax = INT64_MIN | n50, total = 0 | 4 words of select state = 0]
hipError_t launch_init_results(uint64_t* res, uint64_t n_sum, uint64_t nch1, bool with_tail, hipStream_t s);

// time_mode 1 helpers
hipError_t launch_mode1_keys(ReadArrays ra, uint64_t n, const TableParams& tp, const int64_t* tminmax, uint64_t* keys,
                             uint32_t* vals, hipStream_t s);
hipError_t launch_mode1_hours(ReadArrays ra, uint64_t n, const TableParams& tp, const uint64_t* sorted_keys,
                              const uint32_t* sorted_vals, uint32_t* chan_start, uint32_t* first_desc,
                              hipStream_t s);

// time_mode 1 across ranks (ppqc_tables.hip): phases 0, 1, 2 over buf = mode1_dist_words(n_channels) u64
uint64_t mode1_dist_words(uint32_t n_channels);
hipError_t launch_mode1_dist(int phase, ReadArrays ra, uint64_t n, const TableParams& tp, uint64_t* buf, hipStream_t s);

// Hand-written LSD radix sort.  `counts` has 256*(n_waves+1)+256 u32; byte_mask selects which key bytes are sorted on.
// sort_u32: the first pass reads `in`, later passes ping-pong between buf_a and buf_b (`in` may be buf_b: in place);
// *out = the buffer that holds the sorted keys.  sort_u64_pairs sorts keys/vals in place (tmp_* are the ping-pong).
uint64_t sort_count_words(uint64_t n);
hipError_t sort_u32(const uint32_t* in, uint32_t* buf_a, uint32_t* buf_b, uint64_t n, uint32_t* counts,
                    uint32_t byte_mask, hipStream_t s, const uint32_t** out);
// The same with all four passes enqueued and skipped on the device from gate[0] >= every key: no host round trip.
// The bound is the OR of gate[0 .. gate_words) (1 or kGateWords).
hipError_t sort_u32_gated(const uint32_t* in, uint32_t* buf_a, uint32_t* buf_b, uint64_t n, uint32_t* counts,
                          const uint64_t* gate, int gate_words, hipStream_t s);
hipError_t sort_u64_pairs(uint64_t* keys, uint64_t* tmp_keys, uint32_t* vals, uint32_t* tmp_vals, uint64_t n,
                          uint32_t* counts, uint32_t byte_mask, hipStream_t s);

// N50 over ascending-sorted lengths: out[0] = n50, out[1] = total bases.  block_sums has n/2048 + 2 u64.
uint64_t n50_block_words(uint64_t n);
hipError_t launch_n50(const uint32_t* sorted, uint64_t n, uint64_t* block_sums, uint64_t* out, uint64_t* host_out,
                      hipStream_t s);
// after sort_u32_gated(in, buf_a, buf_b, .., gate): picks the buffer the sorted keys ended in from the same gate word
hipError_t launch_n50_gated(const uint32_t* in, const uint32_t* buf_a, const uint32_t* buf_b, const uint64_t* gate,
                            int gate_words, uint64_t n, uint64_t* block_sums, uint64_t* out, uint64_t* host_out,
                            hipStream_t s);   // host_out: optional pinned host word that also receives the N50

// Exact N50 by MSD radix select (multi-GPU path).  state = {prefix, weight of all longer keys}; sel = PPQC_N_SEL u64.
// hist: sel[d] += sum of the lengths whose bits above `shift + nbits` equal the prefix and whose digit is d.
// pick: chooses the digit (see include/pypore_qc.h), updates state, zeroes sel; `last` also writes out[0] = N50, out[1] = total.
hipError_t launch_select_hist(const uint32_t* lens, uint64_t n, int shift, int nbits, const uint64_t* state,
                              uint64_t* sel, hipStream_t s, const uint64_t* gate = nullptr, int gate_words = 1);
hipError_t launch_select_pick(uint64_t* sel, const uint64_t* total, uint64_t* state, int shift, int nbits, bool last,
                              bool first, uint64_t* out, uint64_t* host_out, hipStream_t s, const uint64_t* gate = nullptr,
                              int gate_words = 1);   // total == nullptr: from the first round's histogram; host_out (last round, may
                              // be null): a pinned word that receives the N50 as well; gate: words whose OR bounds every key -- a
                              // round above its top bit does nothing (both kernels), state[3] tells whether the total is known

// ---- FASTQ text on the GPU (ppqc_fastq_gpu.hip) --------------------------------------------------------------------
constexpr uint32_t kNlChunk = 8192;   // bytes of text per wave of the line-index kernels (eight 1 KiB spans in registers)
constexpr uint32_t kNlEntriesPerChunk = 32;
struct FastqScratch {
    uint32_t* counts;       // newlines per chunk
    uint32_t* n_entries;    // odd vectors per chunk (may exceed kNlEntriesPerChunk: then the list is incomplete)
    uint64_t* entries;      // kNlEntriesPerChunk per chunk
    uint64_t* prefix;       // exclusive prefix of counts; [n_chunks] = newlines of the text, [n_chunks + 1] = overflowing chunks
    uint64_t* block_sums;   // scan scratch
    uint64_t* line_off;     // n_lines + 1 line starts (the CSR offsets of the line-mode tile kernel)
    uint64_t* span_sum;     // two words per 1 KiB span of every chunk: the scan pass's span sums (ppqc_fastq_gpu.hip)
};
uint64_t nl_scan_block_words(uint64_t n_chunks);
// lut_dev != nullptr: the pass also leaves the span sums in fs.span_sum
hipError_t launch_nl_scan(const uint8_t* text, uint64_t n, const FastqScratch& fs, uint64_t* host_out, hipEvent_t counted,
                          hipStream_t s, const uint64_t* lut_dev = nullptr, bool notes = true);   // notes = false: count only
// whole lines from the span sums (sequence and quality line of every record) into line_ra.rec_*
hipError_t launch_line_sums(const uint8_t* text, uint64_t n, uint64_t n_records, const uint64_t* line_off,
                            const uint64_t* span_sum, const uint64_t* lut_dev, ReadArrays line_ra, hipStream_t s);
hipError_t launch_nl_place(const uint8_t* text, uint64_t n, uint64_t n_lines, uint64_t n_over, const FastqScratch& fs,
                           uint64_t* errors, hipStream_t s);
// Part of the index-only results of the text's reads from the finalize kernels of the FASTQ paths (on != 0): their hour bins and, with cells,
// their packed (channel, mux, hour) cells -- the million-odd atomics that are best issued from a kernel busy with
// something else.  The extremes follow in extremes_kernel<true> over the run's arrays (launch_batch_extremes, offsets
// == nullptr), which then leaves hours and cells alone.
struct Fq1Index {
    int on;
    TableParams tp;
    uint64_t* cells;   // nullptr: no (channel, mux, hour) cells
};
// one thread per record: validates the four lines, strips line ends, merges the line sums into the per-read arrays,
// parses ch= and start_time= out of the header.  errors[1] counts malformed records.
// channel, mux and start time of every record from its header line (needs the line table only)
hipError_t launch_fastq_header(const uint8_t* text, uint64_t n_records, const uint64_t* line_off, ReadArrays ra,
                               uint64_t read_base, uint8_t mux, const uint64_t* ws_seen, uint64_t* errors, hipStream_t s);
hipError_t launch_fastq_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const uint64_t* line_off,
                                 ReadArrays line_ra, ReadArrays ra, uint64_t read_base, const Partial* head,
                                 const Partial* tail, uint32_t tile_shift, const uint64_t* lut, const QThr& thr,
                                 uint8_t mux, uint64_t* errors, uint32_t n_channels, uint64_t* gate, const Fq1Index& ix,
                                 hipStream_t s, bool whole_lines = false);   // also lens32, the sort gate and (ix) hours and cells

// ---- FASTQ text of short lines in one pass (ppqc_fastq1.hip) ---------------------------------------------------------
constexpr uint32_t kFq1Chunk = 16384;            // bytes of text a wave of fq1_scan_kernel takes at a time (256 per lane)
constexpr uint32_t kFq1LinesPerChunk = 128;      // share of the point table per chunk of a region (lines)
constexpr uint64_t kFq1MaxRegionChunks = 1024;   // a region holds at most 2^24 bytes (24-bit positions in the points)
struct Fq1Region { uint64_t lines, gcs, n, odd, qb, err_a, err_b, flags; };   // totals, then exclusive prefix over the regions
struct Fq1Scratch {
    uint64_t* points;      // 4 words per line end, kFq1LinesPerChunk per chunk
    Fq1Region* regions;
    uint64_t* head;        // lines of the text, flags (2: a region with more lines than its share of the point table)
};
uint64_t fq1_chunks(uint64_t n);
void fq1_shape(uint64_t n, int want_regions, uint32_t* cpr, uint64_t* n_regions);
hipError_t launch_fq1_scan(const uint8_t* text, uint64_t n, const Fq1Scratch& f, uint32_t cpr, uint64_t n_regions,
                           const uint64_t* lut_dev, uint64_t* host_out, uint64_t ticket, hipStream_t s);   // host_out: five pinned
                           // words; [0] lines, [1] flags, and last of all [4] = ticket (what the host polls for)
hipError_t launch_fq1_finalize(const uint8_t* text, uint64_t n, uint64_t n_records, const Fq1Scratch& f, uint32_t cpr,
                               uint64_t n_regions, ReadArrays ra, uint64_t read_base, const uint64_t* lut_dev,
                               const QThr& thr, uint8_t mux, uint64_t* errors, uint32_t n_channels, uint64_t* gate,
                               const Fq1Index& ix, hipStream_t s);   // also writes lens32 and raises the sort gate, like finalize_reads
hipError_t launch_fq1_finalize_early(const uint8_t* text, uint64_t n, uint64_t max_records, const Fq1Scratch& f, uint32_t cpr, uint64_t n_regions, ReadArrays ra,
                                     uint64_t read_base, const uint64_t* lut_dev, const QThr& thr, uint8_t mux,
                                     uint64_t* errors, uint32_t n_channels, uint64_t* gate, const Fq1Index& ix,
                                     uint64_t* host_out, uint64_t ticket, hipStream_t s);   // behind the regions kernel, the record
                                     // count read on the device; host_out / ticket: as launch_fq1_scan (which then gets none)

hipError_t launch_synth(uint8_t* bases, uint8_t* quals, uint64_t first, uint64_t n, uint64_t seed, hipStream_t s);

// ---- device helpers ----------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_add_u64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return v + (((uint64_t)hi << 32) | lo);
}
// Sum over the 64 lanes of a fully active wave; the result is wave-uniform.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v = dpp_add_u64<0x121>(v);   // row_ror:1
    v = dpp_add_u64<0x122>(v);   // row_ror:2
    v = dpp_add_u64<0x124>(v);   // row_ror:4
    v = dpp_add_u64<0x128>(v);   // row_ror:8  -> every lane holds the sum of its 16-lane row
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    uint64_t t = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lo, r * 16);
        uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hi, r * 16);
        t += ((uint64_t)h << 32) | l;
    }
    return t;
}
// Sum of a u32 over the 64 lanes of a fully active wave (wave-uniform result; the sum must fit 32 bits): four row
// rotations leave every lane with its 16-lane row's sum, row_bcast:15 / row_bcast:31 (gfx9 DPP) carry the row sums up
// to lane 63.  Six fused v_add_u32_dpp and one v_readlane, against about 28 instructions for the u64 version.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v = dpp_add_u32<0x121, 0xf>(v);   // row_ror:1
    v = dpp_add_u32<0x122, 0xf>(v);   // row_ror:2
    v = dpp_add_u32<0x124, 0xf>(v);   // row_ror:4
    v = dpp_add_u32<0x128, 0xf>(v);   // row_ror:8
    v = dpp_add_u32<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add_u32<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Inclusive scan of a u32 over the 64 lanes (every partial sum must fit 32 bits): row_shr steps inside each row of 16,
// then row_bcast:15 / row_bcast:31 carry the row totals upwards.
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    v = dpp_add_u32<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add_u32<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add_u32<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add_u32<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add_u32<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add_u32<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}
// Raw sums of the segment [a, e) of the byte streams (e > a) whose record index is g: complete in rec_* when it lies
// in one tile, else the tail partial of its first tile plus the head partials of the following ones, in tile order.
struct SegSums { uint64_t gcs, nn, qb, hi, lo; };
__device__ __forceinline__ SegSums merged_sums(uint64_t a, uint64_t e, uint64_t g, const ReadArrays& ra,
                                               const Partial* __restrict__ head, const Partial* __restrict__ tail,
                                               uint32_t tile_shift) {
    SegSums r;
    const uint64_t ts = a >> tile_shift, te = (e - 1) >> tile_shift;
    if (ts == te) {
        const uint64_t c = ra.rec_cnt[g];
        r.gcs = cnt_gcs(c); r.nn = cnt_n(c); r.qb = cnt_qb(c);
        r.hi = ra.rec_hi[g]; r.lo = ra.rec_lo[g];
    } else {
        Partial p = tail[ts];
        r.gcs = cnt_gcs(p.cnt); r.nn = cnt_n(p.cnt); r.qb = cnt_qb(p.cnt); r.hi = p.hi; r.lo = p.lo;
        // (a read of megabases over 16 KiB tiles has a hundred of them: eight partials asked for at a time, not one
        // memory latency each)
        uint64_t t = ts + 1;
        for (; t + 8 <= te + 1; t += 8) {
            Partial q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = head[t + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r.gcs += cnt_gcs(q[j].cnt); r.nn += cnt_n(q[j].cnt); r.qb += cnt_qb(q[j].cnt); r.hi += q[j].hi; r.lo += q[j].lo;
            }
        }
        for (; t <= te; ++t) {
            p = head[t];
            r.gcs += cnt_gcs(p.cnt); r.nn += cnt_n(p.cnt); r.qb += cnt_qb(p.cnt); r.hi += p.hi; r.lo += p.lo;
        }
    }
    return r;
}
// OR of the gate words, for kernels whose blocks are whole waves: lane l reads word l, six xor-shuffle steps.
__device__ __forceinline__ uint64_t gate_bound(const uint64_t* __restrict__ gate, int words) {
    const int lane = threadIdx.x & 63;
    uint64_t v = gate[lane < words ? lane : 0];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v |= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Mean quality of a read of L > 0 bytes whose fixed-point error sum is S = H * 2^32 + l32, and the flags that hang on it:
// bit 0 = fail (mean_q < fail_q <=> S > L * e(fail_q), exact integers), bits 4..7 = 1 + quality-histogram bin, where
// bin j <=> 6 + j <= mean_q < 7 + j <=> L * e(7 + j) < S <= L * e(6 + j).  The fp64 mean_q is within 1e-9 of the
// exact value, so floor(mean_q - 6) is the bin or a neighbour of it: two or three exact comparisons settle it.
__device__ __forceinline__ uint8_t quality_flags(uint64_t L, uint64_t H, uint64_t l32, const QThr& t, double* mq_out) {
    const double S = (double)H * 4294967296.0 + (double)l32;
    const double mq = -10.0 * log10(S / ((double)L * 4503599627370496.0));
    *mq_out = mq;
    const unsigned __int128 S128 = ((unsigned __int128)H << 32) + l32;
    uint8_t fl = S128 > (unsigned __int128)L * t.fail ? 1 : 0;
    int qbin = -1;
    if (mq > 5.5 && mq < 14.5) {
        int j = (int)floor(mq - 6.0);
        j = j < 0 ? 0 : (j > PPQC_N_QHIST - 1 ? PPQC_N_QHIST - 1 : j);
        const bool a = S128 <= (unsigned __int128)L * t.q[j];        // mean_q >= 6 + j
        const bool b = S128 <= (unsigned __int128)L * t.q[j + 1];    // mean_q >= 7 + j
        if (a && !b) qbin = j;
        else if (!a) { if (j > 0 && S128 <= (unsigned __int128)L * t.q[j - 1]) qbin = j - 1; }
        else if (j < PPQC_N_QHIST - 1 && !(S128 <= (unsigned __int128)L * t.q[j + 2])) qbin = j + 1;
    }
    return (uint8_t)(fl | ((qbin + 1) << 4));
}

// ---- index-only results of a read: what extremes_kernel (ppqc_tables.hip) and fq1_finalize_kernel share -------------
constexpr int kCmhShift = 40;                      // packed (channel, mux, hour) cell: bases below, reads above this bit
constexpr uint32_t kExtremesLdsChannels = 8000;    // per-channel extremes are privatised in LDS up to this many channels
__device__ __forceinline__ int64_t floordiv64(int64_t a, int64_t b) {
    int64_t q = a / b;
    return ((a % b != 0) && ((a < 0) != (b < 0))) ? q - 1 : q;
}
// The hour bin of a read (time_mode 0; -1: none: no good channel, no start time -- INT64_MIN, a FASTQ header without
// start_time= --, or outside the bins) and, with `cells`, its packed (channel, mux, hour) cell.
__device__ __forceinline__ int32_t index_cell(uint32_t ch, uint32_t L, int64_t t, uint32_t m, const TableParams& tp, uint64_t* cells) {
    int32_t hour = -1;
    if (ch >= 1 && ch <= tp.lay.n_channels && t != INT64_MIN && tp.time_mode == 0) {
        const int64_t h = floordiv64(t - tp.ref_time_s + tp.tz_offset_s, (int64_t)tp.bin_seconds);
        hour = (h >= 0 && h < (int64_t)tp.lay.n_time_bins) ? (int32_t)h : -1;
    }
    if (cells != nullptr && hour >= 0 && m >= 1 && m <= tp.lay.n_mux)
        atomicAdd((unsigned long long*)&cells[((uint64_t)(ch - 1) * tp.lay.n_mux + (m - 1)) * tp.lay.n_time_bins + (uint32_t)hour],
                  (1ull << kCmhShift) | (unsigned long long)L);
    return hour;
}
// ---- byte classification shared by the per-read kernels and the FASTQ text kernels ------------------------------
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
#endif

}  // namespace ppqc
