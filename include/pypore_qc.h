/* pypore_qc.h -- C ABI of libpypore_qc.so: MI355X (gfx950) per-read QC statistics for `pypore seqstats`.
 *
 * The reference (rsemeraro/PyPore) is pure Python and has no FFI; the seam this library slots into is
 * the task protocol of lib/seq_routines.py:18-37 (a callable per channel / per file whose result is
 * merged into ChannelDict, :940-942, :1020-1022) and the ChannelDict data contract consumed by
 * result_plotting (:572) and FastStats (:97).  Each entry point below cites the reference code it
 * replaces.  Loaders (h5py / FASTQ) stay on the CPU and hand over reads packed as CSR.
 *
 * Conventions: every call returns 0 on success or a negative PPQC_E_* code; ppqc_last_error() gives a
 * thread-local message; nothing throws across the ABI.  A context is bound to one GPU and must be
 * driven by one thread at a time.  There is NO CPU fallback: without a usable HIP device every
 * compute entry point fails with PPQC_E_NO_DEVICE.
 */
#ifndef PYPORE_QC_H
#define PYPORE_QC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPQC_ABI_VERSION 7

enum {
    PPQC_OK = 0,
    PPQC_E_INVALID = -1,   /* bad argument / config */
    PPQC_E_NO_DEVICE = -2, /* no HIP device, or device index out of range */
    PPQC_E_HIP = -3,       /* a HIP runtime call failed (message has the hipError string) */
    PPQC_E_NOMEM = -4,
    PPQC_E_STATE = -5,     /* call order violated (e.g. fetch before finalize) */
    PPQC_E_RANGE = -6      /* input outside the documented range (e.g. time_mode 1 with a span of 2^40 ticks or more) */
};

/* fixed-point scale of the error-probability sum: e[b] = round(2^52 * 10^(-(b-33)/10)) */
#define PPQC_ERR_SHIFT 52

/* sizes of the fixed tables */
#define PPQC_N_TOTALS 16
#define PPQC_N_LEN_LOG2 40
#define PPQC_N_LEN_LIN 100
#define PPQC_N_QHIST 8
#define PPQC_N_SEL 2048     /* words of the N50 radix-select histogram (11-bit digits) */
#define PPQC_N_GCHIST 20

/* indices into ppqc_tables.totals */
enum {
    PPQC_TOT_N_READS = 0,    /* every read submitted */
    PPQC_TOT_N_BASES = 1,    /* sum of lengths (fail-flagged reads count 0)                */
    PPQC_TOT_N_FAIL = 2,     /* reads with mean_q < fail_q, regardless of mux ("true" fails) */
    PPQC_TOT_N_GC = 3,
    PPQC_TOT_N_N = 4,
    PPQC_TOT_QSUM = 5,       /* sum of integer Phred scores (two's complement)             */
    PPQC_TOT_N_BINNED = 6,   /* reads that reached the hour / channel x mux x hour tables  */
    PPQC_TOT_BAD_CHANNEL = 7,/* channel outside 1..n_channels: counted here and in N_READS only */
    PPQC_TOT_TIME_OOB = 8,   /* hour bin outside [0, n_time_bins): dropped from hour tables only */
    PPQC_TOT_ZERO_LEN = 9,
    PPQC_TOT_BAD_MUX = 10,   /* mux outside 1..n_mux: in channel totals, not in mux/hour/fail tables
                                (lib/fast5_reader.py:130-137 only visits mux '1'..'4') */
    PPQC_TOT_NO_TIME = 11    /* start time INT64_MIN ("none": a FASTQ header without start_time=), valid channel: not in
                                the time range (ppqc_time_range), no hour bin (counted in TIME_OOB too when the mux is
                                valid) */
};

/* Hard-coded in the reference: 512 channels (lib/seq_routines.py:905-906), 4 muxes
 * (lib/fast5_reader.py:130), 48 hourly bins (lib/seq_routines.py:99), fail below 7.0
 * (lib/fast5_reader.py:136), Phred+33 (SeqIO 'fastq', lib/seq_routines.py:50). */
typedef struct ppqc_config {
    uint32_t struct_size;  /* = sizeof(ppqc_config); set by ppqc_config_default */
    uint32_t n_channels;   /* 512 */
    uint32_t n_mux;        /* 4 */
    uint32_t n_time_bins;  /* 48 */
    uint32_t bin_seconds;  /* 3600: ticks per time bin.  Times are integer ticks; a tick is one second for Fast5 input
                              (the reference truncates to whole seconds with mktime, lib/fast5_reader.py:66-67) and may
                              be finer (e.g. 1 us, bin_seconds = 3 600 000 000) for sequencing_summary.txt input whose
                              start times are fractional (lib/seq_routines.py:845-846) */
    uint32_t time_mode;    /* 0: bin = floor((t - ref + tz) / bin_seconds)
                              1: reference quirk, lib/fast5_reader.py:117-126: bin = that value mod 24 ("hour of day"),
                                 then per channel in start-time order every read from the first descent at
                                 index >= 2 onwards gets +24; start times must span < 2^40 ticks */
    int64_t tz_offset_s;   /* local-zone offset (ticks) the reference picks up through time.localtime (:118); 0 = UTC */
    int64_t ref_time_s;    /* RefTime / RefStart in ticks (lib/seq_routines.py:819-828, multi_read_fast5_reader.py:137) */
    double fail_q;         /* 7.0 */
    uint64_t max_reads_hint; /* per-read arrays are pre-sized to this (0 = grow on demand) */
    uint32_t compat_flags; /* PPQC_COMPAT_*: reference behaviours that are defects rather than definitions, selectable so
                              that a drop-in can be bit-identical to the reference where it matters to the caller */
    uint32_t reserved0;    /* 0 */
} ppqc_config;

/* channel_parser (lib/multi_read_fast5_reader.py:64-67) filters a channel's reads by mux only `if len(l) > 1`: the read of
 * a channel that holds exactly ONE read goes into MuxProductivity of every mux '1'..'4' -- whatever its own mux is --
 * and its fail counts once per mux.  With this bit the tables of such channels say what the reference says:
 * chan_fail = n_mux * fail, cmh_reads / cmh_bases = (1, length) at the read's hour bin under every mux, and hour_reads /
 * hour_bases (the sum over channels and muxes, lib/seq_routines.py:103-108) count the read n_mux times.  The totals are
 * not changed.  Multi-GPU: the rule looks at a channel's read count in the whole job (ppqc_compat_dist_count below).
 * The single-read worker (lib/fast5_reader.py:130-145) has no such branch. */
#define PPQC_COMPAT_MULTI_READ_SINGLE 1u

/* One CSR batch of reads (what lib/fast5_reader.py:36-71 / lib/multi_read_fast5_reader.py:89-119
 * extract per read, and the FASTQ record of lib/fastqparser.py:7-16: bases = line 2, quals = line 4).
 * offsets[0] must be 0 and offsets[n_reads] == n_bytes.  For ppqc_submit all pointers are host memory
 * (ideally from ppqc_alloc_pinned); for ppqc_submit_device all are device memory, bases/quals 16-byte
 * aligned.  A 16-byte aligned load may touch up to 15 bytes past n_bytes inside the same 16-byte granule. */
typedef struct ppqc_batch {
    uint64_t n_reads;
    uint64_t n_bytes;
    const uint8_t* bases;         /* ASCII, n_bytes */
    const uint8_t* quals;         /* ASCII Phred+33, n_bytes */
    const uint64_t* offsets;      /* n_reads + 1 */
    const uint32_t* channel;      /* channel_number, 1-based (lib/fast5_reader.py:50) */
    const uint8_t* mux;           /* start_mux (lib/fast5_reader.py:48) */
    const int64_t* start_time_s;  /* mktime(parse(time_stamp)) whole seconds (lib/fast5_reader.py:66-67) */
    const uint8_t* fail_flag;     /* optional (may be NULL): path contains 'fail' => len 0, q 6.99, gc 0
                                     (lib/fast5_reader.py:37-40,52) */
} ppqc_batch;

/* Reads whose statistics were computed elsewhere: the rows of albacore's sequencing_summary.txt, from which
 * summary_reader (lib/seq_routines.py:831-863) builds ChannelDict without touching the sequences.  Host pointers.
 * fail = mean_q < fail_q in fp64 (:855); quality bin = floor(mean_q - 6).  mux is 1 for every read when the caller
 * wants the mux-less tables of that mode (configure n_mux = 1). */
typedef struct ppqc_records {
    uint64_t n_reads;
    const uint32_t* length;       /* sequence_length_template column */
    const double* mean_q;         /* mean_qscore_template column */
    const uint32_t* gc_count;     /* optional (may be NULL = 0) */
    const uint32_t* channel;
    const uint8_t* mux;
    const int64_t* start_time_s;  /* ticks */
} ppqc_records;

/* Per-read results, caller-allocated SoA; any pointer may be NULL to skip that column. */
typedef struct ppqc_per_read {
    uint32_t* length;   /* offsets[i+1]-offsets[i]  (== sequence_length attr, lib/fast5_reader.py:55); 0 if flagged */
    uint32_t* gc_count; /* G,C,S either case (Bio.SeqUtils.GC numerator, lib/fast5_reader.py:58) */
    uint32_t* n_count;  /* N, n */
    int64_t* qsum;      /* sum(byte) - 33*length */
    double* mean_q;     /* -10*log10(S / (length * 2^52)); 0.0 for empty reads; 6.99 if flagged
                           (ONT mean_qscore, the attribute read at lib/fast5_reader.py:56) */
    uint64_t* err_hi;   /* S = err_hi * 2^32 + err_lo exactly; the halves are sums, not a canonical split */
    uint64_t* err_lo;
    uint8_t* flags;     /* bit0: fail (mean_q < fail_q), bit1: fail_flag was set, bit2: submitted as a record */
} ppqc_per_read;

/* Run-level tables, caller-allocated (all uint64_t unless noted; NULL skips a table).
 * Replaces the data part of parsing_func (lib/fast5_reader.py:127-158), FastStats
 * (lib/seq_routines.py:97-144) and result_plotting / ChannelPlotter_f (:378-394, :507-514, :673-680). */
typedef struct ppqc_tables {
    uint64_t* totals;        /* PPQC_N_TOTALS */
    uint64_t* len_hist_log2; /* PPQC_N_LEN_LOG2: bin 0 = len 0, bin k = [2^(k-1), 2^k) */
    uint64_t* len_hist_lin;  /* PPQC_N_LEN_LIN bins of width lin_bin_width = max_len/100 + 1 */
    uint64_t* q_hist;        /* PPQC_N_QHIST: 6+j <= mean_q < 7+j  (xbins 6..14 step 1, lib/seq_routines.py:234) */
    uint64_t* gc_hist;       /* PPQC_N_GCHIST: 5k <= GC% < 5k+5, GC% > 0, 100 -> last (:136, :241) */
    uint64_t* hour_reads;    /* n_time_bins (HoursDict[h][0], lib/seq_routines.py:103-108) */
    uint64_t* hour_bases;    /* n_time_bins (HoursDict[h][1]) */
    uint64_t* chan_reads;    /* n_channels  (ReadPerChannel,  lib/fast5_reader.py:128) */
    uint64_t* chan_bases;    /* n_channels  (BasesPerChannel, :129) */
    uint64_t* chan_fail;     /* n_channels  (FailedReads,     :136-137) */
    uint64_t* cmh_reads;     /* [n_channels][n_mux][n_time_bins] (MuxProductivity, :138-145) */
    uint64_t* cmh_bases;     /* same shape */
    uint64_t* chan_min_len;  /* n_channels, INT64_MAX if the channel is empty (MaxMinvec, lib/seq_routines.py:132-133) */
    uint64_t* chan_max_len;  /* n_channels */
    uint64_t max_len;        /* out */
    uint64_t min_pos_len;    /* out: smallest length > 0, INT64_MAX if none */
    uint64_t lin_bin_width;  /* out */
    uint64_t n50;            /* out: first length, descending, whose running sum c has 2c >= total */
} ppqc_tables;

/* Device-side view of the mergeable tables, for RCCL all-reduce through torch.distributed:
 * sum_buf (n_sum u64, SUM), max_buf (n_max i64, MAX), min_buf (n_min i64, holds the bitwise COMPLEMENT of each minimum,
 * ~INT64_MAX if empty, so it also merges by MAX): min_buf == max_buf + n_max, one all-reduce(MAX) of n_max + n_min words
 * merges both.  ppqc_fetch_tables decodes the minima. */
typedef struct ppqc_device_tables {
    void* sum_buf; uint64_t n_sum;
    void* max_buf; uint64_t n_max;   /* [0] = max_len, [1..] = chan_max_len */
    void* min_buf; uint64_t n_min;   /* [0] = ~min_pos_len, [1..] = ~chan_min_len */
    void* lengths; uint64_t n_lengths; /* uint32 per read (0 for bad-channel / flagged reads): N50 input; complete after
                                          ppqc_finalize_extremes */
    /* offsets (in u64 elements) of each table inside sum_buf */
    uint64_t off_totals, off_len_log2, off_len_lin, off_q_hist, off_gc_hist, off_hour_reads, off_hour_bases,
        off_chan_reads, off_chan_bases, off_chan_fail, off_cmh_reads, off_cmh_bases;
    void* sel_buf; uint64_t n_sel;   /* N50 radix-select histogram (PPQC_N_SEL u64, SUM); sel_buf == sum_buf + n_sum, so
                                        [sum_buf | sel_buf] can be all-reduced in one call of n_sum + n_sel words */
} ppqc_device_tables;

/* GPU time of the stages of the most recent calls, measured with hipEvents on the library's stream. */
typedef struct ppqc_timing {
    double tile_kernel_ms_total;   /* per-read segmented-reduction kernel, summed since create / reset_timings */
    uint64_t tile_kernel_launches;
    uint64_t tile_kernel_bytes;    /* algorithmic bytes of those launches: 2*n_bytes + 50*n_reads */
    double submit_ms_last;         /* whole last submit (H2D + kernels) */
    double finalize_ms_last;       /* extremes + tables + sort + N50 of the last finalize */
    double sort_ms_last;           /* radix sort + N50 part */
} ppqc_timing;

typedef struct ppqc_ctx ppqc_ctx;

const char* ppqc_last_error(void);
int ppqc_abi_version(void);
int ppqc_device_count(int* n);
int ppqc_config_default(ppqc_config* cfg);

/* One context per GPU replaces the pool of Consumer processes of lib/seq_routines.py:916-923. */
int ppqc_create(ppqc_ctx** out, int device, const ppqc_config* cfg);
int ppqc_destroy(ppqc_ctx* ctx);
/* Forget all reads submitted so far (a new run on the same context). */
int ppqc_reset(ppqc_ctx* ctx);

int ppqc_alloc_pinned(void** p, size_t bytes);
int ppqc_free_pinned(void* p);

/* Replaces the per-read work of get_content + the GC() call (lib/fast5_reader.py:36-71).  Asynchronous:
 * H2D copies on one of two staging slots (double buffer), then the kernels; the host batch must stay
 * alive until ppqc_sync.  slot is 0 or 1. */
int ppqc_submit(ppqc_ctx* ctx, const ppqc_batch* host_batch, int slot);
/* Same, for a batch already resident in HBM (kernel-only path). */
int ppqc_submit_device(ppqc_ctx* ctx, const ppqc_batch* device_batch);
/* Replaces the per-row work of summary_reader (lib/seq_routines.py:838-842, :852-855): appends pre-computed reads. */
int ppqc_submit_records(ppqc_ctx* ctx, const ppqc_records* host_records);
/* FASTQ text straight to per-read statistics, tokenised on the GPU (no host parsing at all): replaces
 * SeqIO.parse(handle, "fastq") + GC() of fastq_reader (lib/seq_routines.py:47-53) for a whole file (or any text that
 * consists of complete records).  Strict 4-line records ("\n" or "\r\n"); channel and start time come from the
 * albacore header fields ch= / start_time= (0 / INT64_MIN if absent, as ppqc_fastq_pack), every read gets `mux`.
 * The number of records appended is returned in *n_records.  Malformed records (no '@', no '+', differing captions,
 * sequence and quality lengths differ) are detected on the device: the call itself fails if the line count is not a
 * multiple of 4, otherwise ppqc_fetch_tables / ppqc_read_stats fail with the first bad record's number.  White space
 * where SeqIO (lib/seq_routines.py:49-50) and ppqc_fastq_scan refuse it -- blanks or tabs inside a sequence line, white
 * space inside a quality line, a bare CR anywhere but at a line end -- is found by a balance over the whole text (every
 * white-space byte the scan pass sees must be a stripped line end or lie in a title) and fails the same two calls,
 * without a record number; a \v or \f inside a sequence line, which SeqIO would take for a base, is refused as well.
 * Quality bytes outside '!'..'~' are counted (ppqc_input_errors) and clamped as in ppqc_submit.  The host variant
 * copies through staging slot 0|1 like ppqc_submit; the device variant needs a 16-byte aligned pointer. */
int ppqc_submit_fastq(ppqc_ctx* ctx, const uint8_t* host_text, uint64_t n, int slot, uint8_t mux, uint64_t* n_records);
int ppqc_submit_fastq_device(ppqc_ctx* ctx, const uint8_t* device_text, uint64_t n, uint8_t mux, uint64_t* n_records);
int ppqc_sync(ppqc_ctx* ctx);
/* Blocks until the H2D copies of the newest ppqc_submit / ppqc_submit_fastq on staging slot 0|1 have completed, i.e. until
 * that call's host batch may be overwritten -- without waiting for the kernels behind them as ppqc_sync does.  It is
 * what lets a loader thread refill a pinned host buffer while the other slot's batch is in flight (the reference's
 * counterpart is the worker pool pulling the next task while results are still being merged, lib/seq_routines.py:916-942).
 * Returns at once if the slot has not been used. */
int ppqc_slot_wait(ppqc_ctx* ctx, int slot);
/* Input problems found on the device since create / reset, five words: out[0] malformed rows of device CSR batches,
 * out[1] malformed FASTQ records, out[2] number of the first of those within its text (UINT64_MAX if none), out[3]
 * quality bytes outside '!'..'~' seen by the GPU tokenizer (reported like n_bad_qual of ppqc_fastq_pack), out[4] the
 * white-space balance of the FASTQ texts (see ppqc_submit_fastq; 0 when every white-space byte is where it may be).
 * [0], [1] and [4] also make ppqc_fetch_tables / ppqc_read_stats fail.  Synchronises. */
int ppqc_input_errors(ppqc_ctx* ctx, uint64_t* out5);

int ppqc_n_reads(ppqc_ctx* ctx, uint64_t* n);
/* Per-read records [first, first+n) (the ChrList entries of lib/fast5_reader.py:127, minus the hour). */
int ppqc_read_stats(ppqc_ctx* ctx, const ppqc_per_read* out, uint64_t first, uint64_t n);
/* Per-read hour bin as used by the tables (int32, -1 = not binned); valid after ppqc_finalize_tables. */
int ppqc_read_hours(ppqc_ctx* ctx, int32_t* out, uint64_t first, uint64_t n);

/* Run-level aggregation (replaces parsing_func's aggregation + the sums of FastStats), in phases so a
 * multi-GPU caller can all-reduce between them:
 *   ppqc_finalize_extremes -> [all-reduce MAX of max_buf | min_buf (one call)]
 *   ppqc_finalize_tables   -> [all-reduce SUM sum_buf]
 *   ppqc_finalize_n50(ctx, NULL, 0) or, multi-GPU, with the all-gathered device array of lengths
 *   ppqc_fetch_tables
 * ppqc_finalize runs all four for the single-GPU case.
 * On a context fresh from ppqc_create or ppqc_reset in time_mode 0, ppqc_submit / ppqc_submit_device also start, on the
 * context's auxiliary stream, the part of ppqc_finalize_extremes that needs the CSR index only (lengths, per-channel
 * extremes, time range, hour bins) beside the batch's per-read kernel; ppqc_finalize_extremes then enqueues nothing,
 * and ppqc_finalize_n50 waits on the HOST for the batches' index kernels (microseconds, while the per-read kernel
 * runs) to learn how many radix passes the lengths need.  Records / FASTQ-text input, a reference time set after the
 * batches, a second finalize of one run and time_mode 1 take the finalize-time kernels; the results are the same. */
int ppqc_finalize_extremes(ppqc_ctx* ctx);
/* Smallest and largest start time over the reads with a valid channel (INT64_MAX / INT64_MIN if there is none); valid
 * after ppqc_finalize_extremes.  With ppqc_set_ref_time it gives RefTime = the earliest start when the caller has no
 * better one (the albacore branch does exactly this: ref_start = min start, lib/seq_routines.py:840-841).
 * ppqc_set_ref_time takes effect in the next ppqc_finalize_extremes (time_mode 0) / ppqc_finalize_tables (time_mode 1). */
int ppqc_time_range(ppqc_ctx* ctx, int64_t* out2);
int ppqc_set_ref_time(ppqc_ctx* ctx, int64_t ref_time);
int ppqc_finalize_tables(ppqc_ctx* ctx);
int ppqc_finalize_n50(ppqc_ctx* ctx, const uint32_t* device_lengths, uint64_t n);
int ppqc_fetch_tables(ppqc_ctx* ctx, ppqc_tables* out);
/* The same without the copy: the table pointers of `out` are SET to point into the context's pinned host result
 * buffer (read-only for the caller; overwritten by the next fetch, released by ppqc_destroy).  For callers that consume
 * the tables at once, like the loop of FastStats over ChannelDict (lib/seq_routines.py:97-144). */
int ppqc_fetch_tables_view(ppqc_ctx* ctx, ppqc_tables* out);
int ppqc_finalize(ppqc_ctx* ctx, ppqc_tables* out);
int ppqc_finalize_view(ppqc_ctx* ctx, ppqc_tables* out);   /* ppqc_finalize ending in ppqc_fetch_tables_view */
int ppqc_device_tables_view(ppqc_ctx* ctx, ppqc_device_tables* out);

/* Multi-GPU N50 without gathering the lengths (no reference counterpart): exact MSD radix SELECT of the
 * weighted median.  Round r = 0, 1, 2 looks at key bits [21,32), [10,21), [0,10).  ppqc_n50_select_hist adds, for the
 * local lengths that match the bits chosen so far, the sum of lengths per digit into sel_buf; the caller all-reduces
 * sel_buf (SUM; round 0 rides along with sum_buf); ppqc_n50_select_pick then chooses, from the top digit down, the
 * first digit at which twice the weight of all longer-or-equal reads reaches the merged total (totals[N_BASES] of
 * sum_buf), and clears sel_buf.  After the pick of round 2 the N50 is final and ppqc_fetch_tables may be called.
 * Everything is enqueued on the context's stream; nothing synchronises with the host.  With one GPU the three
 * hist/pick pairs back to back give the same N50 as the sort-based ppqc_finalize_n50. */
int ppqc_n50_select_hist(ppqc_ctx* ctx, int round);
int ppqc_n50_select_pick(ppqc_ctx* ctx, int round);

/* No counterpart in the reference (it has one process per channel and no N50).  Multi-GPU callers whose batches left
 * their index-only results at submit time (see ppqc_finalize_extremes; ppqc_index_early says so) can merge everything
 * that does not depend on the bases WHILE the per-read kernel runs, on a stream of their own:
 *   ppqc_stream_wait_index(ctx, side)                     side waits for the batches' index kernels
 *   all-reduce MAX of [max_buf | min_buf] on side
 *   for round 0, 1, 2: ppqc_n50_select_early(ctx, round, 0, side); all-reduce SUM of sel_buf on side;
 *                      ppqc_n50_select_early(ctx, round, 1, side)     (the total weight is taken from round 0's merged
 *                                                                      histogram, not from the tables)
 *   the context's stream waits for side; ppqc_finalize_extremes; ppqc_finalize_tables; all-reduce SUM of sum_buf;
 *   ppqc_fetch_tables
 * which leaves ONE collective behind the per-read kernel instead of four.  PPQC_E_STATE if the run does not qualify
 * (then use the phases above).
 * The (channel, mux, hour) cells -- 2 x n_channels x n_mux x n_time_bins words, all but a sliver of sum_buf: 13.8 of
 * 13.9 MB at 3000 x 4 x 72 -- depend on the index alone as well, and the batches leave them with their index kernels.
 * Between ppqc_stream_wait_index and the context's stream waiting for side:
 *   ppqc_cells_early(ctx, 0, side)   the cells final in sum_buf[off_cmh_reads, n_sum) behind side
 *   all-reduce SUM of that range on side
 *   ppqc_cells_early(ctx, 1, side)   its copy to the host result buffer, on side
 * and the all-reduce behind ppqc_finalize_tables then covers sum_buf[0, off_cmh_reads) only (a few thousand words), as
 * does the copy of ppqc_fetch_tables.  Every rank may make these calls whenever ppqc_index_early says yes (a rank whose
 * batches could not leave their cells -- 2^24 reads or 2^40 bases and more -- computes them inside the call, later).
 * Not with PPQC_COMPAT_MULTI_READ_SINGLE (PPQC_E_STATE): that rule rewrites cells behind the tables. */
int ppqc_index_early(ppqc_ctx* ctx, int* out);
int ppqc_stream_wait_index(ppqc_ctx* ctx, void* stream);
int ppqc_n50_select_early(ppqc_ctx* ctx, int round, int phase, void* stream);
int ppqc_cells_early(ppqc_ctx* ctx, int phase, void* stream);

/* time_mode 1 when the reads of a channel are spread over several GPUs (the reference's workers each hold one channel's
 * reads, lib/seq_routines.py:909-915, and apply the hour rule of lib/fast5_reader.py:117-126 to them in start-time
 * order).  The rule needs three small per-channel tables the ranks agree on, not the reads themselves:
 *   ppqc_finalize_extremes; [all-reduce MAX of max_buf | min_buf]
 *   ppqc_mode1_dist_phase(ctx, 0)   first[ch] = ~(earliest start time of the channel)        -> all-reduce MAX of `first`
 *   ppqc_mode1_dist_phase(ctx, 1)   ties[ch] = reads starting at that time; per channel and day (PPQC_MODE1_DAYS days
 *                                   from the day before the reference time) the latest and the complemented earliest
 *                                   start time of the other reads   -> all-reduce SUM of `ties`, all-reduce MAX of `days`
 *   ppqc_mode1_dist_phase(ctx, 2)   per channel the start time from which reads are bumped by 24, then every read's hour
 *   ppqc_finalize_tables            (finds the hours done and does not sort)
 * With one rank (no collectives) the hours are those of the sort-based path of ppqc_finalize_tables.  Reads outside the
 * day window are counted in ties[n_channels] (merged like the rest): ppqc_fetch_tables then fails with PPQC_E_RANGE on
 * every rank.  ppqc_mode1_dist_view allocates the tables on first use; all three are int64 device arrays. */
#define PPQC_MODE1_DAYS 64
typedef struct ppqc_mode1_tables {
    void* first; uint64_t n_first;   /* n_channels */
    void* ties;  uint64_t n_ties;    /* n_channels + 1 */
    void* days;  uint64_t n_days;    /* 2 * n_channels * PPQC_MODE1_DAYS */
} ppqc_mode1_tables;
int ppqc_mode1_dist_view(ppqc_ctx* ctx, ppqc_mode1_tables* out);
int ppqc_mode1_dist_phase(ppqc_ctx* ctx, int phase);

/* PPQC_COMPAT_MULTI_READ_SINGLE when a job's reads are spread over several GPUs: `len(l) > 1` of channel_parser
 * (lib/multi_read_fast5_reader.py:64) is the channel's read count in the whole job.  Between ppqc_finalize_extremes and
 * ppqc_finalize_tables, ppqc_compat_dist_count counts this context's reads per channel into a device array of n_channels
 * uint64 (returned in *counts / *n; enqueued on the context's stream); the caller all-reduces it (SUM), and
 * ppqc_finalize_tables then applies the rule to the channels whose merged count is one -- such a channel's only read is
 * on one rank, whose rewritten cells the table all-reduce adds to zeros.  Without this call (one GPU) the context's own
 * channel table is the job's.  ppqc_group_finalize does it by itself. */
int ppqc_compat_dist_count(ppqc_ctx* ctx, void** counts, uint64_t* n);

/* Makes the context's GPU the calling thread's current device.  For loader threads that call ppqc_alloc_pinned /
 * ppqc_free_pinned (hipHostMalloc initialises the CURRENT device of the calling thread, device 0 for a fresh thread,
 * whichever GPU the context lives on).  The reference's counterpart is a worker process inheriting its parent's
 * state at fork (lib/seq_routines.py:916-923). */
int ppqc_bind_thread(ppqc_ctx* ctx);

/* No counterpart in the reference.  Test and measurement hook: where the library has more than one code path for the
 * same result (which of the two per-read kernels takes a tile, the tile size, index kernels at submit or at finalize
 * time, how the GPU FASTQ tokenizer sums its lines), `value` -1 leaves the choice to the library -- the state of every
 * new context -- and other values force one path for this context: "tile_dense" 0|1, "dense_len" bytes, "tile_all" 0|1,
 * "tile_shift" 12..16, "pre" 0|1, "fastq_sums" 0|1|2, "fastq_hdr_aux" 0|1|2, "fastq_one_pass" 0|1, "fq1_regions" n,
 * "sort_beside" 0|1, "n50_select" 0|1 (the local N50 by sort / by radix select), "fastq_extremes" 0|1 (the extremes of a
 * FASTQ text's reads by a kernel of their own / by the tables kernel of ppqc_finalize), "fq1_early" 0|1 (the one-pass finalize kernel launched when the host
 * knows the line count / straight behind the scan) (pypore_amd/csrc/ppqc_internal.hpp, Tuning).  Results never depend on it (the parity tests run the paths against each
 * other); the library reads no environment variable. */
int ppqc_set_tuning(ppqc_ctx* ctx, const char* name, int value);
/* No counterpart in the reference.  Counters of the same code paths, for the tests that must know which one ran:
 * "fastq_one_pass" texts that went through the one-pass kernels, "fastq_declined" texts those kernels handed to the
 * general path, "fastq_general" texts the general path took, "fastq_early" one-pass texts whose finalize kernel ran
 * from the launch made before the line count was known (ppqc_submit_fastq[_device], since ppqc_create). */
int ppqc_path_count(ppqc_ctx* ctx, const char* name, uint64_t* out);

/* Several GPUs driven by ONE process, without torch: the counterpart of gathering the workers' results over the result
 * queue and summing over channels (lib/seq_routines.py:940-942, :103-108) for a caller that binds this library through
 * ctypes / cgo / JNI and has no collective library of its own.  ppqc_group_create takes n contexts on n different GPUs
 * and opens an RCCL communicator over them (ncclCommInitAll; librccl is loaded with dlopen at this call, so the
 * library itself has no link-time dependency on it: PPQC_E_NO_DEVICE if it cannot be loaded).  ppqc_group_finalize runs
 * the phased finalize on every context with the all-reduces in between -- MAX over [max_buf | min_buf], in time_mode 1
 * the three merges of ppqc_mode1_dist_phase, with PPQC_COMPAT_MULTI_READ_SINGLE the channel counts of
 * ppqc_compat_dist_count, SUM over [sum_buf | sel_buf] and the two further N50-select rounds --
 * each on its context's stream, and fetches the merged tables (the same on every context) from the first one.
 * One process per GPU over torch.distributed is the other way (pypore_amd/merge.py); both enqueue the same kernels. */
typedef struct ppqc_group ppqc_group;
int ppqc_group_create(ppqc_group** out, ppqc_ctx* const* ctxs, int n);
int ppqc_group_finalize(ppqc_group* group, ppqc_tables* out);
int ppqc_group_destroy(ppqc_group* group);

/* Enqueue all further work of the context on a caller-owned HIP stream (use_external != 0; `stream` is a hipStream_t,
 * 0 = the null stream) so that it is stream-ordered with the caller's own kernels and RCCL collectives without host
 * synchronisation, or go back to the context's own stream (use_external == 0).  Drains the previous stream first. */
int ppqc_set_stream(ppqc_ctx* ctx, void* stream, int use_external);

int ppqc_timings(ppqc_ctx* ctx, ppqc_timing* out);
int ppqc_reset_timings(ppqc_ctx* ctx);

/* Stand-alone device utilities (no reference counterpart; used by bench.py and the tests).
 * Hand-written LSD radix sort of n uint32 keys in device memory (ascending, in place). */
int ppqc_sort_u32_device(ppqc_ctx* ctx, uint32_t* device_keys, uint64_t n);
/* No counterpart in the reference.  ppqc_sort_u32_device for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the sort's ping-pong buffer and digit counters
 * are a static device array inside the library (about 10.3 MB per GPU, held until the process ends), so the call
 * allocates nothing.  At most PPQC_STATIC_SORT_MAX_KEYS keys (PPQC_E_RANGE beyond).  Every call on one GPU shares that
 * GPU's array: calls that overlap on one GPU (two contexts, two threads) race -- serialise them. */
#define PPQC_STATIC_SORT_MAX_KEYS (1ull << 21)
int ppqc_sort_u32_device_static(ppqc_ctx* ctx, uint32_t* device_keys, uint64_t n);
/* Counter-based synthetic read generator: fills device bases/quals for stream bytes [first, first+n). */
int ppqc_synth_fill_device(ppqc_ctx* ctx, uint8_t* device_bases, uint8_t* device_quals, uint64_t first,
                           uint64_t n, uint64_t seed);

/* Host-side FASTQ -> CSR packer (no GPU involved).  Replaces the record parsing the reference delegates to
 * Biopython: SeqIO.parse(handle, "fastq") at lib/seq_routines.py:50 and SeqIO.read(StringIO(..), "fastq") at
 * lib/fastqparser.py:12; bases = line 2 (lib/fast5_reader.py:57), quals = line 4.
 * ppqc_fastq_scan counts the complete 4-line records in text[0, n) (final != 0: the block ends the file, so a last
 * line without '\n' is complete) and reports how many bytes they span; malformed records fail with PPQC_E_INVALID
 * where SeqIO raises ValueError.  ppqc_fastq_pack then copies sequences and qualities into caller buffers (n_bases
 * bytes each), fills offsets[1..n_records] continuing from the caller-set offsets[0], and optionally the header span,
 * the albacore header fields ch= and start_time= (ISO 8601, UTC seconds; INT64_MIN if absent) and the number of
 * quality bytes outside '!'..'~'. */
int ppqc_fastq_scan(const uint8_t* text, uint64_t n, int final, uint64_t* n_records, uint64_t* n_bases,
                    uint64_t* consumed);
int ppqc_fastq_pack(const uint8_t* text, uint64_t n, uint64_t n_records, uint8_t* bases, uint8_t* quals,
                    uint64_t* offsets, uint64_t* header_off, uint32_t* header_len, uint32_t* channel,
                    int64_t* start_time_s, uint64_t* n_bad_qual);

#ifdef __cplusplus
}
#endif
#endif /* PYPORE_QC_H */
