// Host side of libpypore_qc.so: the extern "C" entry points of include/pypore_qc.h.
// Owns all device memory, one compute stream and two copy streams (double-buffered H2D) per context.
#include <math.h>
#include <stdio.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "ppqc_internal.hpp"

using namespace ppqc;

// Scratch of ppqc_sort_u32_device_static, the entry point for callers that cannot allocate (include/pypore_qc.h):
// [ping-pong keys: kStaticSortMaxKeys u32 | digit counters of the radix passes].  One copy per GPU, held until the
// process ends.  The name is a host-side stand-in, not a device address: host code only ever uses the per-device
// address that hipGetSymbolAddress returned (g_static_scratch), taken in ppqc_create with that device current.
constexpr uint64_t kStaticSortMaxKeys = PPQC_STATIC_SORT_MAX_KEYS;
constexpr uint64_t kStaticSortCountWords = 256ull * 2048 + 256;   // sort_count_words(n) never exceeds this (<= 2048 waves)
constexpr uint64_t kStaticScratchBytes = kStaticSortMaxKeys * 4 + kStaticSortCountWords * 4;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
static unsigned char* g_static_scratch[kMaxDevices];   // per device; written once under g_static_mu

static std::mutex g_static_mu;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int ppqc_set_error(int code, const std::string& msg) { return fail(code, msg); }   // for the other host units
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? PPQC_E_NOMEM : PPQC_E_HIP,                    \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)
#define CHK(expr)                     \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != PPQC_OK) return rc_; \
    } while (0)

// The four streams of a context (main, auxiliary, two copy streams) come from a per-device pool and go back to it when
// the context is destroyed: a process that creates contexts by the hundred (a test run, a service with a context per
// request) would otherwise keep creating and destroying hardware queues, and on this stack a process that has been
// through a few hundred of them -- or three processes with a few each on one GPU -- was seen to stall for good on a wait
// between two of its streams (tests hanging in hipEventSynchronize / hipStreamSynchronize with nothing left to run).
// The pool is as large as the largest number of contexts alive at once and is never freed.
struct StreamSet { hipStream_t own = nullptr, aux = nullptr, copy[2] = {nullptr, nullptr}; };
static std::mutex g_streams_mu;
static std::vector<StreamSet> g_streams[kMaxDevices];
static int take_streams(int device, StreamSet* out) {
    {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        if (device >= 0 && device < kMaxDevices && !g_streams[device].empty()) {
            *out = g_streams[device].back();
            g_streams[device].pop_back();
            return PPQC_OK;
        }
    }
    StreamSet s;
    HIPCHK(hipStreamCreateWithFlags(&s.own, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s.aux, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamCreateWithFlags(&s.copy[i], hipStreamNonBlocking));
    *out = s;
    return PPQC_OK;
}
static void give_streams(int device, const StreamSet& s) {   // the streams are idle (the caller has synchronised the device)
    if (!s.own) return;
    if (device >= 0 && device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        g_streams[device].push_back(s);
        return;
    }
    (void)hipStreamDestroy(s.own); (void)hipStreamDestroy(s.aux);
    for (int i = 0; i < 2; ++i) (void)hipStreamDestroy(s.copy[i]);
}

namespace {

struct Slot {   // device staging for one host batch
    uint8_t *bases = nullptr, *quals = nullptr, *mux = nullptr, *ff = nullptr;
    uint64_t* offsets = nullptr;
    uint32_t* channel = nullptr;
    int64_t* time = nullptr;
    uint64_t cap_bytes = 0, cap_reads = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
    bool used = false;   // `copied` has been recorded at least once
};

struct EvPair {
    hipEvent_t a, b;
    uint64_t bytes;
};

}  // namespace

// ---- the context's state: one phase, and what is known about the buffers ------------------------------------------------
// `state` is the phase of the run-level aggregation: 0 reads only -> 1 extremes -> 2 tables -> 3 N50 (fetchable).  Every
// call that adds reads, and ppqc_reset, puts it back to 0; a phase function checks the phase it needs (PPQC_E_STATE) and
// sets its own.  The booleans beside it are facts about buffers and streams that decide WHERE work is enqueued, never
// WHAT is computed; each has one place that makes it true and a short list that makes it false:
//
//   fact          true from                                   false from                                   read by
//   res_clean     ppqc_create / ppqc_reset (results cleared   ppqc_finalize_extremes (the buffer is in      run_batch (index kernel at submit time),
//                 on the main stream, ev_clean)                use from then on)                            finalize_extremes (no clear needed)
//   pre_ok        ppqc_create / ppqc_reset                    a batch that could not leave its index-only   run_batch, finalize_extremes,
//                                                              results early (time_mode 1, buffer in use),  ppqc_index_early / _stream_wait_index /
//                                                              records / FASTQ text input, a reference      _n50_select_early
//                                                              time that changes after batches
//   pre_run       finalize_extremes when res_clean && pre_ok  ppqc_reset, the next finalize_extremes        finalize_n50 (host knows the largest key)
//                 && reads (this finalize relies on them)
//   lens_early    ppqc_create / ppqc_reset                    records / FASTQ text input (their lens32 is   finalize_extremes, finalize_n50,
//                                                              written by the extremes kernel)              n50_select_hist (where the sort forks)
//   n50_early     ppqc_n50_select_early round 2               any new batch, ppqc_reset                      finalize_tables (the finalize ends there)
//   init_on_aux   finalize_extremes when the clear of the     finalize_extremes when an earlier phase may   n50_select_hist (round 0 may use aux)
//                 results precedes everything on aux           still run (res_busy): clear stays in line
//   res_busy      every finalize phase (mark_busy)            ppqc_sync (and all that call it)              finalize_extremes
//   sum_fetched   ppqc_finalize's early result copy           finalize_extremes, fetch_tables                fetch_tables (no second copy; N50 from the
//                                                                                                           mirror's spare word)
//   m1_phase      ppqc_mode1_dist_phase (0, 1, 2 in order)    finalize_extremes (-1), a reference time      finalize_tables (2: hours are done)
//                                                              that changes (-1: the hours are stale)
//   cells_early   ppqc_create / ppqc_reset (not with          a batch without index kernel, records / FASTQ  run_batch, finalize_tables, ppqc_cells_early
//                 PPQC_COMPAT_MULTI_READ_SINGLE)              text, a run whose reads or bases may overflow
//                                                              a packed cell
//   cells_dirty   a batch that left cells                     every clear of the result buffer               finalize_tables (clears the region first)
//   cells_done    ppqc_cells_early(0), ppqc_finalize          any new batch, ppqc_reset, a clear             finalize_tables (leaves the cells alone)
//   cmh_on_host   ppqc_cells_early(1), ppqc_finalize          as cells_done                                  the result copies (small tables only)
//   cc_ready      ppqc_compat_dist_count                      finalize_extremes                             finalize_tables (job-wide channel counts)
//   index_on_main a FASTQ text that left its index-only      a CSR batch that left its own (auxiliary       ppqc_finalize (cells: unpack and copy on the
//                 results (its finalize kernel, main stream)  stream), ppqc_reset                            copy stream), finalize_n50 (no host wait)
//   gate_max_run  ppqc_create / ppqc_reset                    a CSR batch, records, a text without index     fastq_committed, finalize_extremes / _tables
//                 (every read so far came from a FASTQ text:   results (they raise the gate by OR)           (the gate's largest word is the largest
//                 the gate's words hold maxima)                                                              length: tables_kernel<true>)
//   ext_deferred  a FASTQ text whose extremes were left to    join_index (the extremes kernel, when someone  finalize_tables (tables_kernel<true>),
//                 the tables kernel (gate_max_run)            asks before the tables), finalize_tables,      ppqc_finalize_extremes, _stream_wait_index
//                                                              ppqc_reset, a finalize that takes every read
//   fin_pending / end_split / fin_start / sort_start / sort_end: timing bookkeeping only (resolve_finalize_timings).
//
// tests/test_gpu_sequences.py drives random interleavings of every entry point against the oracle; any flag read before it
// is defined on some path shows up there as a table that differs.
struct ppqc_ctx {
    int device = 0, n_cu = 256;
    Tuning tn{};   // ppqc_set_tuning: all -1 (the library's own choices) unless a test or a measurement says otherwise
    ppqc_config cfg{};
    TableParams tp{};
    QThr qthr{};
    uint64_t* tab_slab = nullptr;   // tables_kernel: one slab of private tables per block (tables_slab_words)
    hipStream_t stream = nullptr, own_stream = nullptr, copy[2] = {nullptr, nullptr};
    hipStream_t aux = nullptr;      // the sort + N50 chain runs here, beside the tables kernel on `stream`
    hipEvent_t ev_join = nullptr, ev_init = nullptr;
    bool init_on_aux = false;   // the last clear of the result buffer went to the auxiliary stream
    bool res_clean = false;  // ppqc_reset cleared the result buffer on the main stream and nothing has used it since
    hipEvent_t fin_start = nullptr, sort_start = nullptr;   // start events of the finalize / sort timings of this step
    bool fin_pending = false; // the last finalize's events have not been turned into timings yet
    uint64_t reads_gen = 0, fin_gen = 0;   // ev_reads is recorded per batch and doubles as the start of a finalize's timings:
                                           // its generation then and at resolve time tell whether it still is that start
    bool end_split = false;  // the finalize ended on two streams: its end is the later of ev_t1 and sort_end
    bool res_busy = false;   // a finalize phase (or a caller's collective on the tables) may still run: set by every
                             // finalize phase, cleared when the stream has been synchronised
    hipEvent_t last_a = nullptr, sort_end = nullptr;   // start event of the newest batch (submit_ms_last)
    hipEvent_t last_b = nullptr;   // behind the newest batch's per-read kernel(s), in front of its finalize_reads
    bool last_dense_alone = false; // ... which was dense_kernel alone (a batch of short reads)
    uint64_t* lut_dev = nullptr;

    ReadArrays ra{};
    uint64_t n_reads = 0, cap_reads = 0;

    Partial *head = nullptr, *tail = nullptr;
    uint64_t cap_tiles = 0;
    DenseList dense{nullptr, 0};   // [2 x 64 counters | entries]: tiles the per-read kernel leaves to dense_kernel
    DenseList slow{nullptr, 0};    // same layout: tiles dense_kernel leaves to its SLOW instantiation

    Slot slot[2];

    // one device allocation [sum | max | min | tmin tmax | n50 total], mirrored by one pinned host buffer
    uint64_t *res_buf = nullptr, *res_host = nullptr, res_words = 0;
    uint64_t *sum_buf = nullptr, *sel_buf = nullptr, *max_buf = nullptr, *min_buf = nullptr, *n50_out = nullptr,
             *sel_state = nullptr;
    int64_t* tminmax = nullptr;
    int state = 0;   // 0 reads only, 1 extremes done, 2 tables done, 3 n50 done
    bool lens_early = true;    // every read of the run came through finalize_reads: lens32 and the sort gate are ready
    hipEvent_t ev_reads = nullptr;   // behind the last finalize_reads of the run
    bool sum_fetched = false;   // ppqc_finalize: everything but the N50 was copied to res_host behind the tables kernel,
                                // and the N50 kernels write their result into res_host themselves
    uint64_t n50 = 0;

    // sort / mode-1 scratch
    uint32_t *sk32 = nullptr, *sk32_tmp = nullptr, *counts = nullptr, *vals = nullptr, *vals_tmp = nullptr;
    uint64_t *sk64 = nullptr, *sk64_tmp = nullptr, *n50_blocks = nullptr;
    uint32_t *chan_start = nullptr, *first_desc = nullptr;
    uint64_t cap_sort32 = 0, cap_sort64 = 0;

    // FASTQ-on-GPU scratch
    FastqScratch fq{};
    ReadArrays lra{};           // rec_* only: raw sums per text line
    uint64_t cap_chunks = 0, cap_lines = 0;
    Fq1Scratch fq1{};           // the one-pass kernels (ppqc_fastq1.hip)
    uint64_t cap_fq1_lines = 0, cap_fq1_regions = 0;
    uint64_t n_fq1 = 0, n_fq1_declined = 0, n_fq_general = 0, n_fq1_early = 0;   // ppqc_path_count
    uint64_t fq1_ticket = 0;      // launches of the one-pass scan so far (the regions kernel hands the newest back, run_fastq_one_pass)
    int64_t fq_line_hint = -1;  // mean bytes per line of the last FASTQ text (-1: none yet); a hint for the scan pass only

    std::vector<EvPair> pending;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    hipEvent_t ev_max = nullptr;   // behind the extremes kernel: fork point of the sort chain when lens32 comes from it
    // Index-only work of a batch (lengths, per-channel extremes, time range, hours, lens32) runs on the auxiliary stream
    // when the batch is submitted, beside its tile kernel (extremes_kernel<true>), as long as the result buffer is as
    // ppqc_reset left it and nothing the hours depend on changes: the finalize then needs no extremes kernel, and the
    // sort + N50 chain, which needs only lens32, runs beside the tile kernel as well.
    bool pre_ok = false;    // every batch since ppqc_reset went that way
    bool index_on_main = false;   // ... the newest of them (a FASTQ text) in line on the main stream, not on the auxiliary one
    bool gate_max_run = true;     // every read of the run so far came from a FASTQ text: the largest gate word is the largest length
    bool ext_deferred = false;    // the extremes of the run's reads (all from FASTQ texts) have been left to the tables kernel of
                                  // ppqc_finalize; whoever needs them before that kernel gets them from join_index
    bool pre_run = false;   // ... and the finalize in progress relies on it (set by ppqc_finalize_extremes)
    bool n50_early = false; // ppqc_n50_select_early has left the N50 of the run's reads (no batch since)
    hipEvent_t ev_pre = nullptr;   // behind the newest of those kernels (auxiliary stream)
    hipEvent_t ev_fq = nullptr;    // FASTQ text: the newlines are counted (the host waits for it)
    hipEvent_t ev_hdr = nullptr;   // FASTQ text: behind the header kernel when it runs on the auxiliary stream
    hipEvent_t ev_clean = nullptr; // behind the clear of the result buffer (ppqc_reset / ppqc_create, main stream)
    uint64_t* maxlen_host = nullptr;   // a few pinned scratch words (line count of the GPU tokenizer)
    uint64_t* m1 = nullptr;            // time_mode 1 across ranks: [first | ties | days] (ppqc_mode1_dist_view)
    int m1_phase = -1;                 // last phase of ppqc_mode1_dist_phase run since ppqc_finalize_extremes (2: hours done)
    // The (channel, mux, hour) cells depend on the CSR index alone: a batch leaves them at submit time with its other
    // index-only results (packed, one atomic per read), and whatever follows -- the unpack, the copy of the 2 x cells words
    // to the host mirror, a caller's all-reduce -- can run beside the per-read kernel (ppqc_cells_early, finalize_impl).
    uint64_t bases_total = 0;          // upper bound of the run's bases so far (cmh_packable)
    bool cells_early = false;          // every batch since ppqc_reset left its cells
    bool cells_dirty = false;          // some batch did: the region is not as the clear left it
    bool cells_done = false;           // the cells are final and unpacked (no batch since): finalize_tables leaves them alone
    bool cmh_on_host = false;          // ... and the host mirror holds them (behind ev_cmhcopy)
    hipEvent_t ev_cells = nullptr, ev_cmhcopy = nullptr;
    uint64_t* cc = nullptr;            // PPQC_COMPAT_MULTI_READ_SINGLE across ranks: reads per channel (ppqc_compat_dist_count)
    bool cc_ready = false;             // ... counted since ppqc_finalize_extremes: ppqc_finalize_tables takes them as the job's
    ppqc_timing timing{};
};

static hipEvent_t get_event(ppqc_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

template <typename T>
static int grow(T** p, uint64_t old_n, uint64_t new_cap, hipStream_t s) {
    T* q = nullptr;
    HIPCHK(hipMalloc((void**)&q, new_cap * sizeof(T) + 64));
    if (*p) {
        hipError_t e = old_n ? hipMemcpyAsync(q, *p, old_n * sizeof(T), hipMemcpyDeviceToDevice, s) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {   // the old array stays valid; the new one must not leak
            (void)hipFree(q);
            return fail(PPQC_E_HIP, std::string("growing a device array: ") + hipGetErrorString(e));
        }
        HIPCHK(hipFree(*p));
    }
    *p = q;
    return PPQC_OK;
}

static int ensure_reads(ppqc_ctx* c, uint64_t need) {
    if (need <= c->cap_reads) return PPQC_OK;
    uint64_t cap = c->cap_reads ? c->cap_reads * 2 : 1024;
    if (cap < need) cap = need;
    const uint64_t n = c->n_reads;
    hipStream_t s = c->stream;
    ReadArrays& r = c->ra;
    CHK(grow(&r.rec_cnt, 0, cap, s)); CHK(grow(&r.rec_hi, 0, cap, s)); CHK(grow(&r.rec_lo, 0, cap, s));
    CHK(grow(&r.len, n, cap, s)); CHK(grow(&r.gc, n, cap, s)); CHK(grow(&r.nn, n, cap, s));
    CHK(grow(&r.qsum, n, cap, s)); CHK(grow(&r.mean_q, n, cap, s));
    CHK(grow(&r.err_hi, n, cap, s)); CHK(grow(&r.err_lo, n, cap, s)); CHK(grow(&r.flags, n, cap, s));
    CHK(grow(&r.channel, n, cap, s)); CHK(grow(&r.mux, n, cap, s)); CHK(grow(&r.time, n, cap, s));
    CHK(grow(&r.hour, n, cap, s)); CHK(grow(&r.lens32, n, cap, s));   // both are filled batch by batch
    c->cap_reads = cap;
    return PPQC_OK;
}

static int ensure_tiles(ppqc_ctx* c, uint64_t n_tiles) {
    if (n_tiles <= c->cap_tiles) return PPQC_OK;
    uint64_t cap = c->cap_tiles ? c->cap_tiles * 2 : 1024;
    if (cap < n_tiles) cap = n_tiles;
    CHK(grow(&c->head, 0, cap, c->stream));
    CHK(grow(&c->tail, 0, cap, c->stream));
    CHK(grow(&c->dense.base, 0, dense_list_words(cap), c->stream));
    HIPCHK(hipMemsetAsync(c->dense.base, 0, 128 * 8, c->stream));   // all counters; every launch leaves the next one's at zero
    c->dense.parity = 0;
    CHK(grow(&c->slow.base, 0, dense_list_words(cap), c->stream));
    HIPCHK(hipMemsetAsync(c->slow.base, 0, 128 * 8, c->stream));
    c->slow.parity = 0;
    c->cap_tiles = cap;
    return PPQC_OK;
}

static int ensure_sort32(ppqc_ctx* c, uint64_t n) {
    if (n <= c->cap_sort32) return PPQC_OK;
    uint64_t cap = n + n / 4 + 1024;
    CHK(grow(&c->sk32, 0, cap, c->stream));
    CHK(grow(&c->sk32_tmp, 0, cap, c->stream));
    CHK(grow(&c->counts, 0, sort_count_words(cap) + 1024, c->stream));
    CHK(grow(&c->n50_blocks, 0, n50_block_words(cap), c->stream));
    HIPCHK(hipMemsetAsync(c->n50_blocks, 0, 8, c->stream));   // the N50 kernel's ticket counter: zero between launches
    HIPCHK(hipStreamSynchronize(c->stream));                  // the kernel may run on the auxiliary stream
    c->cap_sort32 = cap;
    return PPQC_OK;
}

static int ensure_sort64(ppqc_ctx* c, uint64_t n) {
    CHK(ensure_sort32(c, n));
    if (n <= c->cap_sort64) return PPQC_OK;
    uint64_t cap = n + n / 4 + 1024;
    CHK(grow(&c->sk64, 0, cap, c->stream));
    CHK(grow(&c->sk64_tmp, 0, cap, c->stream));
    CHK(grow(&c->vals, 0, cap, c->stream));
    CHK(grow(&c->vals_tmp, 0, cap, c->stream));
    c->cap_sort64 = cap;
    return PPQC_OK;
}

static int input_errors(const uint64_t* e) {
    if (e[0]) return fail(PPQC_E_INVALID, "malformed CSR in a device batch: " + std::to_string(e[0]) + " rows with bad offsets");
    if (e[1])
        return fail(PPQC_E_INVALID, "malformed FASTQ: " + std::to_string(e[1]) + " bad records, the first is record " +
                                        std::to_string(e[2]) + " of its text (header without '@', third line without '+', "
                                        "differing captions, or sequence and quality lengths differ)");
    if (e[6])
        return fail(PPQC_E_INVALID, "malformed FASTQ: " + std::to_string((int64_t)e[6]) + " white-space bytes inside sequence "
                                        "or quality lines, or bare CR line ends (SeqIO refuses them; ppqc_fastq_scan names "
                                        "the record)");
    return PPQC_OK;
}

static int mark_busy(ppqc_ctx* c) {
    c->res_busy = true;
    return PPQC_OK;
}

// The extremes of the run's reads were left to the tables kernel of ppqc_finalize (fastq_committed) and someone needs
// them before that: extremes_kernel<true> over the run's arrays, in line.  (Reads whose extremes are there already are
// taken in again: maxima and minima do not mind.)
static int join_index(ppqc_ctx* c) {
    if (c->ext_deferred && c->n_reads) {
        BatchView v{nullptr, nullptr, nullptr, c->n_reads, 0, nullptr, kDefaultTileShift, nullptr, nullptr, nullptr};
        HIPCHK(launch_batch_extremes(v, 0, c->ra, c->tp, c->max_buf, c->min_buf, c->tminmax, c->res_buf + c->res_words + 4,
                                     c->maxlen_host + 4, nullptr, nullptr, c->stream));
        HIPCHK(hipEventRecord(c->ev_pre, c->stream));
    }
    c->ext_deferred = false;
    return PPQC_OK;
}

static uint64_t q_threshold(double q) { return (uint64_t)llrint(ldexp(pow(10.0, -q / 10.0), PPQC_ERR_SHIFT)); }

static void make_layout(const ppqc_config& cfg, TableLayout* l) {
    l->n_channels = cfg.n_channels; l->n_mux = cfg.n_mux; l->n_time_bins = cfg.n_time_bins;
    uint64_t o = 0;
    l->off_totals = o; o += PPQC_N_TOTALS;
    l->off_len_log2 = o; o += PPQC_N_LEN_LOG2;
    l->off_len_lin = o; o += PPQC_N_LEN_LIN;
    l->off_q_hist = o; o += PPQC_N_QHIST;
    l->off_gc_hist = o; o += PPQC_N_GCHIST;
    l->off_hour_reads = o; o += cfg.n_time_bins;
    l->off_hour_bases = o; o += cfg.n_time_bins;
    l->off_chan_reads = o; o += cfg.n_channels;
    l->off_chan_bases = o; o += cfg.n_channels;
    l->off_chan_fail = o; o += cfg.n_channels;
    const uint64_t cells = (uint64_t)cfg.n_channels * cfg.n_mux * cfg.n_time_bins;
    l->off_cmh_reads = o; o += cells;
    l->off_cmh_bases = o; o += cells;
    l->n_sum = o;
}

template <typename T>
static int slot_grow(T** p, uint64_t n) {
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    HIPCHK(hipMalloc((void**)p, n * sizeof(T) + 64));
    return PPQC_OK;
}

extern "C" {

const char* ppqc_last_error(void) { return g_err.c_str(); }
int ppqc_abi_version(void) { return PPQC_ABI_VERSION; }

int ppqc_device_count(int* n) {
    if (!n) return fail(PPQC_E_INVALID, "null argument");
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *n = 0; return fail(PPQC_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = k;
    return PPQC_OK;
}

int ppqc_config_default(ppqc_config* cfg) {
    if (!cfg) return fail(PPQC_E_INVALID, "null argument");
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(*cfg);
    cfg->n_channels = 512; cfg->n_mux = 4; cfg->n_time_bins = 48; cfg->bin_seconds = 3600;
    cfg->time_mode = 0; cfg->tz_offset_s = 0; cfg->ref_time_s = 0; cfg->fail_q = 7.0; cfg->max_reads_hint = 0;
    cfg->compat_flags = 0; cfg->reserved0 = 0;
    return PPQC_OK;
}

// Everything ppqc_create sets up after its argument checks.  Any failure returns at once: the caller then hands the
// half-built context to ppqc_destroy, which releases whatever is there (every member starts out null).
static int create_impl(ppqc_ctx* c, int device, const ppqc_config* cfg) {
    HIPCHK(hipSetDevice(device));
    {   // per-device address of the static scratch (ppqc_sort_u32_device_static), once per device
        std::lock_guard<std::mutex> lk(g_static_mu);
        if (device < kMaxDevices && !g_static_scratch[device]) {
            unsigned char* p = nullptr;
            HIPCHK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
            size_t sym_bytes = 0;
            HIPCHK(hipGetSymbolSize(&sym_bytes, HIP_SYMBOL(scratch)));
            if (!p || sym_bytes != kStaticScratchBytes)
                return fail(PPQC_E_HIP, "static scratch: symbol address / size not as built");
            g_static_scratch[device] = p;
        }
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        StreamSet ss;
        CHK(take_streams(device, &ss));
        c->own_stream = ss.own; c->aux = ss.aux; c->copy[0] = ss.copy[0]; c->copy[1] = ss.copy[1];
    }
    c->stream = c->own_stream;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&c->slot[i].copied, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->slot[i].done, hipEventDisableTiming));
    }
    HIPCHK(hipEventCreate(&c->ev_t0)); HIPCHK(hipEventCreate(&c->ev_t1));
    HIPCHK(hipEventCreate(&c->ev_s0)); HIPCHK(hipEventCreate(&c->ev_s1));
    HIPCHK(hipEventCreateWithFlags(&c->ev_max, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_clean, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fq, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_hdr, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_cells, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_cmhcopy, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->ev_join));
    HIPCHK(hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->ev_reads));
    HIPCHK(hipHostMalloc((void**)&c->maxlen_host, 64, hipHostMallocDefault));
    memset(c->maxlen_host, 0, 64);   // (the ticket word: pinned memory comes back as the last owner left it)

    // error-probability LUT: e[b] = round(2^52 * 10^(-q/10)), q = clamp(b - 33, 0, 93)
    uint64_t lut[256];
    for (int b = 0; b < 256; ++b) {
        int q = b - 33;
        q = q < 0 ? 0 : (q > 93 ? 93 : q);
        lut[b] = (uint64_t)llrint(ldexp(pow(10.0, -(double)q / 10.0), PPQC_ERR_SHIFT));
    }
    HIPCHK(hipMalloc((void**)&c->lut_dev, sizeof(lut)));
    HIPCHK(hipMemcpy(c->lut_dev, lut, sizeof(lut), hipMemcpyHostToDevice));


    TableParams& tp = c->tp;
    make_layout(*cfg, &tp.lay);
    tp.bin_seconds = cfg->bin_seconds; tp.time_mode = cfg->time_mode; tp.tz_offset_s = cfg->tz_offset_s;
    tp.compat_flags = cfg->compat_flags;
    tp.ref_time_s = cfg->ref_time_s; tp.fail_q = cfg->fail_q;
    c->qthr.fail = q_threshold(cfg->fail_q);
    c->qthr.fail_q = cfg->fail_q;
    for (int j = 0; j <= PPQC_N_QHIST; ++j) c->qthr.q[j] = q_threshold(6.0 + j);
    HIPCHK(hipMalloc((void**)&c->tab_slab, tables_slab_words(tp.lay) * 8));

    const uint64_t nch1 = 1 + (uint64_t)cfg->n_channels;
    c->res_words = tp.lay.n_sum + PPQC_N_SEL + 2 * nch1 + 8;
    // one extra word behind the results: the count of malformed device-CSR rows (never touched by init_results)
    //  [0] malformed device-CSR rows, [1] malformed FASTQ records, [2] number of the first of those,
    //  [3] invalid quality bytes seen by the GPU tokenizer (reported, not fatal)
    // behind the results: 4 input-error words, then at +8 the 64 words of the sort gate (see finalize_reads_kernel)
    HIPCHK(hipMalloc((void**)&c->res_buf, (c->res_words + 8 + kGateWords) * 8));
    HIPCHK(hipMemset(c->res_buf + c->res_words, 0, 16));
    HIPCHK(hipMemset(c->res_buf + c->res_words + 2, 0xFF, 8));
    HIPCHK(hipMemset(c->res_buf + c->res_words + 3, 0, (5 + kGateWords) * 8));   // [3] bad quality bytes, [8..] sort gate
    HIPCHK(hipHostMalloc((void**)&c->res_host, (c->res_words + 8) * 8, hipHostMallocDefault));
    c->sum_buf = c->res_buf;
    c->sel_buf = c->sum_buf + tp.lay.n_sum;
    c->max_buf = c->sel_buf + PPQC_N_SEL;
    c->min_buf = c->max_buf + nch1;
    c->tminmax = (int64_t*)(c->min_buf + nch1);
    c->n50_out = c->min_buf + nch1 + 2;
    c->sel_state = c->n50_out + 2;
    HIPCHK(hipMalloc((void**)&c->chan_start, (uint64_t)cfg->n_channels * 4 + 64));
    HIPCHK(hipMalloc((void**)&c->first_desc, (uint64_t)cfg->n_channels * 4 + 64));
    if (cfg->max_reads_hint) CHK(ensure_reads(c, cfg->max_reads_hint));
    // a new context starts as ppqc_reset leaves one: results cleared, batches may leave their index-only results early
    HIPCHK(launch_init_results(c->res_buf, tp.lay.n_sum + PPQC_N_SEL, nch1, true, c->stream));
    HIPCHK(hipEventRecord(c->ev_clean, c->stream));
    c->res_clean = true;
    c->pre_ok = true;
    c->cells_early = !(cfg->compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE);   // that rule rewrites cells behind the tables
    return PPQC_OK;
}


int ppqc_create(ppqc_ctx** out, int device, const ppqc_config* cfg) {
    if (!out || !cfg) return fail(PPQC_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(ppqc_config)) return fail(PPQC_E_INVALID, "ppqc_config.struct_size mismatch");
    if (cfg->n_channels == 0 || cfg->n_channels > (1u << 20) || cfg->n_mux == 0 || cfg->n_mux > 255 ||
        cfg->n_time_bins == 0 || cfg->n_time_bins > 4096 || cfg->bin_seconds == 0 || cfg->time_mode > 1)
        return fail(PPQC_E_INVALID, "config out of range");
    if (!(cfg->fail_q >= 0.0 && cfg->fail_q <= 93.0)) return fail(PPQC_E_INVALID, "fail_q out of range");
    if (cfg->compat_flags & ~PPQC_COMPAT_MULTI_READ_SINGLE) return fail(PPQC_E_INVALID, "unknown compat_flags bit");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PPQC_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(PPQC_E_NO_DEVICE, "device index out of range");
    ppqc_ctx* c = new (std::nothrow) ppqc_ctx();
    if (!c) return fail(PPQC_E_NOMEM, "out of host memory");
    c->device = device;
    c->cfg = *cfg;
    const int rc = create_impl(c, device, cfg);
    if (rc != PPQC_OK) {
        const std::string msg = g_err;   // ppqc_destroy must not replace the message of the failure
        (void)ppqc_destroy(c);
        (void)hipGetLastError();
        return fail(rc, msg);
    }
    *out = c;
    return PPQC_OK;
}

int ppqc_destroy(ppqc_ctx* c) {
    if (!c) return PPQC_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    ReadArrays& r = c->ra;
    void* ptrs[] = {r.rec_cnt, r.rec_hi, r.rec_lo, r.len, r.gc, r.nn, r.qsum, r.mean_q, r.err_hi, r.err_lo, r.flags,
                    r.channel, r.mux, r.time, r.hour, r.lens32, c->head, c->tail, c->res_buf, c->sk32, c->sk32_tmp, c->counts, c->vals, c->vals_tmp, c->sk64,
                    c->sk64_tmp, c->n50_blocks, c->chan_start, c->first_desc, c->lut_dev, c->dense.base, c->slow.base, c->fq.counts, c->fq.n_entries, c->fq.entries, c->fq.prefix,
                    c->fq.block_sums, c->fq.line_off, c->fq.span_sum, c->lra.rec_cnt, c->lra.rec_hi, c->lra.rec_lo, c->m1, c->cc, c->tab_slab,
                    c->fq1.points, c->fq1.regions, c->fq1.head};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->res_host) (void)hipHostFree(c->res_host);
    if (c->maxlen_host) (void)hipHostFree(c->maxlen_host);
    if (c->ev_max) (void)hipEventDestroy(c->ev_max);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->ev_clean) (void)hipEventDestroy(c->ev_clean);
    if (c->ev_fq) (void)hipEventDestroy(c->ev_fq);
    if (c->ev_hdr) (void)hipEventDestroy(c->ev_hdr);
    if (c->ev_cells) (void)hipEventDestroy(c->ev_cells);
    if (c->ev_cmhcopy) (void)hipEventDestroy(c->ev_cmhcopy);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    if (c->ev_reads) (void)hipEventDestroy(c->ev_reads);
    for (int i = 0; i < 2; ++i) {
        Slot& s = c->slot[i];
        void* sp[] = {s.bases, s.quals, s.mux, s.ff, s.offsets, s.channel, s.time};
        for (void* p : sp) if (p) (void)hipFree(p);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    hipEvent_t evs[] = {c->ev_t0, c->ev_t1, c->ev_s0, c->ev_s1};
    for (auto e : evs) if (e) (void)hipEventDestroy(e);
    {
        StreamSet ss;
        ss.own = c->own_stream; ss.aux = c->aux; ss.copy[0] = c->copy[0]; ss.copy[1] = c->copy[1];
        give_streams(c->device, ss);
    }
    delete c;
    return PPQC_OK;
}

int ppqc_reset(ppqc_ctx* c) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    c->ext_deferred = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    // one launch clears the results and the words behind them (input errors, length bound, sort gate); the finalize that
    // follows finds the buffer clean and enqueues no clear of its own
    HIPCHK(launch_init_results(c->res_buf, c->tp.lay.n_sum + PPQC_N_SEL, 1 + (uint64_t)c->cfg.n_channels, true, c->stream));
    HIPCHK(hipEventRecord(c->ev_clean, c->stream));
    c->res_clean = true;
    c->n_reads = 0;
    c->state = 0;
    c->n50 = 0;
    c->lens_early = true;
    c->pre_ok = true;
    c->index_on_main = false;
    c->gate_max_run = true;
    c->pre_run = false;
    c->n50_early = false;
    c->bases_total = 0;
    c->cells_early = !(c->cfg.compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE);
    c->cells_dirty = c->cells_done = c->cmh_on_host = false;
    return PPQC_OK;
}

int ppqc_alloc_pinned(void** p, size_t bytes) {
    if (!p) return fail(PPQC_E_INVALID, "null argument");
    HIPCHK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return PPQC_OK;
}
int ppqc_free_pinned(void* p) {
    if (p) HIPCHK(hipHostFree(p));
    return PPQC_OK;
}

static int resolve_batch_timings(ppqc_ctx* c, hipEvent_t first_of_last_batch, bool keep_last);

// Enqueue the per-read kernels for a batch whose arrays are all in device memory.
static int run_batch(ppqc_ctx* c, const ppqc_batch* b) {
    const uint64_t R = b->n_reads;
    CHK(ensure_reads(c, c->n_reads + R));
    // 64 KiB tiles (a tile's start-up -- the read search, the first loads -- is 3 % of its time; 12 % at 16 KiB), except
    // for batches too small to give every wave slot of the GPU a few tiles: there the kernel's time is the time of its
    // slowest wave, and smaller tiles shorten that (the config-4 adversarial mix, 240 MB: 0.27 -> 0.43 of the HBM peak at
    // 16 KiB; a 4000-read batch of the streaming path, 40 MB: 2400 waves instead of 600).  Which boundary strategy a tile
    // takes is decided per tile on the device (ppqc_tile.hip), not here.
    uint32_t shift = kDefaultTileShift;
    while (shift > 14 && (b->n_bytes >> shift) < 16384) --shift;
    if (c->tn.tile_shift >= 0) shift = (uint32_t)c->tn.tile_shift;
    if (shift < (uint32_t)kMinTileShift || shift > (uint32_t)kMaxTileShift) shift = kDefaultTileShift;
    const uint64_t n_tiles = (b->n_bytes + (1ull << shift) - 1) >> shift;
    CHK(ensure_tiles(c, n_tiles));
    const uint64_t base = c->n_reads;
    hipStream_t s = c->stream;
    BatchView v{b->bases, b->quals, b->offsets, R, b->n_bytes, b->fail_flag, shift, b->channel, b->mux,
                b->start_time_s};
    EvPair ev{get_event(c), get_event(c), 2 * b->n_bytes + 50 * R};
    if (!ev.a || !ev.b) return fail(PPQC_E_HIP, "hipEventCreate failed");
    HIPCHK(hipEventRecord(ev.a, s));
    hipEvent_t prev_a = c->last_a;
    c->last_a = ev.a;
    HIPCHK(launch_tile_kernel(v, c->lut_dev, c->ra, base, c->head, c->tail, c->dense, c->slow, c->tn, s, false,
                              &c->last_dense_alone));
    HIPCHK(hipEventRecord(ev.b, s));
    c->last_b = ev.b;
    c->pending.push_back(ev);
    const bool pre = c->pre_ok && c->res_clean && c->cfg.time_mode == 0 && c->tn.pre != 0;
    if (!pre) c->pre_ok = false;
    // the batch's cells with its index kernel, as long as no packed cell of the run can overflow (host-known bounds)
    const bool cells = pre && c->cells_early && cmh_packable(c->n_reads + R, c->bases_total + b->n_bytes);
    if (!cells) c->cells_early = false;
    c->bases_total += b->n_bytes;
    c->cells_done = c->cmh_on_host = false;
    HIPCHK(launch_finalize_reads(v, c->ra, base, c->head, c->tail, c->qthr,
                                 c->res_buf + c->res_words, c->cfg.n_channels, c->res_buf + c->res_words + 8, !pre, s));
    if (pre) c->index_on_main = false;
    c->gate_max_run = false;   // finalize_reads and the index kernel raise the gate by OR
    if (pre) {
        // Enqueued behind the batch's kernels in host order (every call in front of the tile kernel's launch delays its
        // start) and ordered on the device behind the batch's start event, i.e. behind whatever produced the index on
        // this stream.  The main stream takes the result in behind finalize_reads, so everything that synchronises with
        // the main stream (ppqc_sync, the slots' reuse, a grow of the arrays) covers this kernel too.
        HIPCHK(hipStreamWaitEvent(c->aux, ev.a, 0));
        HIPCHK(launch_batch_extremes(v, base, c->ra, c->tp, c->max_buf, c->min_buf, c->tminmax,
                                     c->res_buf + c->res_words + 4, c->maxlen_host + 4,
                                     cells ? c->sum_buf + c->tp.lay.off_cmh_bases : nullptr,
                                     c->res_buf + c->res_words + 8, c->aux));
        if (cells) c->cells_dirty = true;
        HIPCHK(hipEventRecord(c->ev_pre, c->aux));
        HIPCHK(hipStreamWaitEvent(s, c->ev_pre, 0));
    }
    // first batch behind ppqc_reset: the last run's batches are complete (the reset synchronised), and the host has
    // nothing to do while this batch runs
    if (c->n_reads == 0) CHK(resolve_batch_timings(c, prev_a, true));
    HIPCHK(hipEventRecord(c->ev_reads, s));
    ++c->reads_gen;
    c->n_reads += R;
    c->state = 0;
    c->n50_early = false;
    return PPQC_OK;
}

static int check_batch(const ppqc_batch* b) {
    if (!b) return fail(PPQC_E_INVALID, "null batch");
    if (b->n_reads >= (1ull << 32)) return fail(PPQC_E_RANGE, "more than 2^32-1 reads in one batch");
    if (b->n_reads && (!b->offsets || !b->channel || !b->mux || !b->start_time_s))
        return fail(PPQC_E_INVALID, "null metadata pointer");
    if (b->n_bytes && (!b->bases || !b->quals)) return fail(PPQC_E_INVALID, "null bases/quals pointer");
    if (b->n_reads == 0 && b->n_bytes != 0) return fail(PPQC_E_INVALID, "bytes without reads");
    return PPQC_OK;
}

int ppqc_submit_device(ppqc_ctx* c, const ppqc_batch* b) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    CHK(check_batch(b));
    if (b->n_reads == 0) return PPQC_OK;
    if (((uintptr_t)b->bases & 15) || ((uintptr_t)b->quals & 15))
        return fail(PPQC_E_INVALID, "device bases/quals must be 16-byte aligned");
    HIPCHK(hipSetDevice(c->device));
    CHK(run_batch(c, b));
    return PPQC_OK;
}

int ppqc_submit(ppqc_ctx* c, const ppqc_batch* b, int si) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    CHK(check_batch(b));
    if (si < 0 || si > 1) return fail(PPQC_E_INVALID, "slot must be 0 or 1");
    const uint64_t R = b->n_reads;
    if (R == 0) return PPQC_OK;
    // host-side validation of the CSR (the reference would fail inside the worker on a malformed read instead)
    if (b->offsets[0] != 0) return fail(PPQC_E_INVALID, "offsets[0] must be 0");
    for (uint64_t i = 0; i < R; ++i) {
        if (b->offsets[i + 1] < b->offsets[i]) return fail(PPQC_E_INVALID, "offsets not monotone");
        if (b->offsets[i + 1] - b->offsets[i] >= 0xFFFFFFFFull) return fail(PPQC_E_RANGE, "read longer than 2^32-2");
    }
    if (b->offsets[R] != b->n_bytes) return fail(PPQC_E_INVALID, "offsets[n_reads] != n_bytes");
    HIPCHK(hipSetDevice(c->device));
    Slot& s = c->slot[si];
    if (s.busy) { HIPCHK(hipEventSynchronize(s.done)); s.busy = false; }
    if (b->n_bytes > s.cap_bytes) {
        const uint64_t cap = b->n_bytes + b->n_bytes / 4 + 4096;
        CHK(slot_grow(&s.bases, cap)); CHK(slot_grow(&s.quals, cap));
        s.cap_bytes = cap;
    }
    if (R > s.cap_reads) {
        const uint64_t cap = R + R / 4 + 64;
        CHK(slot_grow(&s.offsets, cap + 1)); CHK(slot_grow(&s.channel, cap)); CHK(slot_grow(&s.mux, cap));
        CHK(slot_grow(&s.time, cap)); CHK(slot_grow(&s.ff, cap));
        s.cap_reads = cap;
    }
    hipStream_t cs = c->copy[si];
    if (b->n_bytes) {
        HIPCHK(hipMemcpyAsync(s.bases, b->bases, b->n_bytes, hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemcpyAsync(s.quals, b->quals, b->n_bytes, hipMemcpyHostToDevice, cs));
    }
    HIPCHK(hipMemcpyAsync(s.offsets, b->offsets, (R + 1) * 8, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(s.channel, b->channel, R * 4, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(s.mux, b->mux, R, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(s.time, b->start_time_s, R * 8, hipMemcpyHostToDevice, cs));
    if (b->fail_flag) HIPCHK(hipMemcpyAsync(s.ff, b->fail_flag, R, hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(s.copied, cs));
    s.used = true;
    HIPCHK(hipStreamWaitEvent(c->stream, s.copied, 0));
    ppqc_batch d = *b;
    d.bases = s.bases; d.quals = s.quals; d.offsets = s.offsets; d.channel = s.channel; d.mux = s.mux;
    d.start_time_s = s.time; d.fail_flag = b->fail_flag ? s.ff : nullptr;
    CHK(run_batch(c, &d));
    HIPCHK(hipEventRecord(s.done, c->stream));
    s.busy = true;
    return PPQC_OK;
}

// FASTQ text resident in device memory -> per-read arrays (see ppqc_fastq_gpu.hip).  One host synchronisation: the
// number of lines has to be known to size the line table and the per-read arrays.
// Mean bytes per line (headers and '+' lines included) from which the span sums pay: line_sum_kernel costs a wave per
// record whatever its length.  tools/sweep_fastq_sums.sh, texts of 2 to 4 GB, second pass / span sums: mean read length
// 500 b (270 B per line) 3.51 / 4.01 ms, 1 kb (520 B) 1.93 / 2.00, 2 kb (1020 B) 2.75 / 2.50, 4 kb 0.99 / 0.80 (0.8 GB),
// 10 kb 1.65 / 1.27 (4 GB).
constexpr int64_t kFastqSumsMinLine = 768;
constexpr int64_t kFastqNotesMinLine = 320;   // below: more than 25 line ends per 8 KiB chunk on average, the notes hold 32 vectors
// Mean bytes per line for which the one-pass kernels (ppqc_fastq1.hip) are taken: above, the span sums read little more
// than the text once as well and do less arithmetic per byte; below, a region's lines outgrow its share of the point table.
constexpr int64_t kFq1MinLine = 160, kFq1MaxLine = kFastqSumsMinLine;

// lens_written: the kernels have left lens32 and raised the sort gate for these reads, as finalize_reads does; pre: and
// their index-only results (extremes, hours; cells: the packed cells too), as extremes_kernel<true> does for a CSR batch
static int fastq_committed(ppqc_ctx* c, uint64_t R, uint64_t n, uint64_t* n_records, bool lens_written, bool pre, bool cells) {
    if (lens_written && R) {
        // lens32 and the gate are complete behind the text's last kernel: the sort + N50 chain may start here
        HIPCHK(hipEventRecord(c->ev_reads, c->stream));
        ++c->reads_gen;
        if (pre) {
            // hours and cells are there (the text's finalize kernel).  The extremes: when the run holds reads of texts
            // only, the gate's largest word is the largest length, the one thing the tables kernel needs of them -- so
            // they are left to that kernel (tables_kernel<true>, ppqc_finalize_tables), which reads every read anyway:
            // an extremes kernel here would stand in the step's tail, the lengths being the last thing a text yields.
            // Else, or when told so, extremes_kernel<true> over these reads, in line.
            const bool defer = c->gate_max_run && c->tn.fastq_extremes != 0 && tables_ext_fits(c->tp.lay);
            if (defer) {
                c->ext_deferred = true;   // (ev_pre is recorded behind the extremes kernel, should one be wanted: join_index;
                                          // until then ev_reads stands behind everything these reads have left -- each
                                          // record on the stream costs the kernel behind it some 6 us)
            } else {
                CHK(join_index(c));   // (earlier texts of the run first)
                BatchView v{nullptr, nullptr, nullptr, R, n, nullptr, kDefaultTileShift, nullptr, nullptr, nullptr};
                HIPCHK(launch_batch_extremes(v, c->n_reads, c->ra, c->tp, c->max_buf, c->min_buf, c->tminmax,
                                             c->res_buf + c->res_words + 4, c->maxlen_host + 4, nullptr, nullptr, c->stream));
                HIPCHK(hipEventRecord(c->ev_pre, c->stream));
            }
            c->index_on_main = true;   // whoever takes these results on another stream waits for ev_reads (finalize_impl)
            if (cells) c->cells_dirty = true;
            c->last_b = nullptr;   // no per-read kernel whose end the sort chain could start at: behind ev_reads
            c->last_dense_alone = false;
        }
    } else if (R) {
        c->lens_early = false;   // lens32 of these reads is written by the extremes kernel
    }
    if (R && !(lens_written && pre)) c->gate_max_run = false;
    if (R && !cells) c->cells_early = false;
    c->n_reads += R;
    c->state = 0;
    c->n50_early = false;
    if (R && !pre) c->pre_ok = false;
    c->cells_done = c->cmh_on_host = false;
    c->bases_total += n;     // the text holds every base
    if (n_records) *n_records = R;
    return PPQC_OK;
}

// Waits until a kernel on `s` has stored `want` in the pinned word *w (a release store at system scope behind the data
// it announces).  Now and then the stream is asked whether it has ended without the word: a failed launch or a fault.
static int wait_pinned_word(const uint64_t* w, uint64_t want, hipStream_t s) {
    const volatile uint64_t* vw = w;
    for (uint64_t spin = 1;; ++spin) {
        if (*vw == want) break;
        __builtin_ia32_pause();
        if ((spin & 0xFFFF) == 0) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipErrorNotReady) continue;
            if (e != hipSuccess) return fail(PPQC_E_HIP, std::string("while waiting for the FASTQ scan: ") + hipGetErrorString(e));
            if (*vw == want) break;
            return fail(PPQC_E_HIP, "the FASTQ scan ended without handing its line count to the host");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return PPQC_OK;
}

static int fastq_line_count_error(uint64_t n_lines) {
    return fail(PPQC_E_INVALID, "FASTQ text has " + std::to_string(n_lines) + " lines, not a multiple of 4 (blank "
                                    "lines or multi-line records: use ppqc_fastq_scan / ppqc_fastq_pack)");
}

// One pass over the text (ppqc_fastq1.hip).  *done = false: the text is not for these kernels (reads of a few bases
// somewhere in it); nothing has been committed and the caller takes the general path.
static int run_fastq_one_pass(ppqc_ctx* c, const uint8_t* text, uint64_t n, uint8_t mux, uint64_t* n_records, bool* done) {
    *done = false;
    hipStream_t s = c->stream;
    uint32_t cpr = 1;
    uint64_t n_regions = 1;
    fq1_shape(n, c->tn.fq1_regions, &cpr, &n_regions);
    const uint64_t need_lines = n_regions * cpr * kFq1LinesPerChunk;
    if (need_lines > c->cap_fq1_lines) {
        const uint64_t cap = need_lines + need_lines / 4;
        CHK(grow(&c->fq1.points, 0, cap * 4, s));
        c->cap_fq1_lines = cap;
    }
    if (n_regions > c->cap_fq1_regions) {
        CHK(grow(&c->fq1.regions, 0, n_regions + 64, s));
        c->cap_fq1_regions = n_regions + 64;
    }
    if (!c->fq1.head) CHK(grow(&c->fq1.head, 0, 2, s));
    uint64_t* h = c->maxlen_host;   // pinned scratch words: [1] lines, [2] flags, [5] the ticket of the launch they belong to
    // No event behind the regions kernel: an event record between two kernels of a stream costs the second one some 20 us
    // here, and the finalize kernel may follow at once (below).  The host polls the ticket word instead.
    const uint64_t ticket = ++c->fq1_ticket;
    // The index-only results of these reads with the finalize kernel, as a CSR batch leaves them when it is submitted
    // (run_batch): the finalize then needs no extremes kernel and the tables no cell atomics.
    const bool pre = c->pre_ok && c->res_clean && c->cfg.time_mode == 0 && c->tn.pre != 0;
    // The finalize kernel straight behind the scan, before the line count has come back (the GPU would wait some 30 us
    // for the host's answer): the last text's mean line length says about how many records there are; if the arrays
    // hold a few more than that, the kernel takes the count from the regions kernel's words and goes ahead unless it
    // finds the text declined, its lines no multiple of four, or more records than `room` -- the very tests the host
    // makes below on the same words, so both know whether it ran.
    // (That kernel then hands the line count to the host as well, first thing: a regions kernel that has stored to host
    // memory keeps the kernel behind it waiting for some 12 us.)
    uint64_t room = 0;
    bool early = false, early_cells = false;
    if (c->tn.fq1_early != 0 && c->fq_line_hint > 0) {
        const uint64_t guess = n / (4 * (uint64_t)c->fq_line_hint) + 1;
        room = c->cap_reads > c->n_reads ? c->cap_reads - c->n_reads : 0;
        if (room >= (1ull << 30)) room = (1ull << 30) - 1;
        early_cells = pre && c->cells_early && cmh_packable(c->n_reads + 1, c->bases_total + n);
        if (early_cells) {   // the packed cells hold 2^24 - 1 reads
            const uint64_t fit = (1ull << (64 - kCmhShift)) - 1 - c->n_reads;
            if (room > fit) room = fit;
        }
        if (room >= guess + guess / 16) {
            room = guess + guess / 16;   // a thread per record it may hold: 6 % of them idle when the guess is right
            Fq1Index ix{pre ? 1 : 0, c->tp, early_cells ? c->sum_buf + c->tp.lay.off_cmh_bases : nullptr};
            HIPCHK(launch_fq1_scan(text, n, c->fq1, cpr, n_regions, c->lut_dev, nullptr, 0, s));
            HIPCHK(launch_fq1_finalize_early(text, n, room, c->fq1, cpr, n_regions, c->ra, c->n_reads, c->lut_dev,
                                             c->qthr, mux, c->res_buf + c->res_words, c->cfg.n_channels,
                                             c->res_buf + c->res_words + 8, ix, h + 1, ticket, s));
            early = true;
        }
    }
    if (!early) HIPCHK(launch_fq1_scan(text, n, c->fq1, cpr, n_regions, c->lut_dev, h + 1, ticket, s));
    CHK(wait_pinned_word(h + 5, ticket, s));
    const uint64_t n_lines = h[1];
    c->fq_line_hint = n_lines ? (int64_t)(n / n_lines) : -1;
    if (h[2] != 0) { ++c->n_fq1_declined; return PPQC_OK; }
    ++c->n_fq1;
    if (n_lines % 4 != 0) return fastq_line_count_error(n_lines);
    const uint64_t R = n_lines / 4;
    if (R >= (1ull << 30)) return fail(PPQC_E_RANGE, "more than 2^30-1 records in one text");
    *done = true;
    if (early && R <= room) {   // the kernel is running, or done
        ++c->n_fq1_early;
        return fastq_committed(c, R, n, n_records, true, pre, early_cells);
    }
    CHK(ensure_reads(c, c->n_reads + R));
    const bool cells = pre && c->cells_early && cmh_packable(c->n_reads + R, c->bases_total + n);
    Fq1Index ix{pre ? 1 : 0, c->tp, cells ? c->sum_buf + c->tp.lay.off_cmh_bases : nullptr};
    HIPCHK(launch_fq1_finalize(text, n, R, c->fq1, cpr, n_regions, c->ra, c->n_reads, c->lut_dev, c->qthr, mux,
                               c->res_buf + c->res_words, c->cfg.n_channels, c->res_buf + c->res_words + 8, ix, s));
    return fastq_committed(c, R, n, n_records, true, pre, cells);
}

static int run_fastq(ppqc_ctx* c, const uint8_t* text, uint64_t n, uint8_t mux, uint64_t* n_records) {
    if (n_records) *n_records = 0;
    if (n == 0) return PPQC_OK;
    hipStream_t s = c->stream;
    const int op = c->tn.fastq_one_pass;
    if (op == 1 || (op < 0 && c->fq_line_hint >= kFq1MinLine && c->fq_line_hint < kFq1MaxLine)) {
        bool done = false;
        CHK(run_fastq_one_pass(c, text, n, mux, n_records, &done));
        if (done) return PPQC_OK;
    }
    ++c->n_fq_general;
    const uint64_t n_chunks = (n + kNlChunk - 1) / kNlChunk;
    if (n_chunks > c->cap_chunks) {
        const uint64_t cap = n_chunks + n_chunks / 4 + 64;
        CHK(grow(&c->fq.counts, 0, cap, s)); CHK(grow(&c->fq.n_entries, 0, cap, s));
        CHK(grow(&c->fq.entries, 0, cap * kNlEntriesPerChunk, s));
        CHK(grow(&c->fq.prefix, 0, cap + 3, s));
        CHK(grow(&c->fq.block_sums, 0, nl_scan_block_words(cap), s));
        CHK(grow(&c->fq.span_sum, 0, cap * 2 * (kNlChunk / kSpan), s));
        c->cap_chunks = cap;
    }
    uint64_t* h = c->maxlen_host;   // pinned scratch words [1] = newlines, [2] = chunks with too many odd vectors, [3] = last byte
    // Long lines: the scan pass also sums every clean 1 KiB span, and the lines are put together from those span sums
    // instead of a second pass over the text (ppqc_fastq_gpu.hip).  It pays when the two end spans of a line are a small
    // part of it; the scan has to be chosen before the line count is known, so the last text's mean line length decides
    // (the first text of a context is taken to be long; a wrong guess costs the scan pass its extra arithmetic, nothing
    // else).  Tuning::fastq_sums = 0 | 2: never / always (tests run every text both ways).
    const int fs_mode = c->tn.fastq_sums >= 0 ? c->tn.fastq_sums : 1;
    const bool with_sums = fs_mode == 2 || (fs_mode == 1 && (c->fq_line_hint < 0 || c->fq_line_hint >= kFastqSumsMinLine));
    // Short lines (by the same hint): most chunks would overflow their notes and the line starts would come from a
    // second pass anyway, so the scan only counts (0.86 -> 0.45 ms for 2.2 GB of 500-base reads).
    const bool notes = with_sums || fs_mode != 1 || c->fq_line_hint < 0 || c->fq_line_hint >= kFastqNotesMinLine;
    HIPCHK(launch_nl_scan(text, n, c->fq, h + 1, c->ev_fq, s, with_sums ? c->lut_dev : nullptr, notes));   // the scan's top kernel writes them
    HIPCHK(hipEventSynchronize(c->ev_fq));
    const uint64_t n_lines = h[1] + (((uint8_t)h[3]) != '\n' ? 1 : 0);
    const uint64_t n_over = notes ? h[2] : n_chunks;   // chunks with more line ends than their notes hold
    c->fq_line_hint = n_lines ? (int64_t)(n / n_lines) : -1;
    const bool from_sums = with_sums && (fs_mode == 2 || c->fq_line_hint >= kFastqSumsMinLine);
    if (n_lines % 4 != 0) return fastq_line_count_error(n_lines);
    const uint64_t R = n_lines / 4;
    if (R >= (1ull << 30)) return fail(PPQC_E_RANGE, "more than 2^30-1 records in one text");
    if (n_lines > c->cap_lines) {
        const uint64_t cap = n_lines + n_lines / 4 + 64;
        CHK(grow(&c->fq.line_off, 0, cap + 1, s));
        CHK(grow(&c->lra.rec_cnt, 0, cap, s)); CHK(grow(&c->lra.rec_hi, 0, cap, s)); CHK(grow(&c->lra.rec_lo, 0, cap, s));
        c->cap_lines = cap;
    }
    HIPCHK(launch_nl_place(text, n, n_lines, n_over, c->fq, c->res_buf + c->res_words, s));
    CHK(ensure_reads(c, c->n_reads + R));
    const uint32_t shift = kDefaultTileShift;
    CHK(ensure_tiles(c, (n + (1ull << shift) - 1) >> shift));
    // The headers need the line table only.  In front of the line-mode tile kernel their kernel runs on the same stream:
    // beside it on the auxiliary stream (also at the highest stream priority) its blocks trickle in between the tile
    // kernel's for the whole 0.72 ms and fastq_finalize waits 50 us for the last of them.  line_sum_kernel is short and
    // leaves room: beside it the header kernel is hidden (1.29 -> 1.265 ms for the step).  Tuning::fastq_hdr_aux = 0 | 2:
    // always in front / always beside.
    const int hdr_env = c->tn.fastq_hdr_aux >= 0 ? c->tn.fastq_hdr_aux : 1;
    const bool hdr_aux = hdr_env == 2 || (from_sums && hdr_env != 0);
    if (hdr_aux) {
        HIPCHK(hipEventRecord(c->ev_fq, s));
        HIPCHK(hipStreamWaitEvent(c->aux, c->ev_fq, 0));
        HIPCHK(launch_fastq_header(text, R, c->fq.line_off, c->ra, c->n_reads, mux, c->fq.prefix + n_chunks + 2, c->res_buf + c->res_words, c->aux));
        HIPCHK(hipEventRecord(c->ev_hdr, c->aux));
    } else {
        HIPCHK(launch_fastq_header(text, R, c->fq.line_off, c->ra, c->n_reads, mux, c->fq.prefix + n_chunks + 2, c->res_buf + c->res_words, s));
    }
    if (from_sums) {
        HIPCHK(launch_line_sums(text, n, R, c->fq.line_off, c->fq.span_sum, c->lut_dev, c->lra, s));
    } else {
        // line mode: both classifiers run over the same text, every line is a segment
        BatchView v{text, text, c->fq.line_off, n_lines, n, nullptr, shift, nullptr, nullptr, nullptr};
        HIPCHK(launch_tile_kernel(v, c->lut_dev, c->lra, 0, c->head, c->tail, c->dense, c->slow, c->tn, s, true));
    }
    if (hdr_aux) HIPCHK(hipStreamWaitEvent(s, c->ev_hdr, 0));
    // as in run_fastq_one_pass: sort keys, gate and (a context as ppqc_reset left it) hours and cells with the finalize kernel
    const bool pre = c->pre_ok && c->res_clean && c->cfg.time_mode == 0 && c->tn.pre != 0;
    const bool cells = pre && c->cells_early && cmh_packable(c->n_reads + R, c->bases_total + n);
    Fq1Index ix{pre ? 1 : 0, c->tp, cells ? c->sum_buf + c->tp.lay.off_cmh_bases : nullptr};
    HIPCHK(launch_fastq_finalize(text, n, R, c->fq.line_off, c->lra, c->ra, c->n_reads, c->head, c->tail, shift,
                                 c->lut_dev, c->qthr, mux, c->res_buf + c->res_words, c->cfg.n_channels,
                                 c->res_buf + c->res_words + 8, ix, s, from_sums));
    return fastq_committed(c, R, n, n_records, true, pre, cells);
}

int ppqc_submit_fastq_device(ppqc_ctx* c, const uint8_t* text, uint64_t n, uint8_t mux, uint64_t* n_records) {
    if (!c || (!text && n)) return fail(PPQC_E_INVALID, "null argument");
    if ((uintptr_t)text & 15) return fail(PPQC_E_INVALID, "device text must be 16-byte aligned");
    HIPCHK(hipSetDevice(c->device));
    return run_fastq(c, text, n, mux, n_records);
}

int ppqc_submit_fastq(ppqc_ctx* c, const uint8_t* text, uint64_t n, int si, uint8_t mux, uint64_t* n_records) {
    if (!c || (!text && n)) return fail(PPQC_E_INVALID, "null argument");
    if (si < 0 || si > 1) return fail(PPQC_E_INVALID, "slot must be 0 or 1");
    if (n_records) *n_records = 0;
    if (n == 0) return PPQC_OK;
    HIPCHK(hipSetDevice(c->device));
    Slot& s = c->slot[si];
    if (s.busy) { HIPCHK(hipEventSynchronize(s.done)); s.busy = false; }
    if (n > s.cap_bytes) {
        const uint64_t cap = n + n / 4 + 4096;
        CHK(slot_grow(&s.bases, cap)); CHK(slot_grow(&s.quals, cap));
        s.cap_bytes = cap;
    }
    hipStream_t cs = c->copy[si];
    HIPCHK(hipMemcpyAsync(s.bases, text, n, hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(s.copied, cs));
    s.used = true;
    HIPCHK(hipStreamWaitEvent(c->stream, s.copied, 0));
    CHK(run_fastq(c, s.bases, n, mux, n_records));
    HIPCHK(hipEventRecord(s.done, c->stream));
    s.busy = true;
    return PPQC_OK;
}

int ppqc_submit_records(ppqc_ctx* c, const ppqc_records* r) {
    if (!c || !r) return fail(PPQC_E_INVALID, "null argument");
    const uint64_t n = r->n_reads;
    if (n == 0) return PPQC_OK;
    if (n >= (1ull << 32)) return fail(PPQC_E_RANGE, "more than 2^32-1 records");
    if (!r->length || !r->mean_q || !r->channel || !r->mux || !r->start_time_s)
        return fail(PPQC_E_INVALID, "null record column");
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_reads(c, c->n_reads + n));
    // stage the columns in one device buffer (freed after the kernel; this path is I/O bound on the text table)
    const uint64_t o_len = 0, o_q = o_len + n * 4 + 8, o_q8 = (o_q + 7) & ~7ull, o_gc = o_q8 + n * 8,
                   o_ch = o_gc + n * 4, o_t = ((o_ch + n * 4) + 7) & ~7ull, o_mux = o_t + n * 8, total = o_mux + n + 64;
    uint8_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, total));
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(d + o_len, r->length, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d + o_q8, r->mean_q, n * 8, hipMemcpyHostToDevice, s));
    if (r->gc_count) HIPCHK(hipMemcpyAsync(d + o_gc, r->gc_count, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d + o_ch, r->channel, n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d + o_t, r->start_time_s, n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d + o_mux, r->mux, n, hipMemcpyHostToDevice, s));
    HIPCHK(launch_records(c->ra, c->n_reads, n, (const uint32_t*)(d + o_len), (const double*)(d + o_q8),
                          r->gc_count ? (const uint32_t*)(d + o_gc) : nullptr, (const uint32_t*)(d + o_ch),
                          d + o_mux, (const int64_t*)(d + o_t), c->tp.fail_q, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(d));
    c->n_reads += n;
    c->state = 0;
    c->lens_early = false;   // lens32 of these reads is written by the extremes kernel
    c->pre_ok = false;
    c->gate_max_run = false;
    c->cells_early = c->cells_done = c->cmh_on_host = false;
    for (uint64_t i = 0; i < n; ++i) c->bases_total += r->length[i];
    return PPQC_OK;
}

// Event arithmetic costs the host a few microseconds per pair, so none of it happens between a step's last
// synchronisation and the next step's first launch: the finished batches are accounted for behind the next batch's
// launches (or when the timings are asked for), the finalize behind the next finalize's start.
static void resolve_finalize_timings(ppqc_ctx* c) {
    if (!c->fin_pending) return;
    c->fin_pending = false;
    float ms = 0.f;
    hipEvent_t f0 = c->fin_start ? c->fin_start : c->ev_t0, s0 = c->sort_start ? c->sort_start : c->ev_s0;
    // a batch submitted since that finalize has recorded ev_reads again: the interval is gone, the last figures stay
    if ((f0 == c->ev_reads || s0 == c->ev_reads) && c->fin_gen != c->reads_gen) return;
    if (hipEventElapsedTime(&ms, f0, c->ev_t1) == hipSuccess) {
        float ms2 = 0.f;
        if (c->end_split && c->sort_end && hipEventElapsedTime(&ms2, f0, c->sort_end) == hipSuccess && ms2 > ms) ms = ms2;
        c->timing.finalize_ms_last = ms;
    }
    if (hipEventElapsedTime(&ms, s0, c->sort_end ? c->sort_end : c->ev_s1) == hipSuccess) c->timing.sort_ms_last = ms;
    (void)hipGetLastError();   // an unrecorded event is not an error of the caller
}

// Batches whose events have completed; `keep_last`: the newest pair belongs to a batch just enqueued.
static int resolve_batch_timings(ppqc_ctx* c, hipEvent_t first_of_last_batch, bool keep_last) {
    if (first_of_last_batch) {   // newest finished batch: tile kernel + finalize_reads
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, first_of_last_batch, c->ev_reads) == hipSuccess) c->timing.submit_ms_last = ms;
        (void)hipGetLastError();
    }
    const size_t n = c->pending.size() - ((keep_last && !c->pending.empty()) ? 1 : 0);
    for (size_t k = 0; k < n; ++k) {
        EvPair& p = c->pending[k];
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        c->timing.tile_kernel_ms_total += ms;
        c->timing.tile_kernel_launches += 1;
        c->timing.tile_kernel_bytes += p.bytes;
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->pending.erase(c->pending.begin(), c->pending.begin() + n);
    return PPQC_OK;
}

static int resolve_timings(ppqc_ctx* c) {   // everything, behind a synchronisation
    hipEvent_t a = c->last_a;
    c->last_a = nullptr;
    CHK(resolve_batch_timings(c, a, false));
    resolve_finalize_timings(c);
    return PPQC_OK;
}

int ppqc_sync(ppqc_ctx* c) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2; ++i) c->slot[i].busy = false;
    c->res_busy = false;
    return PPQC_OK;
}

int ppqc_slot_wait(ppqc_ctx* c, int si) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (si < 0 || si > 1) return fail(PPQC_E_INVALID, "slot must be 0 or 1");
    if (!c->slot[si].used) return PPQC_OK;
    HIPCHK(hipEventSynchronize(c->slot[si].copied));
    return PPQC_OK;
}

int ppqc_input_errors(ppqc_ctx* c, uint64_t* out5) {
    if (!c || !out5) return fail(PPQC_E_INVALID, "null argument");
    CHK(ppqc_sync(c));
    uint64_t e[7];
    HIPCHK(hipMemcpy(e, c->res_buf + c->res_words, 56, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) out5[k] = e[k];
    out5[4] = e[6];
    return PPQC_OK;
}

int ppqc_n_reads(ppqc_ctx* c, uint64_t* n) {
    if (!c || !n) return fail(PPQC_E_INVALID, "null argument");
    *n = c->n_reads;
    return PPQC_OK;
}

int ppqc_read_stats(ppqc_ctx* c, const ppqc_per_read* o, uint64_t first, uint64_t n) {
    if (!c || !o) return fail(PPQC_E_INVALID, "null argument");
    if (first + n > c->n_reads) return fail(PPQC_E_INVALID, "read range out of bounds");
    CHK(ppqc_sync(c));
    if (n == 0) return PPQC_OK;
    uint64_t bad[7] = {0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(bad, c->res_buf + c->res_words, 56, hipMemcpyDeviceToHost));
    CHK(input_errors(bad));
    const ReadArrays& r = c->ra;
#define D2H(dst, src, T) \
    if (dst) HIPCHK(hipMemcpy(dst, (src) + first, n * sizeof(T), hipMemcpyDeviceToHost))
    D2H(o->length, r.len, uint32_t); D2H(o->gc_count, r.gc, uint32_t); D2H(o->n_count, r.nn, uint32_t);
    D2H(o->qsum, r.qsum, int64_t); D2H(o->mean_q, r.mean_q, double); D2H(o->err_hi, r.err_hi, uint64_t);
    D2H(o->err_lo, r.err_lo, uint64_t); D2H(o->flags, r.flags, uint8_t);
#undef D2H
    if (o->flags)   // the device keeps the read's quality bin in the upper bits (quality_flags)
        for (uint64_t i = 0; i < n; ++i) o->flags[i] &= kFlagsPublic;
    return PPQC_OK;
}

int ppqc_read_hours(ppqc_ctx* c, int32_t* out, uint64_t first, uint64_t n) {
    if (!c || !out) return fail(PPQC_E_INVALID, "null argument");
    if (c->state < 2) return fail(PPQC_E_STATE, "hours are available after ppqc_finalize_tables");
    if (first + n > c->n_reads) return fail(PPQC_E_INVALID, "read range out of bounds");
    CHK(ppqc_sync(c));
    if (n) HIPCHK(hipMemcpy(out, c->ra.hour + first, n * 4, hipMemcpyDeviceToHost));
    return PPQC_OK;
}

// defer_index: the caller (ppqc_finalize) goes on to the tables kernel, which takes the extremes of FASTQ reads along
static int finalize_extremes_impl(ppqc_ctx* c, bool defer_index) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!(c->res_clean && c->pre_ok)) c->ext_deferred = false;   // the extremes kernel below takes every read of the run
    else if (!(defer_index && c->gate_max_run)) CHK(join_index(c));
    resolve_finalize_timings(c);   // its events are about to be recorded again
    c->fin_gen = c->reads_gen;
    c->sum_fetched = false;
    c->m1_phase = -1;
    c->cc_ready = false;
    // The result buffer is cleared on the auxiliary stream: this call is usually enqueued while the per-read kernels of
    // the batch still run, and nothing before the extremes kernel touches the results, so the clear costs no time on
    // the main stream.  If an earlier finalize phase may still be running (no synchronisation since), it stays in line.
    // Straight after ppqc_reset the buffer is clean already (cleared on this stream before the batch): nothing is
    // enqueued between finalize_reads and the extremes kernel, and ev_reads doubles as the start of the timings (every
    // event record or wait on the stream costs about 6 us between two dependent kernels).
    c->fin_start = c->ev_t0;
    c->pre_run = c->res_clean && c->pre_ok && c->n_reads > 0;
    if (c->pre_run) {   // every batch left its extremes, hours and lens32 when it was submitted: nothing to do here
        c->init_on_aux = true;
        c->fin_start = c->ev_reads;
        c->res_clean = false;
        c->state = 1;
        return mark_busy(c);
    }
    if (c->res_clean) {
        c->init_on_aux = true;   // what it stands for holds: the clear precedes ev_reads
        if (c->lens_early && c->n_reads) c->fin_start = c->ev_reads;
        else HIPCHK(hipEventRecord(c->ev_t0, s));
    } else {
        c->init_on_aux = !c->res_busy;
        if (!c->res_busy) {
            HIPCHK(launch_init_results(c->res_buf, c->tp.lay.n_sum + PPQC_N_SEL, 1 + (uint64_t)c->cfg.n_channels, false, c->aux));
            c->cells_dirty = c->cells_done = c->cmh_on_host = false;
            HIPCHK(hipEventRecord(c->ev_init, c->aux));
            HIPCHK(hipEventRecord(c->ev_t0, s));
            HIPCHK(hipStreamWaitEvent(s, c->ev_init, 0));
        } else {
            HIPCHK(hipEventRecord(c->ev_t0, s));
            HIPCHK(launch_init_results(c->res_buf, c->tp.lay.n_sum + PPQC_N_SEL, 1 + (uint64_t)c->cfg.n_channels, false, s));
            c->cells_dirty = c->cells_done = c->cmh_on_host = false;
        }
    }
    c->res_clean = false;
    HIPCHK(launch_extremes(c->ra, c->n_reads, c->tp, c->max_buf, c->min_buf, c->tminmax, !c->lens_early, s));
    if (!c->lens_early) HIPCHK(hipEventRecord(c->ev_max, s));   // fork point of the sort + N50 chain (ppqc_finalize_n50)
    c->state = 1;
    return mark_busy(c);
}

int ppqc_finalize_extremes(ppqc_ctx* c) { return finalize_extremes_impl(c, false); }

int ppqc_time_range(ppqc_ctx* c, int64_t* out2) {
    if (!c || !out2) return fail(PPQC_E_INVALID, "null argument");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    CHK(ppqc_sync(c));
    HIPCHK(hipMemcpy(out2, c->tminmax, 16, hipMemcpyDeviceToHost));
    return PPQC_OK;
}

int ppqc_set_ref_time(ppqc_ctx* c, int64_t ref_time) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (c->n_reads && ref_time != c->tp.ref_time_s) c->pre_ok = false;   // hours of the submitted batches are stale
    if (ref_time != c->tp.ref_time_s) c->m1_phase = -1;   // ... and so are the per-day tables and hours of ppqc_mode1_dist_phase:
                                                          // ppqc_finalize_tables takes the sort path unless the phases run again
    c->cfg.ref_time_s = ref_time;
    c->tp.ref_time_s = ref_time;
    return PPQC_OK;
}

static uint32_t bytes_mask_for(uint64_t maxval) {
    uint32_t m = 0;
    for (int b = 0; b < 8; ++b)
        if ((maxval >> (8 * b)) & 0xFF || (maxval >> (8 * b)) > 0xFF) m |= 1u << b;
    return m ? m : 1u;
}

int ppqc_finalize_tables(ppqc_ctx* c) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t n = c->n_reads;
    if (c->cfg.time_mode == 1 && c->m1_phase >= 0 && c->m1_phase != 2)
        return fail(PPQC_E_STATE, "ppqc_mode1_dist_phase: phases 0, 1, 2 must all run before ppqc_finalize_tables");
    if (c->cfg.time_mode == 1 && n && c->m1_phase != 2) {
        int64_t tmm[2];
        HIPCHK(hipMemcpyAsync(tmm, c->tminmax, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint64_t span = 0;
        if (tmm[0] <= tmm[1]) {
            span = (uint64_t)tmm[1] - (uint64_t)tmm[0];
            // 2^40 - 1 is the key of a read without a start time (mode1_keys_kernel): every timed key must stay below it
            if (span >= (1ull << 40) - 1) return fail(PPQC_E_RANGE, "time_mode 1: start times span 2^40 ticks or more");
        }
        CHK(ensure_sort64(c, n));
        HIPCHK(launch_mode1_keys(c->ra, n, c->tp, c->tminmax, c->sk64, c->vals, s));
        // the bytes of span + 1: in the bytes sorted on, the key of a read without a start time (all ones) then lies above
        // every timed key of its channel, whatever the span (with the bytes of `span` alone a span of 2^(8k) - 1 tied them)
        const uint32_t mask = (bytes_mask_for(span + 1) & 0x1Fu) | ((bytes_mask_for((uint64_t)c->cfg.n_channels + 1) & 0x7u) << 5);
        HIPCHK(sort_u64_pairs(c->sk64, c->sk64_tmp, c->vals, c->vals_tmp, n, c->counts, mask, s));
        HIPCHK(launch_mode1_hours(c->ra, n, c->tp, c->sk64, c->vals, c->chan_start, c->first_desc, s));
    }
    // The cells: there already when every batch left them with its index kernel (then they are unpacked here unless
    // ppqc_cells_early / ppqc_finalize has done that beside the per-read kernel); else by tables_kernel, one packed atomic
    // per read when the run's reads and bases fit the two fields, into a region that is as the clear left it.
    const uint64_t n_cells = (uint64_t)c->cfg.n_channels * c->cfg.n_mux * c->cfg.n_time_bins;
    int cell_mode = 0;
    if (!c->cells_done) {
        if (c->pre_run && c->cells_early) {
            HIPCHK(launch_cmh_unpack(c->tp, c->sum_buf, s));
        } else {
            if (c->cells_dirty) HIPCHK(hipMemsetAsync(c->sum_buf + c->tp.lay.off_cmh_reads, 0, 2 * n_cells * 8, s));
            cell_mode = cmh_packable(n, c->bases_total) ? 1 : 2;
        }
        c->cells_dirty = false;
        c->cells_done = true;
        c->cmh_on_host = false;
    }
    // the largest length: left by an extremes kernel in front of this one, or (the extremes of a run of FASTQ texts are
    // left to this kernel) the largest word of the gate, which the texts' finalize kernels have raised
    const bool with_ext = c->ext_deferred && c->pre_run && c->gate_max_run && n != 0;
    if (!with_ext) CHK(join_index(c));
    HIPCHK(launch_tables(c->ra, n, c->tp, c->sum_buf, with_ext ? c->res_buf + c->res_words + 8 : c->max_buf,
                         with_ext ? kGateWords : 1, c->tab_slab, s, cell_mode, c->cc_ready ? c->cc : nullptr,
                         with_ext ? reinterpret_cast<long long*>(c->max_buf) : nullptr));
    c->ext_deferred = false;
    c->state = 2;
    if (c->n50_early) {   // the N50 is there already (ppqc_n50_select_early): the finalize ends here
        HIPCHK(hipEventRecord(c->ev_t1, s));
        c->end_split = false;
        c->sort_start = c->sort_end = nullptr;
        c->state = 3;
    }
    return mark_busy(c);
}

// The result words the host reads ([sum | sel | max | min | time range | n50, total | select state | input errors]) to the
// pinned mirror, on `st`.  When the mirror holds the cells already (cmh_on_host: 2 x cells words, by far the largest
// part, copied beside the per-read kernel), only what lies in front of them and behind them is copied.
static int copy_results(ppqc_ctx* c, hipStream_t st) {
    const uint64_t end = c->res_words + 7;   // with the seven error words
    if (!c->cmh_on_host) {
        HIPCHK(hipMemcpyAsync(c->res_host, c->res_buf, end * 8, hipMemcpyDeviceToHost, st));
        return PPQC_OK;
    }
    const uint64_t a = c->tp.lay.off_cmh_reads, b = c->tp.lay.n_sum + PPQC_N_SEL;   // the select histogram is not read either
    HIPCHK(hipMemcpyAsync(c->res_host, c->res_buf, a * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->res_host + b, c->res_buf + b, (end - b) * 8, hipMemcpyDeviceToHost, st));
    return PPQC_OK;
}

// The cells are final in the cmh region behind whatever `st` is ordered behind: unpack them there, and (to_host) copy the
// region to the host mirror on `cs`, which waits for the unpack by event.
static int cells_unpack_on(ppqc_ctx* c, hipStream_t st) {
    HIPCHK(launch_cmh_unpack(c->tp, c->sum_buf, st));
    c->cells_dirty = false;
    c->cells_done = true;
    c->cmh_on_host = false;
    return PPQC_OK;
}
static int cells_to_host_on(ppqc_ctx* c, hipStream_t st, hipStream_t cs) {
    const uint64_t n_cells = (uint64_t)c->cfg.n_channels * c->cfg.n_mux * c->cfg.n_time_bins, a = c->tp.lay.off_cmh_reads;
    if (cs != st) {
        HIPCHK(hipEventRecord(c->ev_cells, st));
        HIPCHK(hipStreamWaitEvent(cs, c->ev_cells, 0));
    }
    HIPCHK(hipMemcpyAsync(c->res_host + a, c->res_buf + a, 2 * n_cells * 8, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipEventRecord(c->ev_cmhcopy, cs));
    c->cmh_on_host = true;
    return PPQC_OK;
}

static const int kSelShift[3] = {21, 10, 0}, kSelBits[3] = {11, 11, 10};

static int finalize_n50_impl(ppqc_ctx* c, const uint32_t* dlens, uint64_t n, bool early_fetch) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!dlens) { dlens = c->ra.lens32; n = c->n_reads; }
    if (n >= (1ull << 32)) return fail(PPQC_E_RANGE, "more than 2^32-1 lengths");
    resolve_finalize_timings(c);
    // Local lengths: the chain runs on the auxiliary stream beside the tables kernel (time_mode 1 sorts on the main
    // stream with the same count buffer, so there it stays in line).  Which passes run: decided by the host when it
    // knows the largest key (`pre` below), else on the device by a word that bounds every key (no host round trip).
    // When every read of the run came through finalize_reads, lens32 and that gate are complete behind the last of those
    // kernels (ev_reads); otherwise the extremes kernel writes lens32 and max_buf[0] is the gate (fork at ev_max).
    // Gathered lengths of other ranks are bounded by the merged max_len.
    const bool local = dlens == c->ra.lens32;
    const bool early = local && c->lens_early;
    hipStream_t ss = (local && c->cfg.time_mode == 0) ? c->aux : s;
    const uint64_t* gate = early ? c->res_buf + c->res_words + 8 : c->max_buf;
    const int gate_words = early ? kGateWords : 1;
    CHK(ensure_sort32(c, n ? n : 1));
    // pre: the host knows the largest key (the batch kernels leave it in a pinned word), so exactly the passes the keys
    // need are enqueued.  lens32 has been there since the batches were submitted, but the chain still waits for the end
    // of finalize_reads: run beside the tile kernel its ten launches cost that kernel more time (3 %: every wave
    // resident beside it displaces one of its own) than they take behind it, beside the tables kernel.
    // (not behind a one-pass FASTQ text: its extremes end just in front of this call, the host would wait for them with
    // nothing enqueued behind; the chain starts behind the text's last per-read kernel, gated on the device)
    const bool pre = local && c->pre_run && ss != s && !c->index_on_main;
    // By radix select instead of the sort (three histograms of the lengths and three one-block picks: six short launches
    // that read the lengths at most three times -- a round above the gate's top bit returns at once -- where the sort
    // reads and writes them once per byte of the largest), whenever the histogram and its state are as ppqc_reset left
    // them (pre_run; each pick clears the histogram for the next).  Behind a FASTQ text, whose lengths are the last thing
    // known, the chain is the tail of the step (2 * 10^6 reads: 205 -> 45 us); beside the per-read kernel of a CSR batch
    // it takes less from that kernel (4 * 10^6 reads of 500 bases: 1.227 -> 1.183 ms for the step), and the host does not
    // wait for the largest length.  Tuning::n50_select = 0: the sort.
    const bool by_select = local && c->pre_run && ss != s && c->tn.n50_select != 0;
    // pre: the keys come from the batches' index kernels, so the chain does not wait for finalize_reads either: it starts
    // when the per-read kernel ends and runs beside finalize_reads and the tables
    // ... and beside the per-read kernel itself, right behind the index kernels, when that kernel is dense_kernel alone:
    // it is bound by its instruction issue at three waves per SIMD and leaves registers and bandwidth for the chain's
    // waves, which tile_kernel (four waves of 128 registers, the HBM saturated) does not (measured: 100 us lost for 60 won).
    const bool beside = pre && (c->tn.sort_beside >= 0 ? c->tn.sort_beside != 0 : c->last_dense_alone);
    if (ss != s) HIPCHK(hipStreamWaitEvent(ss, beside ? c->ev_pre : (pre && c->last_b ? c->last_b : (early ? c->ev_reads : c->ev_max)), 0));
    c->sort_start = c->ev_s0;
    if (ss != s && early && c->n_reads && !(pre && c->last_b)) c->sort_start = c->ev_reads;   // the chain starts right behind it
    else HIPCHK(hipEventRecord(c->ev_s0, ss));
    // The N50 word of the mirror lies outside the copied range (the spare word behind the error words), so the copy,
    // which carries the device's not yet final n50_out along, cannot overwrite it.
    const bool early_out = early_fetch && c->state == 2 && ss != s;
    uint64_t* host_out = early_out ? c->res_host + c->res_words + 7 : nullptr;
    if (by_select) {
        for (int r = 0; r < 3; ++r) {
            HIPCHK(launch_select_hist(dlens, n, kSelShift[r], kSelBits[r], c->sel_state, c->sel_buf, ss, gate, gate_words));
            HIPCHK(launch_select_pick(c->sel_buf, nullptr, c->sel_state, kSelShift[r], kSelBits[r], r == 2, r == 0,
                                      c->n50_out, host_out, ss, gate, gate_words));
        }
    } else if (pre) {
        HIPCHK(hipEventSynchronize(c->ev_pre));   // a wait of the host while the GPU is busy with the tile kernel
        const uint64_t m = c->maxlen_host[4];
        const uint32_t mask = m < (1ull << 8) ? 1u : (m < (1ull << 16) ? 3u : (m < (1ull << 24) ? 7u : 15u));
        const uint32_t* sorted = dlens;
        HIPCHK(sort_u32(dlens, c->sk32, c->sk32_tmp, n, c->counts, mask, ss, &sorted));
        HIPCHK(launch_n50(sorted, n, c->n50_blocks, c->n50_out, host_out, ss));
    } else {
        HIPCHK(sort_u32_gated(dlens, c->sk32, c->sk32_tmp, n, c->counts, gate, gate_words, ss));
        HIPCHK(launch_n50_gated(dlens, c->sk32, c->sk32_tmp, gate, gate_words, n, c->n50_blocks, c->n50_out, host_out, ss));
    }
    c->sort_end = c->ev_s1;
    if (ss == s) HIPCHK(hipEventRecord(c->ev_s1, ss));
    if (ss != s) {
        if (early_out) {   // ppqc_finalize: all but the N50 is final behind the tables kernel; copied beside the sort chain
            CHK(copy_results(c, s));
            c->sum_fetched = true;
        }
        HIPCHK(hipEventRecord(c->ev_join, ss));   // also the end of the sort timing
        c->sort_end = c->ev_join;
        HIPCHK(hipEventRecord(c->ev_t1, s));      // end of the main chain; the finalize ends at the later of the two
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    } else {
        HIPCHK(hipEventRecord(c->ev_t1, s));
    }
    c->end_split = ss != s;
    c->state = 3;
    return mark_busy(c);
}

int ppqc_finalize_n50(ppqc_ctx* c, const uint32_t* dlens, uint64_t n) { return finalize_n50_impl(c, dlens, n, false); }

static int fetch_tables_impl(ppqc_ctx* c, ppqc_tables* o, bool view) {
    if (!c || !o) return fail(PPQC_E_INVALID, "null argument");
    if (c->state < 3) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes/tables/n50 first");
    HIPCHK(hipSetDevice(c->device));
    const TableLayout& l = c->tp.lay;
    // stream-ordered behind the last finalize kernel: one host synchronisation for the whole step
    const bool n50_in_mirror = c->sum_fetched;
    if (!c->sum_fetched) CHK(copy_results(c, c->stream));
    c->sum_fetched = false;
    CHK(ppqc_sync(c));
    if (c->cmh_on_host) HIPCHK(hipEventSynchronize(c->ev_cmhcopy));   // the cells' copy ran on another stream
    {   // the device keeps the minima complemented (one all-reduce MAX merges maxima and minima): decode in place
        uint64_t* mnw = c->res_host + c->tp.lay.n_sum + PPQC_N_SEL + 1 + (uint64_t)c->tp.lay.n_channels;
        for (uint64_t k = 0; k < 1 + (uint64_t)c->tp.lay.n_channels; ++k) mnw[k] = ~mnw[k];
    }
    const uint64_t* h = c->res_host;
    CHK(input_errors(h + c->res_words));
    if (c->cfg.time_mode == 1 && c->m1_phase == 2) {
        uint64_t oob = 0;
        HIPCHK(hipMemcpy(&oob, c->m1 + 2 * (uint64_t)c->cfg.n_channels, 8, hipMemcpyDeviceToHost));
        if (oob)
            return fail(PPQC_E_RANGE, "time_mode 1 across ranks: " + std::to_string(oob) + " reads start outside the " +
                                          std::to_string(PPQC_MODE1_DAYS) + "-day window around the reference time");
    }
    const uint64_t nch = l.n_channels, nb = l.n_time_bins, cells = nch * l.n_mux * nb;
    uint64_t* hv = c->res_host;
#define OUT(dst, off, cnt) do { if (view) dst = hv + (off); else if (dst) memcpy(dst, h + (off), (cnt) * 8); } while (0)
    OUT(o->totals, l.off_totals, PPQC_N_TOTALS); OUT(o->len_hist_log2, l.off_len_log2, PPQC_N_LEN_LOG2);
    OUT(o->len_hist_lin, l.off_len_lin, PPQC_N_LEN_LIN); OUT(o->q_hist, l.off_q_hist, PPQC_N_QHIST);
    OUT(o->gc_hist, l.off_gc_hist, PPQC_N_GCHIST); OUT(o->hour_reads, l.off_hour_reads, nb);
    OUT(o->hour_bases, l.off_hour_bases, nb); OUT(o->chan_reads, l.off_chan_reads, nch);
    OUT(o->chan_bases, l.off_chan_bases, nch); OUT(o->chan_fail, l.off_chan_fail, nch);
    OUT(o->cmh_reads, l.off_cmh_reads, cells); OUT(o->cmh_bases, l.off_cmh_bases, cells);
    const uint64_t* mx = h + l.n_sum + PPQC_N_SEL;
    const uint64_t* mn = mx + 1 + nch;
    OUT(o->chan_max_len, l.n_sum + PPQC_N_SEL + 1, nch);
    OUT(o->chan_min_len, l.n_sum + PPQC_N_SEL + 1 + nch + 1, nch);
#undef OUT
    o->max_len = mx[0];
    o->min_pos_len = mn[0];
    o->lin_bin_width = mx[0] / PPQC_N_LEN_LIN + 1;
    o->n50 = c->n50 = n50_in_mirror ? h[c->res_words + 7] : (mn + 1 + nch)[2];
    c->fin_pending = true;   // timings: see resolve_finalize_timings
    return PPQC_OK;
}

int ppqc_fetch_tables(ppqc_ctx* c, ppqc_tables* o) { return fetch_tables_impl(c, o, false); }
int ppqc_fetch_tables_view(ppqc_ctx* c, ppqc_tables* o) { return fetch_tables_impl(c, o, true); }

static int finalize_impl(ppqc_ctx* c, ppqc_tables* o, bool view) {
    CHK(finalize_extremes_impl(c, true));
    // Every batch left its cells with its index kernel: unpack them on the auxiliary stream, right behind those kernels,
    // and copy the region (13.8 MB at 3000 x 4 x 72, the bulk of the result buffer) to the host mirror on a copy stream --
    // all of it beside the per-read kernel, which this call usually finds running.
    if (c->pre_run && c->cells_early && !c->cells_done) {
        // (a FASTQ text left them with its finalize kernel on the main stream: unpack and copy both on the copy stream,
        // behind that kernel; the main stream goes straight on to the tables, the auxiliary one to the N50)
        hipStream_t us = c->aux;
        if (c->index_on_main) {
            us = c->copy[1];
            HIPCHK(hipStreamWaitEvent(us, c->ev_reads, 0));
        }
        CHK(cells_unpack_on(c, us));
        CHK(cells_to_host_on(c, us, c->copy[1]));
    }
    CHK(ppqc_finalize_tables(c));
    CHK(finalize_n50_impl(c, nullptr, 0, true));
    return fetch_tables_impl(c, o, view);
}
int ppqc_finalize(ppqc_ctx* c, ppqc_tables* o) { return finalize_impl(c, o, false); }
int ppqc_finalize_view(ppqc_ctx* c, ppqc_tables* o) { return finalize_impl(c, o, true); }

int ppqc_device_tables_view(ppqc_ctx* c, ppqc_device_tables* o) {
    if (!c || !o) return fail(PPQC_E_INVALID, "null argument");
    const TableLayout& l = c->tp.lay;
    o->sum_buf = c->sum_buf; o->n_sum = l.n_sum;
    o->max_buf = c->max_buf; o->n_max = 1 + (uint64_t)l.n_channels;
    o->min_buf = c->min_buf; o->n_min = 1 + (uint64_t)l.n_channels;
    o->lengths = c->ra.lens32; o->n_lengths = c->n_reads;
    o->off_totals = l.off_totals; o->off_len_log2 = l.off_len_log2; o->off_len_lin = l.off_len_lin;
    o->off_q_hist = l.off_q_hist; o->off_gc_hist = l.off_gc_hist; o->off_hour_reads = l.off_hour_reads;
    o->off_hour_bases = l.off_hour_bases; o->off_chan_reads = l.off_chan_reads; o->off_chan_bases = l.off_chan_bases;
    o->off_chan_fail = l.off_chan_fail; o->off_cmh_reads = l.off_cmh_reads; o->off_cmh_bases = l.off_cmh_bases;
    o->sel_buf = c->sel_buf; o->n_sel = PPQC_N_SEL;
    return PPQC_OK;
}

int ppqc_n50_select_hist(ppqc_ctx* c, int round) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (round < 0 || round > 2) return fail(PPQC_E_INVALID, "round must be 0, 1 or 2");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    HIPCHK(hipSetDevice(c->device));
    // Round 0 needs only the lengths: when finalize_reads left them (lens_early) it runs on the auxiliary stream, beside
    // the extremes kernel, the caller's extremes all-reduce and the tables kernel, and joins the main stream here.
    const bool side = round == 0 && c->lens_early && c->init_on_aux;   // the clear of sel_buf must precede on aux
    hipStream_t ss = side ? c->aux : c->stream;
    if (side) HIPCHK(hipStreamWaitEvent(ss, c->ev_reads, 0));
    if (round == 0) { c->sort_start = c->ev_s0; HIPCHK(hipEventRecord(c->ev_s0, ss)); }
    HIPCHK(launch_select_hist(c->ra.lens32, c->n_reads, kSelShift[round], kSelBits[round], c->sel_state, c->sel_buf, ss));
    if (side) {
        HIPCHK(hipEventRecord(c->ev_join, ss));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    return mark_busy(c);
}

int ppqc_n50_select_pick(ppqc_ctx* c, int round) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (round < 0 || round > 2) return fail(PPQC_E_INVALID, "round must be 0, 1 or 2");
    if (c->state < 2) return fail(PPQC_E_STATE, "call ppqc_finalize_tables first (the pick needs the merged total)");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_select_pick(c->sel_buf, c->sum_buf + c->tp.lay.off_totals + PPQC_TOT_N_BASES, c->sel_state,
                              kSelShift[round], kSelBits[round], round == 2, false, c->n50_out, nullptr, c->stream));
    if (round == 2) {
        HIPCHK(hipEventRecord(c->ev_s1, c->stream));
        c->sort_end = c->ev_s1;
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        c->end_split = false;
        c->state = 3;
    }
    return mark_busy(c);
}

int ppqc_index_early(ppqc_ctx* c, int* out) {
    if (!c || !out) return fail(PPQC_E_INVALID, "null argument");
    *out = (c->pre_ok && c->res_clean && c->cfg.time_mode == 0) ? 1 : 0;   // also with no reads yet: an empty shard
                                                                          // has to take the same path as its peers
    return PPQC_OK;
}

int ppqc_stream_wait_index(ppqc_ctx* c, void* stream) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (!(c->pre_ok && c->res_clean && c->cfg.time_mode == 0))
        return fail(PPQC_E_STATE, "the run's batches did not leave their index-only results at submit time");
    HIPCHK(hipSetDevice(c->device));
    CHK(join_index(c));   // (extremes that were left to the tables kernel: now)
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_clean, 0));   // the clear of the result buffer (an empty run)
    if (c->n_reads) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->ev_pre, 0));
    return PPQC_OK;
}

int ppqc_n50_select_early(ppqc_ctx* c, int round, int phase, void* stream) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (round < 0 || round > 2 || phase < 0 || phase > 1) return fail(PPQC_E_INVALID, "round must be 0..2, phase 0 or 1");
    if (!(c->pre_ok && c->res_clean && c->cfg.time_mode == 0))
        return fail(PPQC_E_STATE, "the run's batches did not leave their index-only results at submit time");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t ss = (hipStream_t)stream;
    if (phase == 0) {
        HIPCHK(launch_select_hist(c->ra.lens32, c->n_reads, kSelShift[round], kSelBits[round], c->sel_state, c->sel_buf, ss));
    } else {
        HIPCHK(launch_select_pick(c->sel_buf, nullptr, c->sel_state, kSelShift[round], kSelBits[round], round == 2,
                                  round == 0, c->n50_out, nullptr, ss));
        if (round == 2) c->n50_early = true;
    }
    return mark_busy(c);
}

int ppqc_cells_early(ppqc_ctx* c, int phase, void* stream) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (phase < 0 || phase > 1) return fail(PPQC_E_INVALID, "phase must be 0 or 1");
    if (c->cfg.compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE)
        return fail(PPQC_E_STATE, "ppqc_cells_early: PPQC_COMPAT_MULTI_READ_SINGLE rewrites cells behind the tables");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t ss = (hipStream_t)stream;
    if (phase == 0) {
        if (!(c->pre_ok && c->res_clean && c->cfg.time_mode == 0))
            return fail(PPQC_E_STATE, "the run's batches did not leave their index-only results at submit time");
        if (c->cells_early) {
            CHK(cells_unpack_on(c, ss));
        } else {
            // A batch could not leave its cells (the run's reads or bases may overflow a packed cell).  Every rank of a
            // job must make the same calls whatever its data, so this call still leaves the cells, later: behind the
            // per-read kernels, from the run's arrays.
            const uint64_t n_cells = (uint64_t)c->cfg.n_channels * c->cfg.n_mux * c->cfg.n_time_bins;
            if (c->n_reads) HIPCHK(hipStreamWaitEvent(ss, c->ev_reads, 0));
            HIPCHK(hipMemsetAsync(c->sum_buf + c->tp.lay.off_cmh_reads, 0, 2 * n_cells * 8, ss));
            HIPCHK(launch_cells(c->ra, c->n_reads, c->tp, c->sum_buf, ss));
            c->cells_dirty = false;
            c->cells_done = true;
            c->cmh_on_host = false;
        }
    } else {
        if (!c->cells_done) return fail(PPQC_E_STATE, "ppqc_cells_early: phase 0 first");
        CHK(cells_to_host_on(c, ss, ss));
    }
    return mark_busy(c);
}

static int mode1_alloc(ppqc_ctx* c) {
    if (c->m1) return PPQC_OK;
    HIPCHK(hipMalloc((void**)&c->m1, mode1_dist_words(c->cfg.n_channels) * 8));
    return PPQC_OK;
}

int ppqc_mode1_dist_view(ppqc_ctx* c, ppqc_mode1_tables* o) {
    if (!c || !o) return fail(PPQC_E_INVALID, "null argument");
    if (c->cfg.time_mode != 1) return fail(PPQC_E_STATE, "ppqc_mode1_dist_*: the context is not in time_mode 1");
    HIPCHK(hipSetDevice(c->device));
    CHK(mode1_alloc(c));
    const uint64_t nch = c->cfg.n_channels;
    o->first = c->m1; o->n_first = nch;
    o->ties = c->m1 + nch; o->n_ties = nch + 1;
    o->days = c->m1 + 2 * nch + 1; o->n_days = 2 * nch * PPQC_MODE1_DAYS;
    return PPQC_OK;
}

int ppqc_mode1_dist_phase(ppqc_ctx* c, int phase) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    if (c->cfg.time_mode != 1) return fail(PPQC_E_STATE, "ppqc_mode1_dist_*: the context is not in time_mode 1");
    if (phase < 0 || phase > 2) return fail(PPQC_E_INVALID, "phase must be 0, 1 or 2");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    if (phase != c->m1_phase + 1) return fail(PPQC_E_STATE, "ppqc_mode1_dist_phase: phases run in the order 0, 1, 2");
    HIPCHK(hipSetDevice(c->device));
    CHK(mode1_alloc(c));
    HIPCHK(launch_mode1_dist(phase, c->ra, c->n_reads, c->tp, c->m1, c->stream));
    c->m1_phase = phase;
    return mark_busy(c);
}

int ppqc_compat_dist_count(ppqc_ctx* c, void** counts, uint64_t* n) {
    if (!c || !counts || !n) return fail(PPQC_E_INVALID, "null argument");
    if (!(c->cfg.compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE))
        return fail(PPQC_E_STATE, "ppqc_compat_dist_count: the context was created without PPQC_COMPAT_MULTI_READ_SINGLE");
    if (c->state < 1) return fail(PPQC_E_STATE, "call ppqc_finalize_extremes first");
    if (c->state >= 2) return fail(PPQC_E_STATE, "ppqc_compat_dist_count comes before ppqc_finalize_tables");
    HIPCHK(hipSetDevice(c->device));
    if (!c->cc) HIPCHK(hipMalloc((void**)&c->cc, (uint64_t)c->cfg.n_channels * 8 + 64));
    HIPCHK(launch_chan_count(c->ra, c->n_reads, c->cfg.n_channels, c->cc, c->stream));
    c->cc_ready = true;
    *counts = c->cc;
    *n = c->cfg.n_channels;
    return mark_busy(c);
}

int ppqc_set_tuning(ppqc_ctx* c, const char* name, int value) {
    if (!c || !name) return fail(PPQC_E_INVALID, "null argument");
    struct { const char* n; int* p; int lo, hi; } knobs[] = {
        {"tile_dense", &c->tn.tile_dense, -1, 1},       {"dense_len", &c->tn.dense_len, -1, 1 << 16},
        {"tile_all", &c->tn.tile_all, -1, 1},           {"tile_shift", &c->tn.tile_shift, -1, kMaxTileShift},
        {"pre", &c->tn.pre, -1, 1},                     {"fastq_sums", &c->tn.fastq_sums, -1, 2},
        {"fastq_hdr_aux", &c->tn.fastq_hdr_aux, -1, 2}, {"sort_beside", &c->tn.sort_beside, -1, 1},
        {"n50_select", &c->tn.n50_select, -1, 1}, {"fq1_early", &c->tn.fq1_early, -1, 1}, {"fastq_extremes", &c->tn.fastq_extremes, -1, 1},
        {"fastq_one_pass", &c->tn.fastq_one_pass, -1, 1}, {"fq1_regions", &c->tn.fq1_regions, -1, 1 << 20},
    };
    for (auto& k : knobs)
        if (strcmp(name, k.n) == 0) {
            if (value < k.lo || value > k.hi || (k.p == &c->tn.tile_shift && value >= 0 && value < kMinTileShift))
                return fail(PPQC_E_INVALID, std::string("ppqc_set_tuning: value out of range for ") + name);
            *k.p = value;
            return PPQC_OK;
        }
    return fail(PPQC_E_INVALID, std::string("ppqc_set_tuning: unknown knob ") + name);
}

int ppqc_path_count(ppqc_ctx* c, const char* name, uint64_t* out) {
    if (!c || !name || !out) return fail(PPQC_E_INVALID, "null argument");
    if (strcmp(name, "fastq_one_pass") == 0) *out = c->n_fq1;
    else if (strcmp(name, "fastq_declined") == 0) *out = c->n_fq1_declined;
    else if (strcmp(name, "fastq_early") == 0) *out = c->n_fq1_early;
    else if (strcmp(name, "fastq_general") == 0) *out = c->n_fq_general;
    else return fail(PPQC_E_INVALID, std::string("ppqc_path_count: unknown counter ") + name);
    return PPQC_OK;
}

int ppqc_bind_thread(ppqc_ctx* c) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    return PPQC_OK;
}

// ---- one process, several GPUs: RCCL through dlopen ---------------------------------------------------------------------
namespace {
typedef void* nccl_comm;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(nccl_comm*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt64 = 4, kNcclUint64 = 5, kNcclSum = 0, kNcclMax = 2;   // rccl.h: ncclDataType_t, ncclRedOp_t
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return PPQC_OK;
    // a librccl already in the process (torch brings its own) is found first by its soname
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(PPQC_E_NO_DEVICE, std::string("librccl could not be loaded: ") + dlerror());
    Rccl r;
    r.lib = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return fail(PPQC_E_NO_DEVICE, "librccl lacks an expected symbol");
    g_rccl = r;
    return PPQC_OK;
}
}  // namespace

struct ppqc_group {
    std::vector<ppqc_ctx*> ctx;
    std::vector<nccl_comm> comm;
};

#define NCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        int e_ = (expr);                                                                                    \
        if (e_ != 0) return fail(PPQC_E_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(e_));       \
    } while (0)

// one all-reduce, in place, of `count` 64-bit words at the same place of every context (picked by `where`)
typedef uint64_t* (*group_where)(ppqc_ctx*, uint64_t);
static uint64_t* at_maxmin(ppqc_ctx* c, uint64_t) { return c->max_buf; }
static uint64_t* at_sum(ppqc_ctx* c, uint64_t) { return c->sum_buf; }
static uint64_t* at_sel(ppqc_ctx* c, uint64_t) { return c->sel_buf; }
static uint64_t* at_m1(ppqc_ctx* c, uint64_t off) { return c->m1 + off; }
static uint64_t* at_cc(ppqc_ctx* c, uint64_t) { return c->cc; }
static int group_all_reduce(ppqc_group* g, group_where where, uint64_t off, uint64_t count, int op) {
    if (count == 0) return PPQC_OK;
    NCCLCHK(g_rccl.GroupStart());
    for (size_t k = 0; k < g->ctx.size(); ++k) {
        void* p = where(g->ctx[k], off);
        const int e = g_rccl.AllReduce(p, p, (size_t)count, op == kNcclSum ? kNcclUint64 : kNcclInt64, op, g->comm[k],
                                       g->ctx[k]->stream);
        if (e != 0) { (void)g_rccl.GroupEnd(); return fail(PPQC_E_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(e)); }
    }
    NCCLCHK(g_rccl.GroupEnd());
    for (ppqc_ctx* c : g->ctx) c->res_busy = true;
    return PPQC_OK;
}

int ppqc_group_create(ppqc_group** out, ppqc_ctx* const* ctxs, int n) {
    if (!out || !ctxs || n < 1) return fail(PPQC_E_INVALID, "null argument or no context");
    *out = nullptr;
    std::vector<int> devs;
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) return fail(PPQC_E_INVALID, "null context in the group");
        for (int d : devs)
            if (d == ctxs[k]->device) return fail(PPQC_E_INVALID, "two contexts of a group on one GPU");
        devs.push_back(ctxs[k]->device);
        const ppqc_config &a = ctxs[0]->cfg, &b = ctxs[k]->cfg;
        if (a.n_channels != b.n_channels || a.n_mux != b.n_mux || a.n_time_bins != b.n_time_bins ||
            a.bin_seconds != b.bin_seconds || a.time_mode != b.time_mode || a.tz_offset_s != b.tz_offset_s ||
            a.fail_q != b.fail_q || a.compat_flags != b.compat_flags)
            return fail(PPQC_E_INVALID, "the contexts of a group must share one configuration");
    }
    CHK(load_rccl());
    ppqc_group* g = new (std::nothrow) ppqc_group();
    if (!g) return fail(PPQC_E_NOMEM, "out of host memory");
    g->ctx.assign(ctxs, ctxs + n);
    g->comm.assign(n, nullptr);
    const int e = g_rccl.CommInitAll(g->comm.data(), n, devs.data());
    if (e != 0) {
        delete g;
        return fail(PPQC_E_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e));
    }
    *out = g;
    return PPQC_OK;
}

int ppqc_group_destroy(ppqc_group* g) {
    if (!g) return PPQC_OK;
    for (size_t k = 0; k < g->comm.size(); ++k) {
        (void)hipSetDevice(g->ctx[k]->device);
        (void)hipStreamSynchronize(g->ctx[k]->stream);
        if (g->comm[k]) (void)g_rccl.CommDestroy(g->comm[k]);
    }
    delete g;
    return PPQC_OK;
}

int ppqc_group_finalize(ppqc_group* g, ppqc_tables* out) {
    if (!g || !out) return fail(PPQC_E_INVALID, "null argument");
    const uint64_t nch1 = 1 + (uint64_t)g->ctx[0]->cfg.n_channels, nch = g->ctx[0]->cfg.n_channels;
    const uint64_t n_sum = g->ctx[0]->tp.lay.n_sum;
    int64_t ref = g->ctx[0]->tp.ref_time_s;
    for (ppqc_ctx* c : g->ctx) {
        if (c->tp.ref_time_s != ref) return fail(PPQC_E_INVALID, "the contexts of a group must share one reference time");
        CHK(ppqc_finalize_extremes(c));
    }
    CHK(group_all_reduce(g, at_maxmin, 0, 2 * nch1, kNcclMax));
    if (g->ctx[0]->cfg.time_mode == 1) {
        for (ppqc_ctx* c : g->ctx) { ppqc_mode1_tables v; CHK(ppqc_mode1_dist_view(c, &v)); CHK(ppqc_mode1_dist_phase(c, 0)); }
        CHK(group_all_reduce(g, at_m1, 0, nch, kNcclMax));
        for (ppqc_ctx* c : g->ctx) CHK(ppqc_mode1_dist_phase(c, 1));
        CHK(group_all_reduce(g, at_m1, nch, nch + 1, kNcclSum));
        CHK(group_all_reduce(g, at_m1, 2 * nch + 1, 2 * nch * PPQC_MODE1_DAYS, kNcclMax));
        for (ppqc_ctx* c : g->ctx) CHK(ppqc_mode1_dist_phase(c, 2));
    }
    if (g->ctx[0]->cfg.compat_flags & PPQC_COMPAT_MULTI_READ_SINGLE) {   // a channel's read count in the whole job
        for (ppqc_ctx* c : g->ctx) { void* p; uint64_t k; CHK(ppqc_compat_dist_count(c, &p, &k)); }
        CHK(group_all_reduce(g, at_cc, 0, nch, kNcclSum));
    }
    for (ppqc_ctx* c : g->ctx) CHK(ppqc_finalize_tables(c));
    for (int r = 0; r < 3; ++r) {
        for (ppqc_ctx* c : g->ctx) CHK(ppqc_n50_select_hist(c, r));
        if (r == 0) CHK(group_all_reduce(g, at_sum, 0, n_sum + PPQC_N_SEL, kNcclSum));
        else CHK(group_all_reduce(g, at_sel, 0, PPQC_N_SEL, kNcclSum));
        for (ppqc_ctx* c : g->ctx) CHK(ppqc_n50_select_pick(c, r));
    }
    for (size_t k = 1; k < g->ctx.size(); ++k) CHK(ppqc_sync(g->ctx[k]));
    return ppqc_fetch_tables(g->ctx[0], out);
}

int ppqc_set_stream(ppqc_ctx* c, void* stream, int use_external) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t ns = use_external ? (hipStream_t)stream : c->own_stream;
    if (ns == c->stream) return PPQC_OK;
    CHK(ppqc_sync(c));
    c->stream = ns;
    return PPQC_OK;
}

int ppqc_timings(ppqc_ctx* c, ppqc_timing* o) {
    if (!c || !o) return fail(PPQC_E_INVALID, "null argument");
    CHK(ppqc_sync(c));
    CHK(resolve_timings(c));
    *o = c->timing;
    return PPQC_OK;
}

int ppqc_reset_timings(ppqc_ctx* c) {
    if (!c) return fail(PPQC_E_INVALID, "null context");
    CHK(ppqc_sync(c));
    CHK(resolve_timings(c));
    c->timing = ppqc_timing{};
    return PPQC_OK;
}

int ppqc_sort_u32_device(ppqc_ctx* c, uint32_t* keys, uint64_t n) {
    if (!c || (!keys && n)) return fail(PPQC_E_INVALID, "null argument");
    if (n >= (1ull << 32)) return fail(PPQC_E_RANGE, "more than 2^32-1 keys");
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_sort32(c, n ? n : 1));
    const uint32_t* out = keys;
    HIPCHK(sort_u32(keys, c->sk32_tmp, keys, n, c->counts, 0xFu, c->stream, &out));   // four passes end in `keys`
    if (out != keys) HIPCHK(hipMemcpyAsync(keys, out, n * 4, hipMemcpyDeviceToDevice, c->stream));
    return PPQC_OK;
}

// The same sort for callers that cannot allocate: the ping-pong buffer and the digit counters are the file-scope
// `scratch` array (per-device address from hipGetSymbolAddress, see create_impl).  No hipMalloc / hipFree here or in
// anything it calls; the kernels take the scratch pointers as arguments exactly as in ppqc_sort_u32_device.  Nothing in
// the scratch has to start at zero (every counter the scan and scatter read is written by the pass's histogram kernel).
int ppqc_sort_u32_device_static(ppqc_ctx* c, uint32_t* keys, uint64_t n) {
    if (!c || (!keys && n)) return fail(PPQC_E_INVALID, "null argument");
    if (n > kStaticSortMaxKeys)
        return fail(PPQC_E_RANGE, "ppqc_sort_u32_device_static: " + std::to_string(n) + " keys exceed the static scratch (" +
                                      std::to_string(kStaticSortMaxKeys) + " keys); use ppqc_sort_u32_device");
    if (c->device < 0 || c->device >= kMaxDevices || !g_static_scratch[c->device])
        return fail(PPQC_E_STATE, "no static scratch address for this device");
    if (sort_count_words(n ? n : 1) > kStaticSortCountWords) return fail(PPQC_E_RANGE, "counter scratch too small");
    HIPCHK(hipSetDevice(c->device));
    unsigned char* base = g_static_scratch[c->device];
    uint32_t* tmp = reinterpret_cast<uint32_t*>(base);
    uint32_t* counts = reinterpret_cast<uint32_t*>(base + kStaticSortMaxKeys * 4);
    const uint32_t* out = keys;
    HIPCHK(sort_u32(keys, tmp, keys, n, counts, 0xFu, c->stream, &out));   // four passes: keys -> tmp -> keys -> tmp -> keys
    if (out != keys) return fail(PPQC_E_STATE, "sorted keys did not end in the caller's buffer");
    return PPQC_OK;
}

int ppqc_synth_fill_device(ppqc_ctx* c, uint8_t* bases, uint8_t* quals, uint64_t first, uint64_t n, uint64_t seed) {
    if (!c || ((!bases || !quals) && n)) return fail(PPQC_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_synth(bases, quals, first, n, seed, c->stream));
    return PPQC_OK;
}

}  // extern "C"
