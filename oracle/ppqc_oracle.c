/* CPU oracle ("port") for the per-read QC statistics path of PyPore `seqstats`, plain C.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product library (pypore_amd/csrc) never calls into it.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and cannot run here (Python 2,
 * h5py/Biopython absent); this file restates the reference source plus the SURVEY.md §8a-11
 * definitions, and is itself pinned against oracle/pypore_oracle.py (the line-by-line Python
 * restatement) and the hand-derived known answers in tests/golden/.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   GC count / percent ............ Bio.SeqUtils.GC (Biopython <= 1.76, third party); call sites
 *                                   lib/fast5_reader.py:58, lib/multi_read_fast5_reader.py:107,
 *                                   lib/seq_routines.py:52
 *   fail defaults (len 0, q 6.99) .. lib/fast5_reader.py:37-40,52
 *   per-channel aggregation ........ lib/fast5_reader.py:117-145 == lib/multi_read_fast5_reader.py:47-77
 *   hour bin + wrap quirk .......... lib/fast5_reader.py:118,122-126
 *   run-level sums ................. lib/seq_routines.py:99-115, 125-137
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ERR_SHIFT 52
#define PHRED 33

typedef unsigned __int128 u128;

void orc_err_lut(uint64_t *lut) {
    for (int b = 0; b < 256; ++b) {
        int q = b - PHRED;
        if (q < 0) q = 0;
        if (q > 93) q = 93;
        lut[b] = (uint64_t)llrint(ldexp(pow(10.0, -(double)q / 10.0), ERR_SHIFT));
    }
}

uint64_t orc_q_threshold(double q) { return (uint64_t)llrint(ldexp(pow(10.0, -q / 10.0), ERR_SHIFT)); }

typedef struct {
    const uint8_t *bases, *quals;
    const uint64_t *off;
    uint64_t r0, r1;
    uint32_t *len, *gc, *n;
    int64_t *qsum;
    uint64_t *err_hi, *err_lo;
    double *mean_q;
    const uint64_t *lut;
    const uint8_t *cls;
} job_t;

static void *per_read_worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (uint64_t r = j->r0; r < j->r1; ++r) {
        uint64_t a = j->off[r], b = j->off[r + 1];
        uint32_t gc = 0, nn = 0;
        uint64_t qs = 0;
        u128 S = 0;
        /* GC(): sum of seq.count(x) for x in "GCgcSs"; N count: 'N' + 'n' (class table = the same predicate).
         * The fixed-point sum is folded into 128 bits every 2048 bytes (2^11 * 2^52 < 2^64). */
        for (uint64_t i = a; i < b;) {
            uint64_t e = i + 2048 < b ? i + 2048 : b, s64 = 0;
            for (; i < e; ++i) {
                uint8_t k = j->cls[j->bases[i]];
                gc += k & 1u;
                nn += k >> 1;
                uint8_t q = j->quals[i];
                qs += q;
                s64 += j->lut[q];
            }
            S += s64;
        }
        uint64_t L = b - a;
        j->len[r] = (uint32_t)L;
        j->gc[r] = gc;
        j->n[r] = nn;
        j->qsum[r] = (int64_t)qs - (int64_t)PHRED * (int64_t)L;
        j->err_hi[r] = (uint64_t)(S >> 64);
        j->err_lo[r] = (uint64_t)S;
        /* mean_q = -10 log10( S / (L 2^52) ), the exact sum rounded once to fp64 */
        j->mean_q[r] = L ? -10.0 * log10((double)S / ((double)L * 4503599627370496.0)) : 0.0;
    }
    return NULL;
}

void orc_per_read(const uint8_t *bases, const uint8_t *quals, const uint64_t *off, uint64_t R, uint32_t *len,
                  uint32_t *gc, uint32_t *n, int64_t *qsum, uint64_t *err_hi, uint64_t *err_lo, double *mean_q,
                  int nthreads) {
    uint64_t lut[256];
    uint8_t cls[256];
    orc_err_lut(lut);
    for (int c = 0; c < 256; ++c)
        cls[c] = (uint8_t)(((c == 'G') | (c == 'C') | (c == 'g') | (c == 'c') | (c == 'S') | (c == 's')) |
                           (((c == 'N') | (c == 'n')) << 1));
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    /* split by bytes, not by reads, so threads get equal work */
    uint64_t total = R ? off[R] - off[0] : 0, r = 0;
    for (int t = 0; t < nthreads; ++t) {
        uint64_t target = off[0] + (total / nthreads) * (t + 1);
        uint64_t r1 = r;
        if (t == nthreads - 1) r1 = R;
        else while (r1 < R && off[r1 + 1] <= target) ++r1;
        jobs[t] = (job_t){bases, quals, off, r, r1, len, gc, n, qsum, err_hi, err_lo, mean_q, lut, cls};
        r = r1;
    }
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, per_read_worker, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ---- N50: sort descending, running sum, first length with 2c >= total ---------------------- */
static int cmp_u32_desc(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x < y) - (x > y);
}

uint64_t orc_n50(const uint32_t *lens, uint64_t n) {
    if (!n) return 0;
    uint32_t *s = (uint32_t *)malloc(n * sizeof(uint32_t));
    memcpy(s, lens, n * sizeof(uint32_t));
    qsort(s, n, sizeof(uint32_t), cmp_u32_desc);
    uint64_t total = 0, c = 0, res = 0;
    for (uint64_t i = 0; i < n; ++i) total += s[i];
    if (total)
        for (uint64_t i = 0; i < n; ++i) {
            c += s[i];
            if (2 * c >= total) { res = s[i]; break; }
        }
    free(s);
    return res;
}

/* ---- run-level tables ----------------------------------------------------------------------- */
typedef struct {
    uint32_t n_channels, n_mux, n_time_bins, bin_seconds, time_mode;
    int64_t tz_offset_s;
    int64_t ref_time_s;
    double fail_q;
    uint32_t compat_flags;   /* bit 0: channel_parser's single-read channel, lib/multi_read_fast5_reader.py:64-67 */
} orc_config;

enum { TOT_N_READS, TOT_N_BASES, TOT_N_FAIL, TOT_N_GC, TOT_N_N, TOT_QSUM, TOT_N_BINNED, TOT_BAD_CHANNEL,
       TOT_TIME_OOB, TOT_ZERO_LEN, TOT_BAD_MUX, TOT_NO_TIME, N_TOTALS = 16 };
#define N_LEN_LOG2 40
#define N_LEN_LIN 100
#define N_QHIST 8
#define N_GCHIST 20

typedef struct { uint32_t ch; int64_t t; uint64_t idx; } key_t_;
static int cmp_key(const void *a, const void *b) {
    const key_t_ *x = (const key_t_ *)a, *y = (const key_t_ *)b;
    if (x->ch != y->ch) return x->ch < y->ch ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int64_t floordiv(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

/* All table pointers are caller-allocated and zero-initialised here. chan_min is filled with INT64_MAX. */
void orc_tables(const orc_config *cfg, uint64_t R, const uint32_t *len_in, const uint32_t *gc_in, const uint32_t *n_in,
                const int64_t *qsum_in, const uint64_t *err_hi, const uint64_t *err_lo, const uint32_t *channel,
                const uint8_t *mux, const int64_t *start_time, const uint8_t *fail_flag,
                const uint8_t *is_record, const double *mean_q,
                int64_t *totals, int64_t *len_log2, int64_t *len_lin, int64_t *q_hist, int64_t *gc_hist,
                int64_t *hour_reads, int64_t *hour_bases, int64_t *chan_reads, int64_t *chan_bases, int64_t *chan_fail,
                int64_t *cmh_reads, int64_t *cmh_bases, int64_t *chan_min, int64_t *chan_max,
                int64_t *scalars /* [max_len, min_pos_len, lin_bin_width, n50] */) {
    const uint32_t nch = cfg->n_channels, nmux = cfg->n_mux, nb = cfg->n_time_bins;
    memset(totals, 0, N_TOTALS * 8); memset(len_log2, 0, N_LEN_LOG2 * 8); memset(len_lin, 0, N_LEN_LIN * 8);
    memset(q_hist, 0, N_QHIST * 8); memset(gc_hist, 0, N_GCHIST * 8);
    memset(hour_reads, 0, nb * 8); memset(hour_bases, 0, nb * 8);
    memset(chan_reads, 0, nch * 8); memset(chan_bases, 0, nch * 8); memset(chan_fail, 0, nch * 8);
    memset(chan_max, 0, nch * 8);
    memset(cmh_reads, 0, (size_t)nch * nmux * nb * 8); memset(cmh_bases, 0, (size_t)nch * nmux * nb * 8);
    for (uint32_t c = 0; c < nch; ++c) chan_min[c] = INT64_MAX;

    uint64_t thr_fail = orc_q_threshold(cfg->fail_q), thr[N_QHIST + 1];
    for (int j = 0; j <= N_QHIST; ++j) thr[j] = orc_q_threshold(6.0 + j);

    int64_t *hour = (int64_t *)malloc((R ? R : 1) * 8);
    uint32_t *len = (uint32_t *)malloc((R ? R : 1) * 4);
    for (uint64_t i = 0; i < R; ++i) {
        int flagged = fail_flag && fail_flag[i];
        len[i] = flagged ? 0 : len_in[i];
        if (start_time[i] == INT64_MIN) { hour[i] = -1; continue; }   /* no start time: no hour bin */
        int64_t sec = start_time[i] - cfg->ref_time_s + cfg->tz_offset_s;
        if (cfg->time_mode == 1) { int64_t h = floordiv(sec, cfg->bin_seconds) % 24; hour[i] = h < 0 ? h + 24 : h; }
        else hour[i] = floordiv(sec, cfg->bin_seconds);
    }
    if (cfg->time_mode == 1 && R) {
        /* per channel, in start-time order: index e > 1 and hour < previous (already fixed) hour => +24 */
        /* reads without a start time are not part of any channel's time order */
        key_t_ *k = (key_t_ *)malloc(R * sizeof(key_t_));
        uint64_t Rt = 0;
        for (uint64_t i = 0; i < R; ++i)
            if (start_time[i] != INT64_MIN) k[Rt++] = (key_t_){channel[i], start_time[i], i};
        qsort(k, Rt, sizeof(key_t_), cmp_key);
        uint64_t s = 0;
        const uint64_t R_all = R;
        R = Rt;
        while (s < R) {
            uint64_t e = s;
            while (e < R && k[e].ch == k[s].ch) ++e;
            for (uint64_t p = s + 2; p < e; ++p)
                if (hour[k[p].idx] < hour[k[p - 1].idx]) hour[k[p].idx] += 24;
            s = e;
        }
        free(k);
        R = R_all;
    }
    int64_t max_len = 0, min_pos = INT64_MAX;
    for (uint64_t i = 0; i < R; ++i) {
        if (channel[i] < 1 || channel[i] > nch) continue;
        if (len[i] > max_len) max_len = len[i];
        if (len[i] > 0 && len[i] < min_pos) min_pos = len[i];
    }
    int64_t lin_w = max_len / N_LEN_LIN + 1;
    uint32_t *sel_len = (uint32_t *)calloc(R ? R : 1, 4);
    totals[TOT_N_READS] = (int64_t)R;
    for (uint64_t i = 0; i < R; ++i) {
        if (channel[i] < 1 || channel[i] > nch) { totals[TOT_BAD_CHANNEL]++; continue; }
        int flagged = fail_flag && fail_flag[i];
        int64_t L = len[i];
        uint32_t c = channel[i] - 1;
        int64_t gc = flagged ? 0 : gc_in[i];
        u128 S = ((u128)err_hi[i] << 64) | err_lo[i];
        int fail, qbin = -1;
        if (flagged) { fail = 6.99 < cfg->fail_q; qbin = 0; }
        else if (is_record && is_record[i]) {   /* sequencing_summary.txt row: float rule, lib/seq_routines.py:855 */
            double q = mean_q[i];
            fail = q < cfg->fail_q;
            if (q >= 6.0 && q < 14.0) qbin = (int)floor(q - 6.0);
        }
        else if (L == 0) fail = 0.0 < cfg->fail_q;
        else {
            fail = S > (u128)L * thr_fail;
            for (int j = 0; j < N_QHIST; ++j)
                if (S <= (u128)L * thr[j] && S > (u128)L * thr[j + 1]) { qbin = j; break; }
        }
        sel_len[i] = (uint32_t)L;
        totals[TOT_N_BASES] += L; totals[TOT_N_FAIL] += fail; totals[TOT_N_GC] += gc;
        totals[TOT_N_N] += flagged ? 0 : n_in[i]; totals[TOT_QSUM] += flagged ? 0 : qsum_in[i];
        totals[TOT_ZERO_LEN] += (L == 0);
        totals[TOT_NO_TIME] += (start_time[i] == INT64_MIN);
        len_log2[L == 0 ? 0 : 64 - __builtin_clzll((unsigned long long)L)]++;
        len_lin[L / lin_w]++;
        if (qbin >= 0) q_hist[qbin]++;
        if (gc > 0 && L != 0) { int64_t b = 20 * gc / L; gc_hist[b > N_GCHIST - 1 ? N_GCHIST - 1 : b]++; }
        chan_reads[c]++; chan_bases[c] += L;
        if (L < chan_min[c]) chan_min[c] = L;
        if (L > chan_max[c]) chan_max[c] = L;
        if (mux[i] < 1 || mux[i] > nmux) { totals[TOT_BAD_MUX]++; continue; }
        chan_fail[c] += fail;
        int64_t h = hour[i];
        if (h < 0 || h >= (int64_t)nb) { totals[TOT_TIME_OOB]++; continue; }
        size_t cell = ((size_t)c * nmux + (mux[i] - 1)) * nb + (size_t)h;
        totals[TOT_N_BINNED]++;
        hour_reads[h]++; hour_bases[h] += L;
        cmh_reads[cell]++; cmh_bases[cell] += L;
    }
    if (cfg->compat_flags & 1u) {
        /* channel_parser, lib/multi_read_fast5_reader.py:64-67: `if len(l) > 1` the reads are filtered by mux, `else`
         * MucList = l -- the only read of a channel is taken under EVERY mux '1'..'4', whatever its own mux: its fail
         * counts once per mux (:69-70), every mux gets {hour: (1, length)} (:71-77), and FastStats' sum over channels
         * and muxes (lib/seq_routines.py:103-108) sees it n_mux times. */
        for (uint64_t i = 0; i < R; ++i) {
            if (channel[i] < 1 || channel[i] > nch) continue;
            uint32_t c = channel[i] - 1;
            if (chan_reads[c] != 1) continue;
            int flagged = fail_flag && fail_flag[i];
            int64_t L = len[i];
            int fail;
            if (flagged) fail = 6.99 < cfg->fail_q;
            else if (is_record && is_record[i]) fail = mean_q[i] < cfg->fail_q;
            else if (L == 0) fail = 0.0 < cfg->fail_q;
            else fail = (((u128)err_hi[i] << 64) | err_lo[i]) > (u128)L * thr_fail;
            chan_fail[c] = (int64_t)nmux * fail;
            int64_t h = hour[i];
            if (h < 0 || h >= (int64_t)nb) continue;
            int normal = mux[i] >= 1 && mux[i] <= nmux;
            for (uint32_t m = 0; m < nmux; ++m) {
                size_t cell = ((size_t)c * nmux + m) * nb + (size_t)h;
                cmh_reads[cell] = 1; cmh_bases[cell] = L;
            }
            hour_reads[h] += (int64_t)nmux - normal;
            hour_bases[h] += ((int64_t)nmux - normal) * L;
        }
    }
    scalars[0] = max_len; scalars[1] = min_pos; scalars[2] = lin_w; scalars[3] = (int64_t)orc_n50(sel_len, R);
    free(sel_len); free(hour); free(len);
}

/* ---- synthetic workload generator (bench.py's cpu_baseline leg only) ------------------------------------------------
 * Host twin of the counter-based generator the benchmark fills HBM with (byte i of each stream depends on (seed, i)
 * alone: word j = i / 4, hb = mix64(seed ^ (2j) M), hq = mix64(seed ^ (2j+1) M), 16 hash bits per byte), so that the CPU
 * leg can regenerate a prefix of the workload in a process that never touches the GPU.  No reference counterpart;
 * tests/test_synth.py holds it to pypore_amd/synth.py:stream_bytes. */
static uint64_t synth_mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
typedef struct { uint8_t *bases, *quals; uint64_t first, n, seed, j0, j1; } synth_job_t;
static void *synth_worker(void *arg) {
    synth_job_t *s = (synth_job_t *)arg;
    const uint64_t M = 0xD1342543DE82EF95ull;
    for (uint64_t j = s->j0; j < s->j1; ++j) {
        uint64_t hb = synth_mix64(s->seed ^ ((2 * j) * M)), hq = synth_mix64(s->seed ^ ((2 * j + 1) * M));
        for (int b = 0; b < 4; ++b) {
            uint64_t i = 4 * j + b;
            if (i < s->first || i >= s->first + s->n) continue;
            uint32_t v = (uint32_t)(hb >> (16 * b)) & 0xFFFFu;
            s->bases[i - s->first] = v < 7u ? 'N' : (v < 26874u ? ((v & 1u) ? 'G' : 'C') : ((v & 1u) ? 'A' : 'T'));
            v = (uint32_t)(hq >> (16 * b)) & 0xFFFFu;
            int q = (int)(4u * ((v & 15u) + ((v >> 4) & 15u) + ((v >> 8) & 15u) + (v >> 12)) / 9u) - 1;
            s->quals[i - s->first] = (uint8_t)(33 + (q < 1 ? 1 : (q > 60 ? 60 : q)));
        }
    }
    return NULL;
}
void orc_synth_fill(uint8_t *bases, uint8_t *quals, uint64_t first, uint64_t n, uint64_t seed, int nthreads) {
    if (!n) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    const uint64_t w0 = first / 4, w1 = (first + n + 3) / 4, per = (w1 - w0 + nthreads - 1) / nthreads;
    pthread_t th[256];
    synth_job_t jobs[256];
    int used = 0;
    for (int t = 0; t < nthreads; ++t) {
        uint64_t a = w0 + per * t, e = a + per < w1 ? a + per : w1;
        if (a >= e) break;
        jobs[used] = (synth_job_t){bases, quals, first, n, seed, a, e};
        pthread_create(&th[used], NULL, synth_worker, &jobs[used]);
        ++used;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
}
