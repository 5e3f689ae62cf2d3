// Host-side 4-line FASTQ tokenizer: text -> CSR (bases, quals, offsets) + header fields, straight into caller buffers
// (pinned for the H2D double buffer).  Replaces the record parsing the reference leaves to Biopython:
// SeqIO.parse(handle, "fastq") at lib/seq_routines.py:50 (albacore FASTQ files) and SeqIO.read(StringIO(...), "fastq")
// at lib/fastqparser.py:12 / lib/fast5_reader.py:76; record layout per lib/fast5_reader.py:57 (bases = line 2) and
// Sanger FASTQ (quals = line 4, same length).  Malformed records are rejected where SeqIO would raise ValueError.
#include <stdint.h>
#include <string.h>

#include <string>

#include "../../include/pypore_qc.h"

int ppqc_set_error(int code, const std::string& msg);   // ppqc_api.hip

namespace {

struct Line { const uint8_t* p; uint64_t len; };

// next line of [cur, end); returns false if no complete line is available (no '\n' and not final)
inline bool next_line(const uint8_t*& cur, const uint8_t* end, bool final, Line* out) {
    if (cur >= end) return false;
    const uint8_t* nl = (const uint8_t*)memchr(cur, '\n', (size_t)(end - cur));
    const uint8_t* e;
    if (nl) e = nl;
    else if (final) e = end;
    else return false;
    out->p = cur;
    out->len = (uint64_t)(e - cur);
    // right-strip ASCII whitespace as SeqIO does with every line; this also turns "\r\n" into a line end
    // (universal newlines, "rU" at lib/seq_routines.py:49)
    while (out->len) {
        const uint8_t ch = out->p[out->len - 1];
        if (ch == ' ' || (ch >= '\t' && ch <= '\r')) out->len--;
        else break;
    }
    cur = nl ? nl + 1 : end;
    return true;
}

inline bool has_byte(const Line& l, uint8_t b) { return l.len && memchr(l.p, b, (size_t)l.len) != nullptr; }

inline const uint8_t* find_field(const uint8_t* h, uint64_t n, const char* key, uint64_t klen) {
    // fields are whitespace separated: " key=value"
    for (uint64_t i = 0; i + klen <= n; ++i)
        if ((i == 0 || h[i - 1] == ' ' || h[i - 1] == '\t') && memcmp(h + i, key, klen) == 0) return h + i + klen;
    return nullptr;
}

inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

// "YYYY-MM-DDTHH:MM:SS[Z]" -> seconds since the epoch, UTC; INT64_MIN if it does not parse
inline int64_t parse_iso8601(const uint8_t* s, uint64_t n) {
    if (n < 19) return INT64_MIN;
    auto num = [&](int a, int b, int64_t* v) {
        int64_t x = 0;
        for (int i = a; i < b; ++i) { if (s[i] < '0' || s[i] > '9') return false; x = x * 10 + (s[i] - '0'); }
        *v = x; return true;
    };
    int64_t Y, M, D, h, m, sec;
    if (!num(0, 4, &Y) || s[4] != '-' || !num(5, 7, &M) || s[7] != '-' || !num(8, 10, &D) || (s[10] != 'T' && s[10] != ' ') ||
        !num(11, 13, &h) || s[13] != ':' || !num(14, 16, &m) || s[16] != ':' || !num(17, 19, &sec))
        return INT64_MIN;
    if (M < 1 || M > 12 || D < 1 || D > 31) return INT64_MIN;
    return days_from_civil(Y, (unsigned)M, (unsigned)D) * 86400 + h * 3600 + m * 60 + sec;
}

}  // namespace

extern "C" {

int ppqc_fastq_scan(const uint8_t* text, uint64_t n, int final, uint64_t* n_records, uint64_t* n_bases,
                    uint64_t* consumed) {
    if ((!text && n) || !n_records || !n_bases || !consumed) return ppqc_set_error(PPQC_E_INVALID, "null argument");
    const uint8_t* cur = text;
    const uint8_t* end = text + n;
    uint64_t recs = 0, bases = 0;
    for (;;) {
        const uint8_t* rec_start = cur;
        // skip blank lines between records (SeqIO tolerates trailing blank lines)
        Line h, s, p, q;
        if (!next_line(cur, end, final != 0, &h)) { cur = rec_start; break; }
        if (h.len == 0) continue;
        if (!next_line(cur, end, final != 0, &s) || !next_line(cur, end, final != 0, &p) ||
            !next_line(cur, end, final != 0, &q)) {
            if (final) return ppqc_set_error(PPQC_E_INVALID, "truncated FASTQ record " + std::to_string(recs));
            cur = rec_start;
            break;
        }
        const std::string where = "FASTQ record " + std::to_string(recs) + ": ";
        if (h.p[0] != '@') return ppqc_set_error(PPQC_E_INVALID, where + "header does not start with '@'");
        if (p.len == 0 || p.p[0] != '+')
            return ppqc_set_error(PPQC_E_INVALID, where + "third line does not start with '+' (multi-line records are not supported)");
        if (p.len > 1 && (p.len != h.len || memcmp(p.p + 1, h.p + 1, (size_t)(h.len - 1)) != 0))
            return ppqc_set_error(PPQC_E_INVALID, where + "sequence and quality captions differ");
        if (has_byte(s, ' ') || has_byte(s, '\t')) return ppqc_set_error(PPQC_E_INVALID, where + "whitespace in the sequence");
        if (has_byte(s, '\r') || has_byte(q, '\r') || has_byte(h, '\r'))
            return ppqc_set_error(PPQC_E_INVALID, where + "bare CR line terminators are not supported");
        if (s.len != q.len) return ppqc_set_error(PPQC_E_INVALID, where + "sequence and quality lengths differ");
        if (s.len >= 0xFFFFFFFFull) return ppqc_set_error(PPQC_E_RANGE, "FASTQ record longer than 2^32-2");
        ++recs;
        bases += s.len;
    }
    *n_records = recs; *n_bases = bases; *consumed = (uint64_t)(cur - text);
    return PPQC_OK;
}

int ppqc_fastq_pack(const uint8_t* text, uint64_t n, uint64_t n_records, uint8_t* bases, uint8_t* quals,
                    uint64_t* offsets, uint64_t* header_off, uint32_t* header_len, uint32_t* channel,
                    int64_t* start_time_s, uint64_t* n_bad_qual) {
    if ((!text && n) || (n_records && (!bases || !quals || !offsets))) return ppqc_set_error(PPQC_E_INVALID, "null argument");
    const uint8_t* cur = text;
    const uint8_t* end = text + n;
    uint64_t pos = offsets ? offsets[0] : 0, bad = 0;
    uint8_t* pb = bases;
    uint8_t* pq = quals;
    for (uint64_t r = 0; r < n_records;) {
        Line h, s, p, q;
        if (!next_line(cur, end, true, &h)) return ppqc_set_error(PPQC_E_INVALID, "FASTQ text shorter than scanned");
        if (h.len == 0) continue;
        if (!next_line(cur, end, true, &s) || !next_line(cur, end, true, &p) || !next_line(cur, end, true, &q) ||
            s.len != q.len)
            return ppqc_set_error(PPQC_E_INVALID, "FASTQ text changed since ppqc_fastq_scan");
        memcpy(pb, s.p, s.len);
        memcpy(pq, q.p, q.len);
        for (uint64_t i = 0; i < q.len; ++i) bad += (q.p[i] < 33) | (q.p[i] > 126);   // SeqIO would reject these
        pb += s.len; pq += s.len; pos += s.len;
        offsets[r + 1] = pos;
        if (header_off) header_off[r] = (uint64_t)(h.p + 1 - text);
        if (header_len) header_len[r] = (uint32_t)(h.len - 1);
        if (channel) {
            const uint8_t* v = find_field(h.p + 1, h.len - 1, "ch=", 3);
            uint32_t c = 0;
            if (v) for (const uint8_t* e = h.p + h.len; v < e && *v >= '0' && *v <= '9'; ++v) c = c * 10 + (uint32_t)(*v - '0');
            channel[r] = c;
        }
        if (start_time_s) {
            const uint8_t* v = find_field(h.p + 1, h.len - 1, "start_time=", 11);
            start_time_s[r] = v ? parse_iso8601(v, (uint64_t)(h.p + h.len - v)) : INT64_MIN;
        }
        ++r;
    }
    if (n_bad_qual) *n_bad_qual = bad;
    return PPQC_OK;
}

}  // extern "C"
