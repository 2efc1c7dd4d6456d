// XYZ text parsing on the host cores (the step before the GPU path; no device code in this file).
// load_txt_points of the reference (utils/data_io.py:8-27) reads a scan line by line in Python: strip,
// skip blank lines and lines that start with '#', '//' or ';', commas count as blanks, the first three
// fields must parse as floats, any other line is skipped.  Here the text is cut at line ends into one
// piece per thread and every "plain" line - bytes from the set  0-9 + - . e E , blank tab VT FF  only - is
// parsed natively (std::from_chars: correctly rounded, like Python's float()).  A line that holds any other
// byte (letters, underscores, non-ASCII, NUL ...) or a field that over/underflows is not decided here: it
// is returned as a flagged row carrying its byte range, and the caller applies the literal Python rule
// to it.  The result is therefore the reference's for every input, fast for the usual ones.
#include <charconv>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "pcfit_internal.h"

namespace pcfit {

static inline bool is_eol_byte(char c) { return c == '\n' || c == '\r'; }

// where one piece of the text puts its rows: at most one per line of the piece
struct RowSink {
    double *rows;
    uint8_t *deferred;
    int64_t n = 0;
    void put(double a, double b, double c, uint8_t d)
    {
        rows[3 * n] = a;
        rows[3 * n + 1] = b;
        rows[3 * n + 2] = c;
        deferred[n] = d;
        n++;
    }
};

static int64_t count_eol(const char *text, size_t begin, size_t end)
{
    int64_t k = 0;
    for (const char c : {'\n', '\r'}) {
        const char *p = text + begin, *e = text + end;
        while (p < e && (p = (const char *)memchr(p, c, (size_t)(e - p))) != nullptr) {
            k++;
            p++;
        }
    }
    return k;
}

// cut points for T pieces: the first line start at or after each even split
static std::vector<size_t> cut_points(const char *text, size_t n_bytes, int T)
{
    std::vector<size_t> cut(T + 1, n_bytes);
    cut[0] = 0;
    for (int t = 1; t < T; t++) {
        size_t p = std::max(cut[t - 1], n_bytes / T * t);
        while (p < n_bytes && !is_eol_byte(text[p])) p++;
        cut[t] = std::min(n_bytes, p + (p < n_bytes ? 1 : 0));
    }
    return cut;
}

static int thread_count(int32_t n_threads, size_t n_bytes)
{
    const int T = n_threads > 0 ? n_threads
                                : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    return (int)std::min<size_t>((size_t)T, n_bytes / (1u << 16) + 1);
}

template <class F> static void run_pieces(int T, F &&body)
{
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++)
        pool.emplace_back([&body, t] { body(t); });
    body(0);
    for (auto &th : pool)
        th.join();
}

static inline bool is_blank(unsigned char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }
static inline bool is_eol(unsigned char c) { return c == '\n' || c == '\r'; }
static inline bool is_plain(unsigned char c)
{
    return (c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E' || c == ',' ||
           is_blank(c);
}

// One field (no blanks inside) by Python's float() grammar restricted to the plain bytes.
// 1 = value stored, 0 = not a float (the line is skipped), -1 = leave the line to the caller.
static int parse_field(const char *b, const char *e, double &out)
{
    if (b < e && *b == '+') {
        b++;
        if (b < e && (*b == '+' || *b == '-'))
            return 0;
    }
    if (b == e)
        return 0;
    // Short decimals take the exact route: a significand below 2^53 and a power of ten up to 10^22 are
    // both doubles, so one multiplication or division rounds once - the correctly rounded result.
    {
        const char *p = b;
        const bool neg = *p == '-';
        if (neg) p++;
        uint64_t m = 0;
        int digits = 0, frac = 0;
        bool seen_dot = false, ok = true;
        for (; p < e; p++) {
            const char c = *p;
            if (c >= '0' && c <= '9') {
                if (digits < 19) {
                    m = m * 10 + (uint64_t)(c - '0');
                    digits += (m != 0 || digits != 0) ? 1 : 0;      // leading zeros do not count
                    frac += seen_dot ? 1 : 0;
                } else {
                    ok = false;                                     // too long for this route
                }
            } else if (c == '.' && !seen_dot) {
                seen_dot = true;
            } else {
                break;
            }
        }
        const int consumed = (int)(p - b) - (neg ? 1 : 0) - (seen_dot ? 1 : 0);
        if (ok && p == e && consumed > 0 && m < (1ull << 53) && frac <= 22) {
            static const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
            const double v = (double)m / P10[frac];
            out = neg ? -v : v;
            return 1;
        }
    }
    const auto r = std::from_chars(b, e, out, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range)
        return -1;                          // float() gives inf or 0.0 there: the caller's rule decides
    if (r.ec != std::errc() || r.ptr != e)
        return 0;
    return 1;
}

static void parse_piece(const char *text, size_t begin, size_t end, RowSink &rows)
{
    size_t p = begin;
    while (p < end) {
        size_t q = p;
        bool plain = true;
        while (q < end && !is_eol((unsigned char)text[q])) {
            plain &= is_plain((unsigned char)text[q]);
            q++;
        }
        // [p, q) is one line without its terminator
        size_t a = p, z = q;
        while (a < z && is_blank((unsigned char)text[a])) a++;
        while (z > a && is_blank((unsigned char)text[z - 1])) z--;
        const size_t line_begin = p, line_end = q;
        p = q + 1;
        if (a == z)
            continue;
        if (!plain) {
            // a comment is recognised by its first bytes, whatever follows; other lines go to the caller.
            // Bytes that Python strips but this file does not call blank (0x1c-0x1f, non-ASCII spaces)
            // may precede the marker: such a line is deferred too, and the caller's rule skips it.
            const char c0 = text[a];
            if (c0 == '#' || c0 == ';' || (c0 == '/' && a + 1 < z && text[a + 1] == '/'))
                continue;
            rows.put(__builtin_nan(""), (double)line_begin, (double)line_end, 1);
            continue;
        }
        double v[3];
        int got = 0, verdict = 1;
        size_t f = a;
        while (f < z && got < 3) {
            while (f < z && (is_blank((unsigned char)text[f]) || text[f] == ',')) f++;
            size_t g = f;
            while (g < z && !is_blank((unsigned char)text[g]) && text[g] != ',') g++;
            if (g == f)
                break;
            verdict = parse_field(text + f, text + g, v[got]);
            if (verdict != 1)
                break;
            got++;
            f = g;
        }
        if (verdict == -1)
            rows.put(__builtin_nan(""), (double)line_begin, (double)line_end, 1);
        else if (verdict == 1 && got == 3)
            rows.put(v[0], v[1], v[2], 0);
    }
}

}  // namespace pcfit

using namespace pcfit;

PCFIT_API int pcfit_count_text_lines(const char *text, size_t n_bytes, int32_t n_threads, int64_t *n_lines)
{
    PCFIT_REQUIRE(n_lines && (text || n_bytes == 0), "count_text_lines: null pointer");
    const int T = thread_count(n_threads, n_bytes);
    std::vector<int64_t> k(T, 0);
    run_pieces(T, [&](int t) { k[t] = count_eol(text, n_bytes / T * t, t + 1 == T ? n_bytes : n_bytes / T * (t + 1)); });
    int64_t total = 1;                                  // the last line may have no terminator
    for (int64_t v : k)
        total += v;
    *n_lines = total;
    return PCFIT_OK;
}

PCFIT_API int pcfit_parse_xyz_text(const char *text, size_t n_bytes, int32_t n_threads, double *rows,
                                   uint8_t *deferred, int64_t capacity, int64_t *n_rows)
{
    PCFIT_REQUIRE(n_rows && (text || n_bytes == 0), "parse_xyz_text: null pointer");
    PCFIT_REQUIRE(capacity >= 0 && (capacity == 0 || (rows && deferred)), "parse_xyz_text: null pointer");
    const int T = thread_count(n_threads, n_bytes);
    const std::vector<size_t> cut = cut_points(text, n_bytes, T);
    // every piece writes at its own base: the lines before it bound the rows before it
    std::vector<int64_t> lines(T, 0), base(T + 1, 0);
    // (a piece ends behind a line end, except the one that reaches the end of the text)
    run_pieces(T, [&](int t) {
        lines[t] = count_eol(text, cut[t], cut[t + 1]) + ((cut[t + 1] == n_bytes && cut[t] < n_bytes) ? 1 : 0);
    });
    for (int t = 0; t < T; t++)
        base[t + 1] = base[t] + lines[t];
    if (base[T] > capacity) {
        *n_rows = base[T];
        set_error("parse_xyz_text: room for %lld rows needed (one per line), %lld offered", (long long)base[T],
                  (long long)capacity);
        return PCFIT_ERR_INVALID;
    }
    std::vector<RowSink> sink(T);
    run_pieces(T, [&](int t) {
        sink[t].rows = rows + 3 * base[t];
        sink[t].deferred = deferred + base[t];
        parse_piece(text, cut[t], cut[t + 1], sink[t]);
    });
    // close the gaps the skipped lines left between the pieces
    int64_t at = sink[0].n;
    for (int t = 1; t < T; t++) {
        if (at != base[t] && sink[t].n > 0) {
            memmove(rows + 3 * at, rows + 3 * base[t], (size_t)sink[t].n * 24);
            memmove(deferred + at, deferred + base[t], (size_t)sink[t].n);
        }
        at += sink[t].n;
    }
    *n_rows = at;
    return PCFIT_OK;
}

// ---- the step after the path: CSV rows with '%.6f' floats ------------------------------------
// pandas' to_csv(float_format="%.6f") formats every float through Python's % operator (save_results,
// utils/data_io.py:37,46): correctly rounded fixed notation, the same digits printf gives.  cols holds n_cols
// columns of n_rows doubles each, one after the other; is_int[c] marks columns printed as integers; NaN
// becomes an empty field (pandas' na_rep).  Rows are formatted by several host threads.
PCFIT_API int pcfit_format_csv_rows(const double *cols, int64_t n_rows, int32_t n_cols, const uint8_t *is_int,
                                    int32_t n_threads, char *out, size_t capacity, size_t *length)
{
    PCFIT_REQUIRE(length && n_rows >= 0 && n_cols >= 1 && (n_rows == 0 || (cols && is_int)),
                  "format_csv_rows: bad arguments");
    const int T = (int)std::min<int64_t>(std::max<int64_t>(n_rows / 2048, 1),
                                         n_threads > 0 ? n_threads
                                                       : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    std::vector<std::string> piece(T);
    run_pieces(T, [&](int t) {
        const int64_t r0 = n_rows * t / T, r1 = n_rows * (t + 1) / T;
        std::string &o = piece[t];
        o.reserve((size_t)(r1 - r0) * (size_t)n_cols * 12);
        char tmp[400];
        for (int64_t r = r0; r < r1; r++) {
            for (int32_t c = 0; c < n_cols; c++) {
                const double v = cols[(size_t)c * (size_t)n_rows + (size_t)r];
                if (c)
                    o.push_back(',');
                if (v != v)
                    continue;
                const int k = is_int[c] ? snprintf(tmp, sizeof tmp, "%lld", (long long)v)
                                        : snprintf(tmp, sizeof tmp, "%.6f", v);
                o.append(tmp, (size_t)k);
            }
            o.push_back('\n');
        }
    });
    size_t total = 0;
    for (const std::string &o : piece)
        total += o.size();
    *length = total;
    if (total > capacity || (total && !out)) {
        set_error("format_csv_rows: %zu bytes needed, %zu offered", total, capacity);
        return PCFIT_ERR_INVALID;
    }
    size_t at = 0;
    for (const std::string &o : piece) {
        memcpy(out + at, o.data(), o.size());
        at += o.size();
    }
    return PCFIT_OK;
}
