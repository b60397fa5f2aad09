// state_json.cc -- minimal, dependency-free JSON walk that materialises only what the descriptor DB needs.
#include "state_json.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace cerebro_hip {
namespace {

constexpr int kMaxDepth = 64;          // nesting cap of skipped values (DataManager's state.json nests 4 deep)
constexpr int64_t kMaxDescriptor = 10240;  // == the largest D chip_create accepts (4 queries x D floats in 160 KiB of LDS)

struct Cur {
    const char *p, *end;
    std::string err;
    int depth = 0;
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool lit(const char *s) { size_t n = std::strlen(s); if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) { p += n; return true; } return false; }
};

// Parses a JSON string; when `out` is null the content is skipped.  Escapes \n \t \" \\ \/ \b \f \r \uXXXX (BMP, as UTF-8).
bool parse_string(Cur &c, std::string *out)
{
    if (c.p >= c.end || *c.p != '"') return c.fail("expected string");
    c.p++;
    while (c.p < c.end && *c.p != '"') {
        char ch = *c.p++;
        if (ch == '\\') {
            if (c.p >= c.end) return c.fail("bad escape");
            char e = *c.p++;
            switch (e) {
                case 'n': ch = '\n'; break;  case 't': ch = '\t'; break;  case 'r': ch = '\r'; break;
                case 'b': ch = '\b'; break;  case 'f': ch = '\f'; break;
                case '"': case '\\': case '/': ch = e; break;
                case 'u': {
                    if (c.end - c.p < 4) return c.fail("bad \\u escape");
                    unsigned v = 0;
                    for (int i = 0; i < 4; i++) {
                        char h = c.p[i];
                        v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : 99);
                    }
                    c.p += 4;
                    if (out) {
                        if (v < 0x80) out->push_back((char)v);
                        else if (v < 0x800) { out->push_back((char)(0xC0 | (v >> 6))); out->push_back((char)(0x80 | (v & 0x3F))); }
                        else { out->push_back((char)(0xE0 | (v >> 12))); out->push_back((char)(0x80 | ((v >> 6) & 0x3F))); out->push_back((char)(0x80 | (v & 0x3F))); }
                    }
                    continue;
                }
                default: return c.fail("bad escape");
            }
        }
        if (out) out->push_back(ch);
    }
    if (c.p >= c.end) return c.fail("unterminated string");
    c.p++;
    return true;
}

bool skip_value(Cur &c);

struct DepthGuard {
    Cur &c;
    explicit DepthGuard(Cur &cc) : c(cc) { c.depth++; }
    ~DepthGuard() { c.depth--; }
};

bool skip_container(Cur &c, char open, char close)
{
    (void)open;
    DepthGuard g(c);
    if (c.depth > kMaxDepth) return c.fail("nesting too deep");
    c.p++;
    c.ws();
    if (c.p < c.end && *c.p == close) { c.p++; return true; }
    for (;;) {
        c.ws();
        if (close == '}') {
            if (!parse_string(c, nullptr)) return false;
            c.ws();
            if (c.p >= c.end || *c.p != ':') return c.fail("expected ':'");
            c.p++;
        }
        if (!skip_value(c)) return false;
        c.ws();
        if (c.p < c.end && *c.p == ',') { c.p++; continue; }
        if (c.p < c.end && *c.p == close) { c.p++; return true; }
        return c.fail("expected ',' or close");
    }
}

bool skip_value(Cur &c)
{
    c.ws();
    if (c.p >= c.end) return c.fail("unexpected end");
    switch (*c.p) {
        case '{': return skip_container(c, '{', '}');
        case '[': return skip_container(c, '[', ']');
        case '"': return parse_string(c, nullptr);
        case 't': return c.lit("true") || c.fail("bad literal");
        case 'f': return c.lit("false") || c.fail("bad literal");
        case 'n': return c.lit("null") || c.fail("bad literal");
        default: {
            char *e = nullptr;
            (void)std::strtod(c.p, &e);
            if (e == c.p) return c.fail("bad number");
            c.p = e;
            return true;
        }
    }
}

// The raw (still JSON-escaped) text of one descriptor's "data" string.  The structural walk only locates it; the numbers
// are converted afterwards, many descriptors at a time (a checkpoint is ~80 KB of decimal text per 4096-D descriptor and
// number conversion is >95 % of the load time).
struct DataSpan {
    const char *b = nullptr, *e = nullptr;  // between the quotes
    bool plain = true;                      // only \n \t \r escapes inside (what nlohmann::json emits for Eigen's text)
    int64_t rows = -1, cols = -1;
    uint64_t stamp = 0;
};

// Finds the closing quote of the string starting at c.p (which must be '"') without materialising it.
bool scan_string(Cur &c, DataSpan &sp)
{
    if (c.p >= c.end || *c.p != '"') return c.fail("expected string");
    const char *q = c.p + 1;
    sp.b = q;
    sp.plain = true;
    for (;;) {
        const char *m = static_cast<const char *>(std::memchr(q, '"', (size_t)(c.end - q)));
        if (!m) return c.fail("unterminated string");
        // a quote preceded by an odd number of backslashes is escaped
        const char *bs = m;
        while (bs > q && bs[-1] == '\\') bs--;
        if (((m - bs) & 1) == 0) { sp.e = m; break; }
        q = m + 1;
    }
    for (const char *t = sp.b; t < sp.e; t++)
        if (*t == '\\') {
            if (t + 1 < sp.e && (t[1] == 'n' || t[1] == 't' || t[1] == 'r')) t++;
            else { sp.plain = false; break; }
        }
    c.p = sp.e + 1;
    return true;
}

// Text -> doubles with the semantics of the reference's std::stod (== strtod, "C" locale): std::from_chars is correctly
// rounded like glibc's strtod and several times faster; anything it rejects (leading '+', hex floats ...) is retried with
// strtod itself.  `escaped`: separators may also be the two-character sequences \n \t \r.
bool convert_values(const char *q, const char *qe, bool escaped, double *dst, int64_t want, const char **err)
{
    int64_t n = 0;
    while (q < qe) {
        const char ch = *q;
        if (ch == ' ' || ch == '\n' || ch == ',' || ch == '\t' || ch == '\r') { q++; continue; }
        if (escaped && ch == '\\' && q + 1 < qe && (q[1] == 'n' || q[1] == 't' || q[1] == 'r')) { q += 2; continue; }
        double v;
        auto r = std::from_chars(q, qe, v);
        const char *next = r.ptr;
        // not a clean token end (e.g. the "0" of a hex float "0x1p-2"): let strtod decide
        const bool clean = next >= qe || *next == ' ' || *next == '\n' || *next == ',' || *next == '\t' || *next == '\r' || *next == '\\';
        if (r.ec != std::errc() || !clean) {
            char *e = nullptr;
            v = std::strtod(q, &e);   // the text is NUL-terminated past qe and strtod stops at the closing quote at the latest
            if (e == q || e > qe) { *err = "wholeImageDescriptor.data: bad number"; return false; }
            next = e;
        }
        if (n >= want) { *err = "wholeImageDescriptor.data: element count != rows*cols"; return false; }
        dst[n++] = v;
        q = next;
    }
    if (n != want) { *err = "wholeImageDescriptor.data: element count != rows*cols"; return false; }
    return true;
}

// {"rows": D, "cols": 1, "data": "..."} -> locates the data text (converted later by convert_values)
bool parse_descriptor(Cur &c, DataSpan &sp)
{
    c.ws();
    if (c.p >= c.end || *c.p != '{') return c.fail("wholeImageDescriptor: expected object");
    c.p++;
    sp.rows = sp.cols = -1;
    bool have_data = false;
    for (;;) {
        c.ws();
        if (c.p < c.end && *c.p == '}') { c.p++; break; }
        std::string key;
        if (!parse_string(c, &key)) return false;
        c.ws();
        if (c.p >= c.end || *c.p != ':') return c.fail("expected ':'");
        c.p++;
        c.ws();
        if (key == "rows" || key == "cols") {
            char *e = nullptr;
            long long v = std::strtoll(c.p, &e, 10);
            if (e == c.p) return c.fail("bad rows/cols");
            c.p = e;
            (key == "rows" ? sp.rows : sp.cols) = v;
        } else if (key == "data") {
            if (!scan_string(c, sp)) return false;
            have_data = true;
        } else if (!skip_value(c)) return false;
        c.ws();
        if (c.p < c.end && *c.p == ',') c.p++;
    }
    if (!have_data || sp.rows <= 0 || sp.cols <= 0) return c.fail("wholeImageDescriptor: missing rows/cols/data");
    return true;
}

bool parse_node(Cur &c, StateDescriptors &out, std::vector<DataSpan> &spans)
{
    c.ws();
    if (c.p >= c.end || *c.p != '{') return c.fail("DataNodes[]: expected object");
    c.p++;
    uint64_t stamp = 0;
    bool have_stamp = false, have_desc = false;
    int available = -1;
    DataSpan sp;
    for (;;) {
        c.ws();
        if (c.p < c.end && *c.p == '}') { c.p++; break; }
        std::string key;
        if (!parse_string(c, &key)) return false;
        c.ws();
        if (c.p >= c.end || *c.p != ':') return c.fail("expected ':'");
        c.p++;
        c.ws();
        if (key == "stampNSec") {
            char *e = nullptr;
            stamp = std::strtoull(c.p, &e, 10);
            if (e == c.p) return c.fail("bad stampNSec");
            c.p = e;
            have_stamp = true;
        } else if (key == "isWholeImageDescriptorAvailable") {
            if (c.lit("true")) available = 1; else if (c.lit("false")) available = 0; else return c.fail("bad bool");
        } else if (key == "wholeImageDescriptor") {
            if (!parse_descriptor(c, sp)) return false;
            have_desc = true;
        } else if (!skip_value(c)) return false;
        c.ws();
        if (c.p < c.end && *c.p == ',') c.p++;
    }
    out.n_nodes++;
    if (have_stamp) out.all_stampNSec.push_back(stamp);
    if (have_desc && available != 0) {   // DataManager::loadStateFromDisk only restores it when the flag is set
        if (!have_stamp) return c.fail("node with descriptor but no stampNSec");
        if (sp.rows > kMaxDescriptor || sp.cols > kMaxDescriptor || sp.rows * sp.cols > kMaxDescriptor)
            return c.fail("wholeImageDescriptor: rows*cols out of range");
        const int D = (int)(sp.rows * sp.cols);
        if (out.D == 0) out.D = D;
        if (D != out.D) return c.fail("descriptor size differs between nodes");
        out.stampNSec.push_back(stamp);
        sp.stamp = stamp;
        spans.push_back(sp);
    }
    return true;
}

}  // namespace

bool parse_state_json(const std::string &text, StateDescriptors &out)
{
    out = StateDescriptors();
    Cur c{text.data(), text.data() + text.size(), {}};
    c.ws();
    if (c.p >= c.end || *c.p != '{') { out.error = "top level: expected object"; return false; }
    c.p++;
    bool seen = false;
    std::vector<DataSpan> spans;
    for (;;) {
        c.ws();
        if (c.p < c.end && *c.p == '}') break;
        std::string key;
        if (!parse_string(c, &key)) break;
        c.ws();
        if (c.p >= c.end || *c.p != ':') { c.fail("expected ':'"); break; }
        c.p++;
        c.ws();
        if (key == "DataNodes") {
            seen = true;
            if (c.p >= c.end || *c.p != '[') { c.fail("DataNodes: expected array"); break; }
            c.p++;
            c.ws();
            if (c.p < c.end && *c.p == ']') c.p++;
            else
                for (;;) {
                    if (!parse_node(c, out, spans)) break;
                    c.ws();
                    if (c.p < c.end && *c.p == ',') { c.p++; continue; }
                    if (c.p < c.end && *c.p == ']') { c.p++; break; }
                    c.fail("DataNodes: expected ',' or ']'");
                    break;
                }
            if (!c.err.empty()) break;
        } else if (!skip_value(c)) break;
        c.ws();
        if (c.p < c.end && *c.p == ',') c.p++;
    }
    if (c.err.empty() && !seen) c.err = "no DataNodes array";
    if (c.err.empty() && !spans.empty()) {
        // The reference rebuilds wholeImageComputedList by iterating the time-sorted data_map (Cerebro.cpp:145-149), so row
        // order is time order whatever the order of the array in the file (stable: equal stamps keep file order).
        std::stable_sort(spans.begin(), spans.end(), [](const DataSpan &a, const DataSpan &b) { return a.stamp < b.stamp; });
        for (size_t i = 0; i < spans.size(); i++) out.stampNSec[i] = spans[i].stamp;
        // second pass: number conversion, descriptors split evenly over the threads
        const size_t n = spans.size(), D = (size_t)out.D;
        out.desc.resize(n * D);
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 32) nt = 32;
        if (nt > n) nt = (unsigned)n;
        if (nt < 1) nt = 1;
        std::atomic<const char *> first_err{nullptr};
        auto work = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi && !first_err.load(std::memory_order_relaxed); i++) {
                const DataSpan &sp = spans[i];
                const char *err = nullptr;
                bool ok;
                if (sp.plain) ok = convert_values(sp.b, sp.e, true, &out.desc[i * D], (int64_t)D, &err);
                else {  // unusual escapes (\uXXXX ...): decode the string first
                    Cur d{sp.b - 1, c.end, {}};
                    std::string data;
                    ok = parse_string(d, &data);
                    if (!ok) err = "wholeImageDescriptor.data: bad string";
                    else ok = convert_values(data.c_str(), data.c_str() + data.size(), false, &out.desc[i * D], (int64_t)D, &err);
                }
                if (!ok) { const char *expect = nullptr; first_err.compare_exchange_strong(expect, err); }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        work(0, n / nt);
        for (std::thread &th : pool) th.join();
        if (const char *e = first_err.load()) c.err = e;
    }
    if (!c.err.empty()) { out.stampNSec.clear(); out.desc.clear(); }
    out.error = c.err;
    return c.err.empty();
}

bool load_state_json(const std::string &path, StateDescriptors &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { out = StateDescriptors(); out.error = "cannot open " + path; return false; }
    std::string text;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        const long sz = std::ftell(f);
        if (sz > 0) text.reserve((size_t)sz);
        std::rewind(f);
    }
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    return parse_state_json(text, out);
}

}  // namespace cerebro_hip
