// state_json.cc -- minimal, dependency-free JSON walk that materialises only what the descriptor DB needs.
#include "state_json.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cerebro_hip {
namespace {

struct Cur {
    const char *p, *end;
    std::string err;
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

bool skip_container(Cur &c, char open, char close)
{
    (void)open;
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

// {"rows": D, "cols": 1, "data": "..."} -> appends rows*cols doubles parsed with strtod (== the reference's std::stod)
bool parse_descriptor(Cur &c, std::vector<double> &vals, int64_t &rows, int64_t &cols)
{
    c.ws();
    if (c.p >= c.end || *c.p != '{') return c.fail("wholeImageDescriptor: expected object");
    c.p++;
    rows = cols = -1;
    std::string data;
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
            (key == "rows" ? rows : cols) = v;
        } else if (key == "data") {
            if (!parse_string(c, &data)) return false;
            have_data = true;
        } else if (!skip_value(c)) return false;
        c.ws();
        if (c.p < c.end && *c.p == ',') c.p++;
    }
    if (!have_data || rows <= 0 || cols <= 0) return c.fail("wholeImageDescriptor: missing rows/cols/data");
    const char *q = data.c_str(), *qe = q + data.size();
    int64_t n = 0;
    while (q < qe) {
        while (q < qe && (*q == ' ' || *q == '\n' || *q == ',' || *q == '\t' || *q == '\r')) q++;
        if (q >= qe) break;
        char *e = nullptr;
        errno = 0;
        const double v = std::strtod(q, &e);
        if (e == q) return c.fail("wholeImageDescriptor.data: bad number");
        vals.push_back(v);
        n++;
        q = e;
    }
    if (n != rows * cols) return c.fail("wholeImageDescriptor.data: element count != rows*cols");
    return true;
}

bool parse_node(Cur &c, StateDescriptors &out)
{
    c.ws();
    if (c.p >= c.end || *c.p != '{') return c.fail("DataNodes[]: expected object");
    c.p++;
    uint64_t stamp = 0;
    bool have_stamp = false, have_desc = false;
    int available = -1;
    std::vector<double> vals;
    int64_t rows = 0, cols = 0;
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
            if (!parse_descriptor(c, vals, rows, cols)) return false;
            have_desc = true;
        } else if (!skip_value(c)) return false;
        c.ws();
        if (c.p < c.end && *c.p == ',') c.p++;
    }
    out.n_nodes++;
    if (have_desc && available != 0) {   // DataManager::loadStateFromDisk only restores it when the flag is set
        if (!have_stamp) return c.fail("node with descriptor but no stampNSec");
        const int D = (int)(rows * cols);
        if (out.D == 0) out.D = D;
        if (D != out.D) return c.fail("descriptor size differs between nodes");
        out.stampNSec.push_back(stamp);
        out.desc.insert(out.desc.end(), vals.begin(), vals.end());
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
                    if (!parse_node(c, out)) break;
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
    out.error = c.err;
    return c.err.empty();
}

bool load_state_json(const std::string &path, StateDescriptors &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { out = StateDescriptors(); out.error = "cannot open " + path; return false; }
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    return parse_state_json(text, out);
}

}  // namespace cerebro_hip
