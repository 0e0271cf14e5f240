// `vgan haplocart -j -jf FILE`: the alignments of the GAM as JSON, one object per line -- what the reference's readGAM writes
// through vg's pb2json while it reads (src/readGAM.h:37-38, src/HaploCart.cpp:146-152,231).  vg and protobuf are not in this
// tree; the text follows protobuf's JSON mapping as vg configures it (MessageToJsonString, preserve_proto_field_names):
// fields in declaration order, proto names, absent fields left out, 64-bit integers as decimal strings, bytes as base64,
// doubles in the shortest form that reads back.  Messages and fields per vg.proto (Alignment, Path, Mapping, Position, Edit);
// Locus (18) and the Struct annotation (100) are skipped.  The byte-exact spacing of pb2json is UNVERIFIED here (no vg to run).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "vgan_gpu.h"

using namespace vgan;

namespace {

enum Kind { K_STRING, K_BYTES, K_INT32, K_INT64, K_BOOL, K_DOUBLE, K_MSG };
struct Msg;
struct Field {
    int num;
    const char *name;
    Kind kind;
    bool repeated;
    const Msg *sub;
};
struct Msg {
    const Field *f;
    int n;
};

extern const Msg M_ALIGNMENT, M_PATH, M_MAPPING, M_POSITION, M_EDIT;
const Field F_EDIT[] = {{1, "from_length", K_INT32, false, nullptr}, {2, "to_length", K_INT32, false, nullptr}, {3, "sequence", K_STRING, false, nullptr}};
const Field F_POSITION[] = {{1, "node_id", K_INT64, false, nullptr}, {2, "offset", K_INT64, false, nullptr}, {4, "is_reverse", K_BOOL, false, nullptr},
                            {5, "name", K_STRING, false, nullptr}};
const Field F_MAPPING[] = {{1, "position", K_MSG, false, &M_POSITION}, {2, "edit", K_MSG, true, &M_EDIT}, {5, "rank", K_INT64, false, nullptr}};
const Field F_PATH[] = {{1, "name", K_STRING, false, nullptr}, {2, "mapping", K_MSG, true, &M_MAPPING}, {3, "is_circular", K_BOOL, false, nullptr},
                        {4, "length", K_INT64, false, nullptr}};
const Field F_ALIGNMENT[] = {{1, "sequence", K_STRING, false, nullptr},
                             {2, "path", K_MSG, false, &M_PATH},
                             {3, "name", K_STRING, false, nullptr},
                             {4, "quality", K_BYTES, false, nullptr},
                             {5, "mapping_quality", K_INT32, false, nullptr},
                             {6, "score", K_INT32, false, nullptr},
                             {7, "query_position", K_INT32, false, nullptr},
                             {9, "sample_name", K_STRING, false, nullptr},
                             {10, "read_group", K_STRING, false, nullptr},
                             {11, "fragment_prev", K_MSG, false, &M_ALIGNMENT},
                             {12, "fragment_next", K_MSG, false, &M_ALIGNMENT},
                             {15, "is_secondary", K_BOOL, false, nullptr},
                             {16, "identity", K_DOUBLE, false, nullptr},
                             {17, "fragment", K_MSG, true, &M_PATH},
                             {19, "refpos", K_MSG, true, &M_POSITION},
                             {20, "read_paired", K_BOOL, false, nullptr},
                             {21, "read_mapped", K_BOOL, false, nullptr},
                             {22, "mate_unmapped", K_BOOL, false, nullptr},
                             {23, "read_on_reverse_strand", K_BOOL, false, nullptr},
                             {24, "mate_on_reverse_strand", K_BOOL, false, nullptr},
                             {25, "soft_clipped", K_BOOL, false, nullptr},
                             {26, "discordant_insert_size", K_BOOL, false, nullptr},
                             {27, "uniqueness", K_DOUBLE, false, nullptr},
                             {28, "correct", K_DOUBLE, false, nullptr},
                             {29, "secondary_score", K_INT32, true, nullptr},
                             {30, "fragment_score", K_DOUBLE, false, nullptr},
                             {31, "mate_mapped_to_disjoint_subgraph", K_BOOL, false, nullptr},
                             {32, "fragment_length_distribution", K_STRING, false, nullptr},
                             {35, "time_used", K_DOUBLE, false, nullptr},
                             {36, "to_correct", K_MSG, false, &M_POSITION},
                             {37, "correctly_mapped", K_BOOL, false, nullptr}};
const Msg M_EDIT = {F_EDIT, 3}, M_POSITION = {F_POSITION, 4}, M_MAPPING = {F_MAPPING, 3}, M_PATH = {F_PATH, 4},
          M_ALIGNMENT = {F_ALIGNMENT, (int)(sizeof F_ALIGNMENT / sizeof F_ALIGNMENT[0])};

struct Rd { // a bounded cursor over wire bytes
    const uint8_t *p, *e;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; p < e && shift < 70; shift += 7) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << (shift & 63);
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    Rd sub() {
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(e - p)) {
            ok = false;
            return Rd{p, p, false};
        }
        Rd c{p, p + n, true};
        p += n;
        return c;
    }
    void skip(int wt) {
        if (wt == 0) varint();
        else if (wt == 1 && e - p >= 8) p += 8;
        else if (wt == 2) sub();
        else if (wt == 5 && e - p >= 4) p += 4;
        else ok = false;
    }
};

void put_string(std::string &o, const uint8_t *p, size_t n) {
    o += '"';
    char buf[8];
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = p[i];
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20 || c == '<' || c == '>') {
                snprintf(buf, sizeof buf, "\\u%04x", c);
                o += buf;
            } else {
                o += (char)c;
            }
        }
    }
    o += '"';
}

void put_base64(std::string &o, const uint8_t *p, size_t n) {
    static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    o += '"';
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        const uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8 | p[i + 2];
        o += T[v >> 18];
        o += T[(v >> 12) & 63];
        o += T[(v >> 6) & 63];
        o += T[v & 63];
    }
    if (i + 1 == n) {
        const uint32_t v = (uint32_t)p[i] << 16;
        o += T[v >> 18];
        o += T[(v >> 12) & 63];
        o += "==";
    } else if (i + 2 == n) {
        const uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8;
        o += T[v >> 18];
        o += T[(v >> 12) & 63];
        o += T[(v >> 6) & 63];
        o += '=';
    }
    o += '"';
}

void put_double(std::string &o, double v) { // protobuf's SimpleDtoa: 15 significant digits if they read back, else 17
    if (std::isnan(v)) {
        o += "\"NaN\"";
        return;
    }
    if (std::isinf(v)) {
        o += v > 0 ? "\"Infinity\"" : "\"-Infinity\"";
        return;
    }
    char buf[40];
    snprintf(buf, sizeof buf, "%.15g", v);
    if (strtod(buf, nullptr) != v) snprintf(buf, sizeof buf, "%.17g", v);
    o += buf;
}

bool put_message(std::string &o, const Msg &m, Rd whole, int depth);

// one occurrence of a scalar field (wire type given)
bool put_scalar(std::string &o, const Field &f, Rd &c, int wt) {
    char buf[32];
    switch (f.kind) {
    case K_STRING:
    case K_BYTES: {
        if (wt != 2) return false;
        Rd s = c.sub();
        if (!c.ok) return false;
        if (f.kind == K_STRING) put_string(o, s.p, (size_t)(s.e - s.p));
        else put_base64(o, s.p, (size_t)(s.e - s.p));
        return true;
    }
    case K_INT32: {
        if (wt != 0) return false;
        snprintf(buf, sizeof buf, "%d", (int)(int32_t)c.varint());
        o += buf;
        return c.ok;
    }
    case K_INT64: {
        if (wt != 0) return false;
        snprintf(buf, sizeof buf, "\"%lld\"", (long long)(int64_t)c.varint());
        o += buf;
        return c.ok;
    }
    case K_BOOL: {
        if (wt != 0) return false;
        o += c.varint() ? "true" : "false";
        return c.ok;
    }
    case K_DOUBLE: {
        if (wt != 1 || c.e - c.p < 8) return false;
        double v;
        memcpy(&v, c.p, 8);
        c.p += 8;
        put_double(o, v);
        return true;
    }
    default: return false;
    }
}

bool put_message(std::string &o, const Msg &m, Rd whole, int depth) {
    if (depth > 8) return false; // (fragment_prev / fragment_next nest Alignments: a bound on hostile input)
    o += '{';
    bool first_field = true;
    for (int fi = 0; fi < m.n; ++fi) {
        const Field &f = m.f[fi];
        Rd c = whole;
        bool open = false; // this field's key (and for a repeated field its '[') is out
        size_t last_scalar = std::string::npos; // a singular field sent twice: the last one counts (proto3)
        while (c.ok && c.p < c.e) {
            const uint64_t key = c.varint();
            if (!c.ok) return false;
            const int num = (int)(key >> 3), wt = (int)(key & 7);
            if (num != f.num) {
                c.skip(wt);
                continue;
            }
            auto begin = [&] {
                if (!open) {
                    if (!first_field) o += ',';
                    first_field = false;
                    o += '"';
                    o += f.name;
                    o += "\":";
                    if (f.repeated) o += '[';
                    open = true;
                } else if (f.repeated) {
                    o += ',';
                }
            };
            if (f.kind == K_MSG) {
                if (wt != 2) return false;
                Rd s = c.sub();
                if (!c.ok) return false;
                begin();
                if (!f.repeated && last_scalar != std::string::npos) o.resize(last_scalar);
                last_scalar = o.size();
                if (!put_message(o, *f.sub, s, depth + 1)) return false;
            } else if (f.repeated && wt == 2 && f.kind != K_STRING && f.kind != K_BYTES) { // packed
                Rd s = c.sub();
                if (!c.ok) return false;
                while (s.p < s.e) {
                    begin();
                    if (!put_scalar(o, f, s, f.kind == K_DOUBLE ? 1 : 0)) return false;
                }
            } else {
                begin();
                if (!f.repeated && last_scalar != std::string::npos) o.resize(last_scalar);
                last_scalar = o.size();
                if (!put_scalar(o, f, c, wt)) return false;
            }
        }
        if (!c.ok) return false;
        if (open && f.repeated) o += ']';
    }
    o += '}';
    return true;
}

} // namespace

// The alignments of a GAM file (gzip / BGZF or plain; groups of `varint count` + length-prefixed items, a leading "GAM" type tag
// per group) as JSON lines into json_path.  *n_out (or NULL): the number of alignments written.
extern "C" int vgan_gam_dump_json(const char *gam_path, const char *json_path, int64_t *n_out) {
    if (!gam_path || !json_path) return fail(VGAN_EINVAL, "vgan_gam_dump_json: null argument");
    gzFile in = gzopen(gam_path, "rb");
    if (!in) return fail(VGAN_EIO, "cannot open %s", gam_path);
    std::vector<uint8_t> bytes;
    {
        std::vector<uint8_t> buf(1 << 20);
        for (;;) {
            const int n = gzread(in, buf.data(), (unsigned)buf.size());
            if (n < 0) {
                gzclose(in);
                return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
            }
            if (n == 0) break;
            bytes.insert(bytes.end(), buf.begin(), buf.begin() + n);
        }
        gzclose(in);
    }
    FILE *out = fopen(json_path, "w");
    if (!out) return fail(VGAN_EIO, "cannot write %s", json_path);
    int64_t n_aln = 0;
    Rd c{bytes.data(), bytes.data() + bytes.size(), true};
    std::string line;
    bool good = true;
    while (good && c.p < c.e) {
        const uint64_t count = c.varint();
        if (!c.ok) break;
        for (uint64_t i = 0; i < count && good; ++i) {
            Rd item = c.sub();
            if (!c.ok) {
                good = false;
                break;
            }
            if (i == 0 && item.e - item.p == 3 && memcmp(item.p, "GAM", 3) == 0) continue; // the group's type tag
            line.clear();
            if (!put_message(line, M_ALIGNMENT, item, 0)) {
                good = false;
                break;
            }
            line += '\n';
            if (fwrite(line.data(), 1, line.size(), out) != line.size()) {
                fclose(out);
                return fail(VGAN_EIO, "cannot write %s", json_path);
            }
            ++n_aln;
        }
    }
    if (fclose(out) != 0) return fail(VGAN_EIO, "cannot write %s", json_path);
    if (!good || !c.ok) return fail(VGAN_EIO, "GAM: malformed group or Alignment message");
    if (n_out) *n_out = n_aln;
    return VGAN_OK;
}
