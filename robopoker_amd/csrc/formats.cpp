// formats.cpp — the artifacts the reference streams to PostgreSQL, as files in the same byte format (SURVEY §8f row f3;
// include/rp_mi355x.h "artifact files").  Host code: this is I/O, not a kernel.
//
// Reference: daybook::Streamable::stream (crates/daybook/src/traits/streamable.rs:36-46) writes each table through
// tokio_postgres::binary_copy::BinaryCopyInWriter into `COPY <table> (<columns>) FROM STDIN BINARY`; the row shapes
// are crates/daybook/src/traits/row.rs:21-57.  tokio-postgres (Cargo.lock:3136-3137, version 0.7.16) is not under
// /root/reference; the stream it produces is PostgreSQL's documented binary COPY format, restated here:
//     signature  "PGCOPY\n\377\r\n\0"   (11 bytes)
//     int32      flags = 0,  int32 header-extension length = 0
//     per tuple  int16 field count, then per field int32 byte length + the value, big-endian
//                (int2 / int4 / int8 two's complement, float4 IEEE-754 bits)
//     trailer    int16 -1
// so a file written here loads with  COPY <table> (<columns>) FROM '<file>' (FORMAT binary)  — byte for byte what
// the reference's writer sends.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rp_internal.h"

namespace {

const unsigned char SIGNATURE[11] = {'P', 'G', 'C', 'O', 'P', 'Y', '\n', 0xff, '\r', '\n', 0};

size_t width_of(char t) {
    switch (t) {
        case 'h': return 2;  // int2
        case 'i': return 4;  // int4
        case 'q': return 8;  // int8
        case 'f': return 4;  // float4
        default: return 0;
    }
}
void put_be(unsigned char* dst, const void* src, size_t w) {  // host is little-endian (x86-64)
    const unsigned char* s = static_cast<const unsigned char*>(src);
    for (size_t i = 0; i < w; ++i) dst[i] = s[w - 1 - i];
}

struct Writer {
    FILE* f = nullptr;
    std::vector<unsigned char> buf;
    bool ok = true;
    bool open(const char* path) {
        f = std::fopen(path, "wb");
        if (!f) return false;
        buf.reserve(1 << 20);
        buf.insert(buf.end(), SIGNATURE, SIGNATURE + 11);
        const unsigned char zeros[8] = {0};
        buf.insert(buf.end(), zeros, zeros + 8);
        return true;
    }
    void flush() {
        if (!buf.empty() && std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) ok = false;
        buf.clear();
    }
    void begin_row(uint16_t fields) {
        if (buf.size() > (1 << 20) - 256) flush();
        const unsigned char n[2] = {(unsigned char)(fields >> 8), (unsigned char)(fields & 0xff)};
        buf.insert(buf.end(), n, n + 2);
    }
    void field(const void* v, size_t w) {
        const unsigned char len[4] = {0, 0, 0, (unsigned char)w};
        buf.insert(buf.end(), len, len + 4);
        unsigned char be[8];
        put_be(be, v, w);
        buf.insert(buf.end(), be, be + w);
    }
    bool close() {
        const unsigned char trailer[2] = {0xff, 0xff};
        buf.insert(buf.end(), trailer, trailer + 2);
        flush();
        if (std::fclose(f) != 0) ok = false;
        f = nullptr;
        return ok;
    }
};

}  // namespace

extern "C" {

int rp_pgcopy_write(const char* path, const char* types, uint64_t n_rows, const void* const* columns) {
    if (!path || !types || (!columns && n_rows)) return rp::fail(RP_ERR_INVALID, "null argument");
    const size_t nc = std::strlen(types);
    if (nc == 0 || nc > 32) return rp::fail(RP_ERR_INVALID, "1..32 columns");
    for (size_t c = 0; c < nc; ++c) {
        if (!width_of(types[c])) return rp::fail(RP_ERR_INVALID, "column type '%c': h = int2, i = int4, q = int8, f = float4", types[c]);
        if (n_rows && !columns[c]) return rp::fail(RP_ERR_INVALID, "column %zu is null", c);
    }
    Writer w;
    if (!w.open(path)) return rp::fail(RP_ERR_INVALID, "cannot create %s", path);
    for (uint64_t r = 0; r < n_rows; ++r) {
        w.begin_row((uint16_t)nc);
        for (size_t c = 0; c < nc; ++c) {
            const size_t wd = width_of(types[c]);
            w.field(static_cast<const unsigned char*>(columns[c]) + r * wd, wd);
        }
    }
    if (!w.close()) return rp::fail(RP_ERR_INVALID, "short write to %s", path);
    return RP_OK;
}

int rp_pgcopy_read(const char* path, const char* types, uint64_t cap, void* const* columns, uint64_t* n_rows) {
    if (!path || !types || !n_rows) return rp::fail(RP_ERR_INVALID, "null argument");
    const size_t nc = std::strlen(types);
    for (size_t c = 0; c < nc; ++c)
        if (!width_of(types[c])) return rp::fail(RP_ERR_INVALID, "column type '%c'", types[c]);
    FILE* f = std::fopen(path, "rb");
    if (!f) return rp::fail(RP_ERR_INVALID, "cannot open %s", path);
    unsigned char head[19];
    int rc = RP_OK;
    uint64_t n = 0;
    if (std::fread(head, 1, 19, f) != 19 || std::memcmp(head, SIGNATURE, 11) != 0) {
        rc = rp::fail(RP_ERR_INVALID, "%s is not a binary COPY stream", path);
    } else {
        const uint32_t ext = (uint32_t)head[15] << 24 | (uint32_t)head[16] << 16 | (uint32_t)head[17] << 8 | head[18];
        if (ext && std::fseek(f, (long)ext, SEEK_CUR) != 0) rc = rp::fail(RP_ERR_INVALID, "truncated header extension");
    }
    while (rc == RP_OK) {
        unsigned char cnt[2];
        if (std::fread(cnt, 1, 2, f) != 2) {
            rc = rp::fail(RP_ERR_INVALID, "%s: missing trailer", path);
            break;
        }
        const int16_t fields = (int16_t)((uint16_t)cnt[0] << 8 | cnt[1]);
        if (fields == -1) break;
        if ((size_t)fields != nc) {
            rc = rp::fail(RP_ERR_INVALID, "%s: tuple %llu has %d fields, expected %zu", path, (unsigned long long)n, fields, nc);
            break;
        }
        for (size_t c = 0; c < nc && rc == RP_OK; ++c) {
            unsigned char len[4], be[8];
            const size_t wd = width_of(types[c]);
            if (std::fread(len, 1, 4, f) != 4 || len[0] || len[1] || len[2] || len[3] != wd || std::fread(be, 1, wd, f) != wd) {
                rc = rp::fail(RP_ERR_INVALID, "%s: tuple %llu field %zu is not a %zu-byte value", path, (unsigned long long)n, c, wd);
                break;
            }
            if (columns && n < cap && columns[c]) put_be(static_cast<unsigned char*>(columns[c]) + n * wd, be, wd);
        }
        if (rc == RP_OK) ++n;
    }
    std::fclose(f);
    *n_rows = n;
    return rc;
}

// Streamable for Lookup (lloyd/src/lookup.rs:141-147): rows (i64::from(iso), i16::from(abs)), abs = street << 8 | index
// (kicker/src/abstraction.rs:65-71,117-121), in the table's (BTreeMap = iterator) order.
int rp_artifact_write_lookup(const char* path, int street, uint64_t n, const int64_t* obs, const uint8_t* abs_index) {
    if (!path || ((!obs || !abs_index) && n)) return rp::fail(RP_ERR_INVALID, "null argument");
    if (street < 0 || street > 3) return rp::fail(RP_ERR_INVALID, "street %d", street);
    Writer w;
    if (!w.open(path)) return rp::fail(RP_ERR_INVALID, "cannot create %s", path);
    for (uint64_t r = 0; r < n; ++r) {
        const int16_t a = (int16_t)((uint16_t)street << 8 | abs_index[r]);
        w.begin_row(2);
        w.field(&obs[r], 8);
        w.field(&a, 2);
    }
    if (!w.close()) return rp::fail(RP_ERR_INVALID, "short write to %s", path);
    return RP_OK;
}

// Streamable for Metric (lloyd/src/metric.rs:219-226 over distances.rs:69-84): rows (i32::from(Pair), dx) for the
// triangular index t ascending; Pair = street << 30 | t (pair.rs:7-16,61-65).
int rp_artifact_write_metric(const char* path, int street, uint32_t K, const float* tri) {
    if (!path || !tri) return rp::fail(RP_ERR_INVALID, "null argument");
    if (street < 0 || street > 3 || K < 2) return rp::fail(RP_ERR_INVALID, "street %d, K %u", street, K);
    Writer w;
    if (!w.open(path)) return rp::fail(RP_ERR_INVALID, "cannot create %s", path);
    const uint32_t T = K * (K - 1) / 2;
    for (uint32_t t = 0; t < T; ++t) {
        const int32_t pair = (int32_t)((uint32_t)street << 30 | t);
        w.begin_row(2);
        w.field(&pair, 4);
        w.field(&tri[t], 4);
    }
    if (!w.close()) return rp::fail(RP_ERR_INVALID, "short write to %s", path);
    return RP_OK;
}

// Streamable for Future (lloyd/src/future.rs:99-111): for each abstraction ascending, its centroid histogram's
// distribution() (bins.rs:113-117: support ascending, density = count as f32 / weight as f32 (bins.rs:58-60), then a
// STABLE sort by density descending) as rows (prev, next, dx); next lives on the following street.
int rp_artifact_write_transitions(const char* path, int street, uint32_t K, uint32_t bins, const uint32_t* counts, const uint64_t* weight) {
    if (!path || !counts || !weight) return rp::fail(RP_ERR_INVALID, "null argument");
    if (street < 0 || street > 2 || bins > 256 || K > 256) return rp::fail(RP_ERR_INVALID, "street %d, K %u, bins %u", street, K, bins);
    Writer w;
    if (!w.open(path)) return rp::fail(RP_ERR_INVALID, "cannot create %s", path);
    std::vector<std::pair<uint32_t, float>> dist;
    for (uint32_t k = 0; k < K; ++k) {
        dist.clear();
        for (uint32_t b = 0; b < bins; ++b)
            if (counts[(size_t)k * bins + b]) dist.emplace_back(b, (float)counts[(size_t)k * bins + b] / (float)weight[k]);
        std::stable_sort(dist.begin(), dist.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
        const int16_t prev = (int16_t)((uint16_t)street << 8 | k);
        for (const auto& e : dist) {
            const int16_t next = (int16_t)((uint16_t)(street + 1) << 8 | e.first);
            w.begin_row(3);
            w.field(&prev, 2);
            w.field(&next, 2);
            w.field(&e.second, 4);
        }
    }
    if (!w.close()) return rp::fail(RP_ERR_INVALID, "short write to %s", path);
    return RP_OK;
}

// From<Edge> for u64 (kicker/src/edge.rs:122-160): the `edge BIGINT` column.  `code` is the 5-bit edge code of a Path
// (edge.rs:101-120): 1 Draw, 2 Fold, 3 Check, 4 Call, 5 Shove, 6..9 Open(2..5 bb), 10..19 Raise(odds) over the grid of
// pokerkit/src/lib.rs:81-97.
static uint64_t edge_code_to_u64(uint32_t code) {
    static const uint64_t OPENS[4] = {2, 3, 4, 5};
    static const uint64_t RAISES[10][2] = {{1, 4}, {1, 3}, {1, 2}, {2, 3}, {3, 4}, {1, 1}, {5, 4}, {3, 2}, {2, 1}, {3, 1}};
    switch (code) {
        case 1: return 0;
        case 2: return 1;
        case 3: return 2;
        case 4: return 3;
        case 5: return 5;
    }
    if (code >= 6 && code < 10) return 6ull | (OPENS[code - 6] << 3);
    if (code >= 10 && code < 20) return 4ull | (RAISES[code - 10][0] << 3) | (RAISES[code - 10][1] << 11);
    return ~0ull;
}

// NlheProfile::rows (nlhe/src/profile.rs:144-163) -> COPY blueprint (past, present, choices, edge, weight, regret, payoff,
// visits) (profile.rs:20-31): one row per (infoset, edge of its choices).  Input = rp_nlhe_export's arrays (9 Encounters per
// infoset, slot a = the a-th edge of `choices`).  only_visited != 0 skips infosets no update has touched (the reference's
// map holds an infoset from its first update on).
int rp_artifact_write_blueprint(const char* path, uint64_t n, const uint64_t* past, const uint32_t* present, const uint64_t* choices,
                                const rp_encounter* enc, int only_visited, uint64_t* rows_written) {
    if (!path || (n && (!past || !present || !choices || !enc))) return rp::fail(RP_ERR_INVALID, "null argument");
    Writer w;
    if (!w.open(path)) return rp::fail(RP_ERR_INVALID, "cannot create %s", path);
    uint64_t rows = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const rp_encounter* e = enc + i * 9;
        if (only_visited && e[0].visits == 0) continue;
        const int64_t p = (int64_t)past[i], c = (int64_t)choices[i];
        const int16_t b = (int16_t)present[i];
        uint64_t path_bits = choices[i];
        for (uint32_t a = 0; a < 9 && path_bits != 0 && (path_bits & 0x1f) != 0; ++a, path_bits >>= 5) {
            const uint64_t ev = edge_code_to_u64((uint32_t)(path_bits & 0x1f));
            if (ev == ~0ull) return rp::fail(RP_ERR_INVALID, "infoset %llu: choices hold an unknown edge code", (unsigned long long)i);
            const int64_t edge = (int64_t)ev;
            const int32_t visits = (int32_t)e[a].visits;
            w.begin_row(8);
            w.field(&p, 8);
            w.field(&b, 2);
            w.field(&c, 8);
            w.field(&edge, 8);
            w.field(&e[a].weight, 4);
            w.field(&e[a].regret, 4);
            w.field(&e[a].payoff, 4);
            w.field(&visits, 4);
            rows += 1;
        }
    }
    if (!w.close()) return rp::fail(RP_ERR_INVALID, "short write to %s", path);
    if (rows_written) *rows_written = rows;
    return RP_OK;
}

}  // extern "C"
