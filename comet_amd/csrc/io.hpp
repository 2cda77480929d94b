// io.hpp — byte sink / source for the on-disk formats of the reference's indexes (VectorIndex embeds
// io.WriterTo / io.ReaderFrom, index.go:58-60). The layouts written here are the reference's own:
// FLAT flat_index.go:348-360, IVFX ivf_index.go:441-462, PQIX pq_index.go:480-505, IVPQ ivfpq_index.go:507-535,
// HNSW hnsw_index.go:701-727 — little-endian (binary.Write(w, binary.LittleEndian, ...)), every scalar 4 bytes
// except the 1-byte `trained` flag and HNSW's float64 levelMult.
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace comet {

// buffered writer over the caller's callback (the cgo shim forwards to io.Writer.Write)
struct Sink {
    comet_write_cb cb; void* user; int64_t total = 0;
    std::vector<uint8_t> buf;
    Sink(comet_write_cb cb_, void* user_) : cb(cb_), user(user_) { buf.reserve(1 << 20); }
    void flush() {
        if (buf.empty()) return;
        if (cb(user, buf.data(), buf.size()) != 0) COMET_FAIL(COMET_ERR_IO, "write failed after %lld bytes", (long long)total);
        buf.clear();
    }
    void put(const void* p, size_t n) {
        total += (int64_t)n;
        if (n >= ((size_t)1 << 20)) { flush(); if (cb(user, p, n) != 0) COMET_FAIL(COMET_ERR_IO, "write failed after %lld bytes", (long long)total); return; }
        if (buf.size() + n > ((size_t)1 << 20)) flush();
        buf.insert(buf.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    }
    void u8(uint8_t v) { put(&v, 1); }
    void u32(uint32_t v) { put(&v, 4); }
    void i32(int32_t v) { put(&v, 4); }
    void f64(double v) { put(&v, 8); }
    void str(const std::string& s) { u32((uint32_t)s.size()); put(s.data(), s.size()); }
};

// reader over the caller's callback (io.ReadFull semantics: the callback fills exactly `len` bytes or fails)
struct Source {
    comet_read_cb cb; void* user; int64_t total = 0;
    Source(comet_read_cb cb_, void* user_) : cb(cb_), user(user_) {}
    void get(void* p, size_t n, const char* what) {
        if (n == 0) return;
        if (cb(user, p, n) != 0) COMET_FAIL(COMET_ERR_IO, "failed to read %s: unexpected EOF", what);
        total += (int64_t)n;
    }
    uint8_t u8(const char* what) { uint8_t v; get(&v, 1, what); return v; }
    uint32_t u32(const char* what) { uint32_t v; get(&v, 4, what); return v; }
    int32_t i32(const char* what) { int32_t v; get(&v, 4, what); return v; }
    double f64(const char* what) { double v; get(&v, 8, what); return v; }
    std::string str(const char* what, size_t max_len = 64) {
        const uint32_t n = u32(what);
        if (n > max_len) COMET_FAIL(COMET_ERR_FORMAT, "%s length %u is not plausible", what, n);
        std::string s(n, '\0'); get(s.data(), n, what); return s;
    }
};

inline const char* metric_name(int metric) {   // DistanceKind strings, distance.go:24-39
    return metric == COMET_L2 ? "l2" : metric == COMET_L2SQ ? "l2_squared" : "cosine";
}

// common header: magic, version 1, dim, distance kind (identical in all five formats)
inline void write_header(Sink& s, const char magic[4], int dim, int metric) {
    s.put(magic, 4); s.u32(1); s.u32((uint32_t)dim); s.str(metric_name(metric));
}
inline void read_header(Source& s, const char magic[4], int dim, int metric) {
    char m[5] = {0, 0, 0, 0, 0};
    s.get(m, 4, "magic number");
    if (std::memcmp(m, magic, 4) != 0) COMET_FAIL(COMET_ERR_FORMAT, "invalid magic number: expected '%.4s', got '%s'", magic, m);
    const uint32_t version = s.u32("version");
    if (version != 1) COMET_FAIL(COMET_ERR_FORMAT, "unsupported version: %u", version);
    const uint32_t d = s.u32("dimensionality");
    if ((int)d != dim) COMET_FAIL(COMET_ERR_FORMAT, "dimension mismatch: index has dim=%d, serialized data has dim=%u", dim, d);
    const std::string kind = s.str("distance kind");
    if (kind != metric_name(metric)) COMET_FAIL(COMET_ERR_FORMAT, "distance kind mismatch: index uses '%s', serialized data uses '%s'", metric_name(metric), kind.c_str());
}
inline void check_param(const char* name, int have, uint32_t got) {
    if ((int)got != have) COMET_FAIL(COMET_ERR_FORMAT, "parameter %s mismatch: index has %s=%d, serialized data has %s=%u", name, name, have, name, got);
}

// Soft-delete tail. WriteTo flushes first, so the bitmap is always EMPTY: roaring's portable format for an empty
// bitmap is the 8-byte preamble {cookie 12346 (SERIAL_COOKIE_NO_RUNCONTAINER), 0 containers} (roaring v1.9.4
// roaringArray.writeTo). The reader accepts any well-formed portable bitmap (array / bitmap / run containers)
// and returns its members, so files holding soft deletes written by other tools load too.
inline void write_empty_bitmap(Sink& s) { s.u32(8); s.u32(12346u); s.u32(0u); }
inline std::vector<uint32_t> read_bitmap(Source& s) {
    const uint32_t size = s.u32("bitmap size");
    if (size > (1u << 30)) COMET_FAIL(COMET_ERR_FORMAT, "bitmap size %u is not plausible", size);
    std::vector<uint8_t> b(size);
    s.get(b.data(), size, "bitmap data");
    std::vector<uint32_t> out;
    // roaring.UnmarshalBinary of zero bytes fails in the reference (no cookie to read: flat_index.go:605-607 returns the wrapped error)
    if (size == 0) COMET_FAIL(COMET_ERR_FORMAT, "failed to deserialize deleted nodes bitmap: empty");
    size_t off = 0;
    auto need = [&](size_t n) { if (off + n > b.size()) COMET_FAIL(COMET_ERR_FORMAT, "failed to deserialize deleted nodes bitmap: truncated"); };
    auto rd32 = [&]() { need(4); uint32_t v; std::memcpy(&v, &b[off], 4); off += 4; return v; };
    auto rd16 = [&]() { need(2); uint16_t v; std::memcpy(&v, &b[off], 2); off += 2; return v; };
    const uint32_t cookie = rd32();
    uint32_t ncont = 0; bool has_run = false; std::vector<uint8_t> run_flags;
    if ((cookie & 0xFFFF) == 12347u) { has_run = true; ncont = (cookie >> 16) + 1; need((ncont + 7) / 8); run_flags.assign(b.begin() + off, b.begin() + off + (ncont + 7) / 8); off += (ncont + 7) / 8; }
    else if (cookie == 12346u) ncont = rd32();
    else COMET_FAIL(COMET_ERR_FORMAT, "failed to deserialize deleted nodes bitmap: bad cookie %u", cookie);
    if (ncont > 65536) COMET_FAIL(COMET_ERR_FORMAT, "failed to deserialize deleted nodes bitmap: %u containers", ncont);
    std::vector<uint16_t> keys(ncont); std::vector<uint32_t> cards(ncont);
    for (uint32_t i = 0; i < ncont; i++) { keys[i] = rd16(); cards[i] = (uint32_t)rd16() + 1; }
    if (!has_run || ncont >= 4) off += (size_t)4 * ncont;   // offset header (always present without runs; with runs only from 4 containers)
    for (uint32_t i = 0; i < ncont; i++) {
        const uint32_t hi = (uint32_t)keys[i] << 16;
        const bool is_run = has_run && ((run_flags[i >> 3] >> (i & 7)) & 1);
        if (is_run) {
            const uint16_t nruns = rd16();
            for (uint16_t r = 0; r < nruns; r++) { const uint16_t start = rd16(), len = rd16(); for (uint32_t v = start; v <= (uint32_t)start + len; v++) out.push_back(hi | v); }
        } else if (cards[i] > 4096) {
            need(8192);
            for (uint32_t w = 0; w < 1024; w++) { uint64_t word; std::memcpy(&word, &b[off + 8 * w], 8); while (word) { const int bit = __builtin_ctzll(word); out.push_back(hi | (w * 64 + bit)); word &= word - 1; } }
            off += 8192;
        } else {
            for (uint32_t j = 0; j < cards[i]; j++) out.push_back(hi | rd16());
        }
    }
    return out;
}

}  // namespace comet
