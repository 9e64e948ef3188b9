// tafl_replay.cpp — the reference's text replay buffer (write_to_file, game/main.rs:86-132) behind the C-ABI.
//
// Host-only code (no device work): the file format is line-based text, so the job is byte-exact formatting and the
// reference's FIFO rule.  What the reference does, restated:
//   * the existing file is read and split into LINES; every line is one element of `entries`            (main.rs:98-101)
//   * if entries.len() >= max_entries, the FIRST element — one line, not one record — is removed        (main.rs:104-106)
//   * the new record = side_len matrix rows (values joined by ','), the vector joined by ',', value1, value2, joined by
//     '\n', is pushed as one element                                                                      (main.rs:109-120)
//   * every element is written back followed by '\n'                                                      (main.rs:125-129)
// so the cap is a cap on lines and takes effect one line per call; this file reproduces exactly that.
#include <cstdint>
#include <cstdio>
#include <deque>
#include <string>
#include <vector>
#include "../../include/taflhip.h"

int tafl_fail_(int code, const char* msg);      // tafl_capi.hip: sets the message tafl_last_error() returns

namespace {

bool read_lines(const char* path, std::deque<std::string>& lines, bool& exists) {
    FILE* f = std::fopen(path, "rb");
    exists = f != nullptr;
    if (!f) return true;                                   // path.exists() == false: start empty (main.rs:98)
    std::string all; char buf[1 << 16]; size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, k);
    const bool ok = !std::ferror(f);
    std::fclose(f);
    if (!ok) return false;
    // str::lines(): split on '\n', a trailing '\r' of a line is dropped, no empty last element after a final '\n'
    size_t pos = 0;
    while (pos < all.size()) {
        size_t nl = all.find('\n', pos);
        size_t end = nl == std::string::npos ? all.size() : nl;
        size_t e2 = end;
        if (e2 > pos && all[e2 - 1] == '\r') --e2;
        lines.emplace_back(all, pos, e2 - pos);
        pos = nl == std::string::npos ? all.size() : nl + 1;
    }
    return true;
}

void join_u8(std::string& out, const uint8_t* v, uint32_t n) {
    char tmp[8];
    for (uint32_t i = 0; i < n; ++i) {
        if (i) out.push_back(',');
        out.append(tmp, (size_t)std::snprintf(tmp, sizeof tmp, "%u", (unsigned)v[i]));
    }
}

// the lines of one record, in order (the reference pushes them as one '\n'-joined element; on the next call they are
// side_len + 3 separate lines again)
void record_lines(std::deque<std::string>& lines, const uint8_t* matrix, uint8_t side_len, const uint8_t* vector, uint32_t vector_len,
                  uint8_t value1, uint8_t value2) {
    for (uint32_t r = 0; r < side_len; ++r) { std::string s; join_u8(s, matrix + (size_t)r * side_len, side_len); lines.push_back(std::move(s)); }
    std::string v; join_u8(v, vector, vector_len); lines.push_back(std::move(v));
    lines.push_back(std::to_string((unsigned)value1));
    lines.push_back(std::to_string((unsigned)value2));
}

bool write_lines(const char* path, const std::deque<std::string>& lines) {
    FILE* f = std::fopen(path, "wb");                      // write + create + truncate (main.rs:123)
    if (!f) return false;
    std::string out;
    for (const std::string& s : lines) { out += s; out.push_back('\n'); }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return std::fclose(f) == 0 && ok;
}

int fail(const char* what, const char* path) { return tafl_fail_(TAFL_ERR_INVALID_ARG, (std::string(what) + ": " + (path ? path : "(null)")).c_str()); }

}  // namespace

extern "C" {

int tafl_replay_append_batch(const char* path, const uint8_t* matrices, uint8_t side_len, uint32_t n, const uint8_t* vectors,
                             const uint32_t* vector_offsets, const uint8_t* values1, const uint8_t* values2, uint64_t max_entries) {
    if (!path || !matrices || !vector_offsets || !values1 || !values2 || side_len == 0) return fail("tafl_replay_append_batch: null argument", path);
    std::deque<std::string> lines; bool exists = false;
    if (!read_lines(path, lines, exists)) return fail("tafl_replay_append_batch: cannot read", path);
    for (uint32_t g = 0; g < n; ++g) {
        if (lines.size() >= max_entries && !lines.empty()) lines.pop_front();       // one LINE per call (main.rs:104-106)
        const uint32_t lo = vector_offsets[g], hi = vector_offsets[g + 1];
        if (hi < lo) return fail("tafl_replay_append_batch: vector_offsets not ascending", path);
        record_lines(lines, matrices + (size_t)g * side_len * side_len, side_len, vectors ? vectors + lo : nullptr, hi - lo, values1[g], values2[g]);
    }
    if (!write_lines(path, lines)) return fail("tafl_replay_append_batch: cannot write", path);
    return TAFL_OK;
}

int tafl_replay_append(const char* path, const uint8_t* matrix, uint8_t side_len, const uint8_t* vector, uint32_t vector_len,
                       uint8_t value1, uint8_t value2, uint64_t max_entries) {
    const uint32_t off[2] = {0u, vector_len};
    return tafl_replay_append_batch(path, matrix, side_len, 1, vector, off, &value1, &value2, max_entries);
}

// Reader (the reference has none): records are parsed from the END of the file, because the line-wise FIFO can only
// damage the oldest record.  Returns the newest `max_records` complete records, oldest first.
int tafl_replay_read(const char* path, uint8_t side_len, uint32_t max_records, uint8_t* matrices, uint8_t* vectors, uint32_t vector_cap,
                     uint32_t* vector_lens, uint8_t* values1, uint8_t* values2, uint32_t* out_n) {
    if (!path || !out_n || side_len == 0) return fail("tafl_replay_read: null argument", path);
    std::deque<std::string> lines; bool exists = false;
    if (!read_lines(path, lines, exists) || !exists) return fail("tafl_replay_read: cannot read", path);
    const size_t per = (size_t)side_len + 3;
    auto parse_u8s = [](const std::string& s, std::vector<uint8_t>& out) -> bool {
        out.clear();
        if (s.empty()) return true;
        size_t i = 0;
        while (true) {
            unsigned v = 0; size_t d = 0;
            while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (unsigned)(s[i] - '0'); ++i; ++d; if (v > 255) return false; }
            if (d == 0) return false;
            out.push_back((uint8_t)v);
            if (i == s.size()) return true;
            if (s[i] != ',') return false;
            ++i;
        }
    };
    const size_t avail = lines.size() / per;
    size_t take = avail < max_records ? avail : max_records;
    // validate from the newest backwards; stop at the first record that does not parse
    std::vector<std::vector<uint8_t>> mats, vecs; std::vector<uint8_t> v1s, v2s;
    std::vector<uint8_t> tmp;
    size_t good = 0;
    for (size_t k = 0; k < take; ++k) {
        const size_t base = lines.size() - (k + 1) * per;
        std::vector<uint8_t> m; bool ok = true;
        for (uint32_t r = 0; r < side_len && ok; ++r) { ok = parse_u8s(lines[base + r], tmp) && tmp.size() == side_len; m.insert(m.end(), tmp.begin(), tmp.end()); }
        std::vector<uint8_t> v, a, b2;
        ok = ok && parse_u8s(lines[base + side_len], v) && parse_u8s(lines[base + side_len + 1], a) && a.size() == 1
                && parse_u8s(lines[base + side_len + 2], b2) && b2.size() == 1;
        if (!ok) break;
        mats.push_back(std::move(m)); vecs.push_back(std::move(v)); v1s.push_back(a[0]); v2s.push_back(b2[0]);
        ++good;
    }
    for (size_t k = 0; k < good; ++k) {                     // oldest first
        const size_t src = good - 1 - k;
        if (matrices) for (size_t i = 0; i < mats[src].size(); ++i) matrices[k * (size_t)side_len * side_len + i] = mats[src][i];
        if (vector_lens) vector_lens[k] = (uint32_t)vecs[src].size();
        if (vectors) {
            if (vecs[src].size() > vector_cap) return fail("tafl_replay_read: vector_cap too small", path);
            for (size_t i = 0; i < vecs[src].size(); ++i) vectors[k * (size_t)vector_cap + i] = vecs[src][i];
        }
        if (values1) values1[k] = v1s[src];
        if (values2) values2[k] = v2s[src];
    }
    *out_n = (uint32_t)good;
    return TAFL_OK;
}

}  // extern "C"
