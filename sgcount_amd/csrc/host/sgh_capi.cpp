// sgh_capi.cpp — flat C entry points over the C++ host (for the Python tests; libsgcount_host.so).
// Every function returns 0 on success, 1 for an sgh::Error, 101 for an sgh::Panic (the reference's exit codes)
// and leaves the message in sgh_last_error().
#include <cstdlib>
#include <cstring>
#include <string>

#include "sgh.hpp"

static thread_local std::string g_err;
template <class F> static int guard(F &&f) {
    try { f(); g_err.clear(); return 0; }
    catch (const sgh::Panic &e) { g_err = e.what(); return 101; }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
}
static std::vector<std::string> split0(const char *blob, int n) {      // n NUL-terminated strings back to back
    std::vector<std::string> v;
    for (int i = 0; i < n; i++) { v.emplace_back(blob); blob += v.back().size() + 1; }
    return v;
}
static int put(const std::string &s, char *out, size_t cap) {
    if (s.size() + 1 > cap) { g_err = "output buffer too small"; return 1; }
    memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}

extern "C" {

const char *sgh_last_error(void) { return g_err.c_str(); }

int sgh_cli(int argc, char **argv) { return sgh::cli_main(argc, argv); }

// entropy offset of every input (offsetter.rs:185-210): reverse_out[i], index_out[i]
int sgh_entropy_offset_group(const char *library_path, const char *inputs_blob, int n_inputs, uint64_t subsample,
                             int *reverse_out, uint64_t *index_out) {
    return guard([&] {
        auto offs = sgh::entropy_offset_group(library_path, split0(inputs_blob, n_inputs), (size_t)subsample);
        for (size_t i = 0; i < offs.size(); i++) { reverse_out[i] = offs[i].reverse; index_out[i] = offs[i].index; }
    });
}

int sgh_positional_entropy(const char *path, uint64_t take, double *out, uint64_t cap, uint64_t *n_out) {
    return guard([&] {
        sgh::FastxReader rd(path);
        auto h = sgh::positional_entropy(rd, (size_t)take);
        if (h.size() > cap) throw sgh::Error("output buffer too small");
        memcpy(out, h.data(), h.size() * sizeof(double));
        *n_out = h.size();
    });
}

int sgh_minimize_mse(const double *ref, uint64_t n_ref, const double *cmp, uint64_t n_cmp, int *reverse, uint64_t *index) {
    return guard([&] {
        auto o = sgh::minimize_mse(std::vector<double>(ref, ref + n_ref), std::vector<double>(cmp, cmp + n_cmp));
        *reverse = o.reverse; *index = o.index;
    });
}

// names joined with '\n'
int sgh_generate_sample_names(const char *paths_blob, int n, char *out, uint64_t cap) {
    return guard([&] {
        auto names = sgh::generate_sample_names(split0(paths_blob, n));
        std::string s;
        for (size_t i = 0; i < names.size(); i++) s += (i ? "\n" : "") + names[i];
        if (put(s, out, cap)) throw sgh::Error(g_err);
    });
}

// gene map: parse `text` (or the file at `path` if text is NULL) and look up `sgrna`; found_out = 0/1
int sgh_genemap_get(const char *path, const char *text, const char *sgrna, char *gene_out, uint64_t cap, int *found_out) {
    return guard([&] {
        sgh::GeneMap g = text ? sgh::GeneMap::from_buffer(text) : sgh::GeneMap::from_path(path);
        const std::string *v = g.get(sgrna);
        *found_out = v != nullptr;
        if (v && put(*v, gene_out, cap)) throw sgh::Error(g_err);
    });
}

// first library alias (file order) missing from the gene map, or found_out = 0
int sgh_genemap_missing(const char *genemap_text, const char *library_path, char *alias_out, uint64_t cap, int *found_out) {
    return guard([&] {
        sgh::GeneMap g = sgh::GeneMap::from_buffer(genemap_text);
        sgh::Library lib = sgh::Library::from_path(library_path);
        const std::string *m = g.missing_alias(lib);
        *found_out = m != nullptr;
        if (m && put(*m, alias_out, cap)) throw sgh::Error(g_err);
    });
}

int sgh_generate_columns(const char *names_blob, int n, int with_genemap, char *out, uint64_t cap) {
    return guard([&] {
        sgh::GeneMap g;
        if (put(sgh::generate_columns(split0(names_blob, n), with_genemap ? &g : nullptr), out, cap)) throw sgh::Error(g_err);
    });
}

// results table from explicit per-sample counts (library order): counts[s * n_guides + i], pooled by id like
// Counter.  genemap_text may be NULL.
int sgh_format_results(const char *library_path, const uint64_t *counts, int n_samples, const char *names_blob,
                       const char *genemap_text, int include_zero, char *out, uint64_t cap) {
    return guard([&] {
        sgh::Library lib = sgh::Library::from_path(library_path);
        std::vector<sgh::SampleCounts> res(n_samples);
        for (int s = 0; s < n_samples; s++)
            for (size_t i = 0; i < lib.ids.size(); i++)
                if (counts[(size_t)s * lib.ids.size() + i]) res[s].by_id[lib.ids[i]] += counts[(size_t)s * lib.ids.size() + i];
        sgh::GeneMap g;
        if (genemap_text) g = sgh::GeneMap::from_buffer(genemap_text);
        if (put(sgh::format_results(res, lib, split0(names_blob, n_samples), genemap_text ? &g : nullptr, include_zero != 0),
                out, cap))
            throw sgh::Error(g_err);
    });
}

// library facts (library.rs): n records, size; ids joined by '\n'
int sgh_library_info(const char *path, uint64_t *n_out, uint64_t *size_out) {
    return guard([&] {
        sgh::Library lib = sgh::Library::from_path(path);
        *n_out = lib.seqs.size(); *size_out = lib.size;
    });
}

// FASTX reader check: number of records and total sequence bytes
int sgh_fastx_stats(const char *path, uint64_t *n_records, uint64_t *seq_bytes, uint64_t *id_bytes) {
    return guard([&] {
        sgh::FastxReader rd(path);
        sgh::RecordView r;
        *n_records = *seq_bytes = *id_bytes = 0;
        while (rd.next(r)) { (*n_records)++; *seq_bytes += r.seq_len; *id_bytes += r.id_len; }
    });
}

// the text path's byte source (sgh.cpp TextFeeder) with malloc'd buffers: walks the whole file the way count() does
// (cut at the last newline, carry the rest) and reports what it would push: parts, bytes, lines, and a checksum of
// (first_line, bytes, newlines) per part folded with the bytes themselves
int sgh_text_feeder_walk(const char *path, uint64_t slice_bytes, uint64_t threads, uint64_t *parts_out, uint64_t *bytes_out,
                         uint64_t *lines_out, uint64_t *fnv_out, int *first_byte_out, int *is_gz_out, uint64_t pgz_chunk, uint64_t *fallbacks_out) {
    return guard([&] {
        sgh::TextFeeder feed(path, (size_t)slice_bytes, 3, (size_t)threads, malloc, free, 0, (size_t)pgz_chunk);
        *first_byte_out = feed.first_byte; *is_gz_out = (feed.is_gz ? 1 : 0) | (feed.is_bgzf ? 2 : 0) | (feed.is_pgz ? 4 : 0);
        struct Fin { sgh::TextFeeder &f; uint64_t *o; ~Fin() { if (o) *o = f.pgz_fallbacks; } } fin{feed, fallbacks_out};
        uint64_t parts = 0, bytes = 0, first_line = 0, h = 1469598103934665603ull;
        size_t carry = 0;
        for (size_t k = 0;; k++) {
            uint8_t *data; size_t len; uint64_t nl; bool eof;
            feed.acquire(k, data, len, nl, eof);
            if (sgh::count_newlines(data, len) != nl) throw sgh::Error("newline count of a slice is wrong");
            uint8_t *part = data - carry;
            size_t part_len = carry + len, tail = 0;
            if (!eof) {
                const void *p = len ? memrchr(data, '\n', len) : nullptr;
                const size_t keep = p ? (size_t)((const uint8_t *)p - part) + 1 : 0;
                tail = part_len - keep;
                if (tail > sgh::TextFeeder::HEAD) throw sgh::Error("line too long");
                memcpy(feed.buffer_of(k + 1) + sgh::TextFeeder::HEAD - tail, part + keep, tail);
                part_len = keep;
            }
            if (part_len) {
                parts++; bytes += part_len;
                for (size_t i = 0; i < part_len; i++) { h ^= part[i]; h *= 1099511628211ull; }
                first_line += nl + (part[part_len - 1] != '\n' ? 1 : 0);
            }
            feed.release_below(k);
            carry = tail;
            if (eof) break;
        }
        *parts_out = parts; *bytes_out = bytes; *lines_out = first_line; *fnv_out = h;
    });
}

// the scan path's record source (sgh_scan.cpp FastqScanner): all records of a plain FASTQ file, in order.  usable_out = 0: the
// scanner declines the file (not plain FASTQ text) and nothing is written.  cap = capacity of `out` in records.
int sgh_scan_records(const char *path, uint32_t L, int reverse, uint32_t offset, int recursion, uint64_t threads, uint64_t block_bytes,
                     uint64_t *out, uint64_t cap, uint64_t *n_out, uint64_t *lines_out, int *usable_out, int source) {
    return guard([&] {
        sgh::ScanParams sp;
        sp.L = L; sp.reverse = reverse != 0; sp.offset = offset; sp.recursion = recursion != 0;
        sgh::FastqScanner sc(path, sp, (size_t)threads, (size_t)block_bytes, 8, source);
        *usable_out = sc.usable ? 1 : 0; *n_out = 0; *lines_out = 0;
        if (!sc.usable) return;
        const uint64_t *r; size_t n;
        uint64_t total = 0;
        while (sc.next(r, n)) {
            if (total + n > cap) throw sgh::Error("sgh_scan_records: output buffer too small");
            if (n) memcpy(out + total * sc.words, r, n * sc.words * 8);
            total += n;
            sc.release();
        }
        *n_out = total; *lines_out = sc.total_lines;
    });
}

// throughput of the text path's byte source alone: every slice acquired and released, nothing looked at (tools/inflate_bench.py)
int sgh_text_feeder_drain(const char *path, uint64_t slice_bytes, uint64_t threads, uint64_t pgz_chunk, uint64_t *bytes_out, uint64_t *lines_out,
                          double *busy_s_out, uint64_t *fallbacks_out, int *kind_out) {
    return guard([&] {
        sgh::TextFeeder feed(path, (size_t)slice_bytes, 3, (size_t)threads, malloc, free, (size_t)threads, (size_t)pgz_chunk);
        uint64_t bytes = 0, lines = 0;
        for (size_t k = 0;; k++) {
            uint8_t *data; size_t len; uint64_t nl; bool eof;
            feed.acquire(k, data, len, nl, eof);
            bytes += len; lines += nl;
            feed.release_below(k + 1);
            if (eof) break;
        }
        *bytes_out = bytes; *lines_out = lines; *busy_s_out = feed.busy_s; *fallbacks_out = feed.pgz_fallbacks;
        *kind_out = (feed.is_gz ? 1 : 0) | (feed.is_bgzf ? 2 : 0) | (feed.is_pgz ? 4 : 0);
    });
}

uint32_t sgh_crc32(uint32_t crc, const uint8_t *buf, uint64_t len) { return sgh::crc32_fast(crc, buf, (size_t)len); }

}  // extern "C"
