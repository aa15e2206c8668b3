// sgh.cpp — C++ host side (see sgh.hpp).  Everything here is plain host code; reads are matched and counted on
// the GPU through the C ABI of include/sgcount_hip.h (count()).
#include "sgh.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <future>
#include <thread>
#include <unordered_set>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../../include/sgcount_hip.h"

namespace sgh {

// =====================================================================================================
// FASTX reader
// =====================================================================================================
struct FastxReader::Impl {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    int fmt = 0;          // 0 unknown, 1 fasta, 2 fastq
    std::string path;

    // reads one line starting at `pos`; keeps bytes from `keep_from` valid across refills.
    // returns false at end of input.  off/len are offsets into buf.
    bool line(size_t &keep_from, size_t &off, size_t &len) {
        size_t scan = pos;
        for (;;) {
            const char *nl = (const char *)memchr(buf.data() + scan, '\n', end - scan);
            if (nl) { off = pos; len = (size_t)(nl - (buf.data() + pos)); pos = off + len + 1; return true; }
            if (eof) {
                if (pos >= end) return false;
                off = pos; len = end - pos; pos = end; return true;
            }
            scan = end;
            // refill: drop everything before keep_from
            if (keep_from > 0) {
                memmove(buf.data(), buf.data() + keep_from, end - keep_from);
                pos -= keep_from; scan -= keep_from; end -= keep_from; shift += keep_from; keep_from = 0;
            }
            if (end == buf.size()) buf.resize(buf.size() * 2);
            const int got = gzread(f, buf.data() + end, (unsigned)std::min<size_t>(buf.size() - end, 1u << 30));
            if (got < 0) throw Error("read error in " + path);
            if (got == 0) eof = true;
            end += (size_t)got;
        }
    }
    size_t shift = 0;     // total bytes dropped by refills during the current record
};

FastxReader::FastxReader(const std::string &path) : p(new Impl) {
    p->path = path;
    p->f = gzopen(path.c_str(), "rb");      // transparently reads plain files too
    if (!p->f) throw Error("No such file or directory (os error 2): " + path);
    gzbuffer(p->f, 1u << 20);
    p->buf.resize(4u << 20);
}

FastxReader::~FastxReader() { if (p && p->f) gzclose(p->f); }

bool FastxReader::next(RecordView &r) {
    Impl &s = *p;
    size_t keep = s.pos, o[4], l[4];
    s.shift = 0;
    size_t base_shift = 0;
    auto get = [&](int k) -> bool {
        const size_t before = s.shift;
        if (!s.line(keep, o[k], l[k])) return false;
        const size_t d = s.shift - before;          // earlier lines moved left by d
        for (int j = 0; j < k; j++) o[j] -= d;
        base_shift += d;
        return true;
    };
    if (!get(0)) return false;
    if (l[0] == 0) {
        // blank lines at the very end are not records (the scanner and the text path cut them off the same way); a blank line with
        // anything behind it is a malformed header, below
        bool only_blank = true;
        size_t bo, bl;
        const size_t pos0 = s.pos;
        size_t keep2 = keep;
        while (only_blank && s.line(keep2, bo, bl)) only_blank = bl == 0;
        if (only_blank) return false;
        (void)pos0;
        throw Panic("malformed FASTX header in " + s.path);
    }
    if (s.fmt == 0) {
        if (l[0] && s.buf[o[0]] == '>') s.fmt = 1;
        else if (l[0] && s.buf[o[0]] == '@') s.fmt = 2;
        else throw Error("not a FASTA/FASTQ file: " + s.path);
    }
    if (l[0] == 0 || s.buf[o[0]] != (s.fmt == 1 ? '>' : '@')) throw Panic("malformed FASTX header in " + s.path);
    if (!get(1)) throw Panic("truncated FASTX record in " + s.path);
    if (s.fmt == 2) {
        if (!get(2) || l[2] == 0 || s.buf[o[2]] != '+') throw Panic("malformed FASTQ record in " + s.path);
        // a stream that ends behind the separator line ends with a record whose quality line is empty — written with its terminator
        // ("+\n\n"), without it ("+\n") or not at all ("+"): the same bytes up to what no reader can tell apart (decision #3)
        if (!get(3)) { o[3] = o[2] + l[2]; l[3] = 0; }
    }
    // a '\r' before the '\n' is part of the line terminator (CRLF files; fxread's behaviour is unpinned)
    for (int k = 0; k < 2; k++) if (l[k] && s.buf[o[k] + l[k] - 1] == '\r') l[k]--;
    r.id = s.buf.data() + o[0] + 1; r.id_len = l[0] - 1;
    r.seq = s.buf.data() + o[1]; r.seq_len = l[1];
    return true;
}

// =====================================================================================================
// Library — src/library.rs
// =====================================================================================================
Library Library::from_path(const std::string &path) {
    Library lib;
    FastxReader rd(path);
    RecordView r;
    while (rd.next(r)) {
        std::string seq(r.seq, r.seq_len);
        if (!lib.index.emplace(seq, lib.seqs.size()).second)           // library.rs:91-96
            throw Panic("Unexpected duplicate sequence in library found: " + seq);
        lib.seqs.push_back(std::move(seq));
        lib.ids.emplace_back(r.id, r.id_len);
    }
    if (lib.seqs.empty()) throw Panic("called `Option::unwrap()` on a `None` value");      // library.rs:74
    for (size_t i = 1; i < lib.seqs.size(); i++)
        if (lib.seqs[i].size() != lib.seqs[i - 1].size()) throw Error("Library sequence sizes are inconsistent");   // :83
    lib.size = lib.seqs[0].size();
    return lib;
}

const std::string *Library::alias(const std::string &seq) const {
    auto it = index.find(seq);
    return it == index.end() ? nullptr : &ids[it->second];
}

// =====================================================================================================
// Offsetter — src/offsetter.rs
// =====================================================================================================
std::string Offset::debug() const { return std::string(reverse ? "Reverse(" : "Forward(") + std::to_string(index) + ")"; }

static int base_map(char c) {                         // offsetter.rs:42-50
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

std::vector<double> positional_entropy(FastxReader &reader, size_t take) {
    RecordView r;
    if (take == 0 || !reader.next(r)) throw Panic("empty reader");            // offsetter.rs:38 expect
    const size_t size = r.seq_len;                                              // first record: size probe, not counted
    std::vector<double> m(size * 4, 0.0);
    size_t seen = 1;
    while (seen < take && reader.next(r)) {                                     // .take(subsample) counts the probe too
        seen++;
        const size_t lim = std::min(r.seq_len, size);
        for (size_t i = 0; i < lim; i++) {
            const int j = base_map(r.seq[i]);
            if (j >= 0) m[i * 4 + j] += 1.0;
            else { m[i * 4] += 1.0; m[i * 4 + 1] += 1.0; m[i * 4 + 2] += 1.0; m[i * 4 + 3] += 1.0; }   // :70-75
        }
    }
    std::vector<double> h(size);
    for (size_t i = 0; i < size; i++) {
        double s = 0.0;
        for (int j = 0; j < 4; j++) s += m[i * 4 + j];
        double e = 0.0;
        for (int j = 0; j < 4; j++) {
            const double pr = m[i * 4 + j] / s;          // 0/0 = NaN propagates like ndarray's division
            if (pr == 0.0) continue;
            e -= pr * std::log(pr);                      // ndarray-stats entropy(): -Σ p ln p
        }
        h[i] = e;
    }
    return h;
}

static std::vector<double> windowed_mse(const std::vector<double> &a, const std::vector<double> &b) {   // :109-120
    const size_t size = b.size() - a.size() + 1;
    std::vector<double> out(size);
    for (size_t x = 0; x < size; x++) {
        double s = 0.0;
        for (size_t k = 0; k < a.size(); k++) { const double d = a[k] - b[x + k]; s += d * d; }
        out[x] = s / (double)a.size();
    }
    return out;
}

static size_t argmin_first(const std::vector<double> &v) {       // ndarray-stats argmin; NaN ⇒ Err ⇒ panic (:123-141)
    size_t a = 0;
    for (size_t i = 0; i < v.size(); i++) {
        if (std::isnan(v[i])) throw Panic("Unexpected minmax error in entropy: Undefined ordering between a tested pair of values.");
        if (v[i] < v[a]) a = i;
    }
    return a;
}

Offset minimize_mse(const std::vector<double> &reference, const std::vector<double> &comparison) {
    if (comparison.size() < reference.size())                    // :154-156
        throw Error("Sequences in reference library are larger than the sequences in input.\nConsider reducing the length of "
                    "your reference sequences (i.e. extracting the variable region of the sgRNA or reducing the length of the "
                    "adapters.)");
    std::vector<double> rev(comparison.rbegin(), comparison.rend());
    const std::vector<double> mf = windowed_mse(reference, comparison), mr = windowed_mse(reference, rev);
    const size_t af = argmin_first(mf), ar = argmin_first(mr);
    Offset o;
    if (mf[af] < mr[ar]) { o.reverse = false; o.index = af; }    // :143 strict <
    else { o.reverse = true; o.index = ar; }
    return o;
}

std::vector<Offset> entropy_offset_group(const std::string &library_path, const std::vector<std::string> &inputs,
                                         size_t subsample) {
    FastxReader ref(library_path);
    const std::vector<double> reference = positional_entropy(ref, (size_t)-1);
    std::vector<Offset> out;
    for (const auto &path : inputs) {
        std::unique_ptr<FastxReader> rd;
        try { rd.reset(new FastxReader(path)); } catch (const Error &) { throw Panic("Unable to open file: " + path); }   // :195
        const std::vector<double> cmp = positional_entropy(*rd, subsample);
        try { out.push_back(minimize_mse(reference, cmp)); }
        catch (const Error &e) { throw Error(std::string("Error in entropy offset calculation:\n\n") + e.what()); }        // :205
    }
    return out;
}

// =====================================================================================================
// GeneMap — src/genemap.rs
// =====================================================================================================
GeneMap GeneMap::from_buffer(const std::string &text) {
    GeneMap g;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        size_t le = e;
        if (le > p && text[le - 1] == '\r') le--;                 // bstr for_byte_line strips \n and \r\n
        const std::string line = text.substr(p, le - p);
        p = e + 1;
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) throw Panic("Missing '\t' in gene map");                 // :58
        const std::string gene = line.substr(0, tab), sgrna = line.substr(tab + 1);
        if (!g.map.emplace(sgrna, gene).second) throw Panic("Duplicate sgRNA key found in gene map: " + sgrna);   // :60-64
    }
    return g;
}

GeneMap GeneMap::from_path(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw Error("Provided gene mapping path doesn't exist: " + path);                  // :37-42
    std::string text;
    char tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) text.append(tmp, n);
    fclose(f);
    return from_buffer(text);
}

const std::string *GeneMap::get(const std::string &sgrna) const {
    auto it = map.find(sgrna);
    return it == map.end() ? nullptr : &it->second;
}

const std::string *GeneMap::missing_alias(const Library &lib) const {
    for (const auto &id : lib.ids)
        if (!get(id)) return &id;
    return nullptr;
}

// =====================================================================================================
// utils / results
// =====================================================================================================
static void trim_end_matches(std::string &s, const char *pat) {       // Rust str::trim_end_matches: repeated
    const size_t n = strlen(pat);
    while (s.size() >= n && n && s.compare(s.size() - n, n, pat) == 0) s.resize(s.size() - n);
}

std::vector<std::string> generate_sample_names(const std::vector<std::string> &paths) {
    std::vector<std::string> base, simple;
    for (size_t i = 0; i < paths.size(); i++) {
        std::string b = paths[i].substr(paths[i].find_last_of('/') == std::string::npos ? 0 : paths[i].find_last_of('/') + 1);
        trim_end_matches(b, ".gz"); trim_end_matches(b, ".fasta"); trim_end_matches(b, ".fastq");
        trim_end_matches(b, ".fa"); trim_end_matches(b, ".fq");
        base.push_back(b);
        simple.push_back("Sample." + std::to_string(i));
    }
    std::unordered_set<std::string> seen(base.begin(), base.end());
    if (seen.size() == base.size()) return base;
    fprintf(stderr, "WARNING: Duplicate Basenames Detected, Using incrementing sample names\n");     // utils.rs:46
    return simple;
}

uint64_t SampleCounts::get_value(const std::string &id) const {
    auto it = by_id.find(id);
    return it == by_id.end() ? 0 : it->second;
}

std::string generate_columns(const std::vector<std::string> &names, const GeneMap *genemap) {
    std::string s = "Guide";
    for (size_t i = 0; i < names.size(); i++) {
        if (i == 0 && genemap) s += "\tGene";
        s += "\t" + names[i];
    }
    return s;
}

std::string format_results(const std::vector<SampleCounts> &results, const Library &library,
                           const std::vector<std::string> &names, const GeneMap *genemap, bool include_zero) {
    std::string out = generate_columns(names, genemap) + "\n";
    for (const auto &alias : library.ids) {
        uint64_t total = 0;
        std::string row = alias;
        for (size_t i = 0; i < results.size(); i++) {
            if (i == 0 && genemap) {                                                  // results.rs:46-62 append_gene
                const std::string *gene = genemap->get(alias);
                if (!gene) throw Panic("Missing sgrna -> gene mapping");
                row += "\t" + *gene;
            }
            const uint64_t c = results[i].get_value(alias);
            row += "\t" + std::to_string(c);
            total += c;
        }
        if (include_zero || total > 0) out += row + "\n";                             // results.rs:90-94
    }
    return out;
}

void write_results(const std::string &path, const std::vector<SampleCounts> &results, const Library &library,
                   const std::vector<std::string> &names, const GeneMap *genemap, bool include_zero) {
    const std::string text = format_results(results, library, names, genemap, include_zero);
    if (path.empty()) { fwrite(text.data(), 1, text.size(), stdout); fflush(stdout); return; }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot create output file: " + path);
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
}

// =====================================================================================================
// count — src/count.rs
// =====================================================================================================
static void sgc_check(int rc, const char *what) {
    if (rc != SGC_OK) {
        const std::string msg = std::string(what) + ": " + sgc_last_error();
        if (rc == SGC_E_DUPLICATE) throw Panic(msg);
        throw Error(msg);
    }
}

// =====================================================================================================
// FASTQ at speed (replaces fxread inside the count loop, src/count.rs:24 + src/counter.rs:211-236)
//
// The host never parses records.  It moves text from the page cache (or from zlib) into pinned buffers with several
// threads, counts the newlines of every block while it is still cache-hot, cuts each slice at its last newline and
// hands [carry | slice] to sgc_sample_push_fastq_part together with the line number it starts at.  Record boundaries,
// marker validation, window extraction and packing happen on the GPU; uploads, ingest kernels and count kernels of
// consecutive parts overlap (upload stream + alternating device buffers inside the library).
// =====================================================================================================

// Number of '\n' in [p, p + n): byte compares summed in 8-bit lanes (the compiler turns the inner loop into vector
// compares; 255 iterations cannot overflow a lane), flushed to a wide sum.
size_t count_newlines(const uint8_t *p, size_t n) {
    size_t total = 0;
    while (n) {
        const size_t m = std::min<size_t>(n, 255 * 64);
        uint8_t acc[64] = {0};
        size_t i = 0;
        for (; i + 64 <= m; i += 64)
            for (int k = 0; k < 64; k++) acc[k] += (uint8_t)(p[i + k] == '\n');
        for (; i < m; i++) total += p[i] == '\n';
        for (int k = 0; k < 64; k++) total += acc[k];
        p += m; n -= m;
    }
    return total;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double unix_s() {
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

TextFeeder::TextFeeder(const std::string &path_, size_t slice_bytes, size_t ring, size_t threads, void *(*alloc)(size_t),
                       void (*release)(void *), size_t inflate_threads, size_t pgz_chunk)
    : path(path_), free_fn(release) {
    pgz_chunk_bytes = pgz_chunk;
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("No such file or directory (os error 2): " + path);
    unsigned char magic[2] = {0, 0};
    const ssize_t got = pread(fd, magic, 2, 0);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); throw Error("cannot stat " + path); }
    file_size = (size_t)sb.st_size;
    is_gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (is_gz) {
        // BGZF?  (first member: FEXTRA with a 'B' 'C' subfield of length 2)
        unsigned char hd[18];
        if (pread(fd, hd, 18, 0) == 18 && hd[2] == 8 && (hd[3] & 4) && hd[10] + 256u * hd[11] >= 6 && hd[12] == 'B' && hd[13] == 'C' && hd[14] == 2 && hd[15] == 0 &&
            file_size >= 28) {
            void *m = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                // BGZF only if EVERY member announces its size (bgzip output concatenated with ordinary gzip output is legal
                // gzip, and zlib — like the reference's flate2 — reads it): hop through the member headers once; anything
                // else takes the sequential inflater below
                const uint8_t *mp = (const uint8_t *)m;
                size_t off = 0;
                bool chain = true;
                while (chain && off < file_size) {
                    const uint8_t *h = mp + off;
                    if (file_size - off < 28 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { chain = false; break; }
                    const size_t xlen = h[10] + 256u * h[11];
                    if (12 + xlen + 8 > file_size - off) { chain = false; break; }
                    size_t bsize = 0;
                    for (size_t x = 12; x + 4 <= 12 + xlen;) {
                        const size_t slen = h[x + 2] + 256u * h[x + 3];
                        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)(h[x + 4] + 256u * h[x + 5]) + 1;
                        x += 4 + slen;
                    }
                    if (bsize < 12 + xlen + 8 || bsize > file_size - off) chain = false;
                    else off += bsize;
                }
                if (chain) { map = mp; is_bgzf = true; madvise(m, file_size, MADV_SEQUENTIAL); }
                else munmap(m, file_size);
            }
        }
    }
    if (is_gz && !is_bgzf && std::max(threads, inflate_threads) > 1 && file_size >= 64 && S_ISREG(sb.st_mode)) {
        // one deflate stream, several cores: chunks of the compressed file are decoded speculatively in parallel and stitched
        // in order (run_pgz)
        void *m = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) { map = (const uint8_t *)m; is_pgz = true; madvise(m, file_size, MADV_SEQUENTIAL); }
    }
    if (is_bgzf) {
        n_slices_known = (size_t)-1;                    // known when the member scan reaches the end of the file
        threads = std::max(threads, inflate_threads);   // inflating is several times the work of copying
    } else if (is_pgz) {
        n_slices_known = (size_t)-1;
        threads = std::max(threads, inflate_threads);
        slice_bytes = std::min<size_t>(slice_bytes, 32u << 20);
        if (!pgz_chunk_bytes) pgz_chunk_bytes = (size_t)1 << 20;        // (larger chunks cost more in buffers than they save in searches: sgh_scan.cpp)
        pgz_chunks = (file_size + pgz_chunk_bytes - 1) / pgz_chunk_bytes;
        pgz_pieces.resize(pgz_chunks);
        pgz_window.assign(32768, 0);
    } else if (is_gz) {
        gz = gzdopen(dup(fd), "rb");
        if (!gz) { close(fd); throw Error("cannot open the gzip stream of " + path); }
        gzbuffer(gz, 1u << 20);
        threads = 1;
        slice_bytes = std::min<size_t>(slice_bytes, 32u << 20);
    } else {
        // small inputs do not need the full-size ring
        const size_t need = std::max<size_t>((file_size + 4095) & ~(size_t)4095, 1u << 16);
        slice_bytes = std::min(slice_bytes, need);
        n_slices_known = (file_size + slice_bytes - 1) / slice_bytes;
        if (n_slices_known == 0) n_slices_known = 1;
        first_byte = got >= 1 ? magic[0] : 0;
        // The readers copy from a mapping of the file, as the scanner reads it (sgh_scan.cpp, head): the first read() of page-cache
        // pages that were written a moment ago moves each of them to the active LRU list under one lock — 16 GB/s however many threads
        // read, 3.05 s instead of 1.38 for the first run over a fresh 31 GB file (BENCH_r03 e2e.plain_gpu_parsed_text) —, a mapping
        // whose pages are dropped without their accessed bits being honoured (MADV_SEQUENTIAL) does not.  pread() stays for what cannot
        // be mapped.
        if (file_size && S_ISREG(sb.st_mode)) {
            void *m = mmap(nullptr, file_size, PROT_READ, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) { map = (const uint8_t *)m; plain_mapped = true; (void)madvise(m, file_size, MADV_SEQUENTIAL); }
        }
    }
    slice = std::max<size_t>(slice_bytes, 1u << 16);
    n_threads = std::max<size_t>(1, threads);
    ring_n = std::max<size_t>(2, ring);
    bufs.assign(ring_n, nullptr);
    for (size_t i = 0; i < ring_n; i++) {
        bufs[i] = (uint8_t *)alloc(HEAD + slice);
        if (!bufs[i]) { shutdown(); throw Error("cannot allocate pinned host buffers"); }
    }
    slots.resize(ring_n);
    plans.resize(ring_n);
    if (is_bgzf) {
        for (size_t t = 0; t < n_threads; t++) workers.emplace_back([this] { run_bgzf(); });
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return (slots[0].ready && slots[0].index == 0) || failed; });
        if (!failed) first_byte = slots[0].len ? bufs[0][HEAD] : 0;
    } else if (is_pgz) {
        n_threads = std::min(n_threads, std::max<size_t>(pgz_chunks, 1));
        for (size_t t = 0; t < n_threads; t++) workers.emplace_back([this] { run_pgz(); });
        std::unique_lock<std::mutex> lk(mu);
        size_t len0 = 0; bool eof0 = false;
        cv.wait(lk, [&] { return failed || slice_ready_locked(0, len0, eof0); });
        if (!failed) first_byte = len0 ? bufs[0][HEAD] : 0;
    } else if (is_gz) {
        // the first byte of the stream decides "FASTQ or not": inflate the first slice eagerly in the producer
        workers.emplace_back([this] { run_gz(); });
        // wait for slice 0 to learn the first byte
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return (slots[0].ready && slots[0].index == 0) || failed; });
        if (!failed) first_byte = slots[0].len ? bufs[0][HEAD] : 0;
    } else {
        for (size_t t = 0; t < n_threads; t++) workers.emplace_back([this] { run_plain(); });
    }
}

void TextFeeder::shutdown() {
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    for (auto &w : workers) if (w.joinable()) w.join();
    workers.clear();
    for (auto &b : bufs) if (b) { free_fn(b); b = nullptr; }
    if (gz) { gzclose(gz); gz = nullptr; }
    if (map) { munmap((void *)map, file_size); map = nullptr; }
    if (fd >= 0) { close(fd); fd = -1; }
}

TextFeeder::~TextFeeder() { shutdown(); }

// plain files: jobs (slice k, sub-range j) are handed out in order; a job of slice k may start once the buffer of
// slice k - ring has been released
void TextFeeder::run_plain() {
    try {
        for (;;) {
            size_t job;
            {
                std::unique_lock<std::mutex> lk(mu);
                job = next_job++;
                const size_t k = job / n_threads;
                if (k >= n_slices_known) return;
                cv.wait(lk, [&] { return stop || k < released + ring_n; });
                if (stop) return;
                Slot &sl = slots[k % ring_n];
                if (sl.index != k) { sl.index = k; sl.ready = false; sl.pending = n_threads; sl.newlines = 0; sl.len = 0; sl.eof = false; }
            }
            const size_t k = job / n_threads, j = job % n_threads;
            const size_t s0 = k * slice, s_len = std::min(slice, file_size - std::min(file_size, s0));
            // sub-range j of the slice, cut at 4 KiB multiples
            const size_t per = ((s_len + n_threads - 1) / n_threads + 4095) & ~(size_t)4095;
            const size_t lo = std::min(s_len, j * per), hi = std::min(s_len, lo + per);
            uint8_t *dst = bufs[k % ring_n] + HEAD;
            const double t0 = now_s();
            uint64_t nl = 0;
            for (size_t off = lo; off < hi;) {
                const size_t want = std::min<size_t>(hi - off, 1u << 20);
                if (plain_mapped) {
                    memcpy(dst + off, map + s0 + off, want);
                    nl += count_newlines(dst + off, want);
                    off += want;
                    continue;
                }
                const ssize_t r = pread(fd, dst + off, want, (off_t)(s0 + off));
                if (r < 0) throw Error("read error in " + path);
                if (r == 0) throw Error("file shrank while reading: " + path);
                nl += count_newlines(dst + off, (size_t)r);
                off += (size_t)r;
            }
            // (the sub-range's page-table entries go here, in parallel, not at exit: its bounds are multiples of 4 KiB but for the file's end)
            if (plain_mapped && hi > lo) (void)madvise((void *)(map + s0 + lo), hi - lo, MADV_DONTNEED);
            const double dt = now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                Slot &sl = slots[k % ring_n];
                sl.newlines += nl;
                busy_s += dt;
                if (--sl.pending == 0) { sl.len = s_len; sl.eof = k + 1 == n_slices_known; sl.ready = true; }
            }
            cv.notify_all();
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
}

// BGZF: the members from scan_off on whose inflated sizes (ISIZE trailers) fit one slice.  Called under the lock, for
// k = 0, 1, 2, ... in order.
void TextFeeder::plan_bgzf_slice(size_t k) {
    std::vector<BlockRef> &pl = plans[k % ring_n];
    pl.clear();
    size_t out = 0;
    while (scan_off < file_size) {
        const uint8_t *h = map + scan_off;
        if (file_size - scan_off < 28 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
            throw Error("corrupt BGZF member header in " + path);
        const size_t xlen = h[10] + 256u * h[11];
        if (12 + xlen + 8 > file_size - scan_off) throw Error("corrupt BGZF member header in " + path);
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + xlen;) {                       // the extra subfields: SI1 SI2 SLEN data
            const size_t slen = h[x + 2] + 256u * h[x + 3];
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)(h[x + 4] + 256u * h[x + 5]) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > file_size - scan_off) throw Error("corrupt BGZF member (no usable BC subfield) in " + path);
        const uint8_t *tr = h + bsize - 8;
        const uint32_t crc = tr[0] | tr[1] << 8 | tr[2] << 16 | (uint32_t)tr[3] << 24;
        const uint32_t isize = tr[4] | tr[5] << 8 | tr[6] << 16 | (uint32_t)tr[7] << 24;
        if (isize > slice) throw Error("BGZF member larger than a slice in " + path);
        if (out + isize > slice) break;
        pl.push_back(BlockRef{scan_off + 12 + xlen, (uint32_t)(bsize - 12 - xlen - 8), (uint32_t)out, isize, crc});
        out += isize;
        scan_off += bsize;
    }
    Slot &sl = slots[k % ring_n];
    sl.index = k; sl.ready = false; sl.pending = n_threads; sl.newlines = 0; sl.len = out; sl.eof = scan_off >= file_size;
    if (sl.eof) n_slices_known = k + 1;
    planned = k + 1;
}

void TextFeeder::run_bgzf() {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = "cannot initialise zlib"; }
        cv.notify_all();
        return;
    }
    try {
        for (;;) {
            size_t job, k;
            {
                std::unique_lock<std::mutex> lk(mu);
                job = next_job++;
                k = job / n_threads;
                // slices are planned in order (the member scan is sequential) by whichever of their jobs gets here first
                cv.wait(lk, [&] { return stop || failed || k >= n_slices_known || (k < released + ring_n && planned >= k); });
                if (stop || failed || k >= n_slices_known) break;
                if (planned == k) { plan_bgzf_slice(k); cv.notify_all(); }
            }
            const size_t j = job % n_threads;
            const std::vector<BlockRef> &pl = plans[k % ring_n];
            const size_t per = (pl.size() + n_threads - 1) / n_threads;
            const size_t lo = std::min(pl.size(), j * per), hi = std::min(pl.size(), lo + per);
            uint8_t *dst = bufs[k % ring_n] + HEAD;
            const double t0 = now_s();
            uint64_t nl = 0;
            for (size_t b = lo; b < hi; b++) {
                const BlockRef &r = pl[b];
                if (r.out_len == 0) continue;                          // e.g. the empty end-of-file member
                inflateReset(&zs);
                zs.next_in = (Bytef *)(map + r.in_off); zs.avail_in = r.in_len;
                zs.next_out = dst + r.out_off; zs.avail_out = r.out_len;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0 || crc32_fast(0u, dst + r.out_off, r.out_len) != r.crc)
                    throw Error("corrupt BGZF member in " + path);
                nl += count_newlines(dst + r.out_off, r.out_len);
            }
            const double dt = now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                Slot &sl = slots[k % ring_n];
                sl.newlines += nl;
                busy_s += dt;
                if (--sl.pending == 0) sl.ready = true;
            }
            cv.notify_all();
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
    inflateEnd(&zs);
}

// gzip streams: one inflating producer (a deflate stream is sequential), slices filled in order
void TextFeeder::run_gz() {
    try {
        for (size_t k = 0;; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || k < released + ring_n; });
                if (stop) return;
                Slot &sl = slots[k % ring_n];
                sl.index = k; sl.ready = false; sl.newlines = 0; sl.len = 0; sl.eof = false;
            }
            uint8_t *dst = bufs[k % ring_n] + HEAD;
            const double t0 = now_s();
            size_t have = 0; uint64_t nl = 0; bool eof = false;
            while (have < slice) {
                const int r = gzread(gz, dst + have, (unsigned)std::min<size_t>(slice - have, 1u << 20));
                if (r < 0) throw Error("read error in " + path);
                if (r == 0) { eof = true; break; }
                nl += count_newlines(dst + have, (size_t)r);
                have += (size_t)r;
            }
            const double dt = now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                Slot &sl = slots[k % ring_n];
                sl.len = have; sl.newlines = nl; sl.eof = eof; sl.ready = true;
                busy_s += dt;
            }
            cv.notify_all();
            if (eof) return;
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
}

// CRC-32 of gzip (reflected polynomial 0xEDB88320), folded 64 bytes at a time with carry-less multiplies (the published
// PCLMULQDQ method; the folding constants are x^k mod P for this polynomial); zlib's table-driven crc32() takes the bytes that are
// left and stands in on other CPUs.  The member CRCs of the parallel gzip reader are computed with it over every resolved byte.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(const uint8_t *buf, size_t len /* multiple of 16, >= 64 */, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len) {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (ok && len >= 64) {
        const size_t bulk = len & ~(size_t)15;
        crc = ~crc32_fold(buf, bulk, ~crc);
        buf += bulk; len -= bulk;
    }
#endif
    while (len) { const uInt m = (uInt)std::min<size_t>(len, 1u << 30); crc = (uint32_t)crc32(crc, buf, m); buf += m; len -= m; }
    return crc;
}

// symbols -> bytes: 32 symbols at a time when none of them is a marker (then it is a plain narrowing), through the table otherwise
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void resolve_symbols_avx2(const uint16_t *src, size_t m, const uint8_t *lut, uint8_t *dst) {
    size_t i = 0;
    for (; i + 32 <= m; i += 32) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 16));
        if (_mm256_testz_si256(_mm256_or_si256(a, b), _mm256_set1_epi16((short)0xFF00))) {
            const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8);
            _mm256_storeu_si256((__m256i *)(dst + i), pk);
        } else {
            for (size_t k = i; k < i + 32; k++) dst[k] = lut[src[k]];
        }
    }
    for (; i < m; i++) dst[i] = lut[src[i]];
}
#endif
void resolve_symbols(const uint16_t *src, size_t m, const uint8_t *lut, uint8_t *dst) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { resolve_symbols_avx2(src, m, lut, dst); return; }
#endif
    for (size_t i = 0; i < m; i++) dst[i] = lut[src[i]];
}

// parallel gzip: is slice k complete?  A slice is full (slice bytes) unless it is the stream's last one
bool TextFeeder::slice_ready_locked(size_t k, size_t &len, bool &eof) const {
    const Slot &sl = slots[k % ring_n];
    const size_t have = sl.index == k ? sl.len : 0;
    if (pgz_total_known) {
        const size_t last = (size_t)(pgz_total / slice);
        if (k > last) return false;
        if (k == last) { len = (size_t)(pgz_total % slice); eof = true; return have == len; }
    }
    len = slice; eof = false;
    return have == slice;
}

// One gzip stream, several threads (the decoder is sgh_inflate.cpp).  Every worker takes the next chunk of the COMPRESSED file,
// looks for a deflate block start inside it and decodes from there to the first block boundary behind the chunk's end — before
// anything in front of the chunk is known, so back-references into the preceding 32 KiB come out as markers.  The chunks are then
// stitched in order (the "chain", one short critical section per chunk): a chunk whose start is where the chunk before ended is
// accepted, its tail is resolved against the window handed over — which yields the window for the next chunk — and its body is
// resolved into the pinned slices outside the lock; a chunk whose speculation failed (no start found, a false start, a block
// that spans whole chunks) is decoded again from the true position with the window known: correct in every case, merely
// sequential for that chunk.  CRC-32 / ISIZE of every member are checked over the resolved bytes (acquire).
void TextFeeder::run_pgz() {
    try {
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || failed || pgz_next >= pgz_chunks || pgz_next < pgz_chain + 2 * n_threads; });
                if (stop || failed || pgz_next >= pgz_chunks) return;
                k = pgz_next++;
            }
            const double t0 = now_s();
            const size_t lo = k * pgz_chunk_bytes, hi = std::min(file_size, lo + pgz_chunk_bytes);
            const size_t max_out = (size_t)1 << 31;
            // the symbol buffer (2 bytes per byte of text) is kept from chunk to chunk: fresh pages for every chunk cost more than decoding
            static thread_local std::vector<uint16_t> sym_keep;
            InflateSpan span;
            span.sym.swap(sym_keep);
            span.sym.clear();
            struct Keep { InflateSpan &s; std::vector<uint16_t> &k; ~Keep() { s.sym.clear(); s.sym.swap(k); } } keep{span, sym_keep};
            auto reset_span = [&]() { span.sym.clear(); span.members.clear(); span.end_of_stream = false; span.start_bit = span.end_bit = 0; };
            bool found = false;
            uint64_t start = 0;
            if (k == 0) {
                // the first member's header, then its first block: nothing lies in front of it
                size_t at = 10;
                if (file_size < 18 || map[0] != 0x1f || map[1] != 0x8b || map[2] != 8) throw Error("not a gzip stream: " + path);
                const unsigned flg = map[3];
                if (flg & 4) at += 2 + (size_t)(map[at] + 256u * map[at + 1]);
                if (flg & 8) { while (at < file_size && map[at]) at++; at++; }
                if (flg & 16) { while (at < file_size && map[at]) at++; at++; }
                if (flg & 2) at += 2;
                if (at >= file_size) throw Error("corrupt gzip header in " + path);
                start = 8ull * at; found = true;
            } else {
                found = find_block_start(map, file_size, 8ull * lo, 8ull * hi, start);
            }
            if (found && inflate_span(map, file_size, start, 8ull * hi, nullptr, span, max_out) != 0) { found = false; reset_span(); }
            // ---- the chain: chunk k's true start, its window, its place in the output
            uint8_t window[32768];
            uint64_t out_off = 0;
            bool empty = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || failed || pgz_chain == k; });
                if (stop || failed) return;
                if (k == 0) pgz_pos = start;
                memcpy(window, pgz_window.data(), 32768);
                const uint64_t pos = pgz_pos;
                if (pgz_eos || (pos >= 8ull * hi && k + 1 < pgz_chunks)) {
                    empty = true;                                  // the stream ended, or a block that began earlier covers this whole chunk
                } else if (!(found && start == pos)) {
                    lk.unlock();
                    reset_span();
                    const int rc = inflate_span(map, file_size, pos, 8ull * hi, k == 0 ? nullptr : window, span, max_out);
                    if (rc != 0) throw Error(std::string(rc == -4 ? "trailing garbage behind the gzip stream in " : "corrupt gzip stream in ") + path);
                    lk.lock();
                    pgz_fallbacks++;
                }
                if (empty) reset_span();
                else {
                    // the window behind this chunk: the last 32 KiB of (window ++ resolved symbols)
                    const size_t n = span.sym.size();
                    uint8_t *nw = pgz_window.data();
                    if (n >= 32768) {
                        for (size_t i = 0; i < 32768; i++) { const uint16_t v = span.sym[n - 32768 + i]; nw[i] = v < 256 ? (uint8_t)v : window[v & 0x7FFFu]; }
                    } else {
                        memmove(nw, nw + n, 32768 - n);
                        for (size_t i = 0; i < n; i++) { const uint16_t v = span.sym[i]; nw[32768 - n + i] = v < 256 ? (uint8_t)v : window[v & 0x7FFFu]; }
                    }
                    out_off = pgz_out;
                    pgz_out += n;
                    pgz_pos = span.end_bit;
                    if (span.end_of_stream) { pgz_eos = true; pgz_total = pgz_out; pgz_total_known = true; }
                }
                if (k + 1 == pgz_chunks && !pgz_eos) throw Error("truncated gzip stream in " + path);
                pgz_chain = k + 1;
            }
            cv.notify_all();
            // ---- the body: symbols -> bytes, into the slices that hold [out_off, out_off + n)
            const size_t n = span.sym.size();
            // symbol -> byte: identity below 256, the window behind a marker
            static thread_local std::vector<uint8_t> lut_keep;
            if (lut_keep.size() != 65536) { lut_keep.assign(65536, 0); for (unsigned i = 0; i < 256; i++) lut_keep[i] = (uint8_t)i; }
            uint8_t *const lut = lut_keep.data();
            memcpy(lut + 0x8000, window, 32768);
            std::vector<PgzPiece> pieces;
            size_t done = 0, mi = 0;
            uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
            uint64_t piece_len = 0;
            while (done < n || mi < span.members.size()) {
                // member ends at the current position close a CRC piece
                while (mi < span.members.size() && span.members[mi].at == done) {
                    pieces.push_back(PgzPiece{crc, piece_len, true, span.members[mi].crc, span.members[mi].isize});
                    crc = (uint32_t)crc32(0L, Z_NULL, 0); piece_len = 0; mi++;
                }
                if (done >= n) break;
                const uint64_t g = out_off + done;
                const size_t s_idx = (size_t)(g / slice), s_off = (size_t)(g % slice);
                size_t m = std::min(n - done, slice - s_off);
                if (mi < span.members.size()) m = std::min(m, span.members[mi].at - done);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || failed || s_idx < released + ring_n; });
                    if (stop || failed) return;
                    Slot &sl = slots[s_idx % ring_n];
                    if (sl.index != s_idx) { sl.index = s_idx; sl.ready = false; sl.pending = 0; sl.newlines = 0; sl.len = 0; sl.eof = false; }
                }
                uint8_t *dst = bufs[s_idx % ring_n] + HEAD + s_off;
                const uint16_t *src = span.sym.data() + done;
                resolve_symbols(src, m, lut, dst);
                crc = crc32_fast(crc, dst, m);
                piece_len += m;
                const uint64_t nl = count_newlines(dst, m);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    Slot &sl = slots[s_idx % ring_n];
                    sl.len += m; sl.newlines += nl;
                }
                cv.notify_all();
                done += m;
            }
            if (piece_len) pieces.push_back(PgzPiece{crc, piece_len, false, 0, 0});
            {
                std::lock_guard<std::mutex> lk(mu);
                pgz_pieces[k] = std::move(pieces);
                pgz_bodies++;
                busy_s += now_s() - t0;
            }
            cv.notify_all();
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
}

bool TextFeeder::acquire(size_t k, uint8_t *&data, size_t &len, uint64_t &newlines, bool &eof) {
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(mu);
    if (is_pgz) {
        size_t l = 0; bool e = false;
        // (the last slice also waits for every chunk to have filed its CRC pieces)
        cv.wait(lk, [&] { return failed || (slice_ready_locked(k, l, e) && (!e || pgz_bodies == pgz_chunks)); });
        wait_s += now_s() - t0;
        if (failed) throw Error(error);
        if (e && !pgz_verified) {
            // every byte of the stream is in place: the CRC-32 and ISIZE of every gzip member, piece by piece
            pgz_verified = true;
            uint32_t crc = 0; uint64_t mlen = 0;
            for (const auto &pcs : pgz_pieces)
                for (const PgzPiece &pc : pcs) {
                    crc = (uint32_t)crc32_combine(crc, pc.crc, (z_off_t)pc.len);
                    mlen += pc.len;
                    if (pc.member_end) {
                        if (crc != pc.want_crc || (uint32_t)mlen != pc.want_isize) throw Error("corrupt gzip stream (CRC / length mismatch) in " + path);
                        crc = 0; mlen = 0;
                    }
                }
        }
        const Slot &sl = slots[k % ring_n];
        data = bufs[k % ring_n] + HEAD; len = l; newlines = (sl.index == k) ? sl.newlines : 0; eof = e;
        return true;
    }
    cv.wait(lk, [&] { return failed || (slots[k % ring_n].index == k && slots[k % ring_n].ready); });
    wait_s += now_s() - t0;
    if (failed) throw Error(error);
    const Slot &sl = slots[k % ring_n];
    data = bufs[k % ring_n] + HEAD; len = sl.len; newlines = sl.newlines; eof = sl.eof;
    return true;
}

void TextFeeder::release_below(size_t k) {
    {
        std::lock_guard<std::mutex> lk(mu);
        if (k > released) released = k;
    }
    cv.notify_all();
}

// Streams one FASTQ file through sgc_sample_push_fastq_part.  Returns false (nothing pushed) if the input is not
// FASTQ (first byte not '@'): the caller then uses the record reader.
static bool count_fastq_text(sgc_sample *smp, const std::string &path, const CountOptions &opt, SampleStats *st) {
    const double t_feed0 = now_s();
    TextFeeder feed(path, std::max<size_t>(opt.chunk_bytes, 1u << 16), 3, opt.io_threads, sgc_alloc_pinned, sgc_free_pinned,
                    opt.inflate_threads);
    if (feed.first_byte != '@') return false;
    const double t_feed1 = now_s();
    double t_first_push = -1;
    uint64_t first_line = 0;
    size_t carry = 0;                                  // bytes of an unfinished line, already sitting in front of the slice
    double t_push = 0, t_upwait = 0;
    for (size_t k = 0;; k++) {
        uint8_t *data; size_t len; uint64_t newlines; bool eof;
        feed.acquire(k, data, len, newlines, eof);
        uint8_t *part = data - carry;
        size_t part_len = carry + len, tail = 0;
        if (!eof) {
            // cut at the last newline of the slice; what follows is carried in front of the next slice
            const void *nl = len ? memrchr(data, '\n', len) : nullptr;
            const size_t keep = nl ? (size_t)((const uint8_t *)nl - part) + 1 : 0;
            tail = part_len - keep;
            if (tail > TextFeeder::HEAD) {
                // the carry area in front of a slice holds one unfinished line of up to HEAD bytes; with nothing pushed yet the
                // record reader (no line limit, like fxread) can still take the whole sample
                if (first_line == 0) return false;
                throw Error("FASTQ line longer than " + std::to_string(TextFeeder::HEAD) + " bytes in " + path + " (--pack device reads lines of any length)");
            }
            memcpy(feed.buffer_of(k + 1) + TextFeeder::HEAD - tail, part + keep, tail);
            part_len = keep;
        } else {
            // trailing blank lines at the very end of the file are not records
            while (part_len >= 2 && part[part_len - 1] == '\n' && part[part_len - 2] == '\n') { part_len--; newlines--; }
        }
        if (part_len) {
            double t0 = now_s();
            sgc_check(sgc_sample_push_fastq_part(smp, part, part_len, SGC_MEM_HOST, first_line, newlines, nullptr),
                      "sgc_sample_push_fastq_part");
            t_push += now_s() - t0;
            if (t_first_push < 0) t_first_push = now_s() - t0;
            first_line += newlines + (part[part_len - 1] != '\n' ? 1 : 0);
            // the buffer of slice k - 1 may be refilled once its upload is through (this part's may still be in flight)
            t0 = now_s();
            sgc_check(sgc_sample_wait_uploads(smp, 1), "sgc_sample_wait_uploads");
            t_upwait += now_s() - t0;
            feed.release_below(k);
        } else {
            sgc_check(sgc_sample_wait_uploads(smp, 0), "sgc_sample_wait_uploads");
            feed.release_below(k);
        }
        carry = tail;
        if (eof) break;
    }
    // (3 mod 4: the stream ends behind a separator line — the last record's quality line is empty; reader decision #3)
    if (first_line % 4 != 0 && first_line % 4 != 3) throw Panic("truncated FASTQ record in " + path);
    if (st) {
        st->text_bytes = feed.is_gz ? 0 : feed.file_size;
        st->reader_threads = feed.n_threads;
        st->read_busy_s = feed.busy_s; st->read_wait_s = feed.wait_s; st->push_s = t_push; st->upload_wait_s = t_upwait;
        st->gz = feed.is_gz; st->bgzf = feed.is_bgzf; st->pgz = feed.is_pgz; st->pgz_fallbacks = feed.pgz_fallbacks;
        st->feeder_setup_s = t_feed1 - t_feed0; st->first_push_s = t_first_push < 0 ? 0 : t_first_push;
    }
    return true;
}

// The host-scan path: the scanner's records, block by block, through a small ring of pinned buffers into
// sgc_sample_push_packed_async (which batches them on the device: one count pass per 2^24 records).
static const size_t SCAN_BUF_RECORDS = 1u << 20;        // records per pinned buffer (8 MB of one-word records)
static const size_t SCAN_BUFS = 4;
static void count_fastq_scan(sgc_sample *smp, FastqScanner &scan, SampleStats *st) {
    const size_t words = scan.words, buf_bytes = SCAN_BUF_RECORDS * words * 8;
    struct Ring {
        uint8_t *b[SCAN_BUFS] = {};
        ~Ring() { for (auto p : b) sgc_free_pinned(p); }
    } ring;
    const double t_a0 = now_s();
    for (size_t i = 0; i < SCAN_BUFS; i++) {
        ring.b[i] = (uint8_t *)sgc_alloc_pinned(buf_bytes);
        if (!ring.b[i]) throw Error("cannot allocate pinned host buffers");
    }
    const double t_a1 = now_s();
    double t_push = 0, t_upwait = 0, t_copy = 0;
    size_t k = 0, fill = 0;                              // current buffer, records in it
    auto push = [&]() {
        if (!fill) return;
        double t0 = now_s();
        sgc_check(sgc_sample_push_packed_async(smp, ring.b[k % SCAN_BUFS], fill), "sgc_sample_push_packed_async");
        t_push += now_s() - t0;
        k++; fill = 0;
        // the buffer about to be refilled was pushed SCAN_BUFS pushes ago (a push is one upload: a batch holds a whole number of buffers)
        t0 = now_s();
        sgc_check(sgc_sample_wait_uploads(smp, (uint32_t)(SCAN_BUFS - 1)), "sgc_sample_wait_uploads");
        t_upwait += now_s() - t0;
    };
    // hybrid library: the reads the scanner set aside, gathered and pushed as bytes now and then (the byte-string chain)
    std::vector<uint8_t> rbytes;
    std::vector<uint64_t> roffs(1, 0);
    auto push_routed = [&]() {
        const uint64_t nr = roffs.size() - 1;
        if (!nr) return;
        const double t0 = now_s();
        sgc_check(sgc_sample_push_windows(smp, rbytes.data(), roffs.data(), nr, SGC_MEM_HOST, scan.window_offset()), "sgc_sample_push_windows");
        sgc_check(sgc_sample_sync(smp), "sgc_sample_sync");            // the vectors are reused
        t_push += now_s() - t0;
        rbytes.clear(); roffs.resize(1);
    };
    const uint64_t *recs; size_t n;
    while (scan.next(recs, n)) {
        if (scan.n_routed) {
            const uint64_t base = rbytes.size(), add = scan.routed_offs[scan.n_routed];
            rbytes.insert(rbytes.end(), scan.routed_bytes, scan.routed_bytes + add);
            for (size_t i = 1; i <= scan.n_routed; i++) roffs.push_back(base + scan.routed_offs[i]);
            if (roffs.size() > (1u << 16)) push_routed();
        }
        const double t0 = now_s();
        size_t done = 0;
        while (done < n) {
            const size_t m = std::min(n - done, SCAN_BUF_RECORDS - fill);
            memcpy(ring.b[k % SCAN_BUFS] + fill * words * 8, recs + done * words, m * words * 8);
            fill += m; done += m;
            if (fill == SCAN_BUF_RECORDS) { t_copy += now_s() - t0; push(); t_copy -= now_s() - t0; }
        }
        t_copy += now_s() - t0;
        scan.release();
    }
    push();
    push_routed();
    sgc_check(sgc_sample_wait_uploads(smp, 0), "sgc_sample_wait_uploads");         // the ring is freed on return
    if (st) {
        st->text_bytes = scan.gz_mode ? 0 : scan.file_size; st->reader_threads = scan.n_threads;
        st->gz = scan.gz_mode; st->bgzf = scan.bgzf_mode; st->pgz = scan.gz_mode && !scan.bgzf_mode; st->pgz_fallbacks = scan.pgz_fallbacks;
        st->read_busy_s = scan.busy_s; st->read_wait_s = scan.wait_s; st->push_s = t_push; st->upload_wait_s = t_upwait;
        st->feeder_setup_s = t_a1 - t_a0; st->host_copy_s = t_copy; st->scan_path = true; st->scan_mapped = scan.used_mapping();
    }
}

static SampleCounts count_sample(sgc_ctx *ctx, const std::string &path, const Offset &off, const Library &library,
                                 const CountOptions &opt, SampleStats *st, std::unique_ptr<FastqScanner> scan) {
    const double t_begin = now_s();
    sgc_sample *smp = nullptr;
    sgc_check(sgc_sample_begin(ctx, &smp, off.reverse, (uint32_t)off.index, opt.position_recursion), "sgc_sample_begin");
    struct Guard { sgc_sample *s; ~Guard() { sgc_sample_free(s); } } guard{smp};
    bool scanned = false;
    if (scan && scan->usable) {
        sgc_lib_info info;
        sgc_check(sgc_library_info(ctx, &info), "sgc_library_info");
        // (a hybrid ctx — path 2 — takes the scanner's records because the scanner routes the reads that need the byte-string chain)
        if (info.record_bytes == scan->words * 8 || (info.path == 2 && scan->routes())) { count_fastq_scan(smp, *scan, st); scanned = true; }
    }
    scan.reset();
    const bool parsed_on_device = !scanned && opt.device_parse && opt.device_pack && count_fastq_text(smp, path, opt, st);
    if (st) st->text_path = parsed_on_device;
    if (!parsed_on_device && !scanned) {
    FastxReader reader(path);                                                         // count.rs:24
    const uint32_t L = (uint32_t)library.size;
    sgc_lib_info info;
    sgc_check(sgc_library_info(ctx, &info), "sgc_library_info");
    // a library of arbitrary bytes or longer than 30 has no packed record format: the device works on the read bytes
    const bool device_pack = opt.device_pack || info.record_bytes == 0;
    const size_t words = info.record_bytes / 8;
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offsets(1, 0), recs;
    bytes.reserve(opt.batch_reads * 160);
    auto flush = [&]() {
        const uint64_t n = offsets.size() - 1;
        if (!n) return;
        if (device_pack) {
            sgc_check(sgc_sample_push_reads(smp, bytes.data(), offsets.data(), n, SGC_MEM_HOST), "sgc_sample_push_reads");
        } else {
            recs.resize(n * words);
            sgc_check(sgc_pack_reads_host(bytes.data(), offsets.data(), n, L, off.reverse, (uint32_t)off.index,
                                          opt.position_recursion, recs.data()), "sgc_pack_reads_host");
            sgc_check(sgc_sample_push_packed(smp, recs.data(), n, SGC_MEM_HOST), "sgc_sample_push_packed");
        }
        sgc_check(sgc_sample_sync(smp), "sgc_sample_sync");
        bytes.clear(); offsets.resize(1);
    };
    RecordView r;
    while (reader.next(r)) {
        bytes.insert(bytes.end(), (const uint8_t *)r.seq, (const uint8_t *)r.seq + r.seq_len);
        offsets.push_back(bytes.size());
        if (offsets.size() > opt.batch_reads) flush();
    }
    flush();
    }
    std::vector<uint64_t> counts(library.seqs.size());
    SampleCounts out;
    const double t_fin = now_s();
    const int frc = sgc_sample_finish(smp, counts.data(), &out.total_reads, &out.matched_reads);
    if (frc == SGC_E_FORMAT) throw Panic(std::string(sgc_last_error()) + " in " + path);     // fxread panics on a malformed record
    sgc_check(frc, "sgc_sample_finish");
    if (st) {
        st->finish_s = now_s() - t_fin;
        st->reads = out.total_reads;
        sgc_timing tm;
        if (sgc_timing_read(ctx, &tm, 1) == SGC_OK) {
            st->h2d_ms = tm.h2d_ms; st->ingest_ms = tm.pack_ms;
            st->count_ms = tm.part_ms + tm.lookup_ms + tm.miss_ms + tm.hist_ms;
        }
        st->wall_s = now_s() - t_begin;
    }
    for (size_t i = 0; i < counts.size(); i++)                                         // id-keyed fold, counter.rs:232-235
        if (counts[i]) out.by_id[library.ids[i]] += counts[i];
    return out;
}

void count(const CountOptions &opt_in) {
    CountOptions opt = opt_in;
    const double t_start = now_s(), t_start_unix = unix_s();
    // The HIP runtime takes 0.06-0.14 s to come up, whoever calls it first: it does so on a thread of its own from the very start,
    // beside the library load, the checks of the inputs and the first blocks of the scanners (nothing of it is observable before
    // sgc_init reports a missing device, below, where the reference would have started counting).
    std::future<int> n_dev_f = std::async(std::launch::async, [] { return sgc_device_count(); });
    const Library library = Library::from_path(opt.library_path);                      // count.rs:87
    if (opt.genemap) {                                                                 // count.rs:90-95
        if (const std::string *missing = opt.genemap->missing_alias(library))
            throw Error("Missing sgRNA aliases in gene map: \"" + *missing + "\"");
    }
    for (const auto &path : opt.input_paths) {                                         // count.rs:62-71, 98-100
        FastxReader rd(path);
        RecordView r;
        if (!rd.next(r)) throw Panic("called `Option::unwrap()` on a `None` value");
        if (library.size > r.seq_len)
            throw Error("Sequences in reference library are larger than the sequences in input.\n\nConsider reducing the length "
                        "of your reference sequences (i.e. extracting the variable region of the sgRNA or reducing the length of "
                        "the adapters.)");
    }
    const double t_lib = now_s();
    // device tables: Library + (unless exact) Permuter, count.rs:103-107.  One context per worker thread (its own
    // stream, scratch and table copy — ~0.15 GB at 100k guides), dealt round-robin over the visible GPUs: with -t N
    // the samples of one GPU overlap too (one sample's inflate and parse run beside another's kernels), which is
    // what the reference's rayon pool over samples gives on CPU cores.
    const size_t n_workers = std::max<size_t>(1, std::min(opt.threads, opt.input_paths.size()));
    const size_t hw = usable_cpus();
    if (opt.io_threads == 0) {
        opt.io_threads = std::max<size_t>(1, std::min<size_t>(8, hw / n_workers));
        opt.inflate_threads = std::max<size_t>(1, std::min<size_t>(16, (hw > 1 ? hw - 1 : 1) / n_workers));      // gzip / BGZF, inflated in parallel
    }
    if (opt.scan_threads == 0) opt.scan_threads = std::max<size_t>(1, std::min<size_t>(16, (hw > 1 ? hw - 1 : 1) / n_workers));
    // Host scan (default for plain FASTQ text when the library has a packed record format: ACGT guides, L <= 30): the
    // scanners of the first samples start NOW, so that reading and packing the text overlap device start-up and the table
    // build below — neither needs the other (count.rs:103-136 builds the Permuter, then fans the samples out; the order of
    // the observable effects stays: a library or gene-map error was raised above, a malformed sample is reported when its
    // turn comes).
    bool packable = opt.host_scan && opt.device_pack && library.size >= 1 && library.size <= SGC_MAX_GUIDE_LEN;
    size_t n_other = 0;                       // guides with a byte outside ACGT
    for (size_t i = 0; packable && i < library.seqs.size(); i++)
        for (char ch : library.seqs[i]) if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') { n_other++; break; }
    // a few such guides: the library becomes a hybrid ctx (sgc_set_library decides by the same rule) and the scanner routes the reads
    // they could influence to the byte-string chain; more than half: the byte-string path alone, text parsed on the GPU
    RouteFilter route_filter;
    const bool hybrid = packable && n_other > 0 && n_other * 2 <= library.seqs.size() && n_other < library.seqs.size();
    if (n_other && !hybrid) packable = false;
    if (hybrid) route_filter = make_route_filter(library.seqs);
    std::vector<std::unique_ptr<FastqScanner>> scanners(opt.input_paths.size());
    auto make_scanner = [&](size_t i) {
        ScanParams sp;
        sp.L = (uint32_t)library.size; sp.reverse = opt.offsets[i].reverse; sp.offset = (uint32_t)opt.offsets[i].index;
        sp.recursion = opt.position_recursion;
        sp.route = hybrid ? &route_filter : nullptr;
        return std::unique_ptr<FastqScanner>(new FastqScanner(opt.input_paths[i], sp, opt.scan_threads, opt.scan_block_bytes, 16384, opt.scan_source));
    };
    if (packable)
        for (size_t i = 0; i < std::min(n_workers, opt.input_paths.size()); i++)
            if (opt.offsets[i].index <= 0xFFFFFFFFull) scanners[i] = make_scanner(i);
    const double t_before_runtime = now_s();                      // (what lies between t_lib and here is the scanners' start, not the runtime)
    int n_dev = n_dev_f.get();
    if (n_dev < 1) n_dev = 1;                                     // sgc_init below reports the missing device
    if (opt.max_devices > 0) n_dev = std::min(n_dev, (int)opt.max_devices);
    const size_t n_ctx = std::min(opt.input_paths.size(), std::max<size_t>((size_t)n_dev, n_workers));
    const double t_runtime = now_s();
    std::string flat;
    flat.reserve(library.seqs.size() * library.size);
    for (const auto &s : library.seqs) flat += s;
    std::vector<sgc_ctx *> ctxs;
    struct CtxGuard { std::vector<sgc_ctx *> &v; ~CtxGuard() { for (auto c : v) if (c) sgc_free(c); } } cg{ctxs};
    std::vector<int> ctx_dev;
    double init_s = 0;
    // Library + Permuter are built ONCE per device (count.rs:103-107 builds them once and lends them to every rayon worker):
    // one thread per device builds its tables — the devices in parallel —, and the other contexts of a device (worker threads
    // beyond the number of GPUs) are clones that share those tables (sgc_ctx_clone).
    const size_t n_primary = std::min(n_ctx, (size_t)n_dev);
    ctxs.assign(n_ctx, nullptr);
    for (size_t k = 0; k < n_ctx; k++) ctx_dev.push_back((int)(k % (size_t)n_dev));
    std::vector<double> dev_init_s(n_primary, 0.0), dev_build_s(n_primary, 0.0);
    std::vector<std::string> build_err(n_primary);
    std::vector<int> build_rc(n_primary, SGC_OK);
    if (!opt.quiet && !opt.exact) fprintf(stderr, "Generating Mismatch Library\n");
    auto build_on = [&](size_t d) {
        const double t0 = now_s();
        sgc_ctx *c = nullptr;
        int rc = sgc_init((int)d, &c);
        dev_init_s[d] = now_s() - t0;
        if (rc == SGC_OK) {
            ctxs[d] = c;
            rc = sgc_set_library(c, (const uint8_t *)flat.data(), (uint32_t)library.seqs.size(), (uint32_t)library.size, !opt.exact);
        }
        if (rc != SGC_OK) { build_rc[d] = rc; build_err[d] = std::string(ctxs[d] ? "sgc_set_library: " : "sgc_init: ") + sgc_last_error(); }
        dev_build_s[d] = now_s() - t0 - dev_init_s[d];
    };
    {
        std::vector<std::thread> builders;
        for (size_t d = 1; d < n_primary; d++) builders.emplace_back(build_on, d);
        build_on(0);
        for (auto &t : builders) t.join();
    }
    for (size_t d = 0; d < n_primary; d++) {
        init_s += dev_init_s[d];
        if (build_rc[d] == SGC_E_DUPLICATE) throw Panic(build_err[d]);
        if (build_rc[d] != SGC_OK) throw Error(build_err[d]);
    }
    for (size_t k = n_primary; k < n_ctx; k++) sgc_check(sgc_ctx_clone(ctxs[k % (size_t)n_dev], &ctxs[k]), "sgc_ctx_clone");
    if (!opt.quiet && !opt.exact) fprintf(stderr, "Finished Mismatch Library\n");
    for (sgc_ctx *c : ctxs) {
        if (!opt.stats_path.empty()) sgc_check(sgc_timing_enable(c, 1), "sgc_timing_enable");
        sgc_check(sgc_set_option(c, "batch_records", (int64_t)(16 * SCAN_BUF_RECORDS)), "sgc_set_option");     // a whole number of pinned buffers
        if (hybrid) sgc_check(sgc_set_option(c, "host_routes", 1), "sgc_set_option");      // this host sets the reads near the non-ACGT guides aside itself
    }
    const double t_tables = now_s();
    // samples in parallel (count.rs:117-136: rayon over samples, pool size -t), results in input order
    const size_t n = opt.input_paths.size();
    std::vector<SampleCounts> results(n);
    std::vector<SampleStats> stats(n);
    std::vector<int> sample_dev(n, -1);
    std::vector<std::string> errors(n);
    std::vector<int> kinds(n, 0);
    std::atomic<size_t> next{0};
    std::vector<std::mutex> dev_mu(ctxs.size());
    auto worker = [&](size_t w) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            // a worker keeps its own context; with fewer workers than contexts (GPUs) the samples are dealt over all of them
            const size_t d = n_workers >= ctxs.size() ? w % ctxs.size() : i % ctxs.size();
            try {
                if (!opt.quiet) fprintf(stderr, "Processing: %s\n", opt.sample_names[i].c_str());
                std::lock_guard<std::mutex> lk(dev_mu[d]);      // one sample at a time per context
                sample_dev[i] = ctx_dev[d];
                std::unique_ptr<FastqScanner> sc = std::move(scanners[i]);
                if (!sc && packable && opt.offsets[i].index <= 0xFFFFFFFFull) sc = make_scanner(i);
                results[i] = count_sample(ctxs[d], opt.input_paths[i], opt.offsets[i], library, opt,
                                          opt.stats_path.empty() ? nullptr : &stats[i], std::move(sc));
                if (!opt.quiet)                                                           // count.rs:34-43
                    fprintf(stderr, "Finished: %s; Fraction mapped: %.3f [%llu / %llu]\n", opt.sample_names[i].c_str(),
                            (double)results[i].matched_reads / (double)results[i].total_reads,
                            (unsigned long long)results[i].matched_reads, (unsigned long long)results[i].total_reads);
            } catch (const Panic &e) { kinds[i] = 2; errors[i] = e.what(); }
            catch (const std::exception &e) { kinds[i] = 1; errors[i] = e.what(); }
        }
    };
    const size_t n_threads = std::max<size_t>(1, std::min(opt.threads, n));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < n; i++) {                            // first failing sample aborts the run (count.rs:136-140)
        if (kinds[i] == 2) throw Panic(errors[i]);
        if (kinds[i] == 1) throw Error(errors[i]);
    }
    const double t_counted = now_s();
    write_results(opt.output_path, results, library, opt.sample_names, opt.genemap, opt.include_zero);
    const double t_end = now_s();
    scanners.clear();
    for (auto c : ctxs) sgc_free(c);            // (the guard above covers the error paths)
    ctxs.clear();
    const double t_freed = now_s();
    if (!opt.stats_path.empty()) {
        FILE *f = fopen(opt.stats_path.c_str(), "wb");
        if (!f) throw Error("cannot create the stats file: " + opt.stats_path);
        std::string per_dev;
        for (size_t d = 0; d < dev_build_s.size(); d++) { char b[48]; snprintf(b, sizeof(b), "%s%.6f", d ? ", " : "", dev_build_s[d]); per_dev += b; }
        // setup_s: everything between the library and the first sample (scanners started, HIP runtime up, contexts, tables).  Its
        // parts: scanner_start_s (the route filter of a hybrid library, the first scanners — for a .gz their constructor returns when
        // the first chunk is inflated), hip_runtime_wait_s (the runtime comes up on a thread of its own from the start: what the main
        // thread still waited for), then the devices side by side: device_init_s / table_build_s are those of the slowest device
        // (init + build), table_build_per_device_s each device's build, contexts_ready_s all of it (the clones of the worker threads included)
        size_t slowest = 0;
        for (size_t d = 1; d < dev_build_s.size(); d++)
            if (dev_init_s[d] + dev_build_s[d] > dev_init_s[slowest] + dev_build_s[slowest]) slowest = d;
        const double slow_init = dev_init_s.empty() ? 0.0 : dev_init_s[slowest], slow_build = dev_build_s.empty() ? 0.0 : dev_build_s[slowest];
        fprintf(f, "{\"library_load_s\": %.6f, \"setup_s\": %.6f, \"scanner_start_s\": %.6f, \"hip_runtime_wait_s\": %.6f, \"contexts_ready_s\": %.6f, \"table_build_s\": %.6f, \"samples_s\": %.6f, \"table_write_s\": %.6f, \"total_s\": %.6f, "
                   "\"device_init_s\": %.6f, \"context_free_s\": %.6f, \"count_entered_unix_s\": %.6f, \"stats_written_unix_s\": %.6f, "
                   "\"devices\": %d, \"contexts\": %zu, \"worker_threads\": %zu, \"table_build_per_device_s\": [%s], \"samples\": [",
                t_lib - t_start, t_tables - t_lib, t_before_runtime - t_lib, t_runtime - t_before_runtime, t_tables - t_runtime, slow_build, t_counted - t_tables, t_end - t_counted, t_end - t_start, slow_init, t_freed - t_end, t_start_unix, unix_s(),
                n_dev, n_ctx, n_threads, per_dev.c_str());
        for (size_t i = 0; i < n; i++) {
            const SampleStats &x = stats[i];
            fprintf(f, "%s{\"device\": %d, \"reads\": %llu, \"text_bytes\": %llu, \"gz\": %s, \"bgzf\": %s, \"parallel_gzip\": %s, \"gzip_chunks_decoded_in_order\": %zu, \"text_path\": %s, \"scan_path\": %s, \"scan_used_mapping\": %s, \"host_copy_s\": %.6f, \"reader_threads\": %zu, "
                       "\"wall_s\": %.6f, \"read_busy_s\": %.6f, \"wait_for_text_s\": %.6f, \"push_s\": %.6f, \"wait_for_upload_s\": %.6f, "
                       "\"finish_s\": %.6f, \"feeder_setup_s\": %.6f, \"first_push_s\": %.6f, \"h2d_ms\": %.3f, \"ingest_kernels_ms\": %.3f, \"count_kernels_ms\": %.3f}",
                    i ? ", " : "", sample_dev[i], (unsigned long long)x.reads, (unsigned long long)x.text_bytes, x.gz ? "true" : "false", x.bgzf ? "true" : "false", x.pgz ? "true" : "false", x.pgz_fallbacks,
                    x.text_path ? "true" : "false", x.scan_path ? "true" : "false", x.scan_mapped ? "true" : "false", x.host_copy_s, x.reader_threads, x.wall_s, x.read_busy_s, x.read_wait_s, x.push_s,
                    x.upload_wait_s, x.finish_s, x.feeder_setup_s, x.first_push_s, x.h2d_ms, x.ingest_ms, x.count_ms);
        }
        fprintf(f, "]}\n");
        fclose(f);
    }
}

}  // namespace sgh
