// sgh.cpp — C++ host side (see sgh.hpp).  Everything here is plain host code; reads are matched and counted on
// the GPU through the C ABI of include/sgcount_hip.h (count()).
#include "sgh.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_set>

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
    if (l[0] == 0 && s.pos >= s.end && s.eof) return false;     // trailing blank line
    if (s.fmt == 0) {
        if (l[0] && s.buf[o[0]] == '>') s.fmt = 1;
        else if (l[0] && s.buf[o[0]] == '@') s.fmt = 2;
        else throw Error("not a FASTA/FASTQ file: " + s.path);
    }
    if (l[0] == 0 || s.buf[o[0]] != (s.fmt == 1 ? '>' : '@')) throw Panic("malformed FASTX header in " + s.path);
    if (!get(1)) throw Panic("truncated FASTX record in " + s.path);
    if (s.fmt == 2) {
        if (!get(2) || l[2] == 0 || s.buf[o[2]] != '+') throw Panic("malformed FASTQ record in " + s.path);
        if (!get(3)) throw Panic("truncated FASTQ record in " + s.path);
    }
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

// FASTQ at speed (replaces fxread for the count loop): the host only inflates/reads text and cuts it at record
// boundaries by counting newlines; record boundaries inside a chunk, window extraction and packing happen on
// the GPU (sgc_sample_push_fastq).  Two pinned buffers alternate so that reading chunk k+1 overlaps counting k.
// Number of '\n' in [p, p + n): byte compares summed in 8-bit lanes (the compiler turns the inner loop into vector
// compares; 255 iterations cannot overflow a lane), flushed to a wide sum.
static size_t count_newlines(const uint8_t *p, size_t n) {
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

// Byte source of the text path: a plain file is read with read(2) straight into the caller's (pinned) buffer; a
// gzip stream (magic 1f 8b) goes through zlib.
struct TextSource {
    FILE *fp = nullptr;
    gzFile gz = nullptr;
    explicit TextSource(const std::string &path) {
        fp = fopen(path.c_str(), "rb");
        if (!fp) throw Error("No such file or directory (os error 2): " + path);
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, fp);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            fclose(fp); fp = nullptr;
            gz = gzopen(path.c_str(), "rb");
            if (!gz) throw Error("No such file or directory (os error 2): " + path);
            gzbuffer(gz, 1u << 20);
        } else {
            rewind(fp);
            setvbuf(fp, nullptr, _IONBF, 0);             // unbuffered: fread becomes read(2) into our buffer
        }
    }
    ~TextSource() { if (fp) fclose(fp); if (gz) gzclose(gz); }
    // up to n bytes; 0 = end of stream
    size_t read(uint8_t *dst, size_t n, const std::string &path) {
        if (gz) {
            const int got = gzread(gz, dst, (unsigned)std::min<size_t>(n, 1u << 30));
            if (got < 0) throw Error("read error in " + path);
            return (size_t)got;
        }
        const size_t got = fread(dst, 1, n, fp);
        if (got == 0 && ferror(fp)) throw Error("read error in " + path);
        return got;
    }
};

// Where to cut a chunk of FASTQ text so that it holds whole 4-line records: count the newlines, then step back over
// the (count mod 4) complete lines — and the unterminated one — that belong to the record the chunk ends inside.
// The final chunk (eof) is taken whole; its last line may lack the newline.
size_t fastq_chunk_cut(const uint8_t *buf, size_t have, bool eof, const std::string &path) {
    const size_t lines = count_newlines(buf, have);
    if (eof) {
        const size_t tail_lines = lines + (have && buf[have - 1] != '\n' ? 1 : 0);
        if (tail_lines % 4 != 0) throw Panic("truncated FASTQ record in " + path);
        return have;
    }
    size_t back = lines & 3, end = have;
    for (;;) {
        const void *nl = end ? memrchr(buf, '\n', end) : nullptr;
        if (!nl) { end = 0; break; }
        end = (size_t)((const uint8_t *)nl - buf);           // index of that newline
        if (back == 0) { end += 1; break; }
        back--;
    }
    if (end == 0) throw Error("FASTQ record larger than the text chunk in " + path);
    return end;
}

static bool count_fastq_text(sgc_sample *smp, const std::string &path, const CountOptions &opt) {
    TextSource src(path);
    const size_t cap = std::max<size_t>(opt.chunk_bytes, 1u << 16);
    uint8_t *buf[2] = {(uint8_t *)sgc_alloc_pinned(cap), (uint8_t *)sgc_alloc_pinned(cap)};
    struct BGuard { uint8_t **b; ~BGuard() { sgc_free_pinned(b[0]); sgc_free_pinned(b[1]); } } bg{buf};
    if (!buf[0] || !buf[1]) throw Error("cannot allocate pinned host buffers");
    size_t have = 0;                                     // bytes carried over (an incomplete record)
    int cur = 0;
    bool eof = false, first_chunk = true;
    while (!eof || have) {
        while (have < cap && !eof) {
            const size_t got = src.read(buf[cur] + have, cap - have, path);
            if (got == 0) eof = true;
            have += got;
        }
        if (first_chunk) {
            if (!have || buf[cur][0] != '@') return false;    // not FASTQ: the caller uses the record reader
            first_chunk = false;
        }
        const size_t cut = fastq_chunk_cut(buf[cur], have, eof, path);
        if (cut) {
            sgc_check(sgc_sample_sync(smp), "sgc_sample_sync");          // the other buffer's count pass may still be running
            sgc_check(sgc_sample_push_fastq(smp, buf[cur], cut, SGC_MEM_HOST, nullptr), "sgc_sample_push_fastq");
        }
        const size_t rest = have - cut;
        memcpy(buf[cur ^ 1], buf[cur] + cut, rest);
        have = rest;
        cur ^= 1;
        if (eof && !have) break;
    }
    return true;
}

static SampleCounts count_sample(sgc_ctx *ctx, const std::string &path, const Offset &off, const Library &library,
                                 const CountOptions &opt) {
    sgc_sample *smp = nullptr;
    sgc_check(sgc_sample_begin(ctx, &smp, off.reverse, (uint32_t)off.index, opt.position_recursion), "sgc_sample_begin");
    struct Guard { sgc_sample *s; ~Guard() { sgc_sample_free(s); } } guard{smp};
    const bool parsed_on_device = opt.device_parse && opt.device_pack && count_fastq_text(smp, path, opt);
    if (!parsed_on_device) {
    FastxReader reader(path);                                                         // count.rs:24
    const uint32_t L = (uint32_t)library.size;
    const size_t words = sgc_record_bytes(L) / 8;
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offsets(1, 0), recs;
    bytes.reserve(opt.batch_reads * 160);
    auto flush = [&]() {
        const uint64_t n = offsets.size() - 1;
        if (!n) return;
        if (opt.device_pack) {
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
    sgc_check(sgc_sample_finish(smp, counts.data(), &out.total_reads, &out.matched_reads), "sgc_sample_finish");
    for (size_t i = 0; i < counts.size(); i++)                                         // id-keyed fold, counter.rs:232-235
        if (counts[i]) out.by_id[library.ids[i]] += counts[i];
    return out;
}

void count(const CountOptions &opt) {
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
    // device tables: Library + (unless exact) Permuter, count.rs:103-107.  One context per worker thread (its own
    // stream, scratch and table copy — ~0.15 GB at 100k guides), dealt round-robin over the visible GPUs: with -t N
    // the samples of one GPU overlap too (one sample's inflate and parse run beside another's kernels), which is
    // what the reference's rayon pool over samples gives on CPU cores.
    int n_dev = sgc_device_count();
    if (n_dev < 1) n_dev = 1;                                     // sgc_init below reports the missing device
    if (const char *v = getenv("SGCOUNT_DEVICES")) n_dev = std::max(1, std::min(n_dev, atoi(v)));
    const size_t n_workers = std::max<size_t>(1, std::min(opt.threads, opt.input_paths.size()));
    const size_t n_ctx = std::min(opt.input_paths.size(), std::max<size_t>((size_t)n_dev, n_workers));
    std::string flat;
    flat.reserve(library.seqs.size() * library.size);
    for (const auto &s : library.seqs) flat += s;
    std::vector<sgc_ctx *> ctxs;
    struct CtxGuard { std::vector<sgc_ctx *> &v; ~CtxGuard() { for (auto c : v) sgc_free(c); } } cg{ctxs};
    for (size_t k = 0; k < n_ctx; k++) {
        sgc_ctx *c = nullptr;
        sgc_check(sgc_init((int)(k % (size_t)n_dev), &c), "sgc_init");
        ctxs.push_back(c);
        if (!opt.quiet && !opt.exact && k == 0) fprintf(stderr, "Generating Mismatch Library\n");
        sgc_check(sgc_set_library(c, (const uint8_t *)flat.data(), (uint32_t)library.seqs.size(), (uint32_t)library.size,
                                  !opt.exact), "sgc_set_library");
        if (!opt.quiet && !opt.exact && k == 0) fprintf(stderr, "Finished Mismatch Library\n");
    }
    // samples in parallel (count.rs:117-136: rayon over samples, pool size -t), results in input order
    const size_t n = opt.input_paths.size();
    std::vector<SampleCounts> results(n);
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
                results[i] = count_sample(ctxs[d], opt.input_paths[i], opt.offsets[i], library, opt);
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
    write_results(opt.output_path, results, library, opt.sample_names, opt.genemap, opt.include_zero);
}

}  // namespace sgh
