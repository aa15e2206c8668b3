// sgh_scan.cpp — FastqScanner: plain FASTQ text -> packed records, on the host, at memory speed.
//
// This is the host half of the count path as the north star draws it ("a host that streams FASTQ, 2-bit-packs reads into
// u64 k-mers and ships pinned batches through a thin C-ABI HIP layer"): it replaces the fxread iterator and
// Counter::apply_trim (reference src/counter.rs:144-204, 211-236) for plain-text samples.  Per read only the L + 2 oriented
// bases [o - 1, o + L + 1) matter (sgc_format.h), so instead of shipping 316 bytes of text per read over PCIe and finding
// the records on the GPU (the `--pack fastq` path, kept as the validated fallback), the reader threads find the lines in
// the page cache and emit ONE 8-byte record per read: 40x fewer bytes on the link.
//
// The file is memory-mapped and never copied.  Why not pread(): the first read() of page-cache pages that were just written
// (a tmpfs file above all: shmem_file_read_iter marks every page accessed) moves each of them to the active LRU list under
// ONE lock — 16 GB/s in all however many threads read (profiles/r03/first_read_probe.txt: that was the 2 s the first run of
// round 2's e2e leg lost).  A mapping does the same when its pages are unmapped with their accessed bits set — unless the
// mapping is marked MADV_SEQUENTIAL (vma_has_recency(): the kernel then ignores those bits) — and MADV_POPULATE_READ /
// MAP_POPULATE touch the pages the same way (FOLL_TOUCH), so the pages are simply faulted in by the scan (64 KiB per
// fault) and dropped block by block with MADV_DONTNEED, in parallel rather than at exit.  Threads take 4 MiB blocks in
// file order:
//   phase 1   list the line starts of the block (AVX2 compare + movemask; SWAR fallback)
//   chain     line number of the block's first line = sum of the line counts of all earlier blocks (published in order)
//   phase 2   walk the lines: marker bytes of header ('@') and separator ('+') lines are verified, every sequence line
//             becomes a record (fast path: all L + 2 window bytes present and ACGT -> pshufb + pext; anything else goes
//             through sgc_pack_one, the same function the device packers restate)
// and the consumer takes the blocks' records in order.  A '\r' before the '\n' belongs to the terminator, trailing blank
// lines at the end of the file are not records, a line count that is no multiple of 4 is a truncated record — the rules
// of the other readers.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../sgc_format.h"
#include "sgh.hpp"

namespace sgh {

static double scan_now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a container sees all the
// host's CPUs in hardware_concurrency() but is throttled to its quota)
size_t usable_cpus() {
    size_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<size_t>(n, (size_t)std::max(1, CPU_COUNT(&set)));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const unsigned long long quota = strtoull(q, nullptr, 10);
            if (quota > 0) n = std::min<size_t>(n, (size_t)std::max<unsigned long long>(1, (quota + period / 2) / period));
        }
        fclose(f);
    }
    return n;
}

// the shadow keys of the guides with exactly one byte outside ACGT (see sgc_bytes.hip): an ACGT window equal to such a guide
// everywhere else is within one substitution of it; a guide with two or more such bytes is at distance >= 2 from every ACGT window
RouteFilter make_route_filter(const std::vector<std::string> &seqs) {
    RouteFilter f;
    std::vector<uint64_t> keys;
    for (const std::string &q : seqs) {
        uint32_t bad = 0, at = 0;
        for (uint32_t k = 0; k < q.size(); k++) if (!(q[k] == 'A' || q[k] == 'C' || q[k] == 'G' || q[k] == 'T')) { bad++; at = k; }
        if (bad != 1 || q.size() > 32) continue;
        uint64_t key = 0;
        for (uint32_t k = 0; k < q.size(); k++) if (k != at) key |= (uint64_t)sgc_base_code((uint8_t)q[k]) << (2 * k);
        for (uint64_t b = 0; b < 4; b++) keys.push_back(key | (b << (2 * at)));
    }
    if (keys.empty()) return f;
    f.log2_words = 6;
    while (f.log2_words < 20 && (1ull << f.log2_words) * 8 < keys.size() * 2 + 64) f.log2_words++;       // >= 32 bits per key
    f.words.assign((size_t)1 << f.log2_words, 0);
    for (uint64_t k : keys) { const uint64_t h2 = sgc_hash2(k); f.words[sgc_bloom_word(h2, f.log2_words)] |= sgc_bloom_mask(h2); }
    return f;
}

// ---- line starts of a byte range ------------------------------------------------------------------------
// appends p + 1 - base for every '\n' at p in [lo, hi)
static void list_newlines_generic(const uint8_t *t, size_t lo, size_t hi, size_t base, std::vector<uint32_t> &out) {
    for (size_t p = lo; p < hi;) {
        const void *q = memchr(t + p, '\n', hi - p);
        if (!q) break;
        p = (size_t)((const uint8_t *)q - t);
        out.push_back((uint32_t)(p + 1 - base));
        p++;
    }
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi"))) static void list_newlines_avx2(const uint8_t *t, size_t lo, size_t hi, size_t base,
                                                                    std::vector<uint32_t> &out) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t n = out.size();
    out.resize(n + 1024);
    uint32_t *w = out.data();
    size_t p = lo;
    for (; p + 64 <= hi; p += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(t + p)), b = _mm256_loadu_si256((const __m256i *)(t + p + 32));
        uint64_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nl)) |
                     ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nl)) << 32);
        if (!m) continue;
        if (n + 64 > out.size()) { out.resize(out.size() * 2 + 64); w = out.data(); }
        const uint32_t rel = (uint32_t)(p + 1 - base);
        while (m) { w[n++] = rel + (uint32_t)__builtin_ctzll(m); m &= m - 1; }
    }
    out.resize(n);
    list_newlines_generic(t, p, hi, base, out);
}
#endif

static bool have_avx2() {
#if defined(__x86_64__)
    static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi");
    return v;
#else
    return false;
#endif
}

// ---- one read -> one record --------------------------------------------------------------------------------
// Fast path: the K = L + 2 window bytes all exist and are ACGT.  Returns false when the general packer must decide.
#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) static bool pack_fast_avx2(const uint8_t *win /* K bytes in FILE order, 32 readable */, uint32_t K,
                                                                bool reverse, uint64_t &span) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)win);
    // by low nibble: 'A' 0x41 -> 1, 'C' 0x43 -> 3, 'T' 0x54 -> 4, 'G' 0x47 -> 7; the other entries hold a byte whose own low
    // nibble differs from their index, so no input byte can equal them
    const __m256i want = _mm256_setr_epi8(1, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0, 1, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i code = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i nib = _mm256_and_si256(v, _mm256_set1_epi8(0x0F));
    const uint32_t ok = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(want, nib), v));
    const uint32_t need = K >= 32 ? 0xFFFFFFFFu : ((1u << K) - 1u);
    if ((ok & need) != need) return false;
    const __m256i c = _mm256_shuffle_epi8(code, nib);
    alignas(32) uint64_t q[4];
    _mm256_store_si256((__m256i *)q, c);
    const uint64_t M = 0x0303030303030303ull;
    uint64_t x = _pext_u64(q[0], M) | (_pext_u64(q[1], M) << 16) | (_pext_u64(q[2], M) << 32) | (_pext_u64(q[3], M) << 48);
    const uint64_t kmask = K >= 32 ? ~0ull : ((1ull << (2 * K)) - 1ull);
    if (reverse) {
        // oriented base w = complement of file base K - 1 - w: reverse the 2-bit groups of the 64-bit word, drop the
        // 32 - K groups that end up below, complement (3 - code)
        x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
        x = __builtin_bswap64(x);
        x = (x >> (64 - 2 * K)) ^ kmask;
    }
    span = x & kmask;
    return true;
}
#endif

struct FastqScanner::Block {
    std::vector<uint32_t> starts;      // line starts relative to the block's first byte (phase 1; dropped after phase 2)
    std::vector<uint64_t> recs;        // phase 2: the records of the sequence lines that START in this block
    uint64_t first_line = 0;           // global number of the block's first line
    uint64_t bad_line = 0;             // 1-based number of the first line whose marker byte is wrong (0 = none)
    bool scanned = false, done = false;
    // hybrid library: the reads this block does NOT turn into records — a guide with bytes outside ACGT could influence them —, as
    // bytes + offsets for the byte-string chain (sgc_sample_push_reads)
    std::vector<uint8_t> routed_bytes;
    std::vector<uint64_t> routed_offs;
};

// A BGZF member (bgzip, htslib: a gzip member whose extra field announces the member's size in a 'BC' subfield): its size in bytes
// and the length of its extra field, or 0 if the `avail` bytes at `h` are not one whole such member.  One walk, with every bound,
// for the pass that cuts the file into chunks of members and for the workers that inflate them.
static size_t bgzf_member_size(const uint8_t *h, size_t avail, size_t &xlen) {
    if (avail < 28 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    xlen = h[10] + 256u * h[11];
    if (12 + xlen + 8 > avail) return 0;
    size_t bsize = 0;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
        const size_t slen = h[x + 2] + 256u * h[x + 3];
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)(h[x + 4] + 256u * h[x + 5]) + 1;
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || bsize > avail) return 0;
    return bsize;
}

FastqScanner::FastqScanner(const std::string &path_, const ScanParams &prm_, size_t threads, size_t block_bytes, size_t max_ahead, int source_)
    : path(path_), prm(prm_) {
    source = source_;
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("No such file or directory (os error 2): " + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); fd = -1; usable = false; return; }
    file_size = (size_t)sb.st_size;
    if (file_size == 0) { usable = false; return; }
    void *m = mmap(nullptr, file_size, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { usable = false; return; }
    map = (const uint8_t *)m;
    (void)madvise(m, file_size, MADV_SEQUENTIAL);            // see the head of this file: no LRU activation when the blocks are unmapped
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
    words = prm.L > SGC_REC8_MAXL ? 2 : 1;
    if (file_size >= 64 && map[0] == 0x1f && map[1] == 0x8b && map[2] == 8 && threads > 1) {
        // gzip.  BGZF (members that announce their size: independent streams) is left to the reader that inflates members in parallel
        // straight into pinned memory; a plain stream is decoded here, chunk by chunk
        gz_mode = true;
        // compressed bytes per chunk: 1 MiB (a block size below that is taken literally: tests).  The chunk's symbols (2 bytes per byte
        // of text) and text live in buffers of the worker: with 8 MiB chunks of text that compresses 19:1 those were 460 MB per
        // thread, and a 30M-read sample took 2.2 s instead of 1.0 (tools/gz_chunk_probe.py: 8 MiB 2.1-2.4 s, 4 MiB 1.5, 2 MiB 1.1,
        // 1 MiB 0.97-1.06, 512 KiB 1.01-1.06, 256 KiB 1.33, 128 KiB 2.0 — below, the search for a block start in every chunk weighs in)
        gz_chunk_bytes = block_bytes < (1u << 20) ? std::max<size_t>(block_bytes, 512) : (size_t)1 << 20;
        n_blocks = (file_size + gz_chunk_bytes - 1) / gz_chunk_bytes;
        // BGZF (bgzip, htslib, Illumina's converters)?  Only if EVERY member announces its size in a 'BC' extra subfield — bgzip output
        // followed by ordinary gzip output is legal gzip and takes the plain-stream decoder: hop through the member headers once
        // and cut the file into runs of whole members of about a chunk each
        {
            size_t off = 0, chunk_start = 0;
            bool chain = true;
            std::vector<size_t> cuts(1, 0);
            while (chain && off < file_size) {
                size_t xlen = 0;
                const size_t bsize = bgzf_member_size(map + off, file_size - off, xlen);
                if (bsize == 0) chain = false;
                else {
                    off += bsize;
                    if (off - chunk_start >= gz_chunk_bytes && off < file_size) { cuts.push_back(off); chunk_start = off; }
                }
            }
            if (chain) { bgzf_mode = true; cuts.push_back(file_size); bgzf_chunk_off.swap(cuts); n_blocks = bgzf_chunk_off.size() - 1; }
        }
        blocks.reset(new Block[n_blocks]);
        gz_pieces.resize(n_blocks);
        gz_window.assign(32768, 0);
        n_threads = std::max<size_t>(1, std::min(threads, n_blocks));
        for (size_t t = 0; t < n_threads; t++) workers.emplace_back([this] { run_gz(); });
        // FASTQ or not is decided by the first byte of the TEXT
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return failed || gz_first_known; });
        if (failed || gz_first_byte != '@') {
            stop = true;
            lk.unlock();
            cv.notify_all();
            for (auto &w : workers) if (w.joinable()) w.join();
            workers.clear();
            usable = false;
        }
        return;
    }
    if (map[0] != '@') { usable = false; return; }          // FASTA, anything else: the other readers decide
    // trailing blank lines at the very end are not records
    end = file_size;
    while (end >= 2 && map[end - 1] == '\n' && map[end - 2] == '\n') end--;
    block = std::max<size_t>(block_bytes, 4096) & ~(size_t)4095;
    n_blocks = (end + block - 1) / block;
    blocks.reset(new Block[n_blocks]);
    ahead = std::max<size_t>(max_ahead, 2);
    n_threads = std::max<size_t>(1, std::min(threads, n_blocks));
    for (size_t t = 0; t < n_threads; t++) workers.emplace_back([this] { run(); });
}

FastqScanner::~FastqScanner() {
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    for (auto &w : workers) if (w.joinable()) w.join();
    blocks.reset();
    if (map) munmap((void *)map, file_size);
    if (fd >= 0) close(fd);
}

void FastqScanner::run() {
    try {
        for (;;) {
            size_t b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || failed || next_block >= n_blocks || next_block < consumed + ahead; });
                if (stop || failed || next_block >= n_blocks) return;
                b = next_block++;
            }
            const double t0 = scan_now_s();
            Block &blk = blocks[b];
            const size_t lo = b * block, hi = std::min(end, lo + block);
            const size_t map_hi = std::min(file_size, (hi + 4095) & ~(size_t)4095);
            // Where the block's bytes come from: the mapping (default), or — options — a pread() into a buffer of the thread's own, which
            // costs less than faulting the mapping in (a copy against a fault per 64 KiB: 0.30 against 0.39 s of busy time per thread
            // on the 100M-read file, profiles/r03/e2e_scan_source.txt) UNLESS the file's pages were just written: then every first
            // read() moves a page to the active LRU list under one lock and the same run takes 2.4 s instead of 0.6 (the head of this
            // file).  "auto" starts with pread() and falls back to the mapping as soon as the threads' average read rate says so; it
            // measured no better than the mapping alone, which is why the mapping is the default.
            static thread_local std::vector<uint8_t> tbuf;
            const uint8_t *t = map;                           // t[absolute offset], valid for offsets in [t_lo, t_hi)
            size_t t_lo = 0, t_hi = file_size;
            const bool by_read = source == 2 || (source == 0 && !auto_map.load(std::memory_order_relaxed));
            if (by_read) {
                const size_t a = lo ? lo - 1 : 0, z = std::min(end, hi + READ_SLACK), len = z - a;
                if (tbuf.size() < len + 64) tbuf.resize(len + 64);
                const double r0 = scan_now_s();
                size_t got = 0;
                while (got < len) {
                    const ssize_t r = pread(fd, tbuf.data() + got, len - got, (off_t)(a + got));
                    if (r < 0) throw Error("read error in " + path);
                    if (r == 0) throw Error("file shrank while reading: " + path);
                    got += (size_t)r;
                }
                memset(tbuf.data() + len, 0, 64);
                const double dr = scan_now_s() - r0;
                t = tbuf.data() - a; t_lo = a; t_hi = z;
                if (source == 0) {
                    const uint64_t nb = read_bytes.fetch_add(len) + len, ns_ = read_ns.fetch_add((uint64_t)(dr * 1e9)) + (uint64_t)(dr * 1e9);
                    if (nb >= (64u << 20) && (double)nb / ((double)ns_ + 1.0) < 2.5) auto_map.store(true, std::memory_order_relaxed);      // < 2.5 GB/s per thread
                }
            }
            // phase 1: the lines that start in [lo, hi): after every '\n' at [lo - 1, hi - 1), and at 0
            blk.starts.reserve((hi - lo) / 64 + 16);
            if (b == 0) blk.starts.push_back(0);
            const size_t s_lo = lo ? lo - 1 : 0, s_hi = hi - 1;
#if defined(__x86_64__)
            if (have_avx2()) list_newlines_avx2(t, s_lo, s_hi, lo, blk.starts);
            else
#endif
                list_newlines_generic(t, s_lo, s_hi, lo, blk.starts);
            {
                std::unique_lock<std::mutex> lk(mu);
                blk.scanned = true;
                while (chain < n_blocks && blocks[chain].scanned) {
                    blocks[chain].first_line = lines_so_far;
                    lines_so_far += blocks[chain].starts.size();
                    chain++;
                }
                cv.notify_all();
                cv.wait(lk, [&] { return stop || failed || chain > b; });
                if (stop || failed) return;
            }
            (void)t_lo;
            extract(blk, t, lo, t_hi, blk.starts.size(), end);
            if (!by_read) (void)madvise((void *)(map + lo), map_hi - lo, MADV_DONTNEED);      // drop the block's page-table entries here, in parallel, not at exit
            const double dt = scan_now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                blk.done = true;
                busy_s += dt;
            }
            cv.notify_all();
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
}

// gz mode: chunk k of the COMPRESSED file.  As TextFeeder::run_pgz (sgh.cpp) up to the stitching chain — speculative decode from a
// block start found inside the chunk, accepted if that is where the chunk before ended, decoded again in order otherwise —, but the
// resolved text goes into a buffer of the thread and is packed right there.  A second chain hands on what only the text in front
// can tell: the number of the chunk's first line and the unfinished line it continues (`gz_carry`).
void FastqScanner::run_gz() {
    try {
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                // (not too far ahead of the chain that stitches the chunks: decoded chunks wait in memory)
                cv.wait(lk, [&] { return stop || failed || gz_next >= n_blocks || gz_next < (bgzf_mode ? gz_lines : gz_chain) + 2 * n_threads; });
                if (stop || failed || gz_next >= n_blocks) return;
                k = gz_next++;
            }
            const double t0 = scan_now_s();
            Block &blk = blocks[k];
            static thread_local std::vector<uint8_t> text_keep, lut_keep;
            constexpr size_t ROOM = 1u << 16;           // in front of the chunk's text: room for the unfinished line it continues
            uint8_t *txt = nullptr;
            size_t n = 0;
            bool last = false;
            std::vector<GzPiece> pieces;
            if (bgzf_mode) {
                // the members [bgzf_chunk_off[k], bgzf_chunk_off[k + 1]): sizes from the ISIZE trailers, then zlib, member by member
                const size_t c_lo = bgzf_chunk_off[k], c_hi = bgzf_chunk_off[k + 1];
                // (the same header walk as the one that cut the chunks, with the same bounds: a member whose size it cannot confirm here
                // — the file changed under the mapping — is an error, never a read past the chunk)
                auto bsize_of = [&](size_t off, size_t &xlen) {
                    const size_t bs = bgzf_member_size(map + off, c_hi - off, xlen);
                    if (bs == 0) throw Error("corrupt BGZF member in " + path);
                    return bs;
                };
                size_t total = 0;
                for (size_t off = c_lo; off < c_hi;) {
                    size_t xlen; const size_t bs = bsize_of(off, xlen);
                    const uint8_t *tr = map + off + bs - 8;
                    total += (size_t)tr[4] | (size_t)tr[5] << 8 | (size_t)tr[6] << 16 | (size_t)tr[7] << 24;
                    off += bs;
                }
                if (text_keep.size() < ROOM + total + 64) text_keep.resize(ROOM + total + 64 + (total >> 3));
                txt = text_keep.data() + ROOM;
                struct RawInflater {                   // one per worker thread, released when the thread ends
                    z_stream zs; bool ready = false;
                    ~RawInflater() { if (ready) inflateEnd(&zs); }
                };
                static thread_local RawInflater inf;
                if (!inf.ready) { memset(&inf.zs, 0, sizeof inf.zs); if (inflateInit2(&inf.zs, -15) != Z_OK) throw Error("zlib: inflateInit2 failed"); inf.ready = true; }
                z_stream &zs = inf.zs;
                size_t at = 0;
                for (size_t off = c_lo; off < c_hi;) {
                    size_t xlen; const size_t bs = bsize_of(off, xlen);
                    const uint8_t *tr = map + off + bs - 8;
                    const uint32_t want_crc = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
                    const size_t isize = (size_t)tr[4] | (size_t)tr[5] << 8 | (size_t)tr[6] << 16 | (size_t)tr[7] << 24;
                    if (inflateReset2(&zs, -15) != Z_OK) throw Error("zlib: inflateReset2 failed");
                    zs.next_in = const_cast<Bytef *>(map + off + 12 + xlen); zs.avail_in = (uInt)(bs - 12 - xlen - 8);
                    zs.next_out = txt + at; zs.avail_out = (uInt)isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0 || crc32_fast(0, txt + at, isize) != want_crc)
                        throw Error("corrupt BGZF member in " + path);
                    at += isize; off += bs;
                }
                n = total;
                memset(txt + n, 0, 64);
                last = k + 1 == n_blocks;
            } else {
            const size_t lo = k * gz_chunk_bytes, hi = std::min(file_size, lo + gz_chunk_bytes);
            const size_t max_out = (size_t)1 << 31;
            static thread_local std::vector<uint16_t> sym_keep;     // kept from chunk to chunk: fresh pages cost more than decoding
            InflateSpan span;
            span.sym.swap(sym_keep);
            span.sym.clear();
            struct Keep { InflateSpan &s; std::vector<uint16_t> &k; ~Keep() { s.sym.clear(); s.sym.swap(k); } } keep{span, sym_keep};
            auto reset_span = [&]() { span.sym.clear(); span.members.clear(); span.end_of_stream = false; span.start_bit = span.end_bit = 0; };
            bool found = false;
            uint64_t start = 0;
            if (k == 0) {
                size_t at = 10;                       // the first member's header, then its first block
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
            // ---- chain A: the chunk's true start and the window in front of it
            uint8_t window[32768];
            bool empty = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || failed || gz_chain == k; });
                if (stop || failed) return;
                if (k == 0) gz_pos = start;
                memcpy(window, gz_window.data(), 32768);
                const uint64_t pos = gz_pos;
                if (gz_eos || (pos >= 8ull * hi && k + 1 < n_blocks)) {
                    empty = true;                     // the stream ended, or a block that began earlier covers this whole chunk
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
                    const size_t n = span.sym.size();
                    uint8_t *nw = gz_window.data();
                    if (n >= 32768) {
                        for (size_t i = 0; i < 32768; i++) { const uint16_t v = span.sym[n - 32768 + i]; nw[i] = v < 256 ? (uint8_t)v : window[v & 0x7FFFu]; }
                    } else {
                        memmove(nw, nw + n, 32768 - n);
                        for (size_t i = 0; i < n; i++) { const uint16_t v = span.sym[i]; nw[32768 - n + i] = v < 256 ? (uint8_t)v : window[v & 0x7FFFu]; }
                    }
                    gz_pos = span.end_bit;
                    if (span.end_of_stream) { gz_eos = true; last = true; }
                }
                if (k + 1 == n_blocks && !gz_eos) throw Error("truncated gzip stream in " + path);
                gz_chain = k + 1;
            }
            cv.notify_all();
            // ---- the text of the chunk, behind room for the unfinished line in front of it; CRC-32 of every member piece
            n = span.sym.size();
            if (text_keep.size() < ROOM + n + 64) text_keep.resize(ROOM + n + 64 + (n >> 3));
            txt = text_keep.data() + ROOM;
            if (lut_keep.size() != 65536) { lut_keep.assign(65536, 0); for (unsigned i = 0; i < 256; i++) lut_keep[i] = (uint8_t)i; }
            memcpy(lut_keep.data() + 0x8000, window, 32768);
            resolve_symbols(span.sym.data(), n, lut_keep.data(), txt);
            memset(txt + n, 0, 64);
            {
                size_t done = 0, mi = 0;
                uint32_t crc = 0;
                uint64_t piece_len = 0;
                while (done < n || mi < span.members.size()) {
                    while (mi < span.members.size() && span.members[mi].at == done) {
                        pieces.push_back(GzPiece{crc, piece_len, true, span.members[mi].crc, span.members[mi].isize});
                        crc = 0; piece_len = 0; mi++;
                    }
                    if (done >= n) break;
                    const size_t m = mi < span.members.size() ? span.members[mi].at - done : n - done;
                    crc = crc32_fast(crc, txt + done, m);
                    piece_len += m; done += m;
                }
                if (piece_len) pieces.push_back(GzPiece{crc, piece_len, false, 0, 0});
            }
            }       // plain gzip stream
            // line starts behind every newline of the chunk (relative to txt; one at n belongs to the next chunk)
            std::vector<uint32_t> nls;
            nls.reserve(n / 64 + 16);
#if defined(__x86_64__)
            if (have_avx2()) list_newlines_avx2(txt, 0, n, 0, nls);
            else
#endif
                list_newlines_generic(txt, 0, n, 0, nls);
            // ---- chain B: line numbers, the unfinished line, and blank lines that may turn out to be the end of the stream
            // Newlines at the end of a chunk beyond the one that ends its last line are blank lines; at the very end of the stream they
            // are not records (the plain scanner cuts them off its mapping).  Whether they are the end only the chunks behind can
            // tell, so they are DEFERRED: counted in gz_pending, neither numbered nor packed, and put back in front of the next chunk
            // that brings anything but newlines (where they are what they are: malformed headers, empty sequence lines).
            std::vector<uint8_t> carry;
            size_t n_txt = n, pend = 0;               // the chunk's text without its deferred newlines; blank lines put back in front of it
            bool final_open = false;                  // the stream ends with a line that has no newline
            size_t n_complete = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || failed || gz_lines == k; });
                if (stop || failed) return;
                if (k == 0) { gz_first_byte = n ? txt[0] : 0; gz_first_known = true; gz_text_ends_nl = true; }
                size_t r = 0;
                while (r < n && txt[n - 1 - r] == '\n') r++;
                const size_t d = r == n ? (n ? r - (gz_text_ends_nl ? 0u : 1u) : 0u) : (r ? r - 1 : 0u);
                n_txt = n - d;
                for (size_t x = 0; x < d; x++) nls.pop_back();
                blk.first_line = lines_so_far;
                if (n_txt) {
                    carry = gz_carry;
                    pend = gz_pending;
                    gz_pending = d;
                    n_complete = pend + nls.size();
                    lines_so_far += n_complete;
                    gz_text_ends_nl = txt[n_txt - 1] == '\n';
                    if (nls.empty()) gz_carry.insert(gz_carry.end(), txt, txt + n_txt);
                    else gz_carry.assign(txt + nls.back(), txt + n_txt);
                    if (gz_carry.size() > GZ_MAX_LINE) throw Error("FASTQ line longer than " + std::to_string(GZ_MAX_LINE) + " bytes in " + path + " (--pack device reads lines of any length)");
                } else {
                    gz_pending += d;
                    if (last) carry = gz_carry;        // (an unfinished last line is packed by the chunk that ends the stream, even an empty one)
                }
                if (last && !gz_carry.empty()) { final_open = true; lines_so_far++; }
                gz_pieces[k] = std::move(pieces);
                gz_lines = k + 1;
            }
            cv.notify_all();
            // ---- pack: the text is (blank lines put back) ++ (unfinished line in front) ++ (chunk); its lines that end inside the chunk
            const size_t front = pend + carry.size();
            uint8_t *C = txt - front;
            if (front > ROOM) {                        // a very long line (or very many blank lines): a buffer of its own
                static thread_local std::vector<uint8_t> big;
                big.resize(front + n_txt + 64);
                memcpy(big.data() + front, txt, n_txt);
                memset(big.data() + front + n_txt, 0, 64);
                C = big.data();
            }
            if (pend) memset(C, '\n', pend);
            if (!carry.empty()) memcpy(C + pend, carry.data(), carry.size());
            const size_t clen = front + n_txt;
            if (clen) {
                blk.starts.reserve(pend + nls.size() + 1);
                for (size_t x = 0; x < pend; x++) blk.starts.push_back((uint32_t)x);
                blk.starts.push_back((uint32_t)pend);
                for (uint32_t s : nls) if (s < n_txt) blk.starts.push_back((uint32_t)(front + s));
                extract(blk, C, 0, clen, n_complete + (final_open ? 1 : 0), clen);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                blk.done = true;
                busy_s += scan_now_s() - t0;
            }
            cv.notify_all();
        }
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; error = e.what(); }
        cv.notify_all();
    }
}

// phase 2 of block b; t[offset] is readable for offsets in [t_lo, t_hi) (the thread's buffer, or the mapping itself); what lies beyond
// — the rest of a line longer than the slack behind the block — is taken from the mapping
void FastqScanner::extract(Block &blk, const uint8_t *t, size_t lo, size_t t_hi, size_t n_pack, size_t text_end) {
    const size_t ns = blk.starts.size();
    const uint32_t L = prm.L, K = L + 2, o = prm.offset;
    const bool rev = prm.reverse, rec = prm.recursion;
    const uint32_t sh = 2 * K;
    // status of a read whose K window bytes are all there and all ACGT: C clean; P and M clean with recursion (o >= 1 is a
    // condition of the fast path), dead without
    const uint64_t fast_status = rec ? 0ull : (uint64_t)K * (1ull + K) * SGC_STATE_DEAD;
    const bool fast_ok = have_avx2() && o >= 1;
    blk.recs.reserve((ns / 4 + 2) * words);
    uint64_t g = blk.first_line;
    for (size_t i = 0; i < n_pack; i++, g++) {
        const size_t s = lo + blk.starts[i];
        const uint32_t ph = (uint32_t)(g & 3u);
        if (ph == 0 || ph == 2) {
            if (t[s] != (ph == 0 ? '@' : '+') && !blk.bad_line) blk.bad_line = g + 1;
            continue;
        }
        if (ph == 3) continue;
        // sequence line [s, e)
        size_t e;
        const uint8_t *src = t;                       // where this line's bytes are read from
        size_t src_hi = t_hi;
        if (i + 1 < ns) e = lo + blk.starts[i + 1] - 1;
        else {
            const void *q = memchr(t + s, '\n', t_hi - s);
            if (q) e = (size_t)((const uint8_t *)q - t);
            else if (t_hi >= text_end) e = text_end;
            else {                                    // the line runs past the thread's buffer: finish it on the mapping
                src = map; src_hi = file_size;
                const void *q2 = memchr(map + t_hi, '\n', text_end - t_hi);
                e = q2 ? (size_t)((const uint8_t *)q2 - map) : text_end;
            }
        }
        size_t n = e - s;
        if (n && src[s + n - 1] == '\r') n--;
        uint64_t span = 0, status = 0;
        bool done = false;
#if defined(__x86_64__)
        if (fast_ok && n >= (size_t)o + L + 1) {
            const size_t w0 = rev ? s + n - o - L - 1 : s + o - 1;           // first window byte in file order
            if (w0 + 32 <= src_hi) {
                done = pack_fast_avx2(src + w0, K, rev, span);
                status = fast_status;
            }
        }
#endif
        if (!done) sgc_pack_one(src + s, n, L, rev ? 1 : 0, o, rec ? 1 : 0, span, status);
        if (prm.route) {
            // hybrid library (sgc_bytes.hip k_bytes_route states the rule): a byte outside ACGT in the span region, or a window that
            // hits the filter of the shadow keys, sends the read to the byte-string chain instead of the packed pass
            bool route = false;
            if (!done) {
                for (uint32_t w = 0; w < K && !route; w++) {
                    const int64_t pp = (int64_t)o - 1 + (int64_t)w;
                    if (pp < 0 || (uint64_t)pp >= n) continue;
                    const uint8_t ch = rev ? src[s + n - 1 - (size_t)pp] : src[s + (size_t)pp];
                    route = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
                }
            }
            if (!route && !prm.route->words.empty()) {
                const uint64_t kmask = sgc_key_mask(L);
                const bool c_ok = (uint64_t)o + L <= n, p_ok = c_ok && rec && (uint64_t)o + 1 + L <= n, m_ok = p_ok && o >= 1;
                const uint64_t keys[3] = {(span >> 2) & kmask, (span >> 4) & kmask, span & kmask};
                const bool ok[3] = {c_ok, p_ok, m_ok};
                for (int k = 0; k < 3 && !route; k++) {
                    if (!ok[k]) continue;
                    const uint64_t h2 = sgc_hash2(keys[k]), m = sgc_bloom_mask(h2);
                    route = (prm.route->words[sgc_bloom_word(h2, prm.route->log2_words)] & m) == m;
                }
            }
            if (route) {
                // only the piece of the read that holds its windows (sgc_sample_push_windows): oriented bases [max(o - 1, 0), min(n, o + L + 1))
                const size_t lo_or = o >= 1 ? (size_t)o - 1 : 0, hi_or = std::min<size_t>(n, (size_t)o + L + 1);
                const size_t plen = hi_or > lo_or ? hi_or - lo_or : 0, pbeg = plen ? (rev ? n - hi_or : lo_or) : 0;
                if (blk.routed_offs.empty()) blk.routed_offs.push_back(0);
                blk.routed_bytes.insert(blk.routed_bytes.end(), src + s + pbeg, src + s + pbeg + plen);
                blk.routed_offs.push_back(blk.routed_bytes.size());
                continue;
            }
        }
        if (words == 2) { blk.recs.push_back(span); blk.recs.push_back(status); }
        else blk.recs.push_back(span | (status << sh));
    }
    std::vector<uint32_t>().swap(blk.starts);
}

bool FastqScanner::next(const uint64_t *&recs, size_t &n_records) {
    const double t0 = scan_now_s();
    std::unique_lock<std::mutex> lk(mu);
    if (consumed >= n_blocks) {
        if (!checked_end) {
            checked_end = true;
            if (gz_mode && !gz_verified) {
                // every byte of the stream went through a worker: the CRC-32 and ISIZE of every gzip member, piece by piece
                gz_verified = true;
                uint32_t crc = 0; uint64_t mlen = 0;
                for (const auto &pcs : gz_pieces)
                    for (const GzPiece &pc : pcs) {
                        crc = (uint32_t)crc32_combine(crc, pc.crc, (z_off_t)pc.len);
                        mlen += pc.len;
                        if (pc.member_end) {
                            if (crc != pc.want_crc || (uint32_t)mlen != pc.want_isize) throw Error("corrupt gzip stream (CRC / length mismatch) in " + path);
                            crc = 0; mlen = 0;
                        }
                    }
            }
            // a line with a wrong marker byte in text that passed its CRC is the sample's fault; in text that did not, the file's (above)
            if (gz_bad_line)
                throw Panic("malformed FASTQ record: line " + std::to_string(gz_bad_line) +
                            " does not start with its marker byte ('@' header / '+' separator) in " + path);
            total_lines = lines_so_far;
            // (3 lines of a record: the stream ends behind its separator line — an empty quality line, written with its terminator
            // ("+\n\n": cut off with the trailing blank lines), without it ("+\n"), or not at all ("+"); reader decision #3)
            if (lines_so_far % 4 != 0 && lines_so_far % 4 != 3) throw Panic("truncated FASTQ record in " + path);
        }
        return false;
    }
    cv.wait(lk, [&] { return failed || blocks[consumed].done; });
    wait_s += scan_now_s() - t0;
    if (failed) throw Error(error);
    Block &blk = blocks[consumed];
    if (blk.bad_line && gz_mode) {
        // Compressed input: a damaged stream usually inflates to text that breaks the 4-line cycle long before the end of the member
        // where its CRC is checked.  Whether to blame the sample (exit code 101, like the reference's reader) or the file ("corrupt
        // gzip stream") is decided there: the blocks from here on are read to the end without handing out their records.
        if (!gz_bad_line) gz_bad_line = blk.bad_line;
    }
    if (gz_bad_line) {
        recs = blk.recs.data(); n_records = 0; routed_bytes = nullptr; routed_offs = nullptr; n_routed = 0;
        return true;
    }
    if (blk.bad_line)
        throw Panic("malformed FASTQ record: line " + std::to_string(blk.bad_line) +
                    " does not start with its marker byte ('@' header / '+' separator) in " + path);
    recs = blk.recs.data();
    n_records = blk.recs.size() / words;
    routed_bytes = blk.routed_bytes.data();
    routed_offs = blk.routed_offs.data();
    n_routed = blk.routed_offs.empty() ? 0 : blk.routed_offs.size() - 1;
    return true;
}

void FastqScanner::release() {
    {
        std::lock_guard<std::mutex> lk(mu);
        if (consumed < n_blocks) {
            std::vector<uint64_t>().swap(blocks[consumed].recs);
            std::vector<uint8_t>().swap(blocks[consumed].routed_bytes);
            std::vector<uint64_t>().swap(blocks[consumed].routed_offs);
            consumed++;
        }
    }
    cv.notify_all();
}

}  // namespace sgh
