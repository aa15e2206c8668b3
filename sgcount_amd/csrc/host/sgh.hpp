// sgh.hpp — C++ host side of the MI355X count path: the callers and data formats either side of the
// C ABI (include/sgcount_hip.h), mirroring the reference's modules one to one:
//
//   FastxReader            fxread::initialize_reader          (call sites src/count.rs:24,64,87)
//   Library                src/library.rs
//   Offset, entropy_offset src/offsetter.rs
//   GeneMap                src/genemap.rs
//   generate_sample_names  src/utils.rs:18-49
//   write_results          src/results.rs
//   count                  src/count.rs:74-148   (matching/counting itself runs on the GPU through the ABI)
//   cli_main               src/main.rs:54-203
//
// Errors: the reference either returns anyhow errors (exit 1, "Error: ...") or panics (exit 101).  Here
// both are C++ exceptions: sgh::Error (exit 1) and sgh::Panic (exit 101), with the reference's messages.
#pragma once
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace sgh {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };   // anyhow::Error → exit 1
struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };   // panic!/assert!/unwrap → exit 101

// ---- FASTX ------------------------------------------------------------------------------------------
struct RecordView { const char *id; size_t id_len; const char *seq; size_t seq_len; };

class FastxReader {                // FASTA (2-line) / FASTQ (4-line), plain or .gz (by suffix), streaming
  public:
    explicit FastxReader(const std::string &path);
    ~FastxReader();
    bool next(RecordView &r);      // views stay valid until the next call
    struct Impl;
  private:
    std::unique_ptr<Impl> p;
};

// ---- Library (src/library.rs) ---------------------------------------------------------------------------
struct Library {
    std::vector<std::string> seqs, ids;                   // file order
    std::unordered_map<std::string, size_t> index;        // seq → first record index
    size_t size = 0;
    static Library from_path(const std::string &path);    // from_reader: dup panic (:91-96), size error (:83)
    const std::string *alias(const std::string &seq) const;
};

// ---- Offset (src/offsetter.rs) ----------------------------------------------------------------------
struct Offset { bool reverse = false; size_t index = 0; std::string debug() const; };
std::vector<double> positional_entropy(FastxReader &reader, size_t take);          // :90-95 (first record = size probe)
Offset minimize_mse(const std::vector<double> &reference, const std::vector<double> &comparison);   // :153-163
std::vector<Offset> entropy_offset_group(const std::string &library_path, const std::vector<std::string> &inputs,
                                         size_t subsample);                        // :185-210

// ---- GeneMap (src/genemap.rs) -------------------------------------------------------------------------
struct GeneMap {
    std::unordered_map<std::string, std::string> map;     // sgrna → gene
    static GeneMap from_path(const std::string &path);    // :21-25
    static GeneMap from_buffer(const std::string &text);  // :28-31
    const std::string *get(const std::string &sgrna) const;
    const std::string *missing_alias(const Library &lib) const;   // :81-86 (library order here)
};

// ---- utils / results -------------------------------------------------------------------------------------
std::vector<std::string> generate_sample_names(const std::vector<std::string> &paths);   // src/utils.rs:18-49

struct SampleCounts {                   // what results.rs reads from a Counter
    std::unordered_map<std::string, uint64_t> by_id;     // counter.rs:18 (id-keyed, pooled)
    uint64_t total_reads = 0, matched_reads = 0;
    uint64_t get_value(const std::string &id) const;     // counter.rs:71-76
};
std::string generate_columns(const std::vector<std::string> &names, const GeneMap *genemap);   // results.rs:32-43
// results.rs:71-99; path empty ⇒ stdout.  Rows in library order (the reference's order is HashMap order).
void write_results(const std::string &path, const std::vector<SampleCounts> &results, const Library &library,
                   const std::vector<std::string> &names, const GeneMap *genemap, bool include_zero);
std::string format_results(const std::vector<SampleCounts> &results, const Library &library,
                           const std::vector<std::string> &names, const GeneMap *genemap, bool include_zero);

// ---- count (src/count.rs) -------------------------------------------------------------------------------
struct CountOptions {
    std::string library_path;
    std::vector<std::string> input_paths, sample_names;
    std::string output_path;             // empty ⇒ stdout
    std::vector<Offset> offsets;
    bool exact = false, position_recursion = true, include_zero = false, quiet = false;
    const GeneMap *genemap = nullptr;
    size_t threads = 1;
    bool device_pack = true;             // raw bytes → records on the GPU (else sgc_pack_reads_host)
    bool device_parse = true;            // FASTQ inputs: ship text, find record boundaries on the GPU
    bool host_scan = true;               // plain FASTQ text + packable library: the host scans and packs (FastqScanner), 8 B/read are shipped
    size_t scan_threads = 0;             // scanner threads per sample (0 = min(16, usable CPUs - 1) / worker threads)
    size_t scan_block_bytes = 4u << 20;  // text per unit of scanner work
    int scan_source = 1;                 // 1 the memory mapping (default: the same speed on a file written a moment ago), 2 pread(), 0 auto (FastqScanner)
    size_t chunk_bytes = 64u << 20;      // bytes of text per slice (one upload) for device_parse
    size_t io_threads = 0;               // reader threads per sample for plain FASTQ (0 = min(8, cores / worker threads))
    size_t inflate_threads = 0;          // inflating threads per BGZF sample (set with io_threads == 0: min(16, cores / worker threads))
    size_t batch_reads = 1u << 20;
    std::string stats_path;              // if set: per-stage timings of the run as JSON (bench.py's e2e block)
    size_t max_devices = 0;              // use at most this many of the visible GPUs (0 = all)
};
void count(const CountOptions &opt);                     // count.rs:74-148

// ---- FASTQ text at speed (the byte source of count()'s text path) ----------------------------------------
size_t count_newlines(const uint8_t *p, size_t n);
// symbols of a speculative decode -> bytes: lut[v] = v below 256, the byte of the window in front of the chunk behind a marker (0x8000 | i)
void resolve_symbols(const uint16_t *src, size_t m, const uint8_t *lut, uint8_t *dst);
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len);     // zlib's crc32(), folded with carry-less multiplies where the CPU has them

struct SampleStats {                    // where one sample's wall time went (host side; device side from sgc_timing)
    double wall_s = 0, read_busy_s = 0, read_wait_s = 0, push_s = 0, upload_wait_s = 0, finish_s = 0;
    double h2d_ms = 0, ingest_ms = 0, count_ms = 0;
    double host_copy_s = 0;             // scan path: block records -> pinned buffers (consumer thread)
    bool scan_path = false;             // the host scanned and packed the text (FastqScanner); text_path: the GPU parsed it
    bool scan_mapped = false;           // ... reading through the memory mapping (else pread() into the threads' buffers)
    double feeder_setup_s = 0, first_push_s = 0;   // pinned ring allocation + reader start; the first push (device scratch allocation)
    uint64_t text_bytes = 0, reads = 0;
    size_t reader_threads = 0;
    bool gz = false, bgzf = false, text_path = false;
    bool pgz = false;                   // a plain gzip stream inflated by several threads (TextFeeder::run_pgz)
    size_t pgz_fallbacks = 0;           // ... chunks of it that had to be decoded in order
};

// ---- deflate from the middle of a stream (sgh_inflate.cpp) -----------------------------------------------------------
// What one speculative decode of a chunk of a gzip file yields: 16-bit symbols (< 256: a byte; 0x8000 | i: byte i of the unknown
// 32 KiB in front of the chunk), the bit positions it covers, and the gzip members that ended inside it.
struct InflateSpan {
    std::vector<uint16_t> sym;
    uint64_t start_bit = 0, end_bit = 0;
    bool end_of_stream = false;
    struct MemberEnd { size_t at; uint32_t crc, isize; };     // a member ends in front of symbol `at`; its trailer says crc / isize
    std::vector<MemberEnd> members;
};
// first plausible (non-final, dynamic-Huffman, text-producing) block start in [from_bit, to_bit)
bool find_block_start(const uint8_t *data, size_t size, uint64_t from_bit, uint64_t to_bit, uint64_t &found);
// blocks from start_bit up to the first block boundary >= stop_bit (or the end of the stream); window: the 32 KiB in front of
// start_bit if known (then no markers are produced), else null.  0 = ok, < 0 = not a valid deflate stream from there.
int inflate_span(const uint8_t *data, size_t size, uint64_t start_bit, uint64_t stop_bit, const uint8_t *window, InflateSpan &out,
                 size_t max_out);

// Fills a small ring of (pinned) buffers with consecutive slices of a file's text and counts the newlines of every
// slice on the way.  Plain files: `threads` readers pread() disjoint sub-ranges of a slice in parallel (page cache →
// buffer is a memory copy, one core does ~5-10 GB/s of it).  gzip streams (by magic number): one inflating producer —
// unless the file is BGZF (gzip members of <= 64 KiB that announce their own size in a 'BC' extra field: bgzip, htslib and
// Illumina's converters write it): the members are independent deflate streams, so the reader threads inflate them in
// parallel straight into the pinned slice (each member's place is known from the ISIZE trailers before anything is inflated).
// Every buffer has HEAD spare bytes in front of the slice, where the consumer parks the unfinished line of the
// slice before.
class TextFeeder {
  public:
    static constexpr size_t HEAD = 4u << 20;      // longest unfinished line that can be carried from one slice to the next
    // threads: readers of a plain file; inflate_threads (>= threads is used): inflaters of a BGZF file
    TextFeeder(const std::string &path, size_t slice_bytes, size_t ring, size_t threads, void *(*alloc)(size_t),
               void (*release)(void *), size_t inflate_threads = 0, size_t pgz_chunk = 0);
    ~TextFeeder();
    // blocks until slice k (k = 0, 1, 2, ... in order) is in its buffer; eof = this is the last slice
    bool acquire(size_t k, uint8_t *&data, size_t &len, uint64_t &newlines, bool &eof);
    uint8_t *buffer_of(size_t k) const { return bufs[k % ring_n]; }     // base of slice k's buffer (slice data at + HEAD)
    void release_below(size_t k);       // the buffers of slices < k may be refilled
    uint8_t first_byte = 0;             // first byte of the text
    bool is_gz = false, is_bgzf = false, is_pgz = false;      // is_pgz: one gzip stream inflated by several threads
    bool plain_mapped = false;           // plain text: the readers copy from a mapping of the file (no first-read LRU activation)
    size_t file_size = 0, n_threads = 1;
    size_t pgz_chunk_bytes = 0;          // compressed bytes per unit of speculative work (0 = chosen from the file size)
    size_t pgz_fallbacks = 0;            // chunks whose speculative start was wrong or missing and that were decoded in order instead
    double busy_s = 0, wait_s = 0;      // Σ reader busy time; time the consumer waited for text
  private:
    struct Slot { size_t index = (size_t)-1, len = 0, pending = 0; uint64_t newlines = 0; bool ready = false, eof = false; };
    struct BlockRef { size_t in_off; uint32_t in_len, out_off, out_len, crc; };     // one BGZF member: deflate bytes in the file -> place in the slice
    void run_plain();
    void run_gz();
    void run_bgzf();
    void run_pgz();                      // plain gzip, several threads: speculative decode of chunks (sgh_inflate.cpp), stitched in order
    bool slice_ready_locked(size_t k, size_t &len, bool &eof) const;
    void plan_bgzf_slice(size_t k);      // under the lock: the members of slice k, from scan_off on
    void shutdown();
    const uint8_t *map = nullptr;        // BGZF: the compressed file, memory-mapped
    size_t scan_off = 0, planned = 0;    // BGZF: next member to plan; slices planned so far
    std::vector<std::vector<BlockRef>> plans;
    // parallel gzip (run_pgz): the chain that stitches the chunks in order
    struct PgzPiece { uint32_t crc; uint64_t len; bool member_end; uint32_t want_crc, want_isize; };
    size_t pgz_chunks = 0, pgz_next = 0, pgz_chain = 0, pgz_bodies = 0;   // chunks in all; next to hand out; stitched so far; resolved and filed so far
    uint64_t pgz_pos = 0, pgz_out = 0, pgz_total = 0;          // true bit position of the next chunk's first block; bytes produced so far; total (when known)
    bool pgz_eos = false, pgz_total_known = false;
    std::vector<uint8_t> pgz_window;                           // the 32 KiB in front of pgz_pos
    std::vector<std::vector<PgzPiece>> pgz_pieces;             // per chunk, for the CRC check at the end
    bool pgz_verified = false;
    std::string path;
    void (*free_fn)(void *);
    int fd = -1;
    gzFile gz = nullptr;
    size_t slice = 0, ring_n = 0, n_slices_known = 0, next_job = 0, released = 0;
    std::vector<uint8_t *> bufs;
    std::vector<Slot> slots;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, failed = false;
    std::string error;
};
// Plain FASTQ text -> packed records on the host (sgh_scan.cpp): the file is memory-mapped, `threads` workers take 4 MiB
// blocks in file order (list the line starts, learn the block's first line number from the blocks before it, verify the
// marker bytes, pack every sequence line's window into a record of sgc_format.h), the consumer takes the blocks' records
// in order (the workers run at most max_ahead_blocks ahead of it: ~110 KB of records per 4 MiB block of 150-base reads).  usable == false: not a regular, non-empty file that starts with '@' (gzip, FASTA, a pipe): use the other readers.
// hybrid libraries (a few guides with bytes outside ACGT): the Bloom filter of the shadow keys — the ACGT windows one substitution
// away from a guide with exactly one such byte (same hashing as the device's: sgc_format.h sgc_hash2 / sgc_bloom_*); may be empty
struct RouteFilter { std::vector<uint64_t> words; uint32_t log2_words = 0; };
RouteFilter make_route_filter(const std::vector<std::string> &seqs);
struct ScanParams {
    uint32_t L = 0; bool reverse = false; uint32_t offset = 0; bool recursion = true;
    const RouteFilter *route = nullptr;      // non-null: reads a non-ACGT guide could influence are set aside as bytes (next(): routed_*)
};
class FastqScanner {
  public:
    // source: where a block's bytes come from — 1 the mapping (page faults), 2 pread() into a buffer of the thread's own, 0 auto
    // (pread, unless the threads' average read rate shows the first-read penalty of freshly written pages: then the mapping)
    FastqScanner(const std::string &path, const ScanParams &prm, size_t threads, size_t block_bytes = 4u << 20, size_t max_ahead_blocks = 16384,
                 int source = 0);
    ~FastqScanner();
    // the records of the next block (n_records records of `words` u64 each), in file order; false = end of file.  Throws
    // Panic on a malformed record (wrong marker byte; a line count that is no multiple of 4 at the end).
    bool next(const uint64_t *&recs, size_t &n_records);
    void release();                     // the block returned by the last next() may be dropped
    // the reads of the block returned by the last next() that were set aside (ScanParams::route): n_routed reads, of read i the piece
    // that holds its windows = routed_bytes[routed_offs[i], routed_offs[i + 1]) (window_offset(): the sample's offset inside a piece)
    const uint8_t *routed_bytes = nullptr; const uint64_t *routed_offs = nullptr; size_t n_routed = 0;
    bool usable = true;
    // the file is ONE gzip stream (not BGZF): its chunks are decoded speculatively by the workers (sgh_inflate.cpp, as TextFeeder::run_pgz
    // does), resolved into a buffer of the thread's own, and packed there — no text leaves the thread, and nothing needs the device,
    // so a .gz sample inflates while the device starts up and the tables are built
    bool gz_mode = false;
    bool bgzf_mode = false;             // gz_mode, and every member announces its size (BGZF): members are independent streams, a chunk
                                        // is a run of whole members, inflated by zlib straight into text (no speculation, no window chain)
    size_t pgz_fallbacks = 0;           // chunks decoded in order after all (speculation failed)
    size_t file_size = 0, n_threads = 0, words = 1;
    uint64_t total_lines = 0;           // after next() returned false
    double busy_s = 0, wait_s = 0;      // Σ worker busy time; time the consumer waited for a block
  private:
    struct Block;
    void run();
    void run_gz();
    // packs the sequence lines among the first n_pack line starts of blk.starts (offsets from t + lo); t[offset] is readable for
    // offsets below t_hi, a line without a newline before text_end ends there
    void extract(Block &blk, const uint8_t *t, size_t lo, size_t t_hi, size_t n_pack, size_t text_end);
    // gz mode: the chain that stitches the chunks (A: deflate position and window; B: line numbers and the unfinished line)
    static constexpr size_t GZ_MAX_LINE = 4u << 20;
    size_t gz_chunk_bytes = 0, gz_next = 0, gz_chain = 0, gz_lines = 0, gz_pending = 0;      // gz_pending: deferred blank lines (run_gz)
    uint64_t gz_pos = 0;
    bool gz_eos = false, gz_first_known = false, gz_verified = false, gz_text_ends_nl = false;
    uint8_t gz_first_byte = 0;
    std::vector<uint8_t> gz_window, gz_carry;
    std::vector<size_t> bgzf_chunk_off;         // BGZF: file offset of the first member of every chunk (+ the end of the file)
    struct GzPiece { uint32_t crc; uint64_t len; bool member_end; uint32_t want_crc, want_isize; };
    std::vector<std::vector<GzPiece>> gz_pieces;
    static constexpr size_t READ_SLACK = 64u << 10;      // bytes read behind a block: the rest of its last line, and the window loads
    int source = 0;
    std::atomic<bool> auto_map{false};
    std::atomic<uint64_t> read_bytes{0}, read_ns{0};
  public:
    bool used_mapping() const { return source == 1 || auto_map.load(); }
    bool routes() const { return prm.route != nullptr; }
    uint32_t window_offset() const { return prm.offset >= 1 ? 1u : 0u; }      // of the pieces in routed_bytes (sgc_sample_push_windows)
  private:
    std::string path;
    ScanParams prm;
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t end = 0, block = 0, n_blocks = 0, ahead = 0;
    std::unique_ptr<Block[]> blocks;
    size_t next_block = 0, chain = 0, consumed = 0;
    uint64_t lines_so_far = 0;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, failed = false, checked_end = false;
    uint64_t gz_bad_line = 0;            // gzip input: the first line with a wrong marker byte — reported once the members' CRCs have been checked
    std::string error;
};
size_t usable_cpus();                                    // affinity mask capped by the cgroup CPU quota

int cli_main(int argc, char **argv);                     // main.rs:142-203; returns the process exit code

}  // namespace sgh
