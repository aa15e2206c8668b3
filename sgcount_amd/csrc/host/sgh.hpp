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
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
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
    bool device_parse = true;            // FASTQ inputs: ship text chunks, find record boundaries on the GPU
    size_t chunk_bytes = 64u << 20;      // text chunk size for device_parse
    size_t batch_reads = 1u << 20;
};
void count(const CountOptions &opt);                     // count.rs:74-148
// text path of count(): byte offset at which a chunk of FASTQ text ends on a whole 4-line record (exposed for tests)
size_t fastq_chunk_cut(const uint8_t *buf, size_t have, bool eof, const std::string &path);
int cli_main(int argc, char **argv);                     // main.rs:142-203; returns the process exit code

}  // namespace sgh
