// sgh_cli.cpp — command line of the MI355X sgRNA counter: the reference's flags (src/main.rs:54-108) and control
// flow (src/main.rs:142-203).  `sgcount-hip -l library.fa -i a.fq.gz b.fq.gz [-a 30] [-x] [-p] [-r] [-g g2s.txt] ...`
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "sgh.hpp"

namespace sgh {

static const char *USAGE =
    "A fast and flexible sgRNA counter (MI355X count path)\n\n"
    "Usage: sgcount-hip [OPTIONS] --library-path <LIBRARY_PATH> --input-paths <INPUT_PATHS>...\n\n"
    "Options:\n"
    "  -l, --library-path <LIBRARY_PATH>     Filepath of the library\n"
    "  -i, --input-paths <INPUT_PATHS>...    Filepath(s) of fastx (fastq, fasta, *.gz) sequences to map\n"
    "  -n, --sample-names <SAMPLE_NAMES>...  Sample Names\n"
    "  -o, --output-path <OUTPUT_PATH>       Output filepath [default: stdout]\n"
    "  -g, --genemap <GENEMAP>               Gene to sgRNA mapping\n"
    "  -a, --offset <OFFSET>                 Adapter Offset\n"
    "  -p, --no-position-recursion           Remove Position Recursion (i.e. offseting sequences by +/- 1 on mismatch condition)\n"
    "  -r, --reverse                         Read Direction (reverse complement reads)\n"
    "  -x, --exact                           Disallow One Off Mismatch\n"
    "  -s, --subsample <SUBSAMPLE>           Number of Reads to Subsample in Determining Offset [default: 5000]\n"
    "  -t, --threads <THREADS>               Number of Threads to Use for Parallel Jobs [default: 1]\n"
    "  -q, --quiet                           Does not show progress\n"
    "  -z, --include-zero                    Include zero count sgRNAs in output table\n"
    "      --pack <scan|fastq|device|host>   Where reads are parsed/packed: scan = the host scans the memory-mapped text with\n"
    "                                        several threads and ships packed records (plain FASTQ; anything else falls\n"
    "                                        through to fastq); fastq = the text is shipped and parsed on the GPU; device =\n"
    "                                        record reader + GPU packer; host = record reader + host packer [default: scan]\n"
    "      --scan-threads <N>                Scanner threads per sample for --pack scan [default: min(16, cpus - 1)/threads]\n"
    "      --scan-source <mmap|read|auto>    How the scanner reads the file: through a memory mapping (as fast on a file written a\n"
    "                                        moment ago as on any other), with pread() into the threads' buffers (~15 % faster on\n"
    "                                        pages that were read before, several times slower on fresh ones), or pread() until\n"
    "                                        it proves slow [default: mmap]\n"
    "      --io-threads <N>                  Reader threads per sample for --pack fastq [default: min(8, cores/threads)]\n"
    "      --chunk-mb <MB>                   Text per upload [default: 64]\n"
    "      --devices <N>                     Use at most N of the visible GPUs [default: all]\n"
    "      --stats-json <PATH>               Write per-stage timings of the run as JSON\n"
    "      --exit <fast|clean>               fast: _exit() once everything is written (skips the HIP runtime's teardown);\n"
    "                                        clean: return through exit() [default: fast]\n"
    "  -h, --help                            Print help\n"
    "  -V, --version                         Print version\n";

static bool file_exists(const std::string &p) { std::ifstream f(p); return f.good(); }

// The argument forms clap 4 accepts for the reference's `Args` (src/main.rs:54-108, clap 4.5 derive): `--long value`, `--long=value`,
// `-s value`, `-svalue`, `-s=value`, bundled shorts (`-xzq`, `-xa 5`, `-xa5`), `--` (no positionals exist: whatever follows is an
// unexpected argument), options with num_args = 1.. (`-i a b -i c`), and clap's errors (exit code 2): a single-valued option or a
// flag given twice, a value attached to a flag, a missing value, an unknown argument.
namespace {
enum Kind { FLAG, ONE, MANY };
struct Spec { char s; const char *l; Kind kind; const char *shown; };      // shown: how clap names the argument in its messages
const Spec SPECS[] = {
    {'l', "library-path", ONE, "--library-path <LIBRARY_PATH>"}, {'i', "input-paths", MANY, "--input-paths <INPUT_PATHS>..."},
    {'n', "sample-names", MANY, "--sample-names <SAMPLE_NAMES>..."}, {'o', "output-path", ONE, "--output-path <OUTPUT_PATH>"},
    {'g', "genemap", ONE, "--genemap <GENEMAP>"}, {'a', "offset", ONE, "--offset <OFFSET>"},
    {'p', "no-position-recursion", FLAG, "--no-position-recursion"}, {'r', "reverse", FLAG, "--reverse"}, {'x', "exact", FLAG, "--exact"},
    {'s', "subsample", ONE, "--subsample <SUBSAMPLE>"}, {'t', "threads", ONE, "--threads <THREADS>"}, {'q', "quiet", FLAG, "--quiet"},
    {'z', "include-zero", FLAG, "--include-zero"}, {'h', "help", FLAG, "--help"}, {'V', "version", FLAG, "--version"},
    // not in the reference: BASELINE.json's name for its default, and the knobs of this implementation (long form only)
    {0, "include-permutations", FLAG, "--include-permutations"}, {0, "pack", ONE, "--pack"}, {0, "scan-threads", ONE, "--scan-threads"},
    {0, "scan-source", ONE, "--scan-source"}, {0, "scan-block-kb", ONE, "--scan-block-kb"}, {0, "io-threads", ONE, "--io-threads"},
    {0, "chunk-mb", ONE, "--chunk-mb"}, {0, "devices", ONE, "--devices"}, {0, "stats-json", ONE, "--stats-json"}, {0, "exit", ONE, "--exit"},
    {0, "dry-run-args", FLAG, "--dry-run-args"},
};
struct UsageError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Occurrence { const Spec *spec; std::vector<std::string> values; };
const Spec *find_long(const std::string &name) { for (const Spec &sp : SPECS) if (name == sp.l) return &sp; return nullptr; }
const Spec *find_short(char c) { for (const Spec &sp : SPECS) if (sp.s && sp.s == c) return &sp; return nullptr; }

// argv -> the occurrences of the arguments, in order (one per appearance; a MANY occurrence holds the values that followed it)
std::vector<Occurrence> tokenize(int argc, char **argv) {
    std::vector<Occurrence> occ;
    const Spec *pending = nullptr;          // an option still taking values: ONE with none yet, or MANY
    auto close_pending = [&]() {
        if (pending && occ.back().values.empty()) throw UsageError(std::string("a value is required for '") + pending->shown + "' but none was supplied");
        pending = nullptr;
    };
    auto open = [&](const Spec *sp, bool has_attached, const std::string &attached) {
        close_pending();
        occ.push_back({sp, {}});
        if (sp->kind == FLAG) {
            if (has_attached) throw UsageError("unexpected value '" + attached + "' for '" + sp->shown + "' found; no more were expected");
            return;
        }
        if (has_attached) { occ.back().values.push_back(attached); if (sp->kind == MANY) pending = sp; }
        else pending = sp;
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--") {                    // ends the values of a pending option; there are no positionals to take what follows
            close_pending();
            if (i + 1 < argc) throw UsageError(std::string("unexpected argument '") + argv[i + 1] + "' found");
            break;
        }
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            const Spec *sp = find_long(name);
            if (!sp) throw UsageError("unexpected argument '--" + name + "' found");
            open(sp, eq != std::string::npos, eq == std::string::npos ? "" : a.substr(eq + 1));
            continue;
        }
        if (a.size() > 1 && a[0] == '-') {  // a bundle of shorts; the first one that takes a value takes the rest of the bundle
            for (size_t k = 1; k < a.size(); k++) {
                const Spec *sp = find_short(a[k]);
                if (!sp) throw UsageError(k == 1 ? "unexpected argument '" + a + "' found" : std::string("unexpected argument '-") + a[k] + "' found");
                if (sp->kind == FLAG) {
                    if (k + 1 < a.size() && a[k + 1] == '=') throw UsageError("unexpected value '" + a.substr(k + 2) + "' for '" + sp->shown + "' found; no more were expected");
                    open(sp, false, "");
                    continue;
                }
                std::string rest = a.substr(k + 1);
                if (!rest.empty() && rest[0] == '=') rest.erase(0, 1);
                open(sp, k + 1 < a.size(), rest);
                break;
            }
            continue;
        }
        if (!pending) throw UsageError("unexpected argument '" + a + "' found");
        occ.back().values.push_back(a);
        if (pending->kind == ONE) pending = nullptr;
    }
    close_pending();
    // clap: only arguments that append (Vec) may appear more than once
    for (size_t x = 0; x < occ.size(); x++)
        for (size_t y = 0; y < x; y++)
            if (occ[x].spec == occ[y].spec && occ[x].spec->kind != MANY)
                throw UsageError(std::string("the argument '") + occ[x].spec->shown + "' cannot be used multiple times");
    return occ;
}
std::string json_str(const std::string &v) {
    std::string o = "\"";
    for (char c : v) { if (c == '"' || c == '\\') o += '\\'; o += c; }
    return o + "\"";
}
const char *USAGE_LINE = "Usage: sgcount-hip [OPTIONS] --library-path <LIBRARY_PATH> --input-paths <INPUT_PATHS>...\n\nFor more information, try '--help'.\n";
}  // namespace

int cli_main(int argc, char **argv) {
    try {
        CountOptions opt;
        std::string genemap_path;
        bool have_offset = false, reverse = false, have_sub = false, dry_run = false;
        size_t offset = 0, subsample = 5000;
        bool have_names = false;
        auto to_num = [&](const std::string &v, const char *flag) -> size_t {
            char *e = nullptr;
            const unsigned long long x = strtoull(v.c_str(), &e, 10);
            if (v.empty() || *e || v[0] == '-' || v[0] == '+' || isspace((unsigned char)v[0]))
                throw UsageError("invalid value '" + v + "' for '" + flag + "': " + (v.empty() ? "cannot parse integer from empty string" : "invalid digit found in string"));
            return (size_t)x;
        };
        std::vector<Occurrence> occ;
        try {
            occ = tokenize(argc, argv);
            // --help / --version win over everything else on the line, as in clap
            for (const Occurrence &o : occ) {
                if (!strcmp(o.spec->l, "help")) { fputs(USAGE, stdout); return 0; }
                if (!strcmp(o.spec->l, "version")) { puts("sgcount-hip 0.1.0 (count path of sgcount 0.1.35)"); return 0; }
            }
            for (const Occurrence &o : occ) {
                const std::string n = o.spec->l, v = o.values.empty() ? "" : o.values[0];
                if (n == "library-path") opt.library_path = v;
                else if (n == "input-paths") opt.input_paths.insert(opt.input_paths.end(), o.values.begin(), o.values.end());
                else if (n == "sample-names") { opt.sample_names.insert(opt.sample_names.end(), o.values.begin(), o.values.end()); have_names = true; }
                else if (n == "output-path") opt.output_path = v;
                else if (n == "genemap") genemap_path = v;
                else if (n == "offset") { offset = to_num(v, o.spec->shown); have_offset = true; }
                else if (n == "no-position-recursion") opt.position_recursion = false;
                else if (n == "reverse") reverse = true;
                else if (n == "exact") opt.exact = true;
                else if (n == "subsample") { subsample = to_num(v, o.spec->shown); have_sub = true; }
                else if (n == "threads") opt.threads = to_num(v, o.spec->shown);
                else if (n == "quiet") opt.quiet = true;
                else if (n == "include-zero") opt.include_zero = true;
                else if (n == "include-permutations") { /* BASELINE.json's name for the reference default; no-op */ }
                else if (n == "pack") {
                    if (v != "host" && v != "device" && v != "fastq" && v != "scan") throw UsageError("invalid value '" + v + "' for '--pack'");
                    opt.device_pack = v != "host";
                    opt.device_parse = v == "fastq" || v == "scan";
                    opt.host_scan = v == "scan";
                }
                else if (n == "scan-threads") opt.scan_threads = to_num(v, "--scan-threads");
                else if (n == "scan-source") {
                    if (v != "auto" && v != "mmap" && v != "read") throw UsageError("invalid value '" + v + "' for '--scan-source'");
                    opt.scan_source = v == "mmap" ? 1 : (v == "read" ? 2 : 0);
                }
                else if (n == "scan-block-kb") opt.scan_block_bytes = to_num(v, "--scan-block-kb") << 10;
                else if (n == "io-threads") opt.io_threads = to_num(v, "--io-threads");
                else if (n == "chunk-mb") opt.chunk_bytes = to_num(v, "--chunk-mb") << 20;
                else if (n == "devices") opt.max_devices = to_num(v, "--devices");
                else if (n == "stats-json") opt.stats_path = v;
                else if (n == "exit") { if (v != "fast" && v != "clean") throw UsageError("invalid value '" + v + "' for '--exit'"); }
                else if (n == "dry-run-args") dry_run = true;
            }
        } catch (const UsageError &e) {
            fprintf(stderr, "error: %s\n\n%s", e.what(), USAGE_LINE);
            return 2;
        }
        (void)have_sub;
        if (opt.library_path.empty() || opt.input_paths.empty()) {
            fprintf(stderr, "error: the following required arguments were not provided:\n%s%s\n%s",
                    opt.library_path.empty() ? "  --library-path <LIBRARY_PATH>\n" : "",
                    opt.input_paths.empty() ? "  --input-paths <INPUT_PATHS>..." : "", USAGE);
            return 2;
        }
        if (dry_run) {
            // what the line parsed to, and nothing else (no file is opened, no device): tests/test_host_cpu.py's table of argument forms
            std::string j = "{\"library_path\": " + json_str(opt.library_path) + ", \"input_paths\": [";
            for (size_t i = 0; i < opt.input_paths.size(); i++) j += (i ? ", " : "") + json_str(opt.input_paths[i]);
            j += "], \"sample_names\": ";
            if (have_names) { j += "["; for (size_t i = 0; i < opt.sample_names.size(); i++) j += (i ? ", " : "") + json_str(opt.sample_names[i]); j += "]"; }
            else j += "null";
            j += ", \"output_path\": " + json_str(opt.output_path) + ", \"genemap\": " + json_str(genemap_path) +
                 ", \"offset\": " + (have_offset ? std::to_string(offset) : "null") + ", \"subsample\": " + (have_sub ? std::to_string(subsample) : "null") +
                 ", \"threads\": " + std::to_string(opt.threads) + ", \"no_position_recursion\": " + (opt.position_recursion ? "false" : "true") +
                 ", \"reverse\": " + (reverse ? "true" : "false") + ", \"exact\": " + (opt.exact ? "true" : "false") +
                 ", \"quiet\": " + (opt.quiet ? "true" : "false") + ", \"include_zero\": " + (opt.include_zero ? "true" : "false") + "}";
            puts(j.c_str());
            return 0;
        }
        for (const auto &p : opt.input_paths)                                          // main.rs:130-140 validate_paths
            if (!file_exists(p)) throw Panic("Provided filepath does not exist: " + p);
        if (have_names) {                                                              // main.rs:151-160
            if (opt.sample_names.size() != opt.input_paths.size())
                throw Panic("Must provide as many sample names as there are input files");
        } else {
            opt.sample_names = generate_sample_names(opt.input_paths);
        }
        if (have_offset) {                                                             // main.rs:163-170
            Offset o; o.reverse = reverse; o.index = offset;
            opt.offsets.assign(opt.input_paths.size(), o);
        } else {                                                                       // main.rs:171-176 (-r is ignored here)
            if (!opt.quiet) fprintf(stderr, "Calculating Offset\n");
            opt.offsets = entropy_offset_group(opt.library_path, opt.input_paths, subsample);
            if (!opt.quiet) {
                std::string s = "[";
                for (size_t i = 0; i < opt.offsets.size(); i++) s += (i ? ", " : "") + opt.offsets[i].debug();
                fprintf(stderr, "Calculated Offsets: %s]\n", s.c_str());
            }
        }
        GeneMap gm;
        if (!genemap_path.empty()) { gm = GeneMap::from_path(genemap_path); opt.genemap = &gm; }   // main.rs:180-183
        count(opt);                                                                    // main.rs:189-200
        return 0;
    } catch (const Panic &e) {
        fprintf(stderr, "thread 'main' panicked:\n%s\n", e.what());
        return 101;
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}

}  // namespace sgh

#ifndef SGH_NO_MAIN
#include <malloc.h>
#include <unistd.h>
int main(int argc, char **argv) {
    // The scanners allocate and free a few hundred KB per block of text, thousands of times, on 15 threads: by default glibc
    // serves every such request with an mmap of its own and gives it back with munmap — page faults on fresh pages every time,
    // and the address-space lock taken for writing beside the HIP runtime, which maps memory all through its start-up.  Keep
    // that memory in the heaps instead.
    // (100M reads, four runs each, alternating: 0.58-0.62 s against 0.56-0.69 with the defaults)
    mallopt(M_MMAP_THRESHOLD, 256 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    const int rc = sgh::cli_main(argc, argv);
    // Everything observable is written and closed by now, the contexts are freed: end the process without the user-space
    // teardown of the HIP runtime (static destructors, atexit handlers: 20-100 ms) — the kernel releases the device either
    // way.  `--exit clean` returns through exit() instead (sanitizers, leak checkers).
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], "--exit=clean") || (i + 1 < argc && !strcmp(argv[i], "--exit") && !strcmp(argv[i + 1], "clean"))) return rc;
    fflush(stdout); fflush(stderr);
    _exit(rc);
}
#endif
