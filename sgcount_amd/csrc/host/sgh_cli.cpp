// sgh_cli.cpp — command line of the MI355X sgRNA counter: the reference's flags (src/main.rs:54-108) and control
// flow (src/main.rs:142-203).  `sgcount-hip -l library.fa -i a.fq.gz b.fq.gz [-a 30] [-x] [-p] [-r] [-g g2s.txt] ...`
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

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

int cli_main(int argc, char **argv) {
    try {
        CountOptions opt;
        std::string genemap_path;
        bool have_offset = false, reverse = false, have_sub = false;
        size_t offset = 0, subsample = 5000;
        bool have_names = false;
        auto need = [&](int &i, const char *flag) -> std::string {
            if (i + 1 >= argc) throw Error(std::string("a value is required for '") + flag + "' but none was supplied");
            return argv[++i];
        };
        auto multi = [&](int &i, std::vector<std::string> &dst) {
            while (i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] != '\0')) dst.push_back(argv[++i]);
        };
        auto to_num = [&](const std::string &v, const char *flag) -> size_t {
            char *e = nullptr;
            const unsigned long long x = strtoull(v.c_str(), &e, 10);
            if (v.empty() || *e || v[0] == '-') throw Error("invalid value '" + v + "' for '" + flag + "': invalid digit found in string");
            return (size_t)x;
        };
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "-l" || a == "--library-path") opt.library_path = need(i, "--library-path <LIBRARY_PATH>");
            else if (a == "-i" || a == "--input-paths") multi(i, opt.input_paths);
            else if (a == "-n" || a == "--sample-names") { multi(i, opt.sample_names); have_names = true; }
            else if (a == "-o" || a == "--output-path") opt.output_path = need(i, "--output-path <OUTPUT_PATH>");
            else if (a == "-g" || a == "--genemap") genemap_path = need(i, "--genemap <GENEMAP>");
            else if (a == "-a" || a == "--offset") { offset = to_num(need(i, "--offset <OFFSET>"), "--offset <OFFSET>"); have_offset = true; }
            else if (a == "-p" || a == "--no-position-recursion") opt.position_recursion = false;
            else if (a == "-r" || a == "--reverse") reverse = true;
            else if (a == "-x" || a == "--exact") opt.exact = true;
            else if (a == "-s" || a == "--subsample") { subsample = to_num(need(i, "--subsample <SUBSAMPLE>"), "--subsample <SUBSAMPLE>"); have_sub = true; }
            else if (a == "-t" || a == "--threads") opt.threads = to_num(need(i, "--threads <THREADS>"), "--threads <THREADS>");
            else if (a == "-q" || a == "--quiet") opt.quiet = true;
            else if (a == "-z" || a == "--include-zero") opt.include_zero = true;
            else if (a == "--include-permutations") { /* BASELINE.json's name for the reference default; no-op */ }
            else if (a == "--pack") {
                const std::string v = need(i, "--pack");
                if (v != "host" && v != "device" && v != "fastq" && v != "scan") throw Error("invalid value '" + v + "' for '--pack'");
                opt.device_pack = v != "host";
                opt.device_parse = v == "fastq" || v == "scan";
                opt.host_scan = v == "scan";
            }
            else if (a == "--scan-threads") opt.scan_threads = to_num(need(i, "--scan-threads"), "--scan-threads");
            else if (a == "--scan-source") {
                const std::string v = need(i, "--scan-source");
                if (v != "auto" && v != "mmap" && v != "read") throw Error("invalid value '" + v + "' for '--scan-source'");
                opt.scan_source = v == "mmap" ? 1 : (v == "read" ? 2 : 0);
            }
            else if (a == "--scan-block-kb") opt.scan_block_bytes = to_num(need(i, "--scan-block-kb"), "--scan-block-kb") << 10;
            else if (a == "--io-threads") opt.io_threads = to_num(need(i, "--io-threads"), "--io-threads");
            else if (a == "--chunk-mb") opt.chunk_bytes = to_num(need(i, "--chunk-mb"), "--chunk-mb") << 20;
            else if (a == "--devices") opt.max_devices = to_num(need(i, "--devices"), "--devices");
            else if (a == "--stats-json") opt.stats_path = need(i, "--stats-json");
            else if (a == "--exit") { const std::string v = need(i, "--exit"); if (v != "fast" && v != "clean") throw Error("invalid value '" + v + "' for '--exit'"); }
            else if (a == "-h" || a == "--help") { fputs(USAGE, stdout); return 0; }
            else if (a == "-V" || a == "--version") { puts("sgcount-hip 0.1.0 (count path of sgcount 0.1.35)"); return 0; }
            else { fprintf(stderr, "error: unexpected argument '%s' found\n\n%s", a.c_str(), USAGE); return 2; }
        }
        (void)have_sub;
        if (opt.library_path.empty() || opt.input_paths.empty()) {
            fprintf(stderr, "error: the following required arguments were not provided:\n%s%s\n%s",
                    opt.library_path.empty() ? "  --library-path <LIBRARY_PATH>\n" : "",
                    opt.input_paths.empty() ? "  --input-paths <INPUT_PATHS>..." : "", USAGE);
            return 2;
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
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], "--exit") && !strcmp(argv[i + 1], "clean")) return rc;
    fflush(stdout); fflush(stderr);
    _exit(rc);
}
#endif
