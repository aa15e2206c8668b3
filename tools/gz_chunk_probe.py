#!/usr/bin/env python3
"""The .gz leg of the bench with different caps on the compressed chunk size of the scanner's gzip mode (experiment)."""
import json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgcount_amd import hostlib, synth
from sgcount_amd.workload import DeviceWorkload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
d = tempfile.mkdtemp(dir="/dev/shm")
wl = DeviceWorkload(1000, 100_000, 20)
lib = os.path.join(d, "lib.fa"); open(lib, "wb").write(synth.library_fasta(wl.lib_seqs))
fq = os.path.join(d, "s.fastq"); bench.write_fastq(wl, n, fq); wl.close()
size = os.path.getsize(fq); parts = 16; per = (size + parts - 1) // parts
gz = os.path.join(d, "s.fastq.gz")
procs = [subprocess.Popen(["bash", "-c", "tail -c +%d %s | head -c %d | gzip -1 > %s.%02d" % (k * per + 1, fq, min(size, (k + 1) * per) - k * per, gz, k)]) for k in range(parts)]
for p in procs: p.wait()
with open(gz, "wb") as o:
    for k in range(parts):
        with open("%s.%02d" % (gz, k), "rb") as f: shutil.copyfileobj(f, o, 1 << 24)
        os.remove("%s.%02d" % (gz, k))
os.remove(fq)
cli = hostlib.cli_path()
print("gz bytes", os.path.getsize(gz))
for kb in (0,):   # (the cap was an experiment: the shipped scanner uses 1 MiB chunks)
    env = dict(os.environ, SGH_EXPERIMENT_GZ_CHUNK_KB=str(kb))
    st = os.path.join(d, "st.json")
    t0 = time.perf_counter()
    subprocess.run([cli, "-l", lib, "-i", gz, "-a", "30", "-q", "-o", os.path.join(d, "o.tsv"), "--stats-json", st], check=True, env=env)
    w = time.perf_counter() - t0
    j = json.load(open(st)); s0 = j["samples"][0]
    print("chunk cap %5d KiB: wall %.3f  setup %.3f  sample %.3f  inflate busy %.2f CPU-s  in-order chunks %s" % (kb, w, j["setup_s"], s0["wall_s"], s0["read_busy_s"], s0.get("gzip_chunks_decoded_in_order")))
shutil.rmtree(d, ignore_errors=True)
