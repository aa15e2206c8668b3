#!/usr/bin/env python3
"""Why is the FIRST read of a freshly written /dev/shm file slow on the GPU box?  (round 3: bench e2e first run 2.9 s vs 1.1 s)
Writes GB of bytes into a tmpfs file the way bench.py does, then reads it with 8 pread threads several times, with and
without a pause, printing GB/s and /proc/meminfo deltas.  No GPU use."""
import os, sys, time, threading, subprocess

def meminfo(keys=("MemFree", "Cached", "Shmem", "ShmemHugePages", "Dirty", "Writeback", "AnonHugePages", "Active(file)", "Inactive(file)", "Active(anon)", "Inactive(anon)", "Unevictable")):
    d = {}
    for line in open("/proc/meminfo"):
        k, v = line.split(":")
        if k in keys:
            d[k] = int(v.split()[0]) // 1024
    return d

def read_all(path, threads=8, chunk=1 << 20):
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    per = (size + threads - 1) // threads
    def work(t):
        buf = bytearray(chunk)
        mv = memoryview(buf)
        off, hi = t * per, min(size, (t + 1) * per)
        while off < hi:
            n = os.preadv(fd, [mv[: min(chunk, hi - off)]], off)
            if n <= 0:
                break
            off += n
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    os.close(fd)
    return size / dt / 1e9

def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16
    d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    print(subprocess.run("mount | grep -E 'shm|tmpfs' | head; df -h %s; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/enabled; cat /proc/sys/kernel/numa_balancing; uname -r" % d, shell=True, capture_output=True, text=True).stdout)
    path = os.path.join(d, "probe_%d.bin" % os.getpid())
    block = os.urandom(1 << 20) * 64
    print("before write", meminfo())
    t0 = time.perf_counter()
    with open(path, "wb", buffering=0) as f:
        for _ in range(int(gb * 1e9 / len(block))):
            f.write(block)
    print("write %.1f GB/s" % (gb / (time.perf_counter() - t0)), meminfo())
    pause = float(sys.argv[3]) if len(sys.argv) > 3 else 0
    if pause:
        time.sleep(pause)
        print("after %.0f s pause" % pause, meminfo())
    for k in range(4):
        print("read %d: %.1f GB/s" % (k, read_all(path)), meminfo(), flush=True)
    # a second file, read FIRST by cat (single kernel reader), then by the threads
    path2 = path + ".2"
    with open(path2, "wb", buffering=0) as f:
        for _ in range(int(gb * 1e9 / len(block))):
            f.write(block)
    t0 = time.perf_counter(); subprocess.run("cat %s > /dev/null" % path2, shell=True); dt = time.perf_counter() - t0
    print("file 2: cat first read %.1f GB/s" % (gb / dt))
    t0 = time.perf_counter(); subprocess.run("cat %s > /dev/null" % path2, shell=True); dt = time.perf_counter() - t0
    print("file 2: cat second read %.1f GB/s" % (gb / dt))
    print("file 2: threads %.1f GB/s" % read_all(path2))
    # mmap first touch of a third file
    import mmap
    path3 = path + ".3"
    with open(path3, "wb", buffering=0) as f:
        for _ in range(int(gb * 1e9 / len(block))):
            f.write(block)
    fd = os.open(path3, os.O_RDONLY)
    size = os.path.getsize(path3)
    mm = mmap.mmap(fd, size, prot=mmap.PROT_READ)
    import numpy as np
    a = np.frombuffer(mm, dtype=np.uint64)
    per = len(a) // 8
    def work(t):
        a[t * per:(t + 1) * per:512].sum()     # one word per 4 KiB page
    for rep in range(2):
        ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        print("file 3: mmap page-touch pass %d: %.2f s (%.1f GB/s of mapping)" % (rep, time.perf_counter() - t0, size / (time.perf_counter() - t0) / 1e9))
    print("file 3: pread threads after mmap touch %.1f GB/s" % read_all(path3))
    del a; mm.close(); os.close(fd)
    for p in (path, path2, path3):
        os.remove(p)

main()
