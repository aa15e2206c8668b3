import subprocess, time
for arg in ([], ["fast"]):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); subprocess.run(["tools/ubench/bin/hip_start_exit"] + arg, capture_output=True); ts.append(time.perf_counter() - t0)
    print(arg, ["%.3f" % x for x in ts])
