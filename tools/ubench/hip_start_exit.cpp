// How long does the smallest possible HIP process live?  (hipSetDevice + one allocation + one kernel-free sync, then exit)
// hipcc -O2 -o hip_start_exit hip_start_exit.cpp ; time ./hip_start_exit [fast]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>
int main(int argc, char **argv) {
    auto t0 = std::chrono::steady_clock::now();
    void *p = nullptr;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&p, 1 << 20) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1;
    auto t1 = std::chrono::steady_clock::now();
    printf("init+malloc %.3f s\n", std::chrono::duration<double>(t1 - t0).count());
    fflush(stdout);
    if (argc > 1 && !strcmp(argv[1], "fast")) _exit(0);
    return 0;
}
