// How many vector instructions does a gfx950 CU issue per cycle, by kind and by the number of waves per SIMD?
// DESIGN.md §6 argues that k_core is "issue-bound" at 0.86-1.07 vector instructions per CU-cycle; whether the ceiling is one per
// cycle per CU (each SIMD one wave64 instruction every four cycles) or two decides what "bound" means.  Every wave runs the
// same unrolled stream of independent instructions of one kind on eight registers; cycles from s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS(NAME, ASM)                                                                                     \
    template <> __device__ __forceinline__ void body<NAME>(uint32_t (&r)[8], uint64_t (&q)[8], uint32_t &s) { \
        _Pragma("unroll") for (int k = 0; k < 4; k++) { REP8(ASM) }                                        \
    }
enum { ADD, XOR, LSHL, MUL24, MUL32, CNDMASK, AND_OR, SHR64, CMP64, POPC, MIX, SALU_MIX, AND, OR, LSHR, BFE, ADD3, LSHLOR, LSHLADD, CMP32, MOV, SUB, MAD24, CND64, CMPCND, MINU, ANDLIT, XOR3, PERM, ADDCO, SHL64, XAD };
template <int OP> __device__ __forceinline__ void body(uint32_t (&r)[8], uint64_t (&q)[8], uint32_t &s);
#define A_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_MUL32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
#define A_SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
#define A_CMP64(i) asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 3) & 7]) : "vcc");
#define A_POPC(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_MIX(i) asm volatile("v_xor_b32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(r[i]), "+s"(s) : "v"(r[(i + 3) & 7]) : "scc");
#define A_SALU(i) asm volatile("s_add_u32 %0, %0, 1\n s_lshl_b32 %0, %0, 1" : "+s"(s) : : "scc");
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_LSHR(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i]));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(r[i]));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_CMP32(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(r[i]), "v"(r[(i + 3) & 7]) : "vcc");
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
#define A_CND64(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[4:5]" : "+v"(r[i]) : "v"(r[(i + 3) & 7]) : "s4", "s5");
#define A_CMPCND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 3) & 7]) : "vcc");
#define A_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 3) & 7]));
#define A_ANDLIT(i) asm volatile("v_and_b32 %0, 0x3ffff0f, %0" : "+v"(r[i]));
#define A_XOR3(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
#define A_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(r[i]), "+v"(r[(i + 1) & 7]) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]), "v"(r[(i+6)&7]) : "vcc");
#define A_SHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
OPS(AND, A_AND) OPS(OR, A_OR) OPS(LSHR, A_LSHR) OPS(BFE, A_BFE) OPS(ADD3, A_ADD3) OPS(LSHLOR, A_LSHLOR) OPS(LSHLADD, A_LSHLADD) OPS(CMP32, A_CMP32)
OPS(MOV, A_MOV) OPS(SUB, A_SUB) OPS(MAD24, A_MAD24) OPS(CND64, A_CND64) OPS(CMPCND, A_CMPCND) OPS(MINU, A_MINU) OPS(ANDLIT, A_ANDLIT) OPS(XOR3, A_XOR3)
OPS(PERM, A_PERM) OPS(SHL64, A_SHL64)
OPS(ADD, A_ADD) OPS(XOR, A_XOR) OPS(LSHL, A_LSHL) OPS(MUL24, A_MUL24) OPS(MUL32, A_MUL32) OPS(CNDMASK, A_CND) OPS(AND_OR, A_ANDOR)
OPS(SHR64, A_SHR64) OPS(CMP64, A_CMP64) OPS(POPC, A_POPC) OPS(MIX, A_MIX) OPS(SALU_MIX, A_SALU)

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t iters, uint32_t *out, unsigned long long *cyc) {
    uint32_t r[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 2654435761u + i; q[i] = ((uint64_t)r[i] << 20) | i; }
    uint32_t s = __builtin_amdgcn_readfirstlane(blockIdx.x);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) body<OP>(r, q, s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = s;
    for (int i = 0; i < 8; i++) acc ^= r[i] ^ (uint32_t)q[i];
    if (acc == 0x12345678u) out[0] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> static void run(const char *name, int per_it) {
    uint32_t *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, 4));
    CK(hipMalloc(&cyc, 256 * 16 * 8));
    const uint32_t iters = 20000;
    printf("%-10s", name);
    for (int threads : {256, 1024, 2048}) {          // 1, 2, 4, 8 waves per SIMD (2048: two workgroups of 1024 per CU)
        const int wgs = threads == 2048 ? 512 : 256, tpb = threads == 2048 ? 1024 : threads;
        CK(hipMemset(cyc, 0, 256 * 16 * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(tpb), 0, 0, iters, out, cyc);       // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(tpb), 0, 0, iters, out, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(256 * 16);
        CK(hipMemcpy(h.data(), cyc, 256 * 16 * 8, hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        const double instr_per_cu = (double)iters * per_it * (threads / 64);
        // s_memtime counts at a constant 100 MHz on this part?  print both views: per s_memtime tick and per ns
        printf("  %4d thr/CU: %.3f instr/CU/ns", threads, instr_per_cu / (ms * 1e6));
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    run<ADD>("add", 32); run<XOR>("xor", 32); run<LSHL>("lshl", 32); run<MUL24>("mul24", 32); run<MUL32>("mul32", 32);
    run<CNDMASK>("cndmask", 32); run<AND_OR>("and_or", 32); run<SHR64>("shr64", 32); run<CMP64>("cmp64", 32); run<POPC>("bcnt", 32);
    run<AND>("and", 32); run<OR>("or", 32); run<LSHR>("lshr", 32); run<BFE>("bfe", 32); run<ADD3>("add3", 32); run<LSHLOR>("lshl_or", 32);
    run<LSHLADD>("lshl_add", 32); run<CMP32>("cmp32", 32); run<MOV>("mov", 32); run<SUB>("sub", 32); run<MAD24>("mad24", 32);
    run<CND64>("cnd_sgpr", 32); run<CMPCND>("cmp+cnd", 32); run<MINU>("min_u32", 32); run<ANDLIT>("and_lit", 32); run<XOR3>("xad", 32);
    run<PERM>("perm", 32); run<SHL64>("shl64", 32);
    run<MIX>("v+s pair", 32); run<SALU_MIX>("salu x2", 64);
    return 0;
}
