// valu_rates.hip — issue cost of the integer VALU instructions the playout kernel is made of, on gfx950.
// One wave per SIMD (1024 workgroups of 64), each running ITER x 64 copies of one instruction in 8 independent dependency
// chains; reports shader-clock cycles per wave-instruction (s_memtime).  Build + run: see tools/measure_valu_rates.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>

#define ITER 2000
// the snippets write vcc and s[20:25] freely: tell the compiler
#define CLOB "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25"

#define KERNEL(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(64) void NAME(unsigned* out, unsigned long long* cyc) {                          \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned b = blockIdx.x | 1u, c = 0x55aa55aau ^ a0;                                                       \
        unsigned long long t0 = __builtin_readcyclecounter();                                                    \
        for (int i = 0; i < ITER; ++i) {                                                                          \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                       \
                asm volatile(ASM(0) : "+v"(a0) : "v"(b), "v"(c) : CLOB); asm volatile(ASM(1) : "+v"(a1) : "v"(b), "v"(c) : CLOB);       \
                asm volatile(ASM(2) : "+v"(a2) : "v"(b), "v"(c) : CLOB); asm volatile(ASM(3) : "+v"(a3) : "v"(b), "v"(c) : CLOB);       \
                asm volatile(ASM(4) : "+v"(a4) : "v"(b), "v"(c) : CLOB); asm volatile(ASM(5) : "+v"(a5) : "v"(b), "v"(c) : CLOB);       \
                asm volatile(ASM(6) : "+v"(a6) : "v"(b), "v"(c) : CLOB); asm volatile(ASM(7) : "+v"(a7) : "v"(b), "v"(c) : CLOB);       \
            }                                                                                                     \
        }                                                                                                         \
        unsigned long long t1 = __builtin_readcyclecounter();                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

#define A_AND(k) "v_and_b32 %0, %0, %1"
#define A_OR3(k) "v_or3_b32 %0, %0, %1, %2"
#define A_BITOP3(k) "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96"
#define A_ALIGNBIT(k) "v_alignbit_b32 %0, %0, %1, %2"
#define A_BFREV(k) "v_bfrev_b32 %0, %0"
#define A_BCNT(k) "v_bcnt_u32_b32 %0, %0, %1"
#define A_FFBH(k) "v_ffbh_u32 %0, %0"
#define A_LSHL(k) "v_lshlrev_b32 %0, %1, %0"
#define A_ADD(k) "v_add_u32 %0, %0, %1"
#define A_SUBCO(k) "v_sub_co_u32 %0, vcc, %0, %1"
#define A_SUBB(k) "v_subb_co_u32 %0, vcc, %0, %1, vcc"
#define A_CNDMASK(k) "v_cndmask_b32 %0, %0, %1, vcc"
#define A_CMP(k) "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc"
#define A_MUL24(k) "v_mul_u32_u24 %0, %0, %1"
#define A_MAD24(k) "v_mad_u32_u24 %0, %0, %1, %2"
#define A_MULLO(k) "v_mul_lo_u32 %0, %0, %1"
#define A_MULHI(k) "v_mul_hi_u32 %0, %0, %1"
#define A_BFE(k) "v_bfe_u32 %0, %0, %1, 5"
#define A_ANDOR(k) "v_and_or_b32 %0, %0, %1, %2"
#define A_LSHLADD(k) "v_lshl_add_u32 %0, %0, 3, %1"
#define A_PERM(k) "v_perm_b32 %0, %0, %1, %2"
#define A_CND64(k) "v_cndmask_b32_e64 %0, %0, %1, s[20:21]"
#define A_CMP64(k) "v_cmp_lt_u32_e64 s[20:21], %0, %1"
#define A_CMPCND64(k) "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]"
#define A_CMPCND64B(k) "v_cmp_lt_u32_e64 s[22:23], %0, %1\n v_and_b32 %0, %0, %2\n v_or_b32 %0, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[22:23]"
#define A_BFI(k) "v_bfi_b32 %0, %1, %0, %2"
#define A_MOV(k) "v_mov_b32 %0, %1"
#define A_ASHR(k) "v_ashrrev_i32 %0, 31, %0"
#define A_XAD(k) "v_xad_u32 %0, %0, %1, %2"
#define A_ADD3(k) "v_add3_u32 %0, %0, %1, %2"
#define A_FFBL(k) "v_ffbl_b32 %0, %0"
#define A_MBCNT(k) "v_mbcnt_lo_u32_b32 %0, %1, %0"
#define A_READFL(k) "v_readfirstlane_b32 s20, %0"
#define A_OR(k) "v_or_b32 %0, %0, %1"
#define A_XOR(k) "v_xor_b32 %0, %0, %1"
#define A_NOT(k) "v_not_b32 %0, %0"
#define A_SUB(k) "v_sub_u32 %0, %0, %1"
#define A_LSHR(k) "v_lshrrev_b32 %0, %1, %0"
#define A_MAX(k) "v_max_u32 %0, %0, %1"
#define A_CMPE32(k) "v_cmp_lt_u32 vcc, %0, %1"
#define A_SVCC_CND(k) "s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %0, %0, %1, vcc"
#define A_SAND_CND(k) "s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]"
#define A_CMP_SAND_CND(k) "v_cmp_lt_u32_e64 s[22:23], %0, %1\n s_and_b64 s[24:25], s[22:23], s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[24:25]"
#define A_LSHLOR(k) "v_lshl_or_b32 %0, %0, 1, %1"
#define A_ADDCO(k) "v_add_co_u32 %0, vcc, %0, %1"
#define A_MOVLIT(k) "v_and_b32 %0, 0x12345678, %0"
#define A_ANDSGPR(k) "v_and_b32 %0, s20, %0"
#define A_CMP_GAP_CND(k) "v_cmp_lt_u32 vcc, %0, %1\n v_and_b32 %0, %0, %2\n v_or_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc"
#define A_CMP_SALU_CND(k) "v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32 %0, %0, %1, vcc"
#define A_CMP_2CND(k) "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc"
#define A_CMP64_SALUVCC_CND64(k) "v_cmp_lt_u32_e64 s[22:23], %0, %1\n s_and_b64 vcc, s[22:23], s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, vcc"
#define A_SUBB_CHAIN(k) "v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %0, vcc, %0, %2, vcc\n v_subb_co_u32 %0, vcc, %0, %1, vcc\n v_subb_co_u32 %0, vcc, %0, %2, vcc"

KERNEL(k_and, A_AND) KERNEL(k_or3, A_OR3) KERNEL(k_bitop3, A_BITOP3) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_bfrev, A_BFREV)
KERNEL(k_bcnt, A_BCNT) KERNEL(k_ffbh, A_FFBH) KERNEL(k_lshl, A_LSHL) KERNEL(k_add, A_ADD) KERNEL(k_subco, A_SUBCO) KERNEL(k_subb, A_SUBB)
KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp_cnd, A_CMP) KERNEL(k_mul24, A_MUL24) KERNEL(k_mad24, A_MAD24) KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI) KERNEL(k_bfe, A_BFE) KERNEL(k_andor, A_ANDOR) KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_perm, A_PERM)
KERNEL(k_cnd64, A_CND64) KERNEL(k_cmp64s, A_CMP64) KERNEL(k_cmpcnd64, A_CMPCND64) KERNEL(k_cmpcnd64b, A_CMPCND64B) KERNEL(k_bfi, A_BFI) KERNEL(k_mov, A_MOV)
KERNEL(k_ashr, A_ASHR) KERNEL(k_xad, A_XAD) KERNEL(k_add3, A_ADD3) KERNEL(k_ffbl, A_FFBL) KERNEL(k_readfl, A_READFL)
KERNEL(k_or, A_OR) KERNEL(k_xor, A_XOR) KERNEL(k_not, A_NOT) KERNEL(k_sub, A_SUB) KERNEL(k_lshr, A_LSHR) KERNEL(k_max, A_MAX) KERNEL(k_cmpe32, A_CMPE32)
KERNEL(k_svcc_cnd, A_SVCC_CND) KERNEL(k_sand_cnd, A_SAND_CND) KERNEL(k_cmp_sand_cnd, A_CMP_SAND_CND) KERNEL(k_lshlor, A_LSHLOR) KERNEL(k_addco, A_ADDCO) KERNEL(k_andlit, A_MOVLIT) KERNEL(k_andsgpr, A_ANDSGPR)
KERNEL(k_cmp_gap_cnd, A_CMP_GAP_CND) KERNEL(k_cmp_salu_cnd, A_CMP_SALU_CND) KERNEL(k_cmp_2cnd, A_CMP_2CND) KERNEL(k_cmp64_saluvcc, A_CMP64_SALUVCC_CND64) KERNEL(k_subb_chain, A_SUBB_CHAIN)

// 64-bit forms: one dependency chain per pair of registers
#define KERNEL64(NAME, ASM)                                                                                       \
    __global__ __launch_bounds__(64) void NAME(unsigned* out, unsigned long long* cyc) {                          \
        unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long b = blockIdx.x | 1u; unsigned c = 3u + (threadIdx.x & 7u);                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                                    \
        for (int i = 0; i < ITER; ++i) {                                                                          \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                       \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : CLOB); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : CLOB);     \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : CLOB); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : CLOB);     \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : CLOB); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : CLOB);     \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : CLOB); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : CLOB);     \
            }                                                                                                     \
        }                                                                                                         \
        unsigned long long t1 = __builtin_readcyclecounter();                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, %2, %0") KERNEL64(k_lshr64, "v_lshrrev_b64 %0, %2, %0") KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 1, %1")
KERNEL64(k_cmp64, "v_cmp_ne_u64 vcc, %0, %1")

typedef void (*kfn)(unsigned*, unsigned long long*);
struct Entry { const char* name; kfn f; int instr_per_slot; };

int main(int argc, char** argv) {
    const int WAVES = argc > 1 ? atoi(argv[1]) : 1;           // resident waves per SIMD
    const int G = 1024 * WAVES;
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(unsigned) * G * 64); hipMalloc(&cyc, sizeof(unsigned long long) * G);
    std::vector<Entry> es = {
        {"v_and_b32", k_and, 1}, {"v_or3_b32", k_or3, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_and_or_b32", k_andor, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_perm_b32", k_perm, 1}, {"v_bfrev_b32", k_bfrev, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_lshlrev_b32", k_lshl, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_add_u32", k_add, 1}, {"v_sub_co_u32", k_subco, 1}, {"v_subb_co_u32", k_subb, 1},
        {"v_cndmask_b32 (vcc)", k_cndmask, 1}, {"v_cmp_lt_u32 + v_cndmask_b32", k_cmp_cnd, 2}, {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1},
        {"v_cndmask_b32_e64 (sgpr pair, not rewritten)", k_cnd64, 1}, {"v_cmp_lt_u32_e64 -> sgpr pair", k_cmp64s, 1}, {"v_cmp_e64 + v_cndmask_e64 back to back", k_cmpcnd64, 2}, {"v_cmp_e64, 2 ALU, v_cndmask_e64", k_cmpcnd64b, 4}, {"v_bfi_b32", k_bfi, 1}, {"v_mov_b32", k_mov, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_xad_u32", k_xad, 1}, {"v_add3_u32", k_add3, 1}, {"v_ffbl_b32", k_ffbl, 1}, {"v_readfirstlane_b32", k_readfl, 1},
        {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_not_b32", k_not, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_max_u32", k_max, 1},
        {"v_cmp_lt_u32_e32 -> vcc", k_cmpe32, 1}, {"s_mov_b64 vcc + v_cndmask_b32 vcc (per VALU)", k_svcc_cnd, 1}, {"s_and_b64 + v_cndmask_e64 (per VALU)", k_sand_cnd, 1},
        {"v_cmp_e64, s_and_b64, v_cndmask_e64 (per VALU)", k_cmp_sand_cnd, 2}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_add_co_u32", k_addco, 1},
        {"v_cmp_e32 vcc, 3 fast ALU, v_cndmask_e32 vcc (per VALU)", k_cmp_gap_cnd, 5}, {"v_cmp_e32 vcc, s_and_b64 vcc, v_cndmask_e32 vcc (per VALU)", k_cmp_salu_cnd, 2},
        {"v_cmp_e32 vcc, 3 x v_cndmask_e32 vcc (per VALU)", k_cmp_2cnd, 4}, {"v_cmp_e64, s_and_b64 vcc, v_cndmask_e64 vcc (per VALU)", k_cmp64_saluvcc, 2},
        {"v_sub_co + 3 x v_subb_co chain (per VALU)", k_subb_chain, 4},
        {"v_and_b32 with 32-bit literal", k_andlit, 1}, {"v_and_b32 with sgpr", k_andsgpr, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1},
        {"v_lshl_add_u64", k_lshladd64, 1}, {"v_cmp_ne_u64", k_cmp64, 1}};
    std::vector<unsigned long long> h(G);
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    printf("{\"waves_per_simd\": %d, \"instructions\": {", WAVES);
    bool first = true;
    for (auto& e : es) {
        hipLaunchKernelGGL(e.f, dim3(G), dim3(64), 0, 0, out, cyc);   // warm-up (code load)
        hipEventRecord(ea, 0);
        hipLaunchKernelGGL(e.f, dim3(G), dim3(64), 0, 0, out, cyc);
        hipEventRecord(eb, 0);
        hipDeviceSynchronize();
        float ms = 0.f; hipEventElapsedTime(&ms, ea, eb);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * G, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        const double per = s / G / ((double)ITER * 64.0) / WAVES;   // SIMD ticks per wave-instruction with WAVES waves interleaved
        // wall clock: ns of SIMD time per wave-instruction = kernel time / (instructions per wave * waves per SIMD)
        const double ns = (double)ms * 1e6 / ((double)ITER * 64.0 * e.instr_per_slot) / WAVES;
        printf("%s\"%s\": [%.2f, %.3f]", first ? "" : ", ", e.name, per / e.instr_per_slot, ns);
        first = false;
    }
    printf("}, \"unit\": \"s_memtime ticks per issue slot (1 instruction, or the 2-instruction pair where named)\"}\n");
    return 0;
}
