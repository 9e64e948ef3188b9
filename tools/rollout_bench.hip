// rollout_bench.hip — standalone timing harness for the playout loop (the dominant code of k_mcts_fused / k_rollout).
// Not part of the product library: it compiles the same device headers into ONE kernel so that a source or flag variant
// builds in seconds, and runs it with 1..4 resident waves per SIMD (65 536 x WAVES games from the Copenhagen start
// position) to measure what occupancy and instruction mix buy.  Results are checked against a checksum so that variants
// can be compared for equality as well as speed.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I alphazeroforhnefatafl_amd/csrc [-DLB=4] [-DVARIANT=..] \
//         -mllvm --amdgpu-sched-strategy=max-ilp -o /tmp/rollout_bench tools/rollout_bench.hip
//   /tmp/rollout_bench [board: 11|13|7] [max_plies=512] [reps=3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "tafl_host.hpp"
#include "tafl_ops.hpp"
using namespace tafl;

#ifndef LB
#define LB 1
#endif
#ifndef EXTRA_GRID
#define EXTRA_GRID 1
#endif
#ifdef TAFL_PROF
// -DTAFL_PROF: per-section shader-clock totals of the playout loop (tafl_bits.hpp TAFL_PROF_*): where the cycles of one ply go
extern "C" __device__ unsigned long long tafl_prof_acc[4096 * 32] = {};
static const char* const PROF_NAME[12] = {"pick", "captures: custodial fields", "captures: king", "captures: shieldwall filter", "removal + repetition tracker", "T-layout upkeep",
                                          "gen (next mover's four reach sets)", "outcome + finish", "  of which enclosure flood", "  of which exit fort", "empty mark", "empty mark"};
#endif

template <int NL, int W, int PRESET>
__global__ __launch_bounds__(64, LB) void k_roll(const Quad* soa, uint32_t n, uint64_t seed, uint32_t sim, uint32_t max_plies, uint64_t base,
                                                 tafl_rollout_result* out) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= n) return;                                  // (-DEXTRA_GRID=k launches k x the blocks needed: the surplus leaves here)
    constexpr Consts<NL> C = preset_consts<NL, W, PRESET>();
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    tafl_rollout_result r;
    Ops<NL, W>::rollout(st, seed, base + g, sim, max_plies, C, r);
    out[g] = r;
}
template <int NL>
__global__ void k_fill(Quad* soa, uint32_t n, DState<NL> st) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g < n) StateIO<NL>::store_soa(soa, n, g, st);
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int NL, int W, int PRESET>
int run(const char* board, uint32_t word_bits, uint32_t max_plies, int reps) {
    tafl_state st; std::string err;
    if (fen_to_state(preset_board(board), 0, word_bits, &st, &err)) { fprintf(stderr, "fen: %s\n", err.c_str()); return 1; }
    DState<NL> ds; state_from_abi<NL>(st, ds);
    for (int waves = 1; waves <= (LB > 4 ? LB : 4); ++waves) {
        const uint32_t n = 65536u * (uint32_t)waves;
        Quad* soa; tafl_rollout_result* out;
        CK(hipMalloc(&soa, sizeof(Quad) * StateIO<NL>::QUADS * n)); CK(hipMalloc(&out, sizeof(tafl_rollout_result) * n));
        hipLaunchKernelGGL((k_fill<NL>), dim3(n / 64), dim3(64), 0, 0, soa, n, ds);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL((k_roll<NL, W, PRESET>), dim3(n / 64 * EXTRA_GRID), dim3(64), 0, 0, soa, n, 3ull, 0u, max_plies, 0ull, out);
            CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r > 0 && ms < best) best = ms;
        }
        std::vector<tafl_rollout_result> h(n);
        CK(hipMemcpy(h.data(), out, sizeof(tafl_rollout_result) * n, hipMemcpyDeviceToHost));
        unsigned long long plies = 0, sum = 0, capped = 0;
        for (uint32_t g = 0; g < n; ++g) { plies += h[g].plies; sum = sum * 1000003ull + (unsigned long long)(h[g].plies * 16u + h[g].reason) + (unsigned long long)(h[g].value + 2); capped += h[g].reason == 14; }
        // checksum over the first 65 536 games only would be waves-independent; print both
        unsigned long long sum0 = 0; for (uint32_t g = 0; g < 65536u; ++g) sum0 = sum0 * 1000003ull + (unsigned long long)(h[g].plies * 16u + h[g].reason) + (unsigned long long)(h[g].value + 2);
#ifdef TAFL_PROF
        {
            static unsigned long long hp[4096 * 32];
            CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(tafl_prof_acc), sizeof hp));
            unsigned long long tot[32] = {0};
            for (int w = 0; w < 4096; ++w) for (int k = 0; k < 32; ++k) tot[k] += hp[w * 32 + k];
            const double iters = (double)tot[31];                  // wave-iterations of the loop over all timed launches
            const double mark = (double)(long long)tot[10] / iters;   // cost of one mark pair
            printf("# waves_per_simd %d: shader-clock ticks per wave-iteration (100 MHz clock; one mark costs %.2f ticks, subtracted)\n", waves, mark);
            double sum = 0;
            for (int k = 0; k < 10; ++k) { const double v = (double)(long long)tot[k] / iters - ((k < 8) ? mark : 0.0); if (k < 8) sum += v; printf("#   %-38s %8.2f\n", PROF_NAME[k], v); }
            printf("#   %-38s %8.2f\n", "sum of sections 0-7", sum);
            memset(hp, 0, sizeof hp); CK(hipMemcpyToSymbol(HIP_SYMBOL(tafl_prof_acc), hp, sizeof hp));
        }
#endif
        printf("{\"board\": \"%s\", \"lb\": %d, \"waves_per_simd\": %d, \"games\": %u, \"ms\": %.4f, \"plies\": %llu, \"Gplies_per_s\": %.3f, "
               "\"ns_per_wave_ply_per_simd\": %.3f, \"capped_frac\": %.3f, \"checksum64k\": \"%016llx\"}\n",
               board, LB, waves, n, best, plies, plies / (best * 1e6), best * 1e6 / ((double)max_plies * waves), (double)capped / n, sum0);
        fflush(stdout);
        CK(hipFree(soa)); CK(hipFree(out));
    }
    return 0;
}

int main(int argc, char** argv) {
    const int board = argc > 1 ? atoi(argv[1]) : 11;
    const uint32_t max_plies = argc > 2 ? (uint32_t)atoi(argv[2]) : 512u;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    if (board == 11) return run<4, 11, PRESET_COPENHAGEN11>("copenhagen", 128, max_plies, reps);
    if (board == 13) return run<8, 15, PRESET_COPENHAGEN13>("copenhagen13", 256, max_plies, reps);
    if (board == 7) return run<2, 7, PRESET_BRANDUBH7>("brandubh", 64, max_plies, reps);
    fprintf(stderr, "board must be 7, 11 or 13\n");
    return 2;
}
