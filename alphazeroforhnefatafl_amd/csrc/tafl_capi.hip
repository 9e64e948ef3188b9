// tafl_capi.hip — HIP kernels (gfx950) + the C-ABI of include/taflhip.h.
//
// Execution shape: ONE GAME PER LANE, 64-lane workgroups (one wavefront), so that a 65 536-game
// batch is 1 024 waves = one wave per SIMD on the 256 CUs x 4 SIMDs of an MI355X.  Whole game
// states live in VGPRs for the duration of a kernel (a random playout never touches HBM between its
// first load and its final 1-byte result).  Batch states are quad-plane SoA in HBM (16 B per lane per
// load, 1 KiB per wave instruction); tree nodes are 64-B records (DESIGN.md "Data layout in HBM").
// No CPU fallback exists: every compute entry point launches kernels or fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "tafl_host.hpp"
#include "tafl_ops.hpp"
#include "tafl_guided.hpp"

using namespace tafl;

// --------------------------------------------------------------------------------------------------
// kernels
// --------------------------------------------------------------------------------------------------
#define TAFL_BLOCK 64
#ifdef TAFL_PROF
// profiling builds only (never the product library): totals of the TAFL_PROF_* section timers
extern "C" __device__ unsigned long long tafl_prof_acc[4096 * 32] = {};
extern "C" int tafl_prof_read(unsigned long long* out, int reset) {
    static unsigned long long h[4096 * 32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(tafl_prof_acc), sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < 32; ++k) { out[k] = 0; for (int w = 0; w < 4096; ++w) out[k] += h[w * 32 + k]; }
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(tafl_prof_acc), h, sizeof h) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifndef TAFL_KATTR
#define TAFL_KATTR
#endif
// minimum waves per SIMD the playout kernels are compiled for (a bound for the register allocator; the preset 11x11 kernel needs 104
// VGPRs and runs four, which is what the pipeline fills: DESIGN.md section 6)
#ifndef TAFL_ROLLOUT_WAVES
#define TAFL_ROLLOUT_WAVES 2
#endif
#define TAFL_MCTS_MAX_SLOTS 8        /* playout slots per game (the pending leaf + up to 7 predicted ones) */
static_assert(TAFL_MCTS_MAX_SLOTS == tafl::kMctsMaxSlots, "slot bound of tafl_ops.hpp");
#define TAFL_MCTS_MAX_PARTS 8         /* partitions of a batch that run the two-kernel pipeline on their own streams */
#define TAFL_MCTS_TRACE_ROUNDS 4096   /* rounds of a search whose work counts are kept for tafl_mcts_round_trace */
#define TAFL_MCTS_UNDO_CAP 16        /* undo records per game and prediction pass (edges and headers each), in LDS: 16 x 14 words x 64 lanes = 56 KiB per tree wave */
#define TAFL_MCTS_UNDO_CAP_FUSED 5   /* the fused kernel predicts one simulation (two slots) and runs eight waves per CU: 17.5 KiB per wave */
#define TAFL_UNDO_LDS_BYTES(cap) ((size_t)(cap) * (tafl::kUndoEWords + tafl::kUndoHWords) * TAFL_BLOCK * sizeof(uint32_t))

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_fill(Quad* soa, uint32_t n, DState<NL> st) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g < n) StateIO<NL>::store_soa(soa, n, g, st);
}

// counts only: one game per lane
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_movegen(Consts<NL> C, const Quad* soa, uint32_t n, uint32_t* counts) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    counts[g] = Ops<NL, W>::movegen(st, C, nullptr);
}

// counts + dense action masks.  A workgroup serves 64 games with ONE WAVE PER BOARD LINE (lane = game, wave i = row i and column i:
// Ops::movegen_line), so the line index is wave-uniform (every bit position a scalar, no divergence between the lines) and the state loads
// stay coalesced (quad-plane SoA: 1 KiB per wave instruction; the waves of a workgroup read the same 4 KiB, from L2 after the first).
// The masks are assembled in LDS (ds_or, odd row stride: no bank conflicts) and streamed out as one contiguous block per workgroup
// (64 x mask_words uint32, fully coalesced).
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK * 15) void k_movegen_masks(Consts<NL> C, const Quad* soa, uint32_t n, uint32_t* counts, uint32_t* masks, uint32_t mw) {
    extern __shared__ uint32_t lds_masks[];                      // [TAFL_BLOCK][mw | 1] masks, then [TAFL_BLOCK] counts
    const uint32_t ldw = mw | 1u, lane = threadIdx.x & 63u, line = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lines = blockDim.x >> 6;     // lines == C.n
    const uint32_t g0 = blockIdx.x * TAFL_BLOCK, g = g0 + lane;
    uint32_t* lds_cnt = lds_masks + TAFL_BLOCK * ldw;
    for (uint32_t i = threadIdx.x; i < TAFL_BLOCK * (ldw + 1u); i += blockDim.x) lds_masks[i] = 0;
    __syncthreads();
    if (g < n) {
        DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
        const uint32_t c = Ops<NL, W>::movegen_line(st, line, C, lds_masks + (size_t)lane * ldw);
        if (c) atomicAdd(&lds_cnt[lane], c);
    }
    __syncthreads();
    if (line == 0 && g < n && counts) counts[g] = lds_cnt[lane];
    const uint32_t games = (n - g0) < TAFL_BLOCK ? (n - g0) : TAFL_BLOCK;
    uint32_t* dst = masks + (size_t)g0 * mw;
    for (uint32_t gi = line; gi < games; gi += lines)                                        // one game per wave and pass: 304 contiguous bytes
        for (uint32_t w = lane; w < mw; w += TAFL_BLOCK) dst[(size_t)gi * mw + w] = lds_masks[(size_t)gi * ldw + w];
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_validate(Consts<NL> C, const Quad* soa, uint32_t n, const tafl_play* plays, uint8_t* codes) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    codes[g] = (uint8_t)Ops<NL, W>::validate(st, plays[g], C);
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_step(Consts<NL> C, Quad* soa, uint32_t n, const tafl_play* plays, tafl_effects* eff) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    tafl_effects e;
    Ops<NL, W>::step(st, plays[g], C, &e);
    StateIO<NL>::store_soa(soa, n, g, st);
    if (eff) eff[g] = e;
}

// game i plays its (rank mod count)-th legal play in canonical order, in two launches:
//   k_select_kth  the dense legal mask is built in LDS by one wave per board line as in k_movegen_masks; every wave then counts the plays in
//                 its share of the mask words, and the first wave (lane = game) walks the partial counts to the chunk that holds the k-th
//                 set bit and finds it there: the chosen dense action index per game (4 B) and the number of plays
//   k_step_action do_valid_play of that action, one game per lane (k_step without the validation): a workgroup of one wave per 64 games,
//                 so that as many games are being applied at once as the device has SIMDs (inside the line-wave workgroup only one wave in
//                 eleven would do this, the longest dependent chain of the call)
// The streamed step of a 256-bit batch whose board has at most 13 columns: the play is validated on the reference's 15-column words (the
// error code of an off-board play depends on that layout, Engine::validate) and applied in the dense 13-column layout of the 13x13 search
// (restride, tafl_core.hpp): six limbs instead of eight keep do_valid_play in registers (k_step<8, 15> carried 720 B of scratch per lane).
template <bool VALIDATE>
__device__ __forceinline__ void step_dense13(const Consts<8>& C, const Consts<6>& Cd, DState<8>& st, tafl_play play, uint32_t action, uint32_t total, tafl_play& pl, tafl_effects& e) {
    using O8 = Ops<8, 15>; using E6 = Engine<6, 13>;
    O8::caps_to_effects(bz<8>(), 0, e);
    pl.from_row = pl.from_col = pl.axis = 0; pl.disp = 0;
    Move m; m.from = m.to = m.dir = m.dist = 0;
    int code;
    if constexpr (VALIDATE) code = Engine<8, 15>::validate(st, play, st.flags & TAFL_F_SIDE, C, &m);
    else {
        if (total == 0) code = TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING ? TAFL_PLAY_GAME_OVER : TAFL_PLAY_NO_PIECE;
        else if (action != O8::NO_ACTION) { m = O8::move_of_action(action, C); pl = O8::to_play(m); code = TAFL_PLAY_OK; }
        else code = TAFL_PLAY_NO_PIECE;
    }
    if (code == TAFL_PLAY_OK) {
        DState<6> d; restride<8, 15, 6, 13>(st, C.n, d);
        Move md = m; md.from = restride_sq<15, 13>(m.from); md.to = restride_sq<15, 13>(m.to);
        StepOut<6> so; Moves<6> nx;
        E6::apply(d, md, Cd, &so, nx);
        restride<6, 13, 8, 15>(d, C.n, st);
        Bits<8> caps = bz<8>(); restride_rows<6, 13, 8, 15>(so.captures, C.n, caps);
        O8::caps_to_effects(caps, so.n_captures, e);
    }
    O8::status_to_effects(st, code, e);
}
__global__ __launch_bounds__(TAFL_BLOCK) void k_step_dense13(Consts<8> C, Consts<6> Cd, Quad* soa, uint32_t n, const tafl_play* plays, tafl_effects* eff) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<8> st; StateIO<8>::load_soa(soa, n, g, st);
    tafl_effects e; tafl_play pl;
    step_dense13<true>(C, Cd, st, plays[g], 0u, 0u, pl, e);
    StateIO<8>::store_soa(soa, n, g, st);
    if (eff) eff[g] = e;
}
__global__ __launch_bounds__(TAFL_BLOCK) void k_step_action_dense13(Consts<8> C, Consts<6> Cd, Quad* soa, uint32_t n, const uint32_t* actions, const uint32_t* totals, tafl_play* out_plays, tafl_effects* eff) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<8> st; StateIO<8>::load_soa(soa, n, g, st);
    tafl_effects e; tafl_play pl; tafl_play none; none.from_row = none.from_col = none.axis = 0; none.disp = 0;
    step_dense13<false>(C, Cd, st, none, actions[g], totals[g], pl, e);
    StateIO<8>::store_soa(soa, n, g, st);
    if (eff) eff[g] = e;
    if (out_plays) out_plays[g] = pl;
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK * 15) void k_select_kth(Consts<NL> C, const Quad* soa, uint32_t n, const uint32_t* ranks, uint32_t* actions, uint32_t* totals, uint32_t mw) {
    extern __shared__ uint32_t lds_masks[];                      // [TAFL_BLOCK][mw | 1] masks, [TAFL_BLOCK] counts, [TAFL_BLOCK][16] partial counts
    const uint32_t ldw = mw | 1u, lane = threadIdx.x & 63u, line = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lines = blockDim.x >> 6;
    const uint32_t g = blockIdx.x * TAFL_BLOCK + lane;
    uint32_t* lds_cnt = lds_masks + TAFL_BLOCK * ldw;
    uint32_t* lds_part = lds_cnt + TAFL_BLOCK;                    // [lane * 17 + line]
    for (uint32_t i = threadIdx.x; i < TAFL_BLOCK * (ldw + 1u); i += blockDim.x) lds_masks[i] = 0;
    __syncthreads();
    if (g < n) {
        DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
        const uint32_t c = Ops<NL, W>::movegen_line(st, line, C, lds_masks + (size_t)lane * ldw);
        if (c) atomicAdd(&lds_cnt[lane], c);
    }
    __syncthreads();
    const uint32_t chunk = (mw + lines - 1u) / lines, w0 = line * chunk, w1 = (w0 + chunk) < mw ? (w0 + chunk) : mw;
    {
        uint32_t pc = 0;
        for (uint32_t w = w0; w < w1; ++w) pc += (uint32_t)__builtin_popcount(lds_masks[(size_t)lane * ldw + w]);
        lds_part[lane * 17u + line] = pc;
    }
    __syncthreads();
    if (line != 0 || g >= n) return;
    const uint32_t total = lds_cnt[lane];
    uint32_t action = Ops<NL, W>::NO_ACTION;
    if (total) {
        uint32_t k = ranks[g] % total, j = 0; bool found = false;
        for (uint32_t q = 0; q < lines; ++q) { const uint32_t pc = lds_part[lane * 17u + q]; if (!found) { if (k < pc) { j = q; found = true; } else k -= pc; } }
        if (found) { const uint32_t a0 = j * chunk, a1 = (a0 + chunk) < mw ? (a0 + chunk) : mw; action = Ops<NL, W>::kth_set_bit(lds_masks + (size_t)lane * ldw, a0, a1, k); }
    }
    actions[g] = action; totals[g] = total;
}
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_step_action(Consts<NL> C, Quad* soa, uint32_t n, const uint32_t* actions, const uint32_t* totals, tafl_play* out_plays, tafl_effects* eff) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    tafl_effects e; tafl_play p;
    Ops<NL, W>::step_action(st, actions[g], totals[g], C, &p, &e);
    StateIO<NL>::store_soa(soa, n, g, st);
    if (eff) eff[g] = e;
    if (out_plays) out_plays[g] = p;
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_side_can_play(Consts<NL> C, const Quad* soa, uint32_t n, uint32_t side, uint8_t* out) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    out[g] = Ops<NL, W>::side_can_play(st, side, C) ? 1 : 0;
}

// For PRESET != 0 every geometry / rule mask is a compile-time literal (no SMEM loads, no SGPR pressure) and rule
// branches that the preset never takes are pruned; PRESET == 0 uses the run-time Consts passed as a kernel argument.
#define TAFL_PICK_CONSTS(C, Carg)                                                   \
    constexpr Consts<NL> C##_ct = preset_consts<NL, W, PRESET>();                   \
    const Consts<NL>& C = (PRESET != PRESET_NONE) ? C##_ct : (Carg)

// game g of the batch (quad-plane SoA in the reference layout <NLS, WS>) in the layout <NL, W> the kernel works in: the same, or the dense
// 13-column layout of the 13x13 preset (restride, tafl_core.hpp)
template <int NLS, int WS, int NL, int W>
__device__ __forceinline__ void load_batch_state(const Quad* soa, uint32_t n, uint32_t g, uint32_t side_len, DState<NL>& st) {
    if constexpr (NLS == NL && WS == W) StateIO<NL>::load_soa(soa, n, g, st);
    else { DState<NLS> t; StateIO<NLS>::load_soa(soa, n, g, t); restride<NLS, WS, NL, W>(t, side_len, st); }
}

template <int NLS, int WS, int NL, int W, int PRESET>
__global__ TAFL_KATTR __launch_bounds__(TAFL_BLOCK) void k_rollout(Consts<NL> Carg, const Quad* soa, uint32_t n, uint64_t seed, uint32_t sim, uint32_t max_plies,
                                                        uint64_t base, tafl_rollout_result* out) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    TAFL_PICK_CONSTS(C, Carg);
    DState<NL> st; load_batch_state<NLS, WS, NL, W>(soa, n, g, C.n, st);
    tafl_rollout_result r;
    Ops<NL, W>::rollout(st, seed, base + g, sim, max_plies, C, r);
    out[g] = r;
}

template <int NL, int W, int PRESET>
__global__ __launch_bounds__(TAFL_BLOCK) void k_random_advance(Consts<NL> Carg, Quad* soa, uint32_t n, uint64_t seed, const uint32_t* plies, uint64_t base) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= n) return;
    TAFL_PICK_CONSTS(C, Carg);
    DState<NL> st; StateIO<NL>::load_soa(soa, n, g, st);
    Ops<NL, W>::random_advance(st, seed, base + g, plies[g], C);
    StateIO<NL>::store_soa(soa, n, g, st);
}

// ---- MCTS kernels ---------------------------------------------------------------------------------
enum { ST_SIMS = 0, ST_ROLLOUTS, ST_PLIES, ST_DEPTH, ST_SCANNED, ST_TERMINAL, ST_FAULTS, ST_SPEC_ISSUED, ST_REASON0 = 8, ST_SPEC_HITS = 24, ST_EXEC = 25, ST_DONE = 26, ST_COUNT = 28 };
// control words of a search in flight (device memory, one set per batch): the width cap of the prediction pass is steered ON THE DEVICE
// from the hit rate of the last window, so that a whole search can be enqueued without a single read-back (tafl_mcts_run_async)
enum { CT_WCAP = 0, CT_LAST_ISSUED, CT_LAST_HITS, CT_NEXT_CHECK, CT_COUNT };
static_assert(CT_COUNT == 4, "SearchPlan::ctrl0");
__device__ __forceinline__ unsigned long long ld_counter(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void stat_add(unsigned long long* stats, int idx, uint32_t v) {
    const uint32_t s = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats[idx], (unsigned long long)s);
}

template <int NLS, int WS, int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_init(Consts<NL> C, const Quad* soa, MctsMem M) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    DState<NL> st; load_batch_state<NLS, WS, NL, W>(soa, M.G, g, C.n, st);
    Ops<NL, W>::mcts_init_game(M, g, st, C);
}

// tree phase of the simulation pipeline: consume finished playouts (backup), run as many further simulations as can be
// served by ready slots, then issue the next slots (tafl_ops.hpp mcts_tree_step)
// SP: a self-play run (tafl_selfplay_run): a game whose search is done plays its most visited root play on the batch state (soa, layout
// <NLS, WS>) and starts its next search in the same launch; its plan counts from the launch in which that search began
template <int NLS, int WS, int NL, int W, int PRESET, bool SP>
__device__ __forceinline__ void mcts_tree_launch(const Consts<NL>& Carg, const MctsMem& M, double c_puct, uint32_t n_sims, uint32_t round, uint32_t planned, uint32_t probe_every,
                                                 uint32_t target, unsigned long long* stats, const unsigned long long* ctrl, uint32_t* work, uint32_t* work_count,
                                                 uint32_t g_begin, uint32_t g_end, Quad* soa, const SelfPlay& sp) {
    const uint32_t g = g_begin + blockIdx.x * TAFL_BLOCK + threadIdx.x;     // this launch serves games g_begin .. g_end - 1
    // the tree phase of one half of the batch runs beside the other half's playouts (2 - 4 waves per SIMD): it is one latency-bound wave
    // per SIMD on the critical path of its half, so its instructions go first
    __builtin_amdgcn_s_setprio(3);
    TAFL_PICK_CONSTS(C, Carg);
    LaneStats ls; ls.sims = ls.rollouts = ls.rollout_plies = ls.depth = ls.scanned = ls.terminal_hits = ls.faults = ls.reason = 0;
    ls.reason_hist4 = 0; ls.spec_issued = ls.spec_hits = 0;
    bool live = g < g_end && (M.sim_next[g] < n_sims || M.kind[g] == 1);
    if constexpr (SP) {
        const bool adv = g < g_end && !live && sp.moves_done[g] < sp.n_moves;
        if (__ballot(live || adv) == 0ull) return;
        int r = 0;
        if (adv) r = Ops<NL, W>::template selfplay_advance<NLS, WS>(M, g, soa, sp, n_sims, round, C);
        live = live || r == 1;                                // the new search takes its first step in this launch
        const unsigned long long fin = __ballot(r == 2);      // games that made their last play
        if ((threadIdx.x & 63u) == 0 && fin) atomicAdd(&stats[ST_DONE], (unsigned long long)__popcll(fin));
    } else {
        (void)soa; (void)sp;
        if (__ballot(live) == 0ull) return;                   // whole wave finished: nothing to do, nothing to count
    }
    // Plan (wave-uniform): inside the plan a game issues ceil(remaining / rounds left) slots.  Past it: 1 = "use every slot that exists"
    // (wasted playouts are free on an emptying device) for short searches and, for long ones, once three quarters of the games are done;
    // until then 0 = every game keeps to what its own hit history allows (a long search whose predictions fail runs far beyond the plan
    // with every game still alive: S = 1000 runs 44 M sims/s this way, 39 M otherwise).  Both only steer WHEN playouts run, never a result.
    uint32_t rounds_left;
    if constexpr (SP) { const uint32_t rel = live ? round - sp.start_round[g] : 0u; rounds_left = rel < planned ? planned - rel : 0u; }      // per game (the device never empties before the run's end)
    else if (round < planned) rounds_left = planned - round;
    else rounds_left = (probe_every == 0u || 4ull * ld_counter(&stats[ST_DONE]) >= 3ull * (unsigned long long)M.G) ? 1u : 0u;
    const uint32_t wcap = (uint32_t)ld_counter(&ctrl[CT_WCAP]);
    // the undo log of the prediction pass: LDS, one log per lane, word-interleaved (tafl_ops.hpp LogMem)
    extern __shared__ uint32_t tree_lds[];
    LogMem lm; lm.base = tree_lds; lm.stride = TAFL_BLOCK; lm.lane = threadIdx.x & 63u; lm.cap = TAFL_MCTS_UNDO_CAP;
    if (live) Ops<NL, W>::mcts_tree_step(M, g, c_puct, n_sims, rounds_left, Ops<NL, W>::mcts_scenarios(rounds_left, planned), wcap, C, ls, lm);
    if constexpr (!SP) {   // games that completed their last simulation in this launch (a finished game is never live again: counted once)
        const unsigned long long fin = __ballot(live && M.sim_next[g] >= n_sims && M.kind[g] != 1);
        if ((threadIdx.x & 63u) == 0 && fin) atomicAdd(&stats[ST_DONE], (unsigned long long)__popcll(fin));
    }
    // dense work lists of the playouts this round has to run, one list per priority class (MctsMem::spec_cls; work[c * stride ..], work_count[c];
    // entry = slot << 27 | game): the playout kernel walks them in class order up to what the device holds at once, so that the most
    // speculative playouts are the ones left for the next round when more is asked for.  One atomic per wave and slot; the loads of all
    // slots, then the atomics of all slots are in flight together (a dependent chain of eight was 8 round trips to L2).
    const uint32_t stride = g_end - g_begin;
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t kd[TAFL_MCTS_MAX_SLOTS], cl[TAFL_MCTS_MAX_SLOTS];
    TAFL_UNROLL for (uint32_t j = 0; j < TAFL_MCTS_MAX_SLOTS; ++j) {
        const size_t o = (size_t)(j < M.spec_k ? j : 0u) * M.G + g;
        kd[j] = (live && j < M.spec_k) ? M.spec_kind[o] : (uint8_t)0;
        cl[j] = (live && j < M.spec_k) ? M.spec_cls[o] : (uint8_t)0;
    }
    // the slot of this game whose requested playout has priority class c (a game's requested playouts have distinct classes)
    uint32_t sl[TAFL_MCTS_MAX_SLOTS];
    TAFL_UNROLL for (uint32_t c = 0; c < TAFL_MCTS_MAX_SLOTS; ++c) {
        sl[c] = 0xFFu;
        TAFL_UNROLL for (uint32_t j = 0; j < TAFL_MCTS_MAX_SLOTS; ++j) sl[c] = (kd[j] == 1 && cl[j] == c) ? j : sl[c];
    }
    unsigned long long bal[TAFL_MCTS_MAX_SLOTS]; uint32_t base[TAFL_MCTS_MAX_SLOTS];
    TAFL_UNROLL for (uint32_t j = 0; j < TAFL_MCTS_MAX_SLOTS; ++j) {
        bal[j] = __ballot(sl[j] != 0xFFu);
        base[j] = 0;
        if (bal[j] != 0ull && (int)lane == __ffsll((long long)bal[j]) - 1) base[j] = atomicAdd(&work_count[j], (uint32_t)__popcll(bal[j]));
    }
    TAFL_UNROLL for (uint32_t j = 0; j < TAFL_MCTS_MAX_SLOTS; ++j) {
        if (bal[j] == 0ull) continue;
        const uint32_t b0 = (uint32_t)__shfl((int)base[j], __ffsll((long long)bal[j]) - 1);
        if ((bal[j] >> lane) & 1ull) work[(size_t)j * stride + b0 + (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1ull))] = (sl[j] << 27) | g;
    }
    stat_add(stats, ST_SIMS, ls.sims); stat_add(stats, ST_DEPTH, ls.depth); stat_add(stats, ST_SCANNED, ls.scanned);
    stat_add(stats, ST_TERMINAL, ls.terminal_hits); stat_add(stats, ST_FAULTS, ls.faults);
    stat_add(stats, ST_ROLLOUTS, ls.rollouts); stat_add(stats, ST_PLIES, ls.rollout_plies);
    stat_add(stats, ST_SPEC_ISSUED, ls.spec_issued); stat_add(stats, ST_SPEC_HITS, ls.spec_hits);
    for (uint32_t r = 0; r < 16; ++r) stat_add(stats, ST_REASON0 + r, (uint32_t)((ls.reason_hist4 >> (4u * r)) & 15ull));
}
template <int NL, int W, int PRESET>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_tree(Consts<NL> Carg, MctsMem M, double c_puct, uint32_t n_sims, uint32_t round, uint32_t planned, uint32_t probe_every,
                                                          uint32_t target, unsigned long long* stats, const unsigned long long* ctrl, uint32_t* work, uint32_t* work_count,
                                                          uint32_t g_begin, uint32_t g_end) {
    SelfPlay none; none.moves_done = nullptr; none.start_round = nullptr; none.plays = nullptr; none.n_moves = 0;
    mcts_tree_launch<NL, W, NL, W, PRESET, false>(Carg, M, c_puct, n_sims, round, planned, probe_every, target, stats, ctrl, work, work_count, g_begin, g_end, nullptr, none);
}
template <int NLS, int WS, int NL, int W, int PRESET>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_tree_selfplay(Consts<NL> Carg, MctsMem M, double c_puct, uint32_t n_sims, uint32_t round, uint32_t planned, uint32_t probe_every,
                                                                   uint32_t target, unsigned long long* stats, const unsigned long long* ctrl, uint32_t* work, uint32_t* work_count,
                                                                   uint32_t g_begin, uint32_t g_end, Quad* soa, SelfPlay sp) {
    mcts_tree_launch<NLS, WS, NL, W, PRESET, true>(Carg, M, c_puct, n_sims, round, planned, probe_every, target, stats, ctrl, work, work_count, g_begin, g_end, soa, sp);
}

// the dominant kernel: one seeded random playout per entry of the round's work list (slot, game), state resident in registers.
// spec_k slots per game put up to spec_k waves on every SIMD.
template <int NL, int W, int PRESET>
__global__ TAFL_KATTR __launch_bounds__(TAFL_BLOCK, TAFL_ROLLOUT_WAVES) void k_mcts_rollout(Consts<NL> Carg, MctsMem M, uint64_t seed, uint64_t base, uint32_t sim_offset,
                                                                       uint32_t max_plies, const uint32_t* work, const uint32_t* work_count, uint32_t* next_count,
                                                                       uint32_t stride, uint32_t capacity, unsigned long long* stats, uint32_t* trace,
                                                                       unsigned long long* ctrl, uint32_t round, uint32_t planned, uint32_t probe_every) {
    // entry i of the concatenated per-class work lists; entries beyond `capacity` (what the device holds at once) wait for the next round
    uint32_t pre[TAFL_MCTS_MAX_SLOTS + 1];
    pre[0] = 0;
    TAFL_UNROLL for (uint32_t t = 0; t < TAFL_MCTS_MAX_SLOTS; ++t) pre[t + 1] = pre[t] + work_count[t];
    const uint32_t cnt = pre[TAFL_MCTS_MAX_SLOTS] < capacity ? pre[TAFL_MCTS_MAX_SLOTS] : capacity;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (trace) { trace[0] = pre[TAFL_MCTS_MAX_SLOTS]; trace[1] = cnt; }                // this round: requested, run
        TAFL_UNROLL for (uint32_t t = 0; t < TAFL_MCTS_MAX_SLOTS; ++t) next_count[t] = 0;     // the next round's counters (the other buffer)
        // Width control of long searches (ctrl is handed to the first partition's launches only): every `probe_every` rounds inside the plan,
        // every few rounds past it, the share of predictions that came true since the last look sets how many predicted simulations a game
        // may run beside the pending one (a prediction costs a child expansion in the tree phase and, when it fails, a playout: S = 1000 runs
        // 50.6 M sims/s with the thresholds below, 45.5 M when the windows with 45 - 75 % hits get three predictions instead of one, S = 256
        // 63.8 M with them and 62.0 M with narrower ones; measured in round 2 with the same rule on the host).
        if (ctrl && probe_every && (unsigned long long)round + 1ull >= ctrl[CT_NEXT_CHECK]) {
            const unsigned long long issued = ld_counter(&stats[ST_SPEC_ISSUED]), hits = ld_counter(&stats[ST_SPEC_HITS]);
            const unsigned long long di = issued - ctrl[CT_LAST_ISSUED], dh = hits - ctrl[CT_LAST_HITS];
            ctrl[CT_LAST_ISSUED] = issued; ctrl[CT_LAST_HITS] = hits;
            unsigned long long wcap = ctrl[CT_WCAP];
            if (di > (unsigned long long)M.G / 4ull) wcap = 100ull * dh > 85ull * di ? M.spec_k - 1u : 100ull * dh > 75ull * di ? 3u : 100ull * dh > 70ull * di ? 2u : 1u;
            else if (wcap < M.spec_k - 1u) wcap += 1ull;                  // hardly anything was predicted: probe one wider
            __hip_atomic_store(&ctrl[CT_WCAP], wcap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctrl[CT_NEXT_CHECK] = (unsigned long long)round + 1ull + (round + 1u < planned ? probe_every : (planned >= 32u ? 4u : 2u));
        }
    }
    if (blockIdx.x * TAFL_BLOCK >= cnt) return;
    const uint32_t i = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    TAFL_PICK_CONSTS(C, Carg);
    const bool has = i < cnt;
    uint32_t cls = 0, off = 0;
    TAFL_UNROLL for (uint32_t t = 1; t < TAFL_MCTS_MAX_SLOTS; ++t) { const bool ge = i >= pre[t]; cls = ge ? t : cls; off = ge ? pre[t] : off; }
    const uint32_t e = has ? work[(size_t)cls * stride + (i - off)] : 0u;
    const uint32_t j = e >> 27, g = e & 0x07FFFFFFu;
    if (has) Ops<NL, W>::mcts_slot_rollout(M, j, g, seed, base + g, sim_offset, max_plies, C);
    if ((threadIdx.x & 63) == 0) atomicAdd(&stats[ST_EXEC], (unsigned long long)__popcll(__ballot(has)));
}

// The fused form of the two kernels above: one wave owns 64 / K games for a whole chunk of rounds and alternates, without any
// grid-wide synchronisation, between the tree phase (its games on the first 64 / K lanes) and the playout phase (K slots x 64 / K
// games on all 64 lanes).  Nothing is shared between waves, so no wave ever waits for another: the tree phase of one wave hides
// under the playouts of the other wave on its SIMD, the per-round launches disappear, and a wave leaves as soon as its own games
// are done.  Same per-game functions, same memory layout, same results as the two-kernel path.
template <int NL, int W, int PRESET, int K>
__global__ TAFL_KATTR __launch_bounds__(TAFL_BLOCK, TAFL_ROLLOUT_WAVES) void k_mcts_fused(Consts<NL> Carg, MctsMem M, double c_puct, uint32_t n_sims, uint64_t seed,
                                                                     uint64_t base, uint32_t sim_offset, uint32_t max_plies, uint32_t max_rounds,
                                                                     unsigned long long* stats) {
    constexpr uint32_t GPW = TAFL_BLOCK / K;                      // games per wave
    const uint32_t lane = threadIdx.x;
    const uint32_t tg = blockIdx.x * GPW + lane;                  // tree phase: lane < GPW serves game tg
    const uint32_t rj = lane / GPW, rg = blockIdx.x * GPW + (lane % GPW);   // playout phase: slot rj of game rg
    TAFL_PICK_CONSTS(C, Carg);
    LaneStats ls; ls.sims = ls.rollouts = ls.rollout_plies = ls.depth = ls.scanned = ls.terminal_hits = ls.faults = ls.reason = 0;
    ls.reason_hist4 = 0; ls.spec_issued = ls.spec_hits = 0;
    uint32_t executed = 0, finished = 0;
    extern __shared__ uint32_t tree_lds[];                        // undo log of the prediction pass (tafl_ops.hpp LogMem)
    LogMem lm; lm.base = tree_lds; lm.stride = TAFL_BLOCK; lm.lane = lane; lm.cap = K > 1 ? TAFL_MCTS_UNDO_CAP_FUSED : 0u;
    for (uint32_t round = 0; round < max_rounds; ++round) {
        const bool live = lane < GPW && tg < M.G && (M.sim_next[tg] < n_sims || M.kind[tg] == 1);
        if (__ballot(live) == 0ull) break;                        // every game of this wave has finished
        if (live) Ops<NL, W>::mcts_tree_step(M, tg, c_puct, n_sims, 0u, 2u, K, C, ls, lm);
        finished += (uint32_t)__popcll(__ballot(live && M.sim_next[tg] >= n_sims && M.kind[tg] != 1));
        if ((round & 3u) == 3u) {                                 // the packed 4-bit reason counters hold 15: at most 2 playouts are consumed per round
            for (uint32_t r = 0; r < 16; ++r) stat_add(stats, ST_REASON0 + r, (uint32_t)((ls.reason_hist4 >> (4u * r)) & 15ull));
            ls.reason_hist4 = 0;
        }
        __threadfence();                                          // slot records written by the tree lanes are read by all lanes
        const bool work = rg < M.G && rj < M.spec_k && M.spec_kind[(size_t)rj * M.G + rg] == 1;
        const unsigned long long wb = __ballot(work);
        if (wb == 0ull) continue;
        if (work) Ops<NL, W>::mcts_slot_rollout(M, rj, rg, seed, base + rg, sim_offset, max_plies, C);
        executed += (uint32_t)__popcll(wb);
        __threadfence();
    }
    stat_add(stats, ST_SIMS, ls.sims); stat_add(stats, ST_DEPTH, ls.depth); stat_add(stats, ST_SCANNED, ls.scanned);
    stat_add(stats, ST_TERMINAL, ls.terminal_hits); stat_add(stats, ST_FAULTS, ls.faults);
    stat_add(stats, ST_ROLLOUTS, ls.rollouts); stat_add(stats, ST_PLIES, ls.rollout_plies);
    stat_add(stats, ST_SPEC_ISSUED, ls.spec_issued); stat_add(stats, ST_SPEC_HITS, ls.spec_hits);
    for (uint32_t r = 0; r < 16; ++r) stat_add(stats, ST_REASON0 + r, (uint32_t)((ls.reason_hist4 >> (4u * r)) & 15ull));
    if (lane == 0 && executed) atomicAdd(&stats[ST_EXEC], (unsigned long long)executed);
    if (lane == 0 && finished) atomicAdd(&stats[ST_DONE], (unsigned long long)finished);
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_root_children(Consts<NL> C, MctsMem M, tafl_root_child* out, uint32_t max_children, uint32_t* out_n) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    out_n[g] = Ops<NL, W>::mcts_root_children(M, g, C, out + (size_t)g * max_children, max_children);
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_root_visits(Consts<NL> C, MctsMem M, uint32_t* out, uint32_t action_size) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    const NodeHdr h = M.hdr[g];
    const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
    for (uint32_t j = 0; j < h.m; ++j) {
        const Edge e = eb[j];
        const NodeHdr ch = M.hdr[(size_t)e.child * M.G + g];
        Move m; m.from = ch.mv_from; m.dir = ch.mv_dir; m.dist = ch.mv_dist; m.to = 0;
        out[(size_t)g * action_size + Ops<NL, W>::action_of(m, C)] = e.n;
    }
}

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_best_play(Consts<NL> C, MctsMem M, tafl_play* out_plays, uint32_t* out_visits) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    const NodeHdr h = M.hdr[g];
    const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
    uint32_t best = 0; tafl_play bp; bp.from_row = bp.from_col = bp.axis = 0; bp.disp = 0;
    for (uint32_t j = 0; j < h.m; ++j) {                       // first maximum (src/mcts.rs:216-227)
        const Edge e = eb[j];
        if (e.n > best) {
            const NodeHdr ch = M.hdr[(size_t)e.child * M.G + g];
            Move m; m.from = ch.mv_from; m.dir = ch.mv_dir; m.dist = ch.mv_dist; m.to = 0;
            best = e.n; bp = Ops<NL, W>::to_play(m);
        }
    }
    out_plays[g] = bp; out_visits[g] = best;
}

// board_to_matrix (game/main.rs:55-83): corners 20, throne 30, soldier +1, king +5, one uint8 per tile, row-major n x n.
// One lane per tile: consecutive lanes write consecutive bytes.
// self-play step on the device: every game plays the most visited root play of its last search (first maximum, src/mcts.rs:216-227)
// on its batch state (do_valid_play); games whose root has no visited child (finished games) stay as they are
// NL, W: the batch layout (the play is applied to the batch state); WA: row stride of the search arena the root's plays are recorded in
template <int NL, int W, int WA>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_play_best(Consts<NL> C, MctsMem M, Quad* soa, tafl_play* out_plays, tafl_effects* eff) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    const NodeHdr h = M.hdr[g];
    const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
    uint32_t best = 0; Move bm; bm.from = bm.to = bm.dir = bm.dist = 0;
    for (uint32_t j = 0; j < h.m; ++j) {
        const Edge e = eb[j];
        if (e.n > best) { const NodeHdr ch = M.hdr[(size_t)e.child * M.G + g]; best = e.n; bm.from = ch.mv_from; bm.dir = ch.mv_dir; bm.dist = ch.mv_dist; }
    }
    if constexpr (WA != W) bm.from = restride_sq<WA, W>(bm.from);
    DState<NL> st; StateIO<NL>::load_soa(soa, M.G, g, st);
    tafl_effects e; Ops<NL, W>::caps_to_effects(bz<NL>(), 0, e);
    tafl_play p; p.from_row = p.from_col = p.axis = 0; p.disp = 0;
    int code = TAFL_PLAY_GAME_OVER;
    if (best > 0 && TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING) {
        bm.to = (uint32_t)((int)bm.from + Engine<NL, W>::delta(bm.dir) * (int)bm.dist);
        p = Ops<NL, W>::to_play(bm);
        StepOut<NL> so; Moves<NL> nx;
        Engine<NL, W>::apply(st, bm, C, &so, nx);
        Ops<NL, W>::caps_to_effects(so.captures, so.n_captures, e);
        StateIO<NL>::store_soa(soa, M.G, g, st);
        code = TAFL_PLAY_OK;
    }
    Ops<NL, W>::status_to_effects(st, code, e);
    if (eff) eff[g] = e;
    if (out_plays) out_plays[g] = p;
}

template <int NL, int W>
__global__ __launch_bounds__(256) void k_encode_boards(Consts<NL> C, const Quad* soa, uint32_t n_games, uint8_t* out) {
    const uint32_t nn = C.n * C.n;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_games * nn) return;
    const uint32_t g = (uint32_t)(i / nn), t = (uint32_t)(i % nn), r = t / C.n, c = t % C.n, bit = r * (uint32_t)W + c;
    const uint32_t wa = bit >> 5, wd = (uint32_t)NL + (bit >> 5);         // absolute state words: att[NL], def[NL], rep[4], meta[4]
    const Quad qa = soa[(size_t)(wa >> 2) * n_games + g], qd = soa[(size_t)(wd >> 2) * n_games + g];
    const uint32_t la = wa & 3, ld = wd & 3;
    const uint32_t aw = la == 0 ? qa.x : la == 1 ? qa.y : la == 2 ? qa.z : qa.w;
    const uint32_t dw = ld == 0 ? qd.x : ld == 1 ? qd.y : ld == 2 ? qd.z : qd.w;
    const Quad meta = soa[(size_t)(2 * NL + 4) / 4 * n_games + g];
    const uint32_t flags = meta.w, krow = TAFL_F_KROW(flags), kcol = TAFL_F_KCOL(flags);
    uint32_t v = 0;
    if ((r == 0 || r == C.n - 1) && (c == 0 || c == C.n - 1)) v = 20;
    if (r == C.n / 2 && c == C.n / 2) v = 30;
    const bool d = (dw >> (bit & 31)) & 1u, a = (aw >> (bit & 31)) & 1u;
    if (d) v += (r == krow && c == kcol) ? 5u : 1u; else if (a) v += 1u;
    out[i] = (uint8_t)v;
}

// probs of src/mcts.py:43-53 for any temperature.
//   temp > 0 : counts ** (1 / temp) (float64 pow of the device math library; exactly the count for temp == 1), summed in ascending action
//              order like Python's sum(), then divided (mcts.py:50-52).
//   temp == 0: one-hot on one of the maxima (mcts.py:44-48).  The reference draws it with the process-global np.random.choice; here it is
//              the first maximum (tie_seed == 0) or the floor(r * ties / 2^32)-th one in ascending action order with r = the taflmix32 word keyed by
//              (tie_seed, global game id): reproducible and independent of the sharding.
__device__ __forceinline__ uint32_t tie_pick(uint64_t tie_seed, uint64_t game_id, uint32_t ties) {
    const uint64_t gk = Engine<2, 7>::game_key(tie_seed, game_id);
    const uint32_t h = Engine<2, 7>::fmix32((uint32_t)gk ^ Engine<2, 7>::fmix32((uint32_t)(gk >> 32) + 0x7A1E5EEDu));
    return Engine<2, 7>::mulhi(h, ties);
}
__device__ __forceinline__ double temp_weight(uint32_t n, double inv_temp) { return inv_temp == 1.0 ? (double)n : pow((double)n, inv_temp); }

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_mcts_policy(Consts<NL> C, MctsMem M, double* out, uint32_t action_size, int one_hot, double inv_temp,
                                                            uint64_t tie_seed, uint64_t base) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    const NodeHdr h = M.hdr[g];
    const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
    double sum = 0.0; uint32_t best = 0, ties = 0;
    for (uint32_t j = 0; j < h.m; ++j) {
        const Edge e = eb[j];
        if (!one_hot) sum += temp_weight(e.n, inv_temp);
        if (e.n > best) { best = e.n; ties = 1; } else if (e.n == best && best > 0) ++ties;
    }
    uint32_t pick = 0;
    if (one_hot && tie_seed != 0 && ties > 1) pick = tie_pick(tie_seed, base + g, ties);
    uint32_t seen = 0;
    for (uint32_t j = 0; j < h.m; ++j) {
        const Edge e = eb[j];
        const NodeHdr ch = M.hdr[(size_t)e.child * M.G + g];
        Move m; m.from = ch.mv_from; m.dir = ch.mv_dir; m.dist = ch.mv_dist; m.to = 0;
        double p;
        if (one_hot) { const bool is_max = best > 0 && e.n == best; p = (is_max && seen == pick) ? 1.0 : 0.0; seen += is_max ? 1u : 0u; }
        else p = temp_weight(e.n, inv_temp) / sum;
        out[(size_t)g * action_size + Ops<NL, W>::action_of(m, C)] = p;
    }
    // all counts zero (the root was terminal): every action is a maximum (np.argwhere order); first, or the seeded choice among all actions
    if (one_hot && best == 0) out[(size_t)g * action_size + (tie_seed != 0 ? tie_pick(tie_seed, base + g, action_size) : 0u)] = 1.0;
}

// --------------------------------------------------------------------------------------------------
// host objects
// --------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
// for the other translation units of the library (tafl_replay.cpp): same per-thread message as tafl_last_error()
int tafl_fail_(int code, const char* msg) { return fail(code, msg ? msg : ""); }
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(TAFL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

enum { KC_MOVEGEN = 0, KC_STEP, KC_ROLLOUT, KC_MCTS_TREE, KC_MCTS_ROLLOUT, KC_COUNT };

struct TimedSpan { hipEvent_t a, b; int cls; };

struct tafl_ctx {
    tafl_rules rules;
    uint32_t n, word_bits, nl, w;
    int device;
    hipStream_t stream;
    bool own_stream;
    Consts<2> c2; Consts<4> c4; Consts<8> c8;
    Consts<6> c6;                    // the same rules in the dense 13-column layout (dense13: 256-bit words, side_len <= 13)
    bool dense13;
    int preset;                      // PRESET_* detected at ctx_create: selects kernels with compile-time constants
    uint32_t live_batches;           // batches created on this context and not yet destroyed (tafl_ctx_destroy refuses while > 0)
    uint32_t rollout_capacity;       // playouts k_mcts_rollout holds on the device at once (occupancy x CUs x 64 lanes); 0 = not asked yet
    bool timing;
    std::vector<TimedSpan> spans;
    double acc_ms[KC_COUNT]; uint64_t acc_n[KC_COUNT];
    // the same spans as intervals on one clock (milliseconds since `t_ref`, recorded by tafl_timing_reset): launches of a class that
    // overlap on different streams are counted once by tafl_timing_get_union
    hipEvent_t t_ref; bool has_ref;
    std::vector<std::pair<float, float>> ivals[KC_COUNT];
};

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) return -1;
        cap = bytes; return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// a search in flight on a batch: tafl_mcts_run_async enqueues the whole plan, tafl_mcts_wait joins it (and runs the stragglers' rounds)
struct SearchPart { uint32_t g0, g1, cap, grid_tree, grid_roll; hipStream_t s; uint32_t* wl; uint32_t* wc; };
struct SearchPlan {
    bool active, fused;
    tafl_mcts_params p; uint64_t base;
    MctsMem M;
    uint32_t parts, slots, planned, probe_every, next_round, max_rounds;
    unsigned long long ctrl0[4];     // initial control words (CT_*): source of an asynchronous copy
    SelfPlay selfplay;               // n_moves != 0: a self-play run (tafl_selfplay_run)
    SearchPart P[TAFL_MCTS_MAX_PARTS];
};

struct tafl_batch {
    tafl_ctx* ctx;
    uint32_t n;
    // searches run on streams of the BATCH (created on first use), forked from the context's stream when the search is enqueued and joined
    // by tafl_mcts_wait: two batches of one context search side by side
    hipStream_t sstream[TAFL_MCTS_MAX_PARTS];
    hipEvent_t ev_fork[TAFL_MCTS_MAX_PARTS], ev_start, ev_half;
    uint32_t n_sstreams;
    bool half_recorded;              // ev_half sits in the stream of the search in flight (half of its planned rounds are enqueued before it)
    SearchPlan plan;
    DevBuf ctrl;
    Quad* soa;                       // quad-plane SoA: [QUADS][n]
    DevBuf plays, effects, counts, masks, codes, ranks, results, out_plays, u8out, plies;
    // MCTS
    MctsMem mem; bool has_mem; uint32_t reserved_sims;
    DevBuf node_state, hdr, edges, node_top, edge_top, leaf, kind, fault, stats, children, children_n, visits;
    DevBuf best_plays, best_visits, enc, policy;
    DevBuf work, work_count, trace, sim_base, sp_moves_done, sp_start_round, sp_plays;
    uint32_t trace_rounds;           // rounds of the last two-kernel search recorded in `trace` (requested / run playouts per round)
    DevBuf sim_next, spec_state, spec_meta, spec_value, spec_kind, spec_reason, spec_plies, spec_ref, spec_cls, spec_pend, spec_bias;
    uint32_t spec_k;                 // playout slots per game that exist (TAFL_MCTS_MAX_SLOTS)
    tafl_mcts_stats last_stats; bool ran;
    bool stats_ok;                   // the counters of the last finished search / self-play run can be read (a self-play run leaves no tree: ran = false)
    // guided MCTS (external evaluator)
    GuidedMem gmem; bool g_has; uint32_t g_max_sims;
    DevBuf g_node_state, g_hdr, g_pedge, g_edges, g_node_top, g_edge_top, g_leaf, g_kind, g_fault, g_sims, g_stats, g_priors, g_values, g_boards, g_sides, g_wait;
};

static int quads_of(const tafl_ctx* c) { return (2 * (int)c->nl + 8) / 4; }
static uint32_t grid_of(uint32_t n) { return (n + TAFL_BLOCK - 1) / TAFL_BLOCK; }

#define DISPATCH_NLW(ctx, STMT)                                                                     \
    do {                                                                                            \
        if ((ctx)->nl == 2) { constexpr int NL = 2, W = 7; const Consts<2>& CC = (ctx)->c2; (void)CC; (void)W; STMT; }        \
        else if ((ctx)->nl == 4) { constexpr int NL = 4, W = 11; const Consts<4>& CC = (ctx)->c4; (void)CC; (void)W; STMT; }  \
        else { constexpr int NL = 8, W = 15; const Consts<8>& CC = (ctx)->c8; (void)CC; (void)W; STMT; }                      \
    } while (0)

// same, for the hot kernels that also exist specialised on a compile-time preset
#define DISPATCH_PRESET(ctx, STMT)                                                                                        \
    do {                                                                                                                  \
        if ((ctx)->nl == 2) { constexpr int NL = 2, W = 7; const Consts<2>& CC = (ctx)->c2; (void)CC; (void)W;            \
            if ((ctx)->preset == PRESET_BRANDUBH7) { constexpr int PRESET = PRESET_BRANDUBH7; STMT; } else { constexpr int PRESET = PRESET_NONE; STMT; } } \
        else if ((ctx)->nl == 4) { constexpr int NL = 4, W = 11; const Consts<4>& CC = (ctx)->c4; (void)CC; (void)W;      \
            if ((ctx)->preset == PRESET_COPENHAGEN11) { constexpr int PRESET = PRESET_COPENHAGEN11; STMT; } else { constexpr int PRESET = PRESET_NONE; STMT; } } \
        else { constexpr int NL = 8, W = 15; const Consts<8>& CC = (ctx)->c8; (void)CC; (void)W;                          \
            if ((ctx)->preset == PRESET_COPENHAGEN13) { constexpr int PRESET = PRESET_COPENHAGEN13; STMT; } else { constexpr int PRESET = PRESET_NONE; STMT; } } \
    } while (0)

// the kernels that work on the search arena (and the playouts): NL, W, CC = the arena layout, NLS, WS = the batch layout.  They differ
// for the 13x13 preset only, which is searched in the dense 13-column layout (6 limbs instead of the reference's 8)
#define DISPATCH_ARENA(ctx, STMT)                                                                   \
    do {                                                                                            \
        if ((ctx)->nl == 2) { constexpr int NL = 2, W = 7, NLS = 2, WS = 7; const Consts<2>& CC = (ctx)->c2; (void)CC; (void)W; (void)NLS; (void)WS; STMT; }        \
        else if ((ctx)->nl == 4) { constexpr int NL = 4, W = 11, NLS = 4, WS = 11; const Consts<4>& CC = (ctx)->c4; (void)CC; (void)W; (void)NLS; (void)WS; STMT; }  \
        else if ((ctx)->preset == PRESET_COPENHAGEN13) { constexpr int NL = 6, W = 13, NLS = 8, WS = 15; const Consts<6>& CC = (ctx)->c6; (void)CC; (void)W; (void)NLS; (void)WS; STMT; } \
        else { constexpr int NL = 8, W = 15, NLS = 8, WS = 15; const Consts<8>& CC = (ctx)->c8; (void)CC; (void)W; (void)NLS; (void)WS; STMT; }                      \
    } while (0)
#define DISPATCH_ARENA_PRESET(ctx, STMT)                                                                                  \
    do {                                                                                                                  \
        if ((ctx)->nl == 2) { constexpr int NL = 2, W = 7, NLS = 2, WS = 7; const Consts<2>& CC = (ctx)->c2; (void)CC; (void)W; (void)NLS; (void)WS;            \
            if ((ctx)->preset == PRESET_BRANDUBH7) { constexpr int PRESET = PRESET_BRANDUBH7; STMT; } else { constexpr int PRESET = PRESET_NONE; STMT; } } \
        else if ((ctx)->nl == 4) { constexpr int NL = 4, W = 11, NLS = 4, WS = 11; const Consts<4>& CC = (ctx)->c4; (void)CC; (void)W; (void)NLS; (void)WS;      \
            if ((ctx)->preset == PRESET_COPENHAGEN11) { constexpr int PRESET = PRESET_COPENHAGEN11; STMT; } else { constexpr int PRESET = PRESET_NONE; STMT; } } \
        else if ((ctx)->preset == PRESET_COPENHAGEN13) { constexpr int NL = 6, W = 13, NLS = 8, WS = 15, PRESET = PRESET_COPENHAGEN13; const Consts<6>& CC = (ctx)->c6; (void)CC; (void)W; (void)NLS; (void)WS; STMT; } \
        else { constexpr int NL = 8, W = 15, NLS = 8, WS = 15, PRESET = PRESET_NONE; const Consts<8>& CC = (ctx)->c8; (void)CC; (void)W; (void)NLS; (void)WS; STMT; } \
    } while (0)
template <int NLS> static const Consts<NLS>& batch_consts(const tafl_ctx* c) {
    if constexpr (NLS == 2) return c->c2; else if constexpr (NLS == 4) return c->c4; else return c->c8;
}
static int arena_quads(const tafl_ctx* c) { return c->preset == PRESET_COPENHAGEN13 ? (2 * 6 + 8) / 4 : quads_of(c); }

struct SpanGuard {
    tafl_ctx* c; int idx;
    hipStream_t st;
    SpanGuard(tafl_ctx* ctx, int cls, hipStream_t on = nullptr) : c(ctx), idx(-1), st(on ? on : ctx->stream) {
        if (!c->timing) return;
        TimedSpan s; s.cls = cls;
        if (hipEventCreate(&s.a) != hipSuccess) return;
        if (hipEventCreate(&s.b) != hipSuccess) { (void)hipEventDestroy(s.a); return; }
        (void)hipEventRecord(s.a, st);
        c->spans.push_back(s); idx = (int)c->spans.size() - 1;
    }
    ~SpanGuard() { if (idx >= 0) (void)hipEventRecord(c->spans[idx].b, st); }
};

static void drain_spans(tafl_ctx* c) {
    for (auto& s : c->spans) {
        (void)hipEventSynchronize(s.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->acc_ms[s.cls] += ms; c->acc_n[s.cls] += 1;
            float t0 = 0.f;
            if (c->has_ref && hipEventElapsedTime(&t0, c->t_ref, s.a) == hipSuccess) c->ivals[s.cls].push_back(std::make_pair(t0, t0 + ms));
        }
        (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
    }
    c->spans.clear();
}

// --------------------------------------------------------------------------------------------------
// C-ABI
// --------------------------------------------------------------------------------------------------
extern "C" {

const char* tafl_last_error(void) { return g_err.c_str(); }
int tafl_abi_version(void) { return TAFLHIP_ABI_VERSION; }
int tafl_preset_rules(const char* name, tafl_rules* out) {
    if (!name || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    if (preset_rules(name, out)) return fail(TAFL_ERR_INVALID_ARG, std::string("unknown ruleset preset: ") + name);
    return TAFL_OK;
}
const char* tafl_preset_board(const char* name) { return preset_board(name); }

int tafl_ctx_create(const tafl_rules* rules, uint8_t side_len, uint32_t word_bits, int device, void* stream, tafl_ctx** out) {
    if (!rules || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    int l64, rw;
    if (word_params(word_bits, &l64, &rw)) return fail(TAFL_ERR_INVALID_ARG, "word_bits must be 64, 128 or 256");
    if (side_len < 3 || (int)side_len > rw || side_len > 15) return fail(TAFL_ERR_INVALID_ARG, "side_len does not fit the board word");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(TAFL_ERR_NO_DEVICE, "no HIP device visible: taflhip has no CPU path");
    if (device < 0 || device >= ndev) return fail(TAFL_ERR_INVALID_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    tafl_ctx* c = new (std::nothrow) tafl_ctx();
    if (!c) return fail(TAFL_ERR_OOM, "out of host memory");
    c->rules = *rules; c->n = side_len; c->word_bits = word_bits; c->nl = (uint32_t)l64 * 2; c->w = (uint32_t)rw; c->device = device;
    c->timing = false; c->rollout_capacity = 0; c->live_batches = 0; c->has_ref = false;
    c->preset = detect_preset(*rules, side_len, word_bits);
    for (int i = 0; i < KC_COUNT; ++i) { c->acc_ms[i] = 0; c->acc_n[i] = 0; }
    int rc = 0;
    if (c->nl == 2) rc = make_consts<2, 7>(*rules, side_len, c->c2);
    else if (c->nl == 4) rc = make_consts<4, 11>(*rules, side_len, c->c4);
    else rc = make_consts<8, 15>(*rules, side_len, c->c8);
    c->dense13 = c->nl == 8 && side_len <= 13;      // the 13-column layout holds the board: streamed steps and the 13x13 preset's searches use it
    if (!rc && c->dense13) rc = make_consts<6, 13>(*rules, side_len, c->c6);
    if (rc) { delete c; return fail(TAFL_ERR_INVALID_ARG, "bad rules / geometry"); }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(TAFL_ERR_HIP, "hipStreamCreate failed"); } c->own_stream = true; }
    *out = c;
    return TAFL_OK;
}

int tafl_ctx_destroy(tafl_ctx* c) {
    if (!c) return TAFL_OK;
    if (c->live_batches != 0) return fail(TAFL_ERR_INVALID_ARG, "tafl_ctx_destroy: batches of this context are still alive (destroy them first)");
    (void)hipSetDevice(c->device);
    drain_spans(c);
    if (c->has_ref) (void)hipEventDestroy(c->t_ref);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TAFL_OK;
}

void* tafl_ctx_stream(tafl_ctx* c) { return c ? (void*)c->stream : nullptr; }

uint32_t tafl_action_size(const tafl_ctx* c) { return c ? c->n * c->n * 2u * (c->n - 1) : 0; }
uint32_t tafl_action_mask_words(const tafl_ctx* c) { return (tafl_action_size(c) + 31) / 32; }
int tafl_action_encode(const tafl_ctx* c, tafl_play p, uint32_t* action) {
    if (!c || !action) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    const uint32_t dist = (uint32_t)(p.disp < 0 ? -(int)p.disp : (int)p.disp), r = p.from_row, cc = p.from_col, m = c->n - 1;
    if (r >= c->n || cc >= c->n || dist == 0) return fail(TAFL_ERR_INVALID_ARG, "play outside the action space");
    const bool vert = p.axis == TAFL_AXIS_VERTICAL;
    const uint32_t room = vert ? (p.disp > 0 ? m - r : r) : (p.disp > 0 ? m - cc : cc);
    if (dist > room) return fail(TAFL_ERR_INVALID_ARG, "play outside the action space");
    const uint32_t slot = vert ? (p.disp > 0 ? dist - 1 : (m - r) + dist - 1) : (p.disp > 0 ? m + dist - 1 : m + (m - cc) + dist - 1);
    *action = (r * c->n + cc) * 2u * m + slot;
    return TAFL_OK;
}
int tafl_action_decode(const tafl_ctx* c, uint32_t a, tafl_play* play) {
    if (!c || !play || a >= tafl_action_size(c)) return fail(TAFL_ERR_INVALID_ARG, "action out of range");
    const uint32_t m = c->n - 1, per = 2u * m, sq = a / per, slot = a % per, r = sq / c->n, cc = sq % c->n;
    play->from_row = (uint8_t)r; play->from_col = (uint8_t)cc;
    if (slot < m - r) { play->axis = TAFL_AXIS_VERTICAL; play->disp = (int8_t)(slot + 1); }
    else if (slot < m) { play->axis = TAFL_AXIS_VERTICAL; play->disp = (int8_t)(-(int)(slot - (m - r) + 1)); }
    else if (slot < m + (m - cc)) { play->axis = TAFL_AXIS_HORIZONTAL; play->disp = (int8_t)(slot - m + 1); }
    else { play->axis = TAFL_AXIS_HORIZONTAL; play->disp = (int8_t)(-(int)(slot - m - (m - cc) + 1)); }
    return TAFL_OK;
}

int tafl_state_from_fen(const tafl_ctx* c, const char* fen, uint8_t side, tafl_state* out) {
    if (!c || !fen || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    std::string err;
    if (fen_to_state(fen, side, c->word_bits, out, &err)) return fail(TAFL_ERR_PARSE, err);
    if (out->side_len != c->n) return fail(TAFL_ERR_PARSE, "FEN side length differs from the context's side_len");
    return TAFL_OK;
}

// BoardState::to_fen for one ABI state (host only).  Returns the length written (without the terminating NUL).
int tafl_state_to_fen(const tafl_state* st, uint32_t word_bits, char* out, uint32_t cap) {
    if (!st || !out || cap == 0) return fail(TAFL_ERR_INVALID_ARG, "tafl_state_to_fen: null argument");
    std::string f;
    if (state_to_fen(st, word_bits, &f)) return fail(TAFL_ERR_INVALID_ARG, "tafl_state_to_fen: bad word size / side length");
    if (f.size() + 1 > cap) return fail(TAFL_ERR_CAPACITY, "tafl_state_to_fen: buffer too small");
    memcpy(out, f.c_str(), f.size() + 1);
    return (int)f.size();
}

int tafl_batch_create(tafl_ctx* c, uint32_t n, tafl_batch** out) {
    if (!c || !out || n == 0) return fail(TAFL_ERR_INVALID_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    tafl_batch* b = new (std::nothrow) tafl_batch();
    if (!b) return fail(TAFL_ERR_OOM, "out of host memory");
    b->ctx = c; b->n = n; b->has_mem = false; b->reserved_sims = 0; b->ran = false; b->soa = nullptr; b->g_has = false; b->g_max_sims = 0; b->trace_rounds = 0;
#ifdef TAFL_EXPERIMENT_SPEC_K
    b->spec_k = TAFL_EXPERIMENT_SPEC_K;      // measurement builds only
#else
    b->spec_k = TAFL_MCTS_MAX_SLOTS;
#endif
    b->n_sstreams = 0; b->half_recorded = false; b->stats_ok = false;
    memset(&b->mem, 0, sizeof b->mem); memset(&b->last_stats, 0, sizeof b->last_stats); memset(&b->plan, 0, sizeof b->plan);
    const size_t bytes = (size_t)quads_of(c) * n * sizeof(Quad);
    if (hipMalloc((void**)&b->soa, bytes) != hipSuccess) { delete b; return fail(TAFL_ERR_OOM, "hipMalloc(batch states) failed"); }
    if (hipMemsetAsync(b->soa, 0, bytes, c->stream) != hipSuccess) { (void)hipFree(b->soa); delete b; return fail(TAFL_ERR_HIP, "hipMemsetAsync failed"); }
    c->live_batches += 1;
    *out = b;
    return TAFL_OK;
}

int tafl_batch_destroy(tafl_batch* b) {
    if (!b) return TAFL_OK;
    if (b->ctx->live_batches > 0) b->ctx->live_batches -= 1;
    (void)hipSetDevice(b->ctx->device);
    for (uint32_t k = 0; k < b->n_sstreams; ++k) {           // a search in flight is abandoned: let its launches drain, then free
        (void)hipStreamSynchronize(b->sstream[k]); (void)hipStreamDestroy(b->sstream[k]); (void)hipEventDestroy(b->ev_fork[k]);
    }
    if (b->n_sstreams) { (void)hipEventDestroy(b->ev_start); (void)hipEventDestroy(b->ev_half); }
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->soa) (void)hipFree(b->soa);
    DevBuf* bufs[] = {&b->plays, &b->effects, &b->counts, &b->masks, &b->codes, &b->ranks, &b->results, &b->out_plays, &b->u8out, &b->plies,
                      &b->node_state, &b->hdr, &b->edges, &b->node_top, &b->edge_top, &b->leaf, &b->kind, &b->fault, &b->stats,
                      &b->children, &b->children_n, &b->visits, &b->sim_next, &b->spec_state, &b->spec_meta, &b->spec_value, &b->spec_kind, &b->spec_reason,
                      &b->spec_plies, &b->spec_ref, &b->spec_cls, &b->spec_pend, &b->spec_bias, &b->best_plays, &b->best_visits, &b->enc, &b->policy, &b->work, &b->work_count, &b->trace, &b->ctrl, &b->sim_base, &b->sp_moves_done, &b->sp_start_round, &b->sp_plays,
                      &b->g_node_state, &b->g_hdr, &b->g_pedge, &b->g_edges, &b->g_node_top, &b->g_edge_top, &b->g_leaf, &b->g_kind, &b->g_fault, &b->g_sims,
                      &b->g_stats, &b->g_priors, &b->g_values, &b->g_boards, &b->g_sides, &b->g_wait};
    for (DevBuf* d : bufs) d->release();
    delete b;
    return TAFL_OK;
}

uint32_t tafl_batch_size(const tafl_batch* b) { return b ? b->n : 0; }

int tafl_sync(tafl_ctx* c) {
    if (!c) return fail(TAFL_ERR_INVALID_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_batch_reset_fen(tafl_batch* b, const char* fen, uint8_t side) {
    if (!b || !fen) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx;
    tafl_state st;
    int rc = tafl_state_from_fen(c, fen, side, &st);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    DISPATCH_NLW(c, {
        DState<NL> ds; state_from_abi<NL>(st, ds);
        hipLaunchKernelGGL((k_fill<NL, W>), dim3(grid_of(b->n)), dim3(TAFL_BLOCK), 0, c->stream, b->soa, b->n, ds);
    });
    HIPCHK(hipGetLastError());
    return TAFL_OK;
}

int tafl_batch_upload(tafl_batch* b, const tafl_state* states, uint32_t first, uint32_t count) {
    if (!b || !states || count == 0 || first > b->n || count > b->n - first) return fail(TAFL_ERR_INVALID_ARG, "bad range");
    tafl_ctx* c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int Q = quads_of(c);
    std::vector<Quad> stage((size_t)Q * count);
    for (uint32_t i = 0; i < count; ++i) {
        if (states[i].side_len != c->n) return fail(TAFL_ERR_INVALID_ARG, "state.side_len differs from the context's side_len");
        uint32_t v[24];
        DISPATCH_NLW(c, { DState<NL> ds; state_from_abi<NL>(states[i], ds); StateIO<NL>::pack(ds, v); });
        for (int q = 0; q < Q; ++q) { Quad t; t.x = v[4 * q]; t.y = v[4 * q + 1]; t.z = v[4 * q + 2]; t.w = v[4 * q + 3]; stage[(size_t)q * count + i] = t; }
    }
    for (int q = 0; q < Q; ++q)
        HIPCHK(hipMemcpyAsync(b->soa + (size_t)q * b->n + first, stage.data() + (size_t)q * count, sizeof(Quad) * count, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_batch_download(tafl_batch* b, tafl_state* states, uint32_t first, uint32_t count) {
    if (!b || !states || count == 0 || first > b->n || count > b->n - first) return fail(TAFL_ERR_INVALID_ARG, "bad range");
    tafl_ctx* c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int Q = quads_of(c);
    std::vector<Quad> stage((size_t)Q * count);
    for (int q = 0; q < Q; ++q)
        HIPCHK(hipMemcpyAsync(stage.data() + (size_t)q * count, b->soa + (size_t)q * b->n + first, sizeof(Quad) * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t v[24];
        for (int q = 0; q < Q; ++q) { const Quad t = stage[(size_t)q * count + i]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
        DISPATCH_NLW(c, { DState<NL> ds; StateIO<NL>::unpack(v, ds); state_to_abi<NL>(ds, (uint8_t)c->n, states[i]); });
    }
    return TAFL_OK;
}

// ---- hot path --------------------------------------------------------------------------------------
#define NEED(buf, bytes) do { if ((buf).ensure(bytes)) return fail(TAFL_ERR_OOM, "hipMalloc(workspace) failed"); } while (0)

int tafl_movegen(tafl_batch* b, uint32_t* out_counts, uint32_t* out_masks) {
    if (!b) return fail(TAFL_ERR_INVALID_ARG, "null batch");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, mw = tafl_action_mask_words(c);
    HIPCHK(hipSetDevice(c->device));
    NEED(b->counts, sizeof(uint32_t) * n);
    if (out_masks) NEED(b->masks, sizeof(uint32_t) * (size_t)n * mw);
    {
        SpanGuard sg(c, KC_MOVEGEN);
        if (out_masks) {
            DISPATCH_NLW(c, hipLaunchKernelGGL((k_movegen_masks<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK * c->n), TAFL_BLOCK * ((mw | 1u) + 1u) * sizeof(uint32_t), c->stream,
                                               CC, b->soa, n, (uint32_t*)b->counts.p, (uint32_t*)b->masks.p, mw));
        } else {
            DISPATCH_NLW(c, hipLaunchKernelGGL((k_movegen<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n, (uint32_t*)b->counts.p));
        }
    }
    HIPCHK(hipGetLastError());
    if (out_counts) HIPCHK(hipMemcpyAsync(out_counts, b->counts.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    if (out_masks) HIPCHK(hipMemcpyAsync(out_masks, b->masks.p, sizeof(uint32_t) * (size_t)n * mw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_validate(tafl_batch* b, const tafl_play* plays, uint8_t* out_codes) {
    if (!b || !plays || !out_codes) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->plays, sizeof(tafl_play) * n); NEED(b->codes, n);
    HIPCHK(hipMemcpyAsync(b->plays.p, plays, sizeof(tafl_play) * n, hipMemcpyHostToDevice, c->stream));
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_validate<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n,
                                       (const tafl_play*)b->plays.p, (uint8_t*)b->codes.p));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_codes, b->codes.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_step(tafl_batch* b, const tafl_play* plays, tafl_effects* out_effects) {
    if (!b || !plays) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->plays, sizeof(tafl_play) * n);
    if (out_effects) NEED(b->effects, sizeof(tafl_effects) * n);
    HIPCHK(hipMemcpyAsync(b->plays.p, plays, sizeof(tafl_play) * n, hipMemcpyHostToDevice, c->stream));
    {
        SpanGuard sg(c, KC_STEP);
        if (c->dense13) hipLaunchKernelGGL(k_step_dense13, dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, c->c8, c->c6, b->soa, n,
                                           (const tafl_play*)b->plays.p, out_effects ? (tafl_effects*)b->effects.p : nullptr);
        else DISPATCH_NLW(c, hipLaunchKernelGGL((k_step<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n,
                                           (const tafl_play*)b->plays.p, out_effects ? (tafl_effects*)b->effects.p : nullptr));
    }
    HIPCHK(hipGetLastError());
    if (out_effects) HIPCHK(hipMemcpyAsync(out_effects, b->effects.p, sizeof(tafl_effects) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_step_kth(tafl_batch* b, const uint32_t* ranks, tafl_play* out_plays, tafl_effects* out_effects) {
    if (!b || !ranks) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->ranks, sizeof(uint32_t) * n); NEED(b->counts, sizeof(uint32_t) * 2 * (size_t)n);      // counts: chosen action and number of plays per game
    if (out_plays) NEED(b->out_plays, sizeof(tafl_play) * n);
    if (out_effects) NEED(b->effects, sizeof(tafl_effects) * n);
    HIPCHK(hipMemcpyAsync(b->ranks.p, ranks, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream));
    {
        SpanGuard sg(c, KC_STEP);
        const uint32_t mw = tafl_action_mask_words(c);
        DISPATCH_NLW(c, hipLaunchKernelGGL((k_select_kth<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK * c->n), TAFL_BLOCK * ((mw | 1u) + 1u + 17u) * sizeof(uint32_t), c->stream,
                                           CC, b->soa, n, (const uint32_t*)b->ranks.p, (uint32_t*)b->counts.p, (uint32_t*)b->counts.p + n, mw));
        if (c->dense13) hipLaunchKernelGGL(k_step_action_dense13, dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, c->c8, c->c6, b->soa, n, (const uint32_t*)b->counts.p,
                                           (const uint32_t*)b->counts.p + n, out_plays ? (tafl_play*)b->out_plays.p : nullptr, out_effects ? (tafl_effects*)b->effects.p : nullptr);
        else DISPATCH_NLW(c, hipLaunchKernelGGL((k_step_action<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n, (const uint32_t*)b->counts.p,
                                           (const uint32_t*)b->counts.p + n, out_plays ? (tafl_play*)b->out_plays.p : nullptr, out_effects ? (tafl_effects*)b->effects.p : nullptr));
    }
    HIPCHK(hipGetLastError());
    if (out_plays) HIPCHK(hipMemcpyAsync(out_plays, b->out_plays.p, sizeof(tafl_play) * n, hipMemcpyDeviceToHost, c->stream));
    if (out_effects) HIPCHK(hipMemcpyAsync(out_effects, b->effects.p, sizeof(tafl_effects) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_side_can_play(tafl_batch* b, uint8_t side, uint8_t* out) {
    if (!b || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->u8out, n);
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_side_can_play<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n,
                                       side ? 1u : 0u, (uint8_t*)b->u8out.p));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, b->u8out.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_rollout(tafl_batch* b, uint64_t seed, uint32_t sim, uint32_t max_plies, uint64_t game_id_base, tafl_rollout_result* out) {
    if (!b || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->results, sizeof(tafl_rollout_result) * n);
    {
        SpanGuard sg(c, KC_ROLLOUT);
        DISPATCH_ARENA_PRESET(c, hipLaunchKernelGGL((k_rollout<NLS, WS, NL, W, PRESET>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n, seed, sim, max_plies,
                                           game_id_base, (tafl_rollout_result*)b->results.p));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, b->results.p, sizeof(tafl_rollout_result) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_random_advance(tafl_batch* b, uint64_t seed, const uint32_t* plies, uint64_t game_id_base) {
    if (!b || !plies) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->plies, sizeof(uint32_t) * n);
    HIPCHK(hipMemcpyAsync(b->plies.p, plies, sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream));
    DISPATCH_PRESET(c, hipLaunchKernelGGL((k_random_advance<NL, W, PRESET>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, n, seed,
                                       (const uint32_t*)b->plies.p, game_id_base));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

// ---- MCTS ----------------------------------------------------------------------------------------------
int tafl_mcts_reserve(tafl_batch* b, uint32_t max_sims) {
    if (!b || max_sims == 0 || max_sims > 60000) return fail(TAFL_ERR_INVALID_ARG, "max_sims must be in 1..60000");
    tafl_ctx* c = b->ctx; const size_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    if (b->has_mem && b->reserved_sims >= max_sims) return TAFL_OK;
    const size_t node_cap = (size_t)max_sims + 1, edge_cap = 4 * ((size_t)max_sims + 1);
    NEED(b->node_state, node_cap * n * arena_quads(c) * sizeof(Quad));
    NEED(b->hdr, node_cap * n * sizeof(NodeHdr));
    NEED(b->edges, edge_cap * n * sizeof(Edge));
    NEED(b->node_top, n * 4); NEED(b->edge_top, n * 4); NEED(b->leaf, n * 4);
    NEED(b->kind, n); NEED(b->fault, n);
    NEED(b->stats, sizeof(unsigned long long) * ST_COUNT);
    const size_t k = b->spec_k;
    NEED(b->sim_next, n * 4); NEED(b->spec_state, k * n * arena_quads(c) * sizeof(Quad)); NEED(b->spec_value, k * n); NEED(b->spec_kind, k * n); NEED(b->spec_meta, k * n * 4);
    NEED(b->spec_reason, k * n); NEED(b->spec_plies, k * n * 4); NEED(b->spec_ref, k * n * 4); NEED(b->spec_cls, k * n); NEED(b->spec_pend, n * 4); NEED(b->spec_bias, n * 4);
    NEED(b->work, k * n * 4); NEED(b->work_count, 4 * 2 * TAFL_MCTS_MAX_SLOTS * TAFL_MCTS_MAX_PARTS); NEED(b->trace, 8 * TAFL_MCTS_TRACE_ROUNDS);
    NEED(b->ctrl, sizeof(unsigned long long) * CT_COUNT); NEED(b->sim_base, n * 4);
    b->mem.node_state = (Quad*)b->node_state.p; b->mem.hdr = (NodeHdr*)b->hdr.p; b->mem.edges = (Edge*)b->edges.p;
    b->mem.node_top = (uint32_t*)b->node_top.p; b->mem.edge_top = (uint32_t*)b->edge_top.p; b->mem.leaf = (uint32_t*)b->leaf.p;
    b->mem.kind = (uint8_t*)b->kind.p; b->mem.fault = (uint8_t*)b->fault.p;
    b->mem.sim_base = (uint32_t*)b->sim_base.p; b->mem.sim_next = (uint32_t*)b->sim_next.p; b->mem.spec_state = (Quad*)b->spec_state.p; b->mem.spec_value = (int8_t*)b->spec_value.p; b->mem.spec_meta = (uint32_t*)b->spec_meta.p;
    b->mem.spec_kind = (uint8_t*)b->spec_kind.p; b->mem.spec_reason = (uint8_t*)b->spec_reason.p; b->mem.spec_plies = (uint32_t*)b->spec_plies.p;
    b->mem.spec_ref = (uint32_t*)b->spec_ref.p; b->mem.spec_cls = (uint8_t*)b->spec_cls.p; b->mem.spec_pend = (uint32_t*)b->spec_pend.p; b->mem.spec_bias = (uint32_t*)b->spec_bias.p;
    b->mem.spec_k = b->spec_k;
    b->mem.G = b->n; b->mem.node_cap = (uint32_t)node_cap; b->mem.edge_cap = (uint32_t)edge_cap;
    b->mem.flags = 0;
    b->has_mem = true; b->reserved_sims = max_sims;
    return TAFL_OK;
}

// ---- the search driver -------------------------------------------------------------------------------------------------------------
// tafl_mcts_run_async enqueues a whole search - the planned rounds plus the rounds its stragglers usually need - on the batch's own
// streams WITHOUT reading anything back: plan and width control live on the device (k_mcts_tree / k_mcts_rollout read the counters
// themselves), launches for a finished batch return at once.  tafl_mcts_wait joins the streams, reads the counters and, while games are
// still unfinished, runs further rounds: a search always completes (or the call fails), however many rounds its slowest game needs.
// tafl_mcts_run = the two back to back.  Results never depend on how a search was enqueued.
static int search_streams(tafl_batch* b, uint32_t parts) {
    if (b->n_sstreams == 0) {
        HIPCHK(hipEventCreateWithFlags(&b->ev_start, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ev_half, hipEventDisableTiming));
    }
    while (b->n_sstreams < parts) {
        const uint32_t k = b->n_sstreams;
        HIPCHK(hipStreamCreateWithFlags(&b->sstream[k], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->ev_fork[k], hipEventDisableTiming));
        b->n_sstreams = k + 1;
    }
    return TAFL_OK;
}

// rounds [next_round, next_round + count) of the two-kernel pipeline: per round and partition one tree launch (one game per lane: backups,
// real selections, slot matching, prediction of the next slots, dense work lists) and one playout launch over the work lists
static int mcts_enqueue_rounds(tafl_batch* b, uint32_t count, bool stagger) {
    tafl_ctx* c = b->ctx; SearchPlan& sp = b->plan; const tafl_mcts_params* p = &sp.p;
    unsigned long long* st = (unsigned long long*)b->stats.p; unsigned long long* ctrl = (unsigned long long*)b->ctrl.p;
    const MctsMem& M = sp.M;
    for (uint32_t r = 0; r < count; ++r) {
        const uint32_t i = sp.next_round;
        if (i >= sp.max_rounds) return fail(TAFL_ERR_CAPACITY, "tafl_mcts: the search did not finish within its round bound");
        b->trace_rounds = i + 1;
        for (uint32_t k = 0; k < sp.parts; ++k) {
            const SearchPart& pk = sp.P[k];
            uint32_t* wc_now = pk.wc + (i & 1u) * TAFL_MCTS_MAX_SLOTS; uint32_t* wc_next = pk.wc + ((i + 1u) & 1u) * TAFL_MCTS_MAX_SLOTS;
            {
                SpanGuard sg(c, KC_MCTS_TREE, pk.s);
                if (sp.selfplay.n_moves)
                    DISPATCH_ARENA_PRESET(c, hipLaunchKernelGGL((k_mcts_tree_selfplay<NLS, WS, NL, W, PRESET>), dim3(pk.grid_tree), dim3(TAFL_BLOCK), TAFL_UNDO_LDS_BYTES(TAFL_MCTS_UNDO_CAP), pk.s, CC, M, p->c_puct, p->n_sims, i, sp.planned, sp.probe_every,
                                                          sp.slots, st, ctrl, pk.wl, wc_now, pk.g0, pk.g1, b->soa, sp.selfplay));
                else
                    DISPATCH_ARENA_PRESET(c, hipLaunchKernelGGL((k_mcts_tree<NL, W, PRESET>), dim3(pk.grid_tree), dim3(TAFL_BLOCK), TAFL_UNDO_LDS_BYTES(TAFL_MCTS_UNDO_CAP), pk.s, CC, M, p->c_puct, p->n_sims, i, sp.planned, sp.probe_every,
                                                          sp.slots, st, ctrl, pk.wl, wc_now, pk.g0, pk.g1));
            }
            // partition k+1 starts behind partition k's first tree launch: from then on the tree phases are spread over a round
            if (stagger && k + 1 < sp.parts) {
                if (hipEventRecord(b->ev_fork[k + 1], pk.s) != hipSuccess || hipStreamWaitEvent(sp.P[k + 1].s, b->ev_fork[k + 1], 0) != hipSuccess) return fail(TAFL_ERR_HIP, "tafl_mcts: stream fork failed");
            }
            {
                SpanGuard sg(c, KC_MCTS_ROLLOUT, pk.s);
                uint32_t* tr = (k == 0 && i < TAFL_MCTS_TRACE_ROUNDS) ? (uint32_t*)b->trace.p + 2 * i : nullptr;      // the first partition's rounds are traced
                DISPATCH_ARENA_PRESET(c, hipLaunchKernelGGL((k_mcts_rollout<NL, W, PRESET>), dim3(pk.grid_roll), dim3(TAFL_BLOCK), 0, pk.s, CC, M, p->seed,
                                                      sp.base, p->sim_offset, p->max_rollout_plies, pk.wl, wc_now, wc_next, pk.g1 - pk.g0, pk.cap, st, tr,
                                                      k == 0 ? ctrl : nullptr, i, sp.planned, sp.probe_every));
            }
        }
        stagger = false;
        sp.next_round = i + 1;
        if (!b->half_recorded && 2u * (i + 1u) >= sp.planned) {          // a search started "after" this one begins here (tafl_mcts_run_async_after)
            if (hipEventRecord(b->ev_half, sp.P[0].s) != hipSuccess) return fail(TAFL_ERR_HIP, "tafl_mcts: hipEventRecord failed");
            b->half_recorded = true;
        }
    }
    HIPCHK(hipGetLastError());
    return TAFL_OK;
}
static int mcts_enqueue_fused(tafl_batch* b, uint32_t rounds) {
    tafl_ctx* c = b->ctx; SearchPlan& sp = b->plan; const tafl_mcts_params* p = &sp.p; const uint32_t n = b->n, bps = grid_of(n);
    unsigned long long* st = (unsigned long long*)b->stats.p; const MctsMem& M = sp.M; hipStream_t s0 = sp.P[0].s;
    SpanGuard sg(c, KC_MCTS_ROLLOUT, s0);
    constexpr int NL = 2, W = 7; const Consts<2>& CC = c->c2;
    if (c->preset == PRESET_BRANDUBH7) {
        if (sp.slots == 2) hipLaunchKernelGGL((k_mcts_fused<NL, W, PRESET_BRANDUBH7, 2>), dim3((n + 31) / 32), dim3(TAFL_BLOCK), TAFL_UNDO_LDS_BYTES(TAFL_MCTS_UNDO_CAP_FUSED), s0, CC, M, p->c_puct, p->n_sims, p->seed, sp.base, p->sim_offset, p->max_rollout_plies, rounds, st);
        else hipLaunchKernelGGL((k_mcts_fused<NL, W, PRESET_BRANDUBH7, 1>), dim3(bps), dim3(TAFL_BLOCK), 0, s0, CC, M, p->c_puct, p->n_sims, p->seed, sp.base, p->sim_offset, p->max_rollout_plies, rounds, st);
    } else {
        if (sp.slots == 2) hipLaunchKernelGGL((k_mcts_fused<NL, W, PRESET_NONE, 2>), dim3((n + 31) / 32), dim3(TAFL_BLOCK), TAFL_UNDO_LDS_BYTES(TAFL_MCTS_UNDO_CAP_FUSED), s0, CC, M, p->c_puct, p->n_sims, p->seed, sp.base, p->sim_offset, p->max_rollout_plies, rounds, st);
        else hipLaunchKernelGGL((k_mcts_fused<NL, W, PRESET_NONE, 1>), dim3(bps), dim3(TAFL_BLOCK), 0, s0, CC, M, p->c_puct, p->n_sims, p->seed, sp.base, p->sim_offset, p->max_rollout_plies, rounds, st);
    }
    sp.next_round += rounds;
    return TAFL_OK;
}

int tafl_mcts_wait(tafl_batch* b);

static int mcts_begin(tafl_batch* b, const tafl_mcts_params* p, uint64_t game_id_base, tafl_batch* after, uint32_t n_moves = 0) {
    if (!b || !p) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    if (p->flags & ~(uint32_t)TAFL_MCTS_FLAGS_KNOWN) return fail(TAFL_ERR_UNSUPPORTED, "tafl_mcts_params.flags: unknown bits set");
    if (p->n_sims == 0) return fail(TAFL_ERR_INVALID_ARG, "n_sims must be > 0");
    if (after && after->ctx->device != b->ctx->device) return fail(TAFL_ERR_INVALID_ARG, "tafl_mcts_run_async_after: the two batches live on different devices");
    int rc = TAFL_OK;
    if (b->plan.active && (rc = tafl_mcts_wait(b)) != TAFL_OK) return rc;       // one search per batch at a time
    if ((rc = tafl_mcts_reserve(b, p->n_sims)) != TAFL_OK) return rc;
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    SearchPlan& sp = b->plan;
    sp.p = *p; sp.base = game_id_base; sp.next_round = 0; sp.fused = false;
    sp.selfplay.n_moves = n_moves; sp.selfplay.moves_done = nullptr; sp.selfplay.start_round = nullptr; sp.selfplay.plays = nullptr;
    MctsMem& M = sp.M; M = b->mem;
    M.node_cap = p->n_sims + 1; M.edge_cap = b->mem.edge_cap; M.flags = p->flags & TAFL_MCTS_FLAG_FPU_INF;
    // tuning fields of `flags` (results never depend on them): pipeline and playout slots per game
    const uint32_t pipe = TAFL_MCTS_TUNE_PIPELINE_OF(p->flags);
    uint32_t slots = TAFL_MCTS_TUNE_SLOTS_OF(p->flags);
    if (pipe > TAFL_MCTS_PIPELINE_TWO_KERNEL) return fail(TAFL_ERR_UNSUPPORTED, "tafl_mcts_params.flags: unknown pipeline");
    // default: the two-kernel pipeline; 7x7 boards have playouts so short (Brandubh: ~65 plies) that the per-round launches and the
    // exposed tree phase cost more than the fused kernel's two waves per SIMD (measured: 92 M vs 31 M sims/s on 7x7)
    // The fused kernel is built for 64-bit boards only (7x7): on wider boards its tree phase does not fit the register file beside the
    // playout loop (256 VGPRs and spills) and the two-kernel pipeline is faster anyway (13x13: 44.9 M vs 37.4 M sims/s).
    if (pipe == TAFL_MCTS_PIPELINE_FUSED && c->nl != 2) return fail(TAFL_ERR_UNSUPPORTED, "the fused pipeline exists for 64-bit boards (word_bits 64) only");
    if (n_moves && pipe == TAFL_MCTS_PIPELINE_FUSED) return fail(TAFL_ERR_UNSUPPORTED, "tafl_selfplay_run uses the two-kernel pipeline");
    const bool fused = !n_moves && (pipe == TAFL_MCTS_PIPELINE_FUSED || (pipe == TAFL_MCTS_PIPELINE_DEFAULT && c->nl == 2 && slots <= 2));
    if (fused && slots == 0) slots = 2;
    if (fused && slots > 2) return fail(TAFL_ERR_UNSUPPORTED, "the fused pipeline has 1 or 2 playout slots per game");
    uint32_t capacity = 0;
    if (!fused) {
        if (c->rollout_capacity == 0) {
            int blocks = 0; hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, c->device));
            DISPATCH_ARENA_PRESET(c, { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_mcts_rollout<NL, W, PRESET>, TAFL_BLOCK, 0) != hipSuccess) blocks = 0; });
            if (blocks < 1) blocks = 8;
            c->rollout_capacity = (uint32_t)blocks * (uint32_t)prop.multiProcessorCount * TAFL_BLOCK;
        }
        capacity = c->rollout_capacity;
        // batches that are searched side by side (tafl_mcts_run_async) each take their share of the device: a round that fills its share
        // EXACTLY keeps every SIMD at the same number of waves - a few workgroups more and some SIMDs carry one wave more than the others,
        // and the whole round takes that wave's time (1 / 2 / 3 / 4 waves per SIMD: 1.0 / 1.5 / 2.0 / 2.55 ms per 512-ply round)
        if (TAFL_MCTS_TUNE_SHARE_OF(p->flags) > 1) { capacity = capacity / TAFL_MCTS_TUNE_SHARE_OF(p->flags) / TAFL_BLOCK * TAFL_BLOCK; if (capacity < TAFL_BLOCK) capacity = TAFL_BLOCK; }
        if (slots == 0) {                                           // what fills the device exactly: 4 slots per game at 65 536 games on 11x11
            slots = capacity / n;
            if (slots < 1) slots = 1;
        }
    }
    if (slots > b->spec_k) slots = b->spec_k;
    sp.fused = fused; sp.slots = slots;
    // The batch is cut into partitions (two by default) that run the same pipeline on their own streams, started one tree launch apart: the
    // tree phase of a partition (latency- and divergence-bound, one wave per 64 games) then runs under the playouts of the others instead
    // of on an idle device, and the partitions' rounds interleave instead of ending together.  Each partition may fill its share of the
    // device.  Small batches stay in one piece.
    uint32_t parts = (!fused && n >= 8192u) ? 2u : 1u;             // measured at 65 536 games, S = 64: 1: 56.9, 2: 64.6, 3: 61.2, 4: 61.0, 8: 35.7 M sims/s
    if (!fused && TAFL_MCTS_TUNE_PARTS_OF(p->flags)) { parts = TAFL_MCTS_TUNE_PARTS_OF(p->flags); if (parts > TAFL_MCTS_MAX_PARTS) parts = TAFL_MCTS_MAX_PARTS; }
    if (parts > grid_of(n)) parts = grid_of(n);
    sp.parts = parts;
    if ((rc = search_streams(b, parts)) != TAFL_OK) return rc;
    // the search starts behind everything enqueued on the context's stream so far (uploads, steps ...) and, if asked for, behind the first
    // half of another batch's search in flight
    HIPCHK(hipEventRecord(b->ev_start, c->stream));
    HIPCHK(hipStreamWaitEvent(b->sstream[0], b->ev_start, 0));
    if (after && after != b && after->plan.active && after->half_recorded) HIPCHK(hipStreamWaitEvent(b->sstream[0], after->ev_half, 0));
    b->half_recorded = false;
    hipStream_t s0 = b->sstream[0];
    unsigned long long* st = (unsigned long long*)b->stats.p;
    HIPCHK(hipMemsetAsync(st, 0, sizeof(unsigned long long) * ST_COUNT, s0));
    M.spec_k = fused ? slots : b->spec_k;
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_init<NLS, WS, NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, s0, CC, b->soa, M));
    b->ran = false; b->trace_rounds = 0;
    if (fused) {
        // Fused pipeline (k_mcts_fused): one wave owns 64 / K games for a whole chunk of rounds and leaves as soon as its games are done;
        // n_sims + 1 rounds always suffice (every round completes at least one simulation of every live game)
        sp.P[0].s = s0; sp.planned = p->n_sims + 1; sp.max_rounds = p->n_sims + 1; sp.probe_every = 0;
        uint32_t left = p->n_sims + 1, chunk_len = 8;
        while (left > 0) {
            const uint32_t rounds = chunk_len < left ? chunk_len : left;
            if ((rc = mcts_enqueue_fused(b, rounds)) != TAFL_OK) return rc;
            left -= rounds;
            if (chunk_len < 16) chunk_len *= 2;
            if (!b->half_recorded && 2u * sp.next_round >= sp.planned) { HIPCHK(hipEventRecord(b->ev_half, s0)); b->half_recorded = true; }
        }
        HIPCHK(hipGetLastError());
        sp.active = true;
        return TAFL_OK;
    }
    // Two-kernel pipeline.  The search is planned for ceil(n_sims / slots) rounds (+ slack, below); every game issues ceil(remaining / rounds left) slots, so
    // games that lost a round to a misprediction catch up instead of trailing behind in nearly empty rounds.
    // A long search gets one round of slack in eight: its rounds are then not quite full, so that a game that lost a round to a failed
    // prediction finds room for the extra playout that lets it catch up inside the plan (S = 256: 95.5 -> 98.3 M sims/s, S = 1000:
    // 73.2 -> 74.0 M; a short search loses more to the longer plan than it wins: S = 64 97.1 -> 91.9 M with 18 rounds instead of 16).
    const uint32_t tight = (p->n_sims + slots - 1) / slots;
    const uint32_t planned = tight + (tight >= 32u ? tight / 8u : 0u);
    uint32_t* wlist = (uint32_t*)b->work.p; uint32_t* wcount = (uint32_t*)b->work_count.p;
    {
        const uint32_t waves = grid_of(n), per = waves / parts, extra = waves % parts;
        uint32_t w0 = 0; size_t wl_off = 0;
        for (uint32_t k = 0; k < parts; ++k) {
            SearchPart& P = sp.P[k];
            const uint32_t wk = per + (k < extra ? 1u : 0u);
            P.g0 = w0 * TAFL_BLOCK; P.g1 = (w0 + wk) * TAFL_BLOCK < n ? (w0 + wk) * TAFL_BLOCK : n;
            w0 += wk;
            const uint32_t cnt = P.g1 - P.g0;
            P.cap = parts > 1 ? (uint32_t)((unsigned long long)capacity * cnt / n / TAFL_BLOCK * TAFL_BLOCK) : capacity;
            if (P.cap < TAFL_BLOCK) P.cap = TAFL_BLOCK;
            const unsigned long long most = (unsigned long long)cnt * M.spec_k;
            P.grid_tree = grid_of(cnt);
            P.grid_roll = (uint32_t)(((most < P.cap ? most : P.cap) + TAFL_BLOCK - 1) / TAFL_BLOCK);
            P.s = b->sstream[k];
            P.wl = wlist + wl_off; wl_off += (size_t)cnt * M.spec_k;
            P.wc = wcount + (size_t)k * 2 * TAFL_MCTS_MAX_SLOTS;              // two counter sets per partition: the playout launch of a round clears the next round's
        }
    }
    // every round runs min(capacity, requested) playouts, the pending leaf of every waiting game first (class 0): progress is guaranteed
    // a large batch always has stragglers that need a few rounds more than the plan (65 536 games: 5 - 6)
    const uint32_t tail_guess = n >= 32768u ? 6u : n >= 4096u ? 4u : n >= 512u ? 2u : 1u;
    const unsigned long long rounds_bound = ((unsigned long long)p->n_sims + 2 + tail_guess) * (1 + (unsigned long long)n / (capacity ? capacity : 1));
    sp.max_rounds = rounds_bound > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)rounds_bound;
    sp.planned = planned;
    // long searches: width control every 16 rounds (k_mcts_rollout)
    sp.probe_every = planned >= 64 ? 16u : 0u;
    sp.ctrl0[CT_WCAP] = M.spec_k - 1u; sp.ctrl0[CT_LAST_ISSUED] = sp.ctrl0[CT_LAST_HITS] = 0ull; sp.ctrl0[CT_NEXT_CHECK] = sp.probe_every ? sp.probe_every : ~0ull;
    HIPCHK(hipMemcpyAsync(b->ctrl.p, sp.ctrl0, sizeof sp.ctrl0, hipMemcpyHostToDevice, s0));      // (the source lives in the batch until the search is joined)
    HIPCHK(hipMemsetAsync(b->trace.p, 0, 8 * TAFL_MCTS_TRACE_ROUNDS, s0));
    HIPCHK(hipMemsetAsync(wcount, 0, sizeof(uint32_t) * 2 * TAFL_MCTS_MAX_SLOTS * TAFL_MCTS_MAX_PARTS, s0));
    uint32_t first = planned + tail_guess;
    if (n_moves) {
        // a self-play run: n_moves searches per game, each game at its own pace (k_mcts_tree_selfplay); most games need planned + 0..1 rounds per search
        NEED(b->sp_moves_done, sizeof(uint32_t) * (size_t)n); NEED(b->sp_start_round, sizeof(uint32_t) * (size_t)n); NEED(b->sp_plays, sizeof(tafl_play) * (size_t)n * n_moves);
        sp.selfplay.moves_done = (uint32_t*)b->sp_moves_done.p; sp.selfplay.start_round = (uint32_t*)b->sp_start_round.p; sp.selfplay.plays = (tafl_play*)b->sp_plays.p;
        HIPCHK(hipMemsetAsync(b->sp_moves_done.p, 0, sizeof(uint32_t) * (size_t)n, s0)); HIPCHK(hipMemsetAsync(b->sp_start_round.p, 0, sizeof(uint32_t) * (size_t)n, s0));
        HIPCHK(hipMemsetAsync(b->sp_plays.p, 0, sizeof(tafl_play) * (size_t)n * n_moves, s0));
        const unsigned long long all = (unsigned long long)sp.max_rounds * n_moves;
        sp.max_rounds = all > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)all;
        const unsigned long long want = (unsigned long long)(planned + 1u) * n_moves + tail_guess;
        first = want > sp.max_rounds ? sp.max_rounds : (uint32_t)want;
    }
    if (first > sp.max_rounds) first = sp.max_rounds;
    if ((rc = mcts_enqueue_rounds(b, first, parts > 1)) != TAFL_OK) return rc;
    sp.active = true;
    return TAFL_OK;
}

int tafl_mcts_run_async(tafl_batch* b, const tafl_mcts_params* p, uint64_t game_id_base) { return mcts_begin(b, p, game_id_base, nullptr); }
int tafl_mcts_run_async_after(tafl_batch* b, const tafl_mcts_params* p, uint64_t game_id_base, tafl_batch* other) { return mcts_begin(b, p, game_id_base, other); }

int tafl_mcts_wait(tafl_batch* b) {
    if (!b) return fail(TAFL_ERR_INVALID_ARG, "null batch");
    SearchPlan& sp = b->plan;
    if (!sp.active) return b->ran ? TAFL_OK : fail(TAFL_ERR_INVALID_ARG, "tafl_mcts_wait: no search was started on this batch");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    sp.active = false;                                                      // whatever happens below, the plan is over
    for (;;) {
        for (uint32_t k = 0; k < sp.parts; ++k) HIPCHK(hipStreamSynchronize(sp.P[k].s));
        unsigned long long h[ST_COUNT];
        HIPCHK(hipMemcpyAsync(h, b->stats.p, sizeof h, hipMemcpyDeviceToHost, sp.P[0].s));
        HIPCHK(hipStreamSynchronize(sp.P[0].s));
        if (h[ST_DONE] >= (unsigned long long)n) break;                     // every game has consumed its last playout
        if (sp.fused) return fail(TAFL_ERR_HIP, "tafl_mcts_wait: the fused search ended with unfinished games");
        // stragglers: a few more rounds, then look again (a round of a nearly finished batch is a lone wave per partition)
        const int rc = mcts_enqueue_rounds(b, sp.planned >= 32 ? 4u : 2u, false);
        if (rc) return rc;
    }
    b->ran = true; b->stats_ok = true;
    return TAFL_OK;
}

int tafl_mcts_run(tafl_batch* b, const tafl_mcts_params* p, uint64_t game_id_base) {
    const int rc = mcts_begin(b, p, game_id_base, nullptr);
    return rc ? rc : tafl_mcts_wait(b);
}

// n_moves x { tafl_mcts_run with sim_offset + move * n_sims; tafl_mcts_play_best } for every game, without leaving the device and without a
// barrier between the moves: a game starts its next search as soon as its own is done (SelfPlay, tafl_ops.hpp)
int tafl_selfplay_run(tafl_batch* b, const tafl_mcts_params* p, uint32_t n_moves, uint64_t game_id_base, tafl_play* out_plays) {
    if (!b || !p || n_moves == 0) return fail(TAFL_ERR_INVALID_ARG, "tafl_selfplay_run: bad argument");
    if ((unsigned long long)n_moves * p->n_sims + p->sim_offset > 0xFFFFFFFFull) return fail(TAFL_ERR_INVALID_ARG, "tafl_selfplay_run: sim_offset + n_moves * n_sims exceeds 32 bits");
    int rc = mcts_begin(b, p, game_id_base, nullptr, n_moves);
    if (rc == TAFL_OK) rc = tafl_mcts_wait(b);
    b->ran = false;                                          // the trees belong to roots that have been played away from
    if (rc) return rc;
    if (out_plays) {
        tafl_ctx* c = b->ctx;
        HIPCHK(hipMemcpyAsync(out_plays, b->sp_plays.p, sizeof(tafl_play) * (size_t)b->n * n_moves, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return TAFL_OK;
}

// every reader of a search's results joins a search in flight first; non-zero = there is no finished search to read
static int search_done(tafl_batch* b) {
    if (b->plan.active && tafl_mcts_wait(b) != TAFL_OK) return 1;
    return b->ran ? 0 : 1;
}

// measurement: playouts requested / run in every round of the last two-kernel search (0 rounds after a fused search)
int tafl_mcts_round_trace(tafl_batch* b, uint32_t* requested, uint32_t* run, uint32_t cap, uint32_t* n_rounds) {
    if (!b || !n_rounds) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    if (b->plan.active && tafl_mcts_wait(b) != TAFL_OK) return TAFL_ERR_HIP;
    if (!b->stats_ok) return fail(TAFL_ERR_INVALID_ARG, "no MCTS run on this batch");
    tafl_ctx* c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    const uint32_t k = b->trace_rounds < TAFL_MCTS_TRACE_ROUNDS ? b->trace_rounds : TAFL_MCTS_TRACE_ROUNDS;
    std::vector<uint32_t> h((size_t)2 * (k ? k : 1));
    if (k) { HIPCHK(hipMemcpyAsync(h.data(), b->trace.p, sizeof(uint32_t) * 2 * k, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    for (uint32_t i = 0; i < k && i < cap; ++i) { if (requested) requested[i] = h[2 * i]; if (run) run[i] = h[2 * i + 1]; }
    *n_rounds = k;
    return TAFL_OK;
}

int tafl_mcts_get_stats(tafl_batch* b, tafl_mcts_stats* out) {
    if (!b || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    if (b->plan.active && tafl_mcts_wait(b) != TAFL_OK) return TAFL_ERR_HIP;
    if (!b->stats_ok) return fail(TAFL_ERR_INVALID_ARG, "no MCTS run on this batch");
    tafl_ctx* c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    unsigned long long h[ST_COUNT];
    HIPCHK(hipMemcpyAsync(h, b->stats.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof *out);
    out->sims = h[ST_SIMS]; out->rollouts = h[ST_ROLLOUTS]; out->rollout_plies = h[ST_PLIES]; out->tree_depth_sum = h[ST_DEPTH];
    out->children_scanned = h[ST_SCANNED]; out->terminal_hits = h[ST_TERMINAL]; out->faults = h[ST_FAULTS];
    out->spec_issued = h[ST_SPEC_ISSUED]; out->spec_hits = h[ST_SPEC_HITS];
    for (int i = 0; i < 16; ++i) out->reason_hist[i] = h[ST_REASON0 + i];
    return TAFL_OK;
}

int tafl_mcts_root_children(tafl_batch* b, tafl_root_child* out, uint32_t max_children, uint32_t* out_n) {
    if (!b || !out || !out_n || max_children == 0 || search_done(b)) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->children, sizeof(tafl_root_child) * (size_t)n * max_children); NEED(b->children_n, sizeof(uint32_t) * n);
    HIPCHK(hipMemsetAsync(b->children.p, 0, sizeof(tafl_root_child) * (size_t)n * max_children, c->stream));
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_root_children<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->mem,
                                       (tafl_root_child*)b->children.p, max_children, (uint32_t*)b->children_n.p));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, b->children.p, sizeof(tafl_root_child) * (size_t)n * max_children, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_n, b->children_n.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t g = 0; g < n; ++g) if (out_n[g] > max_children) return fail(TAFL_ERR_CAPACITY, "max_children too small for some game");
    return TAFL_OK;
}

int tafl_mcts_root_visits(tafl_batch* b, uint32_t* out) {
    if (!b || !out || search_done(b)) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, as = tafl_action_size(c);
    HIPCHK(hipSetDevice(c->device));
    NEED(b->visits, sizeof(uint32_t) * (size_t)n * as);
    HIPCHK(hipMemsetAsync(b->visits.p, 0, sizeof(uint32_t) * (size_t)n * as, c->stream));
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_root_visits<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->mem, (uint32_t*)b->visits.p, as));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, b->visits.p, sizeof(uint32_t) * (size_t)n * as, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

// probs of src/mcts.py:40-53 computed on the host from the device's root visit counts (float64, same op order)
int tafl_mcts_policy(tafl_batch* b, double temp, double* out) {
    if (!b || !out || search_done(b) || temp < 0) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, as = tafl_action_size(c);
    std::vector<uint32_t> counts((size_t)n * as);
    int rc = tafl_mcts_root_visits(b, counts.data());
    if (rc) return rc;
    for (uint32_t g = 0; g < n; ++g) {
        const uint32_t* cg = counts.data() + (size_t)g * as; double* og = out + (size_t)g * as;
        if (temp == 0) {
            uint32_t best = 0, arg = 0;
            for (uint32_t a = 0; a < as; ++a) if (cg[a] > best) { best = cg[a]; arg = a; }
            for (uint32_t a = 0; a < as; ++a) og[a] = 0.0;
            og[arg] = 1.0;
        } else {
            const double ex = 1.0 / temp; double sum = 0.0;
            for (uint32_t a = 0; a < as; ++a) { og[a] = pow((double)cg[a], ex); sum += og[a]; }
            for (uint32_t a = 0; a < as; ++a) og[a] = og[a] / sum;
        }
    }
    return TAFL_OK;
}

int tafl_mcts_best_play(tafl_batch* b, tafl_play* out_plays, uint32_t* out_visits) {
    if (!b || !out_plays || search_done(b)) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->best_plays, sizeof(tafl_play) * n); NEED(b->best_visits, sizeof(uint32_t) * n);
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_best_play<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->mem,
                                       (tafl_play*)b->best_plays.p, (uint32_t*)b->best_visits.p));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_plays, b->best_plays.p, sizeof(tafl_play) * n, hipMemcpyDeviceToHost, c->stream));
    if (out_visits) HIPCHK(hipMemcpyAsync(out_visits, b->best_visits.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_mcts_play_best(tafl_batch* b, tafl_play* out_plays, tafl_effects* out_effects) {
    if (!b || search_done(b)) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    if (out_plays) NEED(b->best_plays, sizeof(tafl_play) * n);
    if (out_effects) NEED(b->effects, sizeof(tafl_effects) * n);
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_play_best<NLS, WS, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, batch_consts<NLS>(c), b->mem, b->soa,
                                       out_plays ? (tafl_play*)b->best_plays.p : nullptr, out_effects ? (tafl_effects*)b->effects.p : nullptr));
    HIPCHK(hipGetLastError());
    if (out_plays) HIPCHK(hipMemcpyAsync(out_plays, b->best_plays.p, sizeof(tafl_play) * n, hipMemcpyDeviceToHost, c->stream));
    if (out_effects) HIPCHK(hipMemcpyAsync(out_effects, b->effects.p, sizeof(tafl_effects) * n, hipMemcpyDeviceToHost, c->stream));
    if (out_plays || out_effects) HIPCHK(hipStreamSynchronize(c->stream));
    b->ran = false;                                   // the tree belongs to the previous roots
    return TAFL_OK;
}

// ---- training-tensor writers: outputs may be HOST or DEVICE pointers --------------------
int tafl_encode_boards(tafl_batch* b, uint8_t* out, int out_is_device) {
    if (!b || !out) return fail(TAFL_ERR_INVALID_ARG, "null argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n; const size_t total = (size_t)n * c->n * c->n;
    HIPCHK(hipSetDevice(c->device));
    uint8_t* dst = out;
    if (!out_is_device) { NEED(b->enc, total); dst = (uint8_t*)b->enc.p; }
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_encode_boards<NL, W>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, CC, b->soa, n, dst));
    HIPCHK(hipGetLastError());
    if (!out_is_device) HIPCHK(hipMemcpyAsync(out, dst, total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_mcts_policy_device_ex(tafl_batch* b, double temp, uint64_t tie_seed, uint64_t game_id_base, double* out, int out_is_device) {
    if (!b || !out || search_done(b)) return fail(TAFL_ERR_INVALID_ARG, "bad argument / no MCTS run");
    if (!(temp >= 0.0)) return fail(TAFL_ERR_INVALID_ARG, "temp must be >= 0");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, as = tafl_action_size(c); const size_t bytes = sizeof(double) * (size_t)n * as;
    HIPCHK(hipSetDevice(c->device));
    double* dst = out;
    if (!out_is_device) { NEED(b->policy, bytes); dst = (double*)b->policy.p; }
    HIPCHK(hipMemsetAsync(dst, 0, bytes, c->stream));
    const double inv = temp == 0.0 ? 1.0 : 1.0 / temp;                  // 1. / temp of mcts.py:50
    DISPATCH_ARENA(c, hipLaunchKernelGGL((k_mcts_policy<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->mem, dst, as, temp == 0.0 ? 1 : 0, inv,
                                       tie_seed, game_id_base));
    HIPCHK(hipGetLastError());
    if (!out_is_device) HIPCHK(hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}
int tafl_mcts_policy_device(tafl_batch* b, double temp, double* out, int out_is_device) { return tafl_mcts_policy_device_ex(b, temp, 0, 0, out, out_is_device); }

}  // extern "C"

// ---- guided MCTS: external evaluator (tafl_guided.hpp) -------------------------------------------------------------------
enum { GS_SIMS = 0, GS_PREDICTS, GS_TERMINAL, GS_FAULTS, GS_DEPTH, GS_WAITING, GS_COUNT };

template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_gmcts_init(Consts<NL> C, const Quad* soa, GuidedMem M) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    DState<NL> st; StateIO<NL>::load_soa(soa, M.G, g, st);
    Guided<NL, W>::init_game(M, g, st);
}
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_gmcts_step(Consts<NL> C, GuidedMem M, const float* priors, const float* values, uint32_t A, double c_puct,
                                                           uint32_t n_sims, unsigned long long* stats) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    GuidedStats gs; gs.sims = gs.predicts = gs.terminal_hits = gs.faults = gs.depth = 0;
    Guided<NL, W>::step(M, g, priors ? priors + (size_t)g * A : nullptr, values ? values[g] : 0.f, A, c_puct, n_sims, C, gs);
    if (gs.sims) atomicAdd(&stats[GS_SIMS], (unsigned long long)gs.sims);
    if (gs.predicts) atomicAdd(&stats[GS_PREDICTS], (unsigned long long)gs.predicts);
    if (gs.terminal_hits) atomicAdd(&stats[GS_TERMINAL], (unsigned long long)gs.terminal_hits);
    if (gs.faults) atomicAdd(&stats[GS_FAULTS], (unsigned long long)gs.faults);
    if (gs.depth) atomicAdd(&stats[GS_DEPTH], (unsigned long long)gs.depth);
    if (M.kind[g] == 1) atomicAdd(&stats[GS_WAITING], 1ull);
}
// network input of the waiting leaves: board_to_matrix planes (game/main.rs:55-83), side to move, waiting flag; one thread per tile
template <int NL, int W>
__global__ __launch_bounds__(256) void k_gmcts_leaves(Consts<NL> C, GuidedMem M, uint8_t* boards, uint8_t* sides, uint8_t* waiting) {
    const uint32_t nn = C.n * C.n;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M.G * nn) return;
    const uint32_t g = (uint32_t)(i / nn), t = (uint32_t)(i % nn), r = t / C.n, c = t % C.n, bit = r * (uint32_t)W + c;
    const bool wait = M.kind[g] == 1;
    const uint32_t L = wait ? M.leaf[g] : 0u;
    const uint32_t* rec = (const uint32_t*)(M.node_state + ((size_t)L * M.G + g) * StateIO<NL>::QUADS);   // att[NL], def[NL], rep[4], meta[4]
    const uint32_t aw = rec[bit >> 5], dw = rec[NL + (bit >> 5)], flags = rec[2 * NL + 7];
    uint32_t v = 0;
    if ((r == 0 || r == C.n - 1) && (c == 0 || c == C.n - 1)) v = 20;
    if (r == C.n / 2 && c == C.n / 2) v = 30;
    const bool d = (dw >> (bit & 31)) & 1u, a = (aw >> (bit & 31)) & 1u;
    if (d) v += (r == TAFL_F_KROW(flags) && c == TAFL_F_KCOL(flags)) ? 5u : 1u; else if (a) v += 1u;
    boards[i] = (uint8_t)v;
    if (t == 0) { sides[g] = (uint8_t)((flags & TAFL_F_SIDE) ? TAFL_DEFENDER : TAFL_ATTACKER); waiting[g] = wait ? 1 : 0; }
}
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_gmcts_root_children(Consts<NL> C, GuidedMem M, tafl_root_child* out, uint32_t max_children, uint32_t* out_n) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    out_n[g] = Guided<NL, W>::root_children(M, g, out + (size_t)g * max_children, max_children);
}
// dense root visit counts and the probs of src/mcts.py:43-53 (any temperature; temp == 0 as in k_mcts_policy)
template <int NL, int W>
__global__ __launch_bounds__(TAFL_BLOCK) void k_gmcts_root_dense(Consts<NL> C, GuidedMem M, uint32_t* visits, double* probs, uint32_t A, int one_hot, double inv_temp,
                                                                 uint64_t tie_seed, uint64_t base) {
    const uint32_t g = blockIdx.x * TAFL_BLOCK + threadIdx.x;
    if (g >= M.G) return;
    const GNode h = M.hdr[g];
    if (!h.expanded) return;
    const GEdge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
    // the legal edges are stored in ascending action order, zeros included: a zero count adds 0.0 to the sum (0 ** x == 0 for x > 0)
    double sum = 0.0; uint32_t best = 0, ties = 0; bool any_n = false;
    for (uint32_t j = 0; j < h.n_legal; ++j) {
        const GEdge e = eb[j];
        if (!one_hot && e.n != 0) sum += temp_weight(e.n, inv_temp);
        if (e.n > best) { best = e.n; ties = 1; } else if (e.n == best && best > 0) ++ties;
        any_n |= e.n != 0;
    }
    uint32_t pick = 0;
    if (one_hot && tie_seed != 0 && ties > 1) pick = tie_pick(tie_seed, base + g, ties);
    uint32_t seen = 0;
    for (uint32_t j = 0; j < h.n_legal; ++j) {
        const GEdge e = eb[j];
        if (visits) visits[(size_t)g * A + e.action] = e.n;
        if (probs && any_n) {
            double p;
            if (one_hot) { const bool is_max = e.n == best; p = (is_max && seen == pick) ? 1.0 : 0.0; seen += is_max ? 1u : 0u; }
            else p = e.n != 0 ? temp_weight(e.n, inv_temp) / sum : 0.0;
            probs[(size_t)g * A + e.action] = p;
        }
    }
}

extern "C" {

int tafl_gmcts_begin(tafl_batch* b, uint32_t max_sims, uint32_t edges_per_node) {
    if (!b || max_sims == 0 || edges_per_node == 0) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_begin: bad argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n; const size_t q = (size_t)quads_of(c);
    HIPCHK(hipSetDevice(c->device));
    const uint32_t node_cap = max_sims + 1;
    const unsigned long long ecap = (unsigned long long)node_cap * edges_per_node;
    if (ecap > 0xFFFFFFFFull) return fail(TAFL_ERR_CAPACITY, "tafl_gmcts_begin: edge arena too large");
    NEED(b->g_node_state, sizeof(Quad) * q * node_cap * n); NEED(b->g_hdr, sizeof(GNode) * (size_t)node_cap * n);
    NEED(b->g_pedge, sizeof(uint32_t) * (size_t)node_cap * n); NEED(b->g_edges, sizeof(GEdge) * (size_t)ecap * n);
    NEED(b->g_node_top, 4 * (size_t)n); NEED(b->g_edge_top, 4 * (size_t)n); NEED(b->g_leaf, 4 * (size_t)n); NEED(b->g_kind, n); NEED(b->g_fault, n);
    NEED(b->g_sims, 4 * (size_t)n); NEED(b->g_stats, sizeof(unsigned long long) * GS_COUNT);
    GuidedMem& M = b->gmem;
    M.node_state = (Quad*)b->g_node_state.p; M.hdr = (GNode*)b->g_hdr.p; M.pedge = (uint32_t*)b->g_pedge.p; M.edges = (GEdge*)b->g_edges.p;
    M.node_top = (uint32_t*)b->g_node_top.p; M.edge_top = (uint32_t*)b->g_edge_top.p; M.leaf = (uint32_t*)b->g_leaf.p; M.kind = (uint8_t*)b->g_kind.p;
    M.fault = (uint8_t*)b->g_fault.p; M.sims_done = (uint32_t*)b->g_sims.p; M.G = n; M.node_cap = node_cap; M.edge_cap = (uint32_t)ecap;
    HIPCHK(hipMemsetAsync(b->g_stats.p, 0, sizeof(unsigned long long) * GS_COUNT, c->stream));
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_gmcts_init<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->soa, M));
    HIPCHK(hipGetLastError());
    b->g_has = true; b->g_max_sims = max_sims;
    return TAFL_OK;
}

int tafl_gmcts_step(tafl_batch* b, const float* priors, const float* values, int in_is_device, double c_puct, uint32_t n_sims, uint32_t* out_waiting) {
    if (!b || !b->g_has) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_step: tafl_gmcts_begin first");
    if (n_sims > b->g_max_sims) return fail(TAFL_ERR_CAPACITY, "tafl_gmcts_step: n_sims exceeds the reserved simulations");
    if ((priors == nullptr) != (values == nullptr)) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_step: priors and values go together");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, A = tafl_action_size(c);
    HIPCHK(hipSetDevice(c->device));
    const float* dp = priors; const float* dv = values;
    if (priors && !in_is_device) {
        NEED(b->g_priors, sizeof(float) * (size_t)n * A); NEED(b->g_values, sizeof(float) * (size_t)n);
        HIPCHK(hipMemcpyAsync(b->g_priors.p, priors, sizeof(float) * (size_t)n * A, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(b->g_values.p, values, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        dp = (const float*)b->g_priors.p; dv = (const float*)b->g_values.p;
    }
    unsigned long long* st = (unsigned long long*)b->g_stats.p;
    HIPCHK(hipMemsetAsync(st + GS_WAITING, 0, sizeof(unsigned long long), c->stream));
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_gmcts_step<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->gmem, dp, dv, A, c_puct, n_sims, st));
    HIPCHK(hipGetLastError());
    if (out_waiting) {
        unsigned long long w = 0;
        HIPCHK(hipMemcpyAsync(&w, st + GS_WAITING, sizeof w, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *out_waiting = (uint32_t)w;
    }
    return TAFL_OK;
}

int tafl_gmcts_leaves(tafl_batch* b, uint8_t* boards, uint8_t* sides, uint8_t* waiting, int out_is_device) {
    if (!b || !b->g_has || !boards || !sides || !waiting) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_leaves: bad argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n; const size_t total = (size_t)n * c->n * c->n;
    HIPCHK(hipSetDevice(c->device));
    uint8_t *db = boards, *ds = sides, *dw = waiting;
    if (!out_is_device) { NEED(b->g_boards, total); NEED(b->g_sides, n); NEED(b->g_wait, n); db = (uint8_t*)b->g_boards.p; ds = (uint8_t*)b->g_sides.p; dw = (uint8_t*)b->g_wait.p; }
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_gmcts_leaves<NL, W>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, CC, b->gmem, db, ds, dw));
    HIPCHK(hipGetLastError());
    if (!out_is_device) {
        HIPCHK(hipMemcpyAsync(boards, db, total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(sides, ds, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(waiting, dw, n, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}

int tafl_gmcts_root_children(tafl_batch* b, tafl_root_child* out, uint32_t max_children, uint32_t* out_n) {
    if (!b || !b->g_has || !out || !out_n || max_children == 0) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_root_children: bad argument");
    tafl_ctx* c = b->ctx; const uint32_t n = b->n;
    HIPCHK(hipSetDevice(c->device));
    NEED(b->children, sizeof(tafl_root_child) * (size_t)n * max_children); NEED(b->children_n, sizeof(uint32_t) * n);
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_gmcts_root_children<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->gmem, (tafl_root_child*)b->children.p,
                                       max_children, (uint32_t*)b->children_n.p));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, b->children.p, sizeof(tafl_root_child) * (size_t)n * max_children, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_n, b->children_n.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t g = 0; g < n; ++g) if (out_n[g] > max_children) return fail(TAFL_ERR_CAPACITY, "tafl_gmcts_root_children: max_children too small");
    return TAFL_OK;
}

static int gmcts_dense(tafl_batch* b, uint32_t* visits, double* probs, double temp, int out_is_device, uint64_t tie_seed = 0, uint64_t game_id_base = 0) {
    tafl_ctx* c = b->ctx; const uint32_t n = b->n, A = tafl_action_size(c);
    HIPCHK(hipSetDevice(c->device));
    const size_t vb = sizeof(uint32_t) * (size_t)n * A, pb = sizeof(double) * (size_t)n * A;
    uint32_t* dv = visits; double* dp = probs;
    if (!out_is_device) { if (visits) { NEED(b->visits, vb); dv = (uint32_t*)b->visits.p; } if (probs) { NEED(b->policy, pb); dp = (double*)b->policy.p; } }
    if (dv) HIPCHK(hipMemsetAsync(dv, 0, vb, c->stream));
    if (dp) HIPCHK(hipMemsetAsync(dp, 0, pb, c->stream));
    DISPATCH_NLW(c, hipLaunchKernelGGL((k_gmcts_root_dense<NL, W>), dim3(grid_of(n)), dim3(TAFL_BLOCK), 0, c->stream, CC, b->gmem, dv, dp, A, temp == 0.0 ? 1 : 0, temp == 0.0 ? 1.0 : 1.0 / temp, tie_seed, game_id_base));
    HIPCHK(hipGetLastError());
    if (!out_is_device) {
        if (visits) HIPCHK(hipMemcpyAsync(visits, dv, vb, hipMemcpyDeviceToHost, c->stream));
        if (probs) HIPCHK(hipMemcpyAsync(probs, dp, pb, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return TAFL_OK;
}
int tafl_gmcts_root_visits(tafl_batch* b, uint32_t* out, int out_is_device) {
    if (!b || !b->g_has || !out) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_root_visits: bad argument");
    return gmcts_dense(b, out, nullptr, 1.0, out_is_device);
}
int tafl_gmcts_policy_ex(tafl_batch* b, double temp, uint64_t tie_seed, uint64_t game_id_base, double* out, int out_is_device) {
    if (!b || !b->g_has || !out) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_policy: bad argument");
    if (!(temp >= 0.0)) return fail(TAFL_ERR_INVALID_ARG, "temp must be >= 0");
    return gmcts_dense(b, nullptr, out, temp, out_is_device, tie_seed, game_id_base);
}
int tafl_gmcts_policy(tafl_batch* b, double temp, double* out, int out_is_device) { return tafl_gmcts_policy_ex(b, temp, 0, 0, out, out_is_device); }
int tafl_gmcts_get_stats(tafl_batch* b, tafl_gmcts_stats* out) {
    if (!b || !b->g_has || !out) return fail(TAFL_ERR_INVALID_ARG, "tafl_gmcts_get_stats: bad argument");
    tafl_ctx* c = b->ctx;
    HIPCHK(hipSetDevice(c->device));
    unsigned long long h[GS_COUNT];
    HIPCHK(hipMemcpyAsync(h, b->g_stats.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof *out);
    out->sims = h[GS_SIMS]; out->predicts = h[GS_PREDICTS]; out->terminal_hits = h[GS_TERMINAL]; out->faults = h[GS_FAULTS];
    out->select_depth_sum = h[GS_DEPTH]; out->waiting = h[GS_WAITING];
    return TAFL_OK;
}

// ---- timing ---------------------------------------------------------------------------------------------
int tafl_timing_enable(tafl_ctx* c, int enable) { if (!c) return fail(TAFL_ERR_INVALID_ARG, "null ctx"); c->timing = enable != 0; return TAFL_OK; }
int tafl_timing_reset(tafl_ctx* c) {
    if (!c) return fail(TAFL_ERR_INVALID_ARG, "null ctx");
    (void)hipSetDevice(c->device);
    drain_spans(c);
    for (int i = 0; i < KC_COUNT; ++i) { c->acc_ms[i] = 0; c->acc_n[i] = 0; c->ivals[i].clear(); }
    if (!c->has_ref) { if (hipEventCreate(&c->t_ref) != hipSuccess) return fail(TAFL_ERR_HIP, "hipEventCreate failed"); c->has_ref = true; }
    HIPCHK(hipEventRecord(c->t_ref, c->stream));
    HIPCHK(hipEventSynchronize(c->t_ref));
    return TAFL_OK;
}
// wall-clock time during which AT LEAST ONE launch of the class was running (union of the spans' intervals), and the sum of the spans:
// sum / union = how many launches of the class were in flight on average (partitions on their own streams overlap)
int tafl_timing_get_union(tafl_ctx* c, int cls, double* union_ms, double* sum_ms) {
    if (!c || cls < 0 || cls >= KC_COUNT) return fail(TAFL_ERR_INVALID_ARG, "bad kernel class");
    (void)hipSetDevice(c->device);
    drain_spans(c);
    std::vector<std::pair<float, float>> v = c->ivals[cls];
    std::sort(v.begin(), v.end());
    double u = 0.0, s = 0.0; float lo = 0.f, hi = -1.f;
    for (const auto& iv : v) {
        s += (double)iv.second - (double)iv.first;
        if (hi < lo) { lo = iv.first; hi = iv.second; }
        else if (iv.first <= hi) { if (iv.second > hi) hi = iv.second; }
        else { u += (double)hi - (double)lo; lo = iv.first; hi = iv.second; }
    }
    if (hi >= lo) u += (double)hi - (double)lo;
    if (union_ms) *union_ms = u;
    if (sum_ms) *sum_ms = s;
    return TAFL_OK;
}
int tafl_timing_get(tafl_ctx* c, int cls, double* total_ms, uint64_t* launches) {
    if (!c || cls < 0 || cls >= KC_COUNT) return fail(TAFL_ERR_INVALID_ARG, "bad kernel class");
    (void)hipSetDevice(c->device);
    drain_spans(c);
    if (total_ms) *total_ms = c->acc_ms[cls];
    if (launches) *launches = c->acc_n[cls];
    return TAFL_OK;
}

}  // extern "C"
