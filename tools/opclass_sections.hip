// opclass_sections.hip — the sections of one playout ply (tafl_fast.hpp: Fast::ply_of<MOVER>) as separate kernels, so that their
// machine code can be counted by opcode class (tools/isa_opclass.py): pick (RNG + move pick), apply_pre (move, captures, king capture,
// shieldwall filter, repetition tracker), T-layout upkeep, gen (opponent's plays), outcome (filters + outcome_early + apply_finish),
// each for an attacker's ply (MOVER 0) and a defender's (MOVER 1).
// Inputs come from memory so that nothing folds away; the load / store instructions at both ends are not part of a ply (they are
// MEM in the histogram).  Never run: compile with  hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S
#include <hip/hip_runtime.h>
#include "tafl_fast.hpp"
using namespace tafl;
#ifndef SEC_NL
#define SEC_NL 4
#define SEC_W 11
#define SEC_PRESET 1
#endif
constexpr int NL = SEC_NL, W = SEC_W;
using E = Engine<NL, W>; using FX = Fast<NL, W>; using S = DState<NL>; using B = Bits<NL>;
struct Blob { S st; B attT, defT; FX::Gen g; Move m; uint32_t idx, rk; typename E::ApplyCtx ax; typename E::Outcome o; int sw; };
#define CC preset_consts<NL, W, SEC_PRESET>()
template <uint32_t MOVER> __device__ __forceinline__ void t_upkeep(Blob& x) {
    const Move m = x.m; const auto& ax = x.ax; B attT = x.attT, defT = x.defT;
    constexpr int BK = 2 * W;
    const uint32_t tT = FX::n_to_t(m.to);
    const B mvT = bit_at<NL>(FX::n_to_t(m.from)) | bit_at<NL>(tT);
    if constexpr (MOVER != 0) defT = defT ^ mvT; else attT = attT ^ mvT;
    const uint32_t cu = ax.cust;
    const uint64_t cf = ((uint64_t)(cu & 1u) << (BK + 1)) | ((uint64_t)((cu >> 1) & 1u) << (BK - 1))
                      | ((uint64_t)((cu >> 2) & 1u) << (BK + W)) | ((uint64_t)((cu >> 3) & 1u) << (BK - W));
    const B cT = deposit64<BK, NL>(cf, tT);
    if constexpr (MOVER != 0) attT = andn(attT, cT); else defT = andn(defT, cT);
    if (ax.ncap != (uint32_t)__builtin_popcount(cu)) {
        B c = ax.caps;
        while (any(c)) { const uint32_t i = lsb(c); c = andn(c, bit_at<NL>(i)); const B cb = bit_at<NL>(FX::n_to_t(i)); attT = andn(attT, cb); defT = andn(defT, cb); }
    }
    x.attT = attT; x.defT = defT;
}
template <uint32_t MOVER> __device__ __forceinline__ void outcome(Blob& x) {
    const auto C = CC;
    bool skip_encl = false, skip_fort = false;
    if constexpr (MOVER == 0) skip_encl = C.rules.enclosure_win == TAFL_ENCL_WITHOUT_EDGE_ACCESS && (x.g.edge_hit || any(x.st.def & C.edge));
    else if (C.rules.exit_fort) skip_fort = !FX::fort_candidate(x.st, x.attT, x.defT, C);
    x.ax.mover = MOVER;
    const typename E::Outcome o = E::outcome_early(x.st, x.ax, C, skip_encl, skip_fort);
    E::apply_finish(x.st, x.ax, o, o.over ? 1u : x.g.total, C);
}
extern "C" {
__global__ void sec_pick(Blob* b) { Blob& x = b[threadIdx.x]; const auto C = CC; const uint32_t idx = E::mulhi(E::fmix32(x.rk), x.g.total); x.rk += 0x85EBCA77u; x.m = FX::pick(x.st, x.attT, x.defT, x.g, idx, C); }
__global__ void sec_apply_pre_att(Blob* b) { Blob& x = b[threadIdx.x]; const auto C = CC; E::apply_pre(x.st, x.m, C, x.ax, 0u); }
__global__ void sec_apply_pre_def(Blob* b) { Blob& x = b[threadIdx.x]; const auto C = CC; E::apply_pre(x.st, x.m, C, x.ax, 1u); }
__global__ void sec_gen_att(Blob* b) { Blob& x = b[threadIdx.x]; const auto C = CC; const auto fc = make_fast_consts<NL>(C); FX::gen(x.st, x.attT, x.defT, 0u, C, fc, x.g); }
__global__ void sec_gen_def(Blob* b) { Blob& x = b[threadIdx.x]; const auto C = CC; const auto fc = make_fast_consts<NL>(C); FX::gen(x.st, x.attT, x.defT, 1u, C, fc, x.g); }
__global__ void sec_outcome_att(Blob* b) { outcome<0>(b[threadIdx.x]); }
__global__ void sec_outcome_def(Blob* b) { outcome<1>(b[threadIdx.x]); }
__global__ void sec_t_upkeep_att(Blob* b) { t_upkeep<0>(b[threadIdx.x]); }
__global__ void sec_t_upkeep_def(Blob* b) { t_upkeep<1>(b[threadIdx.x]); }
}
