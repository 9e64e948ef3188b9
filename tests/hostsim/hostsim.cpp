// hostsim.cpp — TEST HARNESS ONLY.  Compiles the product's per-game device functions
// (alphazeroforhnefatafl_amd/csrc/tafl_ops.hpp) for the HOST with g++ and drives them in plain
// CPU loops, so that the bit-parallel engine can be differential-tested against the literal oracle
// in the build container (which has no GPU).  The product library never links this file and has no
// CPU path; the GPU parity tests (tests/test_gpu_parity.py) check the real kernels.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../alphazeroforhnefatafl_amd/csrc/tafl_ops.hpp"
#include "../../alphazeroforhnefatafl_amd/csrc/tafl_guided.hpp"

using namespace tafl;

static uint32_t g_spec_k = 4;          // playout slots per game that exist in the MCTS pipeline (1 = no speculation)
static uint32_t g_scen = 0;            // scenario passes of the prediction (0: the product's policy, Ops::mcts_scenarios)
static uint32_t g_spec_target = 0;     // slots per game and round the search is planned for (0: no plan, issue what is allowed)
static uint32_t g_capacity = 0;        // playouts a round may run (0: all that are requested), like the device capacity of k_mcts_rollout
static uint32_t g_log_cap = 16;        // undo records of each kind a prediction pass may write (the device's LDS holds 16 per lane; an overflow ends the pass early)
static std::vector<uint32_t> g_round_work;   // playouts executed per round of the last hs_mcts call (cost-model experiments)
static bool g_force_generic = false;   // differential tests: generic Engine::rollout vs the fast playout engine

// DENSE13: the position arrives in the reference's U256 / 15-column layout and is searched in the dense 13-column layout (6 limbs),
// as the library does for the 13x13 preset (restride, tafl_core.hpp)
template <int NL, int W, bool DENSE13 = false>
struct Host {
    using O = Ops<NL, W>;
    using S = DState<NL>;
    using K = Consts<NL>;
    static int consts(const tafl_rules* r, uint8_t n, K& C) { return make_consts<NL, W>(*r, n, C); }
    static void load(const tafl_state& a, S& s) {
        if constexpr (DENSE13) { DState<8> t; state_from_abi<8>(a, t); restride<8, 15, NL, W>(t, 13, s); }
        else state_from_abi<NL>(a, s);
    }

    static int movegen(const tafl_rules* r, uint8_t n, const tafl_state* st, uint32_t cnt, uint32_t* counts, uint32_t* masks, uint32_t mw) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) {
            S s; state_from_abi<NL>(st[g], s);
            uint32_t* m = masks ? masks + (size_t)g * mw : nullptr;
            if (m) memset(m, 0, sizeof(uint32_t) * mw);
            const uint32_t c = O::movegen(s, C, m);
            if (counts) counts[g] = c;
        }
        return 0;
    }
    static int validate(const tafl_rules* r, uint8_t n, const tafl_state* st, uint32_t cnt, const tafl_play* plays, uint8_t* codes) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) { S s; state_from_abi<NL>(st[g], s); codes[g] = (uint8_t)O::validate(s, plays[g], C); }
        return 0;
    }
    static int step(const tafl_rules* r, uint8_t n, tafl_state* st, uint32_t cnt, const tafl_play* plays, tafl_effects* eff) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) { S s; state_from_abi<NL>(st[g], s); O::step(s, plays[g], C, eff ? &eff[g] : nullptr); state_to_abi<NL>(s, n, st[g]); }
        return 0;
    }
    static int step_kth(const tafl_rules* r, uint8_t n, tafl_state* st, uint32_t cnt, const uint32_t* ranks, tafl_play* out_plays, tafl_effects* eff) {
        K C; if (consts(r, n, C)) return -1;
        const uint32_t mw = ((uint32_t)n * n * 2u * (n - 1u) + 31u) / 32u; std::vector<uint32_t> mask(mw);
        for (uint32_t g = 0; g < cnt; ++g) { S s; state_from_abi<NL>(st[g], s); std::fill(mask.begin(), mask.end(), 0u); O::step_kth(s, ranks[g], C, out_plays ? &out_plays[g] : nullptr, eff ? &eff[g] : nullptr, mask.data(), mw); state_to_abi<NL>(s, n, st[g]); }
        return 0;
    }
    static int side_can_play(const tafl_rules* r, uint8_t n, const tafl_state* st, uint32_t cnt, uint8_t side, uint8_t* out) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) { S s; state_from_abi<NL>(st[g], s); out[g] = O::side_can_play(s, side ? 1u : 0u, C) ? 1 : 0; }
        return 0;
    }
    static int rollout(const tafl_rules* r, uint8_t n, const tafl_state* st, uint32_t cnt, uint64_t seed, uint32_t sim, uint32_t max_plies, uint64_t base, tafl_rollout_result* out) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) { S s; load(st[g], s); O::rollout(s, seed, base + g, sim, max_plies, C, out[g], g_force_generic); }
        return 0;
    }
    static int random_advance(const tafl_rules* r, uint8_t n, tafl_state* st, uint32_t cnt, uint64_t seed, const uint32_t* plies, uint64_t base) {
        K C; if (consts(r, n, C)) return -1;
        for (uint32_t g = 0; g < cnt; ++g) { S s; state_from_abi<NL>(st[g], s); O::random_advance(s, seed, base + g, plies[g], C, g_force_generic); state_to_abi<NL>(s, n, st[g]); }
        return 0;
    }
    // n_moves != 0: a self-play run as tafl_selfplay_run drives it (k_mcts_tree_selfplay): `st_io` holds the batch, is advanced in place, and
    // `plays_out` [n_moves * G] receives the plays
    static int mcts(const tafl_rules* r, uint8_t n, const tafl_state* st, uint32_t G, const tafl_mcts_params* p, uint64_t base,
                    tafl_root_child* out_children, uint32_t max_children, uint32_t* out_n, tafl_mcts_stats* stats,
                    uint32_t n_moves = 0, tafl_state* st_io = nullptr, tafl_play* plays_out = nullptr) {
        K C; if (consts(r, n, C)) return -1;
        using IO = StateIO<NL>;
        // same host loop as tafl_mcts_run's two-kernel pipeline: g_spec_k slots exist per game, the search is planned for
        // ceil(n_sims / g_spec_target) rounds (g_spec_target = 0: every game issues as many slots as it may, every round)
        MctsMem M; M.G = G; M.node_cap = p->n_sims + 1; M.edge_cap = 4 * (p->n_sims + 1); M.spec_k = g_spec_k; M.flags = p->flags & TAFL_MCTS_FLAG_FPU_INF;
        std::vector<Quad> ns((size_t)M.node_cap * G * IO::QUADS), sst((size_t)M.spec_k * G * IO::QUADS);
        std::vector<NodeHdr> hdr((size_t)M.node_cap * G);
        std::vector<Edge> edges((size_t)M.edge_cap * G);
        std::vector<uint32_t> ntop(G), etop(G), leaf(G), simn(G), spend(G), splies((size_t)M.spec_k * G), smeta((size_t)M.spec_k * G), sref((size_t)M.spec_k * G);
        std::vector<uint8_t> kind(G), fault(G), skind((size_t)M.spec_k * G), sreason((size_t)M.spec_k * G), scls((size_t)M.spec_k * G);
        std::vector<int8_t> sval((size_t)M.spec_k * G);
        // the undo log of the prediction pass: one lane's scratch (the device keeps 64 of them side by side in LDS)
        std::vector<uint32_t> logw((size_t)g_log_cap * (kUndoEWords + kUndoHWords) + 1);
        LogMem lm; lm.base = logw.data(); lm.stride = 1; lm.lane = 0; lm.cap = g_spec_k > 1 ? g_log_cap : 0;
        M.node_state = ns.data(); M.hdr = hdr.data(); M.edges = edges.data(); M.node_top = ntop.data(); M.edge_top = etop.data();
        M.leaf = leaf.data(); M.kind = kind.data(); M.fault = fault.data();
        M.sim_next = simn.data(); M.spec_state = sst.data(); M.spec_value = sval.data(); M.spec_kind = skind.data(); M.spec_reason = sreason.data(); M.spec_meta = smeta.data();
        M.spec_plies = splies.data(); M.spec_ref = sref.data(); M.spec_cls = scls.data(); M.spec_pend = spend.data();
        std::vector<uint32_t> simbase(G), sbias(G); M.sim_base = simbase.data(); M.spec_bias = sbias.data();
        memset(stats, 0, sizeof *stats);
        for (uint32_t g = 0; g < G; ++g) { S s; load(st[g], s); O::mcts_init_game(M, g, s, C); }
        // self-play: the batch as the device holds it (quad-plane SoA in the reference layout), per-game counters, the plays
        constexpr int NLB = DENSE13 ? 8 : NL, WB = DENSE13 ? 15 : W;
        std::vector<Quad> soa((size_t)StateIO<NLB>::QUADS * G);
        std::vector<uint32_t> mdone(G, 0), sround(G, 0);
        SelfPlay sp; sp.moves_done = mdone.data(); sp.start_round = sround.data(); sp.plays = plays_out; sp.n_moves = n_moves;
        if (n_moves) {
            for (uint32_t g = 0; g < G; ++g) { DState<NLB> t; state_from_abi<NLB>(st[g], t); StateIO<NLB>::store_soa(soa.data(), G, g, t); }
            memset(plays_out, 0, sizeof(tafl_play) * (size_t)n_moves * G);
        }
        g_round_work.clear();
        uint32_t round_no = 0, sp_done = 0;
        const uint32_t planned_sp = g_spec_target ? (p->n_sims + g_spec_target - 1) / g_spec_target : 0;
        auto tree = [&](uint32_t rounds_left) {
            for (uint32_t g = 0; g < G; ++g) {
                LaneStats ls; memset(&ls, 0, sizeof ls);
                if (n_moves) {                              // as k_mcts_tree_selfplay: advance, then the plan of the game's own search
                    const int rr = O::template selfplay_advance<NLB, WB>(M, g, soa.data(), sp, p->n_sims, round_no, C);
                    if (rr == 2) ++sp_done;
                    const uint32_t rel = round_no - sround[g];
                    rounds_left = g_spec_target ? (rel < planned_sp ? planned_sp - rel : 1u) : 0u;
                    if (!(simn[g] < p->n_sims || kind[g] == 1)) continue;
                }
                O::mcts_tree_step(M, g, p->c_puct, p->n_sims, rounds_left, g_scen ? g_scen : O::mcts_scenarios(rounds_left, planned_sp), g_spec_k, C, ls, lm);
                stats->sims += ls.sims; stats->tree_depth_sum += ls.depth; stats->children_scanned += ls.scanned;
                stats->terminal_hits += ls.terminal_hits; stats->faults += ls.faults;
                stats->rollouts += ls.rollouts; stats->rollout_plies += ls.rollout_plies;
                for (int q = 0; q < 16; ++q) stats->reason_hist[q] += (ls.reason_hist4 >> (4 * q)) & 15u;
                stats->spec_issued += ls.spec_issued; stats->spec_hits += ls.spec_hits;
            }
        };
        const uint32_t planned = g_spec_target ? (p->n_sims + g_spec_target - 1) / g_spec_target : 0;
        for (uint32_t i = 0; i < (p->n_sims + 2) * (g_capacity ? 1 + G / g_capacity : 1) * (n_moves ? n_moves : 1u) + (n_moves ? n_moves : 0u); ++i) {
            round_no = i;
            tree(g_spec_target ? (i < planned ? planned - i : 1u) : 0u);
            // class-major like the device's per-class work lists; with a capacity, playouts beyond it wait for the next round
            uint32_t work = 0;
            for (uint32_t c = 0; c < kMctsMaxSlots; ++c)
                for (uint32_t g = 0; g < G; ++g) {
                    if (!(simn[g] < p->n_sims || kind[g] == 1)) continue;
                    uint32_t found = 0, slot = 0;
                    for (uint32_t j = 0; j < M.spec_k; ++j) if (skind[(size_t)j * G + g] == 1 && scls[(size_t)j * G + g] == c) { ++found; slot = j; }
                    if (found > 1) return -5;                 // a game's requested playouts must have distinct classes
                    if (!found) continue;
                    if (g_capacity && work >= g_capacity) continue;
                    ++work; O::mcts_slot_rollout(M, slot, g, p->seed, base + g, p->sim_offset, p->max_rollout_plies, C);
                }
            if (work == 0 && (!n_moves || sp_done >= G)) break;
            g_round_work.push_back(work);
        }
        if (n_moves) {
            if (sp_done < G) return -3;
            for (uint32_t g = 0; g < G; ++g) { DState<NLB> t; StateIO<NLB>::load_soa(soa.data(), G, g, t); state_to_abi<NLB>(t, n, st_io[g]); }
            return 0;
        }
        for (uint32_t g = 0; g < G; ++g) if (simn[g] != p->n_sims) return -3;
        // the speculation pass must leave no trace: no edge may point at a slot-only child
        for (uint32_t g = 0; g < G; ++g)
            for (uint32_t k = 0; k < ntop[g]; ++k) { const NodeHdr& h = hdr[(size_t)k * G + g]; for (uint32_t j = 0; j < h.m; ++j) if (edges[(size_t)g * M.edge_cap + h.edge_base + j].child >= ntop[g]) return -4; }
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t k = O::mcts_root_children(M, g, C, out_children + (size_t)g * max_children, max_children);
            if (out_n) out_n[g] = k;
        }
        return 0;
    }
};

// guided MCTS session on host memory: the same Guided<NL,W>::step the k_gmcts_step kernel runs, one game after the other
struct GSessionBase {
    virtual ~GSessionBase() {}
    virtual uint32_t step(const float* priors, const float* values, double c_puct, uint32_t n_sims) = 0;
    virtual void leaves(uint8_t* boards, uint8_t* sides, uint8_t* waiting) = 0;
    virtual void root_children(tafl_root_child* out, uint32_t max_children, uint32_t* out_n) = 0;
    uint64_t sims = 0, predicts = 0, terminal_hits = 0, faults = 0;
};
template <int NL, int W>
struct GSession : GSessionBase {
    using GD = Guided<NL, W>;
    using IO = StateIO<NL>;
    Consts<NL> C; GuidedMem M; uint32_t A, n;
    std::vector<Quad> ns; std::vector<GNode> hdr; std::vector<uint32_t> pedge, ntop, etop, leaf, simsd; std::vector<GEdge> edges; std::vector<uint8_t> kind, fault;
    int init(const tafl_rules* r, uint8_t side, const tafl_state* st, uint32_t G, uint32_t max_sims, uint32_t edges_per_node) {
        if (make_consts<NL, W>(*r, side, C)) return -1;
        n = side; A = (uint32_t)side * side * 2u * (side - 1u);
        M.G = G; M.node_cap = max_sims + 1; M.edge_cap = (max_sims + 1) * edges_per_node;
        ns.resize((size_t)M.node_cap * G * IO::QUADS); hdr.resize((size_t)M.node_cap * G); pedge.resize((size_t)M.node_cap * G); edges.resize((size_t)M.edge_cap * G);
        ntop.resize(G); etop.resize(G); leaf.resize(G); simsd.resize(G); kind.resize(G); fault.resize(G);
        M.node_state = ns.data(); M.hdr = hdr.data(); M.pedge = pedge.data(); M.edges = edges.data(); M.node_top = ntop.data(); M.edge_top = etop.data();
        M.leaf = leaf.data(); M.kind = kind.data(); M.fault = fault.data(); M.sims_done = simsd.data();
        for (uint32_t g = 0; g < G; ++g) { DState<NL> s; state_from_abi<NL>(st[g], s); GD::init_game(M, g, s); }
        return 0;
    }
    uint32_t step(const float* priors, const float* values, double c_puct, uint32_t n_sims) override {
        uint32_t waiting = 0;
        for (uint32_t g = 0; g < M.G; ++g) {
            GuidedStats gs; memset(&gs, 0, sizeof gs);
            GD::step(M, g, priors ? priors + (size_t)g * A : nullptr, values ? values[g] : 0.f, A, c_puct, n_sims, C, gs);
            sims += gs.sims; predicts += gs.predicts; terminal_hits += gs.terminal_hits; faults += gs.faults;
            waiting += M.kind[g] == 1;
        }
        return waiting;
    }
    void leaves(uint8_t* boards, uint8_t* sides, uint8_t* waiting) override {
        for (uint32_t g = 0; g < M.G; ++g) {
            const bool w = M.kind[g] == 1; const uint32_t L = w ? M.leaf[g] : 0u;
            DState<NL> s; IO::load_rec(M.node_state + ((size_t)L * M.G + g) * IO::QUADS, s);
            for (uint32_t r = 0; r < n; ++r) for (uint32_t c = 0; c < n; ++c) {
                const uint32_t bit = r * (uint32_t)W + c; uint32_t v = 0;
                if ((r == 0 || r == n - 1u) && (c == 0 || c == n - 1u)) v = 20;
                if (r == n / 2u && c == n / 2u) v = 30;
                if (test(s.def, bit)) v += (r == TAFL_F_KROW(s.flags) && c == TAFL_F_KCOL(s.flags)) ? 5u : 1u; else if (test(s.att, bit)) v += 1u;
                boards[((size_t)g * n + r) * n + c] = (uint8_t)v;
            }
            sides[g] = (uint8_t)((s.flags & TAFL_F_SIDE) ? TAFL_DEFENDER : TAFL_ATTACKER); waiting[g] = w ? 1 : 0;
        }
    }
    void root_children(tafl_root_child* out, uint32_t max_children, uint32_t* out_n) override {
        for (uint32_t g = 0; g < M.G; ++g) out_n[g] = GD::root_children(M, g, out + (size_t)g * max_children, max_children);
    }
};

#define DISPATCH(call)                                              \
    switch (word_bits) {                                            \
        case 64:  return Host<2, 7>::call;                          \
        case 128: return Host<4, 11>::call;                         \
        case 256: return Host<8, 15>::call;                         \
        default:  return -2;                                        \
    }
// rollouts and searches of 13x13 positions in the dense layout when hs_set_dense13(1) (the other entry points have no dense form)
static bool g_dense13 = false;
#define DISPATCH_DENSE(call)                                        \
    if (g_dense13 && word_bits == 256 && n == 13) return Host<6, 13, true>::call; \
    DISPATCH(call)

extern "C" {
void* hs_gmcts_new(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t G, uint32_t max_sims, uint32_t edges_per_node) {
    GSessionBase* s = nullptr; int rc = -2;
    if (word_bits == 64) { auto* x = new GSession<2, 7>(); rc = x->init(r, n, st, G, max_sims, edges_per_node); s = x; }
    else if (word_bits == 128) { auto* x = new GSession<4, 11>(); rc = x->init(r, n, st, G, max_sims, edges_per_node); s = x; }
    else if (word_bits == 256) { auto* x = new GSession<8, 15>(); rc = x->init(r, n, st, G, max_sims, edges_per_node); s = x; }
    if (rc) { delete s; return nullptr; }
    return s;
}
void hs_gmcts_free(void* h) { delete (GSessionBase*)h; }
uint32_t hs_gmcts_step(void* h, const float* priors, const float* values, double c_puct, uint32_t n_sims) { return ((GSessionBase*)h)->step(priors, values, c_puct, n_sims); }
void hs_gmcts_leaves(void* h, uint8_t* boards, uint8_t* sides, uint8_t* waiting) { ((GSessionBase*)h)->leaves(boards, sides, waiting); }
void hs_gmcts_root_children(void* h, tafl_root_child* out, uint32_t max_children, uint32_t* out_n) { ((GSessionBase*)h)->root_children(out, max_children, out_n); }
void hs_gmcts_counts(void* h, uint64_t* out4) { GSessionBase* s = (GSessionBase*)h; out4[0] = s->sims; out4[1] = s->predicts; out4[2] = s->terminal_hits; out4[3] = s->faults; }
void hs_force_generic(int on) { g_force_generic = on != 0; }
int hs_selfplay(const tafl_rules* r, uint8_t n, uint32_t word_bits, tafl_state* st, uint32_t cnt, const tafl_mcts_params* p, uint64_t base, uint32_t n_moves, tafl_play* plays, tafl_mcts_stats* stats) {
    DISPATCH_DENSE(mcts(r, n, st, cnt, p, base, nullptr, 0, nullptr, stats, n_moves, st, plays))
}
void hs_set_spec_k(uint32_t k) { g_spec_k = k < 1 ? 1 : (k > 8 ? 8 : k); }
void hs_set_spec_target(uint32_t t) { g_spec_target = t > 8 ? 8 : t; }
void hs_set_scenarios(uint32_t s) { g_scen = s > 2 ? 2 : s; }
void hs_set_capacity(uint32_t c) { g_capacity = c; }
void hs_set_log_cap(uint32_t c) { g_log_cap = c; }
uint32_t hs_round_work(uint32_t* out, uint32_t cap) { const uint32_t n = (uint32_t)g_round_work.size(); for (uint32_t i = 0; i < n && i < cap; ++i) out[i] = g_round_work[i]; return n; }
int hs_movegen(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t cnt, uint32_t* counts, uint32_t* masks, uint32_t mw) { DISPATCH(movegen(r, n, st, cnt, counts, masks, mw)) }
int hs_validate(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t cnt, const tafl_play* plays, uint8_t* codes) { DISPATCH(validate(r, n, st, cnt, plays, codes)) }
int hs_step(const tafl_rules* r, uint8_t n, uint32_t word_bits, tafl_state* st, uint32_t cnt, const tafl_play* plays, tafl_effects* eff) { DISPATCH(step(r, n, st, cnt, plays, eff)) }
int hs_step_kth(const tafl_rules* r, uint8_t n, uint32_t word_bits, tafl_state* st, uint32_t cnt, const uint32_t* ranks, tafl_play* out_plays, tafl_effects* eff) { DISPATCH(step_kth(r, n, st, cnt, ranks, out_plays, eff)) }
int hs_side_can_play(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t cnt, uint8_t side, uint8_t* out) { DISPATCH(side_can_play(r, n, st, cnt, side, out)) }
int hs_rollout(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t cnt, uint64_t seed, uint32_t sim, uint32_t max_plies, uint64_t base, tafl_rollout_result* out) { DISPATCH_DENSE(rollout(r, n, st, cnt, seed, sim, max_plies, base, out)) }
int hs_random_advance(const tafl_rules* r, uint8_t n, uint32_t word_bits, tafl_state* st, uint32_t cnt, uint64_t seed, const uint32_t* plies, uint64_t base) { DISPATCH(random_advance(r, n, st, cnt, seed, plies, base)) }
int hs_mcts(const tafl_rules* r, uint8_t n, uint32_t word_bits, const tafl_state* st, uint32_t cnt, const tafl_mcts_params* p, uint64_t base, tafl_root_child* out_children, uint32_t max_children, uint32_t* out_n, tafl_mcts_stats* stats) { DISPATCH_DENSE(mcts(r, n, st, cnt, p, base, out_children, max_children, out_n, stats)) }
void hs_set_dense13(int on) { g_dense13 = on != 0; }
}

#ifdef TAFL_STAT
// statistics builds only (ablate experiments): event counters of the TAFL_STAT_HIT marks
extern "C" { unsigned long long tafl_stat_acc[32] = {}; }
#endif
