// tafl_ops.hpp — per-game operations behind the C-ABI entry points, written once as
// __host__ __device__ functions: the HIP kernels (tafl_kernels.hip) run them one game per lane,
// tests/hostsim runs the identical code in a CPU loop for differential testing against the oracle.
//
// MCTS: the arithmetic of src/mcts.py:55-136 (select :104-123, expand :83-102, backup :127-136)
// on an explicit per-game tree (src/mcts.rs:9-28), iterative instead of recursive.  Data layout
// in DESIGN.md "MCTS arena".
#pragma once
#include <math.h>
#include "tafl_core.hpp"
#include "tafl_fast.hpp"

namespace tafl {

// ---- state <-> 16-byte quads -------------------------------------------------------------------------
struct Quad { uint32_t x, y, z, w; };
template <int NL> struct StateIO {
    static constexpr int WORDS = 2 * NL + 8;
    static constexpr int QUADS = WORDS / 4;
    static TAFL_HD void pack(const DState<NL>& s, uint32_t* v) {
        TAFL_UNROLL for (int i = 0; i < NL; ++i) { v[i] = s.att.w[i]; v[NL + i] = s.def.w[i]; }
        TAFL_UNROLL for (int i = 0; i < 4; ++i) v[2 * NL + i] = s.rep[i];
        v[2 * NL + 4] = s.turn; v[2 * NL + 5] = s.psc; v[2 * NL + 6] = s.reps; v[2 * NL + 7] = s.flags;
    }
    static TAFL_HD void unpack(const uint32_t* v, DState<NL>& s) {
        TAFL_UNROLL for (int i = 0; i < NL; ++i) { s.att.w[i] = v[i]; s.def.w[i] = v[NL + i]; }
        TAFL_UNROLL for (int i = 0; i < 4; ++i) s.rep[i] = v[2 * NL + i];
        s.turn = v[2 * NL + 4]; s.psc = v[2 * NL + 5]; s.reps = v[2 * NL + 6]; s.flags = v[2 * NL + 7];
    }
    // batch states: quad-plane SoA, quad q of game g at base[q * n + g]  (coalesced 16 B per lane)
    static TAFL_HD void load_soa(const Quad* base, uint32_t n, uint32_t g, DState<NL>& s) {
        uint32_t v[WORDS];
        TAFL_UNROLL for (int q = 0; q < QUADS; ++q) { const Quad t = base[(size_t)q * n + g]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
        unpack(v, s);
    }
    static TAFL_HD void store_soa(Quad* base, uint32_t n, uint32_t g, const DState<NL>& s) {
        uint32_t v[WORDS]; pack(s, v);
        TAFL_UNROLL for (int q = 0; q < QUADS; ++q) { Quad t; t.x = v[4 * q]; t.y = v[4 * q + 1]; t.z = v[4 * q + 2]; t.w = v[4 * q + 3]; base[(size_t)q * n + g] = t; }
    }
    // tree-node states: one contiguous record of QUADS quads per node
    static TAFL_HD void load_rec(const Quad* rec, DState<NL>& s) {
        uint32_t v[WORDS];
        TAFL_UNROLL for (int q = 0; q < QUADS; ++q) { const Quad t = rec[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
        unpack(v, s);
    }
    static TAFL_HD void store_rec(Quad* rec, const DState<NL>& s) {
        uint32_t v[WORDS]; pack(s, v);
        TAFL_UNROLL for (int q = 0; q < QUADS; ++q) { Quad t; t.x = v[4 * q]; t.y = v[4 * q + 1]; t.z = v[4 * q + 2]; t.w = v[4 * q + 3]; rec[q] = t; }
    }
};

// ---- MCTS arena -------------------------------------------------------------------------------------------
struct NodeHdr {                 // 32 bytes
    uint32_t parent;             // node id of the parent (0 for the root)
    uint32_t edge_base;          // first edge of this node inside the game's edge arena
    uint32_t ns;                 // Ns[s]                                   mcts.py:22
    uint16_t pslot;              // index of the edge (parent -> this) in the parent's edge array
    uint16_t m;                  // visited children = a prefix of the canonical legal list
    uint16_t n_legal;            // |Vs[s]|; Ps[s][a] = 1.0 / n_legal       mcts.py:87-90
    uint16_t cap;                // allocated edges
    uint16_t mv_from;            // play that led here (from tile bit index)
    uint8_t  mv_dir, mv_dist;
    uint16_t cur_from;           // canonical cursor: last expanded child's play
    uint8_t  cur_dir, cur_dist;
    uint8_t  term;               // 0 not ended, 1 Es=+1, 2 Es=-1, 3 Es=1e-4 (draw)   mcts.py:77-81
    uint8_t  expanded;           // s in Ps                                   mcts.py:83
    uint8_t  _pad[2];
};
struct Edge { double q; uint32_t n; uint32_t child; };   // Qsa, Nsa (mcts.py:20-21), next state     16 bytes

struct MctsMem {
    Quad* node_state;            // [(k * G + g) * QUADS]
    NodeHdr* hdr;                // [k * G + g]
    Edge* edges;                 // [g * edge_cap + e]
    uint32_t* node_top;          // [G]
    uint32_t* edge_top;          // [G]
    uint32_t* leaf;              // [G] leaf of the simulation whose playout value is pending
    uint8_t* kind;               // [G] 0 nothing pending, 1 rollout value pending, 2 terminal value pending
    int8_t* rvalue;              // [G] playout value handed to the backup
    uint8_t* fault;              // [G]
    // ---- simulation pipeline (DESIGN.md "speculative playout slots") ------------------------------------------
    uint32_t* sim_next;          // [G] simulations completed so far
    Quad* spec_state;            // [(j * G + g) * QUADS] leaf state of slot j
    int8_t* spec_value;          // [j * G + g] playout value of slot j
    uint8_t* spec_kind;          // [j * G + g] 0 unused, 1 playout requested, 2 value ready, 3 no playout needed (terminal child)
    uint8_t* spec_reason;        // [j * G + g] playout termination reason
    uint32_t* spec_meta;         // [j * G + g] slot j > 0: the play that leads to its leaf and the leaf's legal-play count:
                                 //     from | dir << 8 | dist << 10 | n_legal << 16 (lets the real expansion of that child reuse the state)
    uint32_t* spec_plies;        // [j * G + g] plies of the playout
    uint32_t* spec_parent;       // [G] slot j is child (spec_o0 + j) of this node ...
    int32_t* spec_o0;            // [G] ... -1: slot 0 is the (unexpanded) root itself
    uint32_t* spec_first;        // [G] simulation index of slot 0
    uint8_t* spec_n;             // [G] slots issued
    uint8_t* spec_cool;          // [G] speculation is skipped while > 0 (set after a misprediction: phases of the search in which a
                                 //     visited child beats the unvisited ones make the prediction fail repeatedly)
    uint32_t G, node_cap, edge_cap, spec_k, spec_cooldown;
};

struct LaneStats {
    uint32_t sims, rollouts, rollout_plies, depth, scanned, terminal_hits, faults;
    uint32_t reason;             // playout termination reason of this lane (valid when rollouts == 1)
    uint64_t reason_hist4;       // tree phase: 16 x 4-bit counters of the termination reasons of the playouts CONSUMED by this lane
                                 // (packed: a per-lane array indexed at run time would live in scratch)
    uint32_t spec_issued, spec_hits;
};

#define TAFL_MCTS_EPS 1e-8       /* src/mcts.py:6 */
#define TAFL_DRAW_VALUE 1e-4     /* getGameEnded draw convention, DESIGN.md */

template <int NL, int W>
struct Ops {
    using E = Engine<NL, W>;
    using S = DState<NL>;
    using K = Consts<NL>;
    using IO = StateIO<NL>;

    static TAFL_HD void caps_to_effects(const Bits<NL>& caps, uint32_t ncap, tafl_effects& e) {
        TAFL_UNROLL for (int i = 0; i < TAFL_MAX_LIMBS; ++i) e.captures[i] = 0;
        TAFL_UNROLL for (int i = 0; i < NL / 2; ++i) e.captures[i] = (uint64_t)caps.w[2 * i] | ((uint64_t)caps.w[2 * i + 1] << 32);
        e.n_captures = (uint8_t)ncap;
    }
    static TAFL_HD void status_to_effects(const S& st, int code, tafl_effects& e) {
        e.code = (uint8_t)code; e.status = (uint8_t)TAFL_F_STATUS(st.flags); e.reason = (uint8_t)TAFL_F_REASON(st.flags);
        e.winner = (uint8_t)((e.status == TAFL_STATUS_WIN && TAFL_F_WINNER(st.flags)) ? TAFL_DEFENDER : TAFL_ATTACKER);
        e._pad[0] = e._pad[1] = e._pad[2] = 0;
    }
    static TAFL_HD tafl_play to_play(const Move& m) {
        tafl_play p; p.from_row = (uint8_t)(m.from / (uint32_t)W); p.from_col = (uint8_t)(m.from % (uint32_t)W);
        p.axis = (uint8_t)(m.dir >= 2 ? TAFL_AXIS_HORIZONTAL : TAFL_AXIS_VERTICAL);
        p.disp = (int8_t)((m.dir & 1) ? -(int)m.dist : (int)m.dist);
        return p;
    }
    // dense action index (include/taflhip.h): (from tile) * 2(n-1) + slot, slots ordered V+,V-,H+,H- by distance
    static TAFL_HD uint32_t action_of(const Move& m, const K& C) {
        const uint32_t r = m.from / (uint32_t)W, c = m.from % (uint32_t)W, nm = C.n - 1;
        const uint32_t slot = m.dir == 0 ? m.dist - 1 : m.dir == 1 ? (nm - r) + m.dist - 1 : m.dir == 2 ? nm + m.dist - 1 : nm + (nm - c) + m.dist - 1;
        return (r * C.n + c) * 2u * nm + slot;
    }

    // tafl_movegen: count + dense action mask (mask may be null; it must be zero-initialised by the caller)
    static TAFL_HD uint32_t movegen(const S& st, const K& C, uint32_t* mask) {
        Moves<NL> mv;
        E::movegen(st, st.flags & TAFL_F_SIDE, C, mv);
        if (mask) {
            TAFL_UNROLL
            for (int d = 0; d < 4; ++d) {
                Bits<NL> r = mv.reach[d];
                while (any(r)) {
                    const uint32_t to = lsb(r);
                    r = andn(r, bit_at<NL>(to));
                    const Move m = E::resolve(st, (uint32_t)d, to, C);
                    const uint32_t a = action_of(m, C);
                    if (a < C.n * C.n * 2u * (C.n - 1)) mask[a >> 5] |= 1u << (a & 31);   // never write outside the game's mask
                }
            }
        }
        return mv.total;
    }
    static TAFL_HD int validate(const S& st, tafl_play p, const K& C) { return E::validate(st, p, st.flags & TAFL_F_SIDE, C, nullptr); }
    static TAFL_HD bool side_can_play(const S& st, uint32_t side, const K& C) {
        Moves<NL> mv; E::movegen(st, side, C, mv); return mv.total != 0;
    }
    // tafl_step: do_play (logic.rs:827-834)
    static TAFL_HD void step(S& st, tafl_play p, const K& C, tafl_effects* eff) {
        Move m; m.from = m.to = m.dir = m.dist = 0;
        const int code = E::validate(st, p, st.flags & TAFL_F_SIDE, C, &m);
        tafl_effects e; caps_to_effects(bz<NL>(), 0, e);
        if (code == TAFL_PLAY_OK) {
            StepOut<NL> so; Moves<NL> nx;
            E::apply(st, m, C, &so, nx);
            caps_to_effects(so.captures, so.n_captures, e);
        }
        status_to_effects(st, code, e);
        if (eff) *eff = e;
    }
    // inverse of action_of
    static TAFL_HD Move move_of_action(uint32_t a, const K& C) {
        const uint32_t nm = C.n - 1, per = 2u * nm;
        const uint32_t tile = a / per, slot = a % per, r = tile / C.n, c = tile % C.n;
        Move m; m.from = r * (uint32_t)W + c;
        if (slot < nm - r) { m.dir = 0; m.dist = slot + 1; }
        else if (slot < nm) { m.dir = 1; m.dist = slot - (nm - r) + 1; }
        else if (slot < per - c) { m.dir = 2; m.dist = slot - nm + 1; }
        else { m.dir = 3; m.dist = slot - (per - c) + 1; }
        m.to = (uint32_t)((int)m.from + E::delta(m.dir) * (int)m.dist);
        return m;
    }
    // The (rank mod count)-th legal play in canonical order = the (rank mod count)-th set bit of the dense action mask (the
    // action index preserves the canonical order).  `mask`: zeroed scratch of mask_words words (LDS on the device).
    static TAFL_HD void step_kth(S& st, uint32_t rank, const K& C, tafl_play* out_play, tafl_effects* eff, uint32_t* mask, uint32_t mask_words) {
        const uint32_t total = movegen(st, C, mask);
        tafl_effects e; caps_to_effects(bz<NL>(), 0, e);
        tafl_play pl; pl.from_row = pl.from_col = pl.axis = 0; pl.disp = 0;
        int code;
        if (total == 0) code = TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING ? TAFL_PLAY_GAME_OVER : TAFL_PLAY_NO_PIECE;
        else {
            uint32_t k = rank % total, a = 0; bool found = false;
            for (uint32_t w = 0; w < mask_words; ++w) {
                const uint32_t v = mask[w], c = (uint32_t)__builtin_popcount(v);
                if (!found) { if (k < c) { a = w * 32u + nth_set_bit32(v, k); found = true; } else k -= c; }
            }
            if (found) {
                const Move m = move_of_action(a, C);
                pl = to_play(m);
                StepOut<NL> so; Moves<NL> nx;
                E::apply(st, m, C, &so, nx);
                caps_to_effects(so.captures, so.n_captures, e);
                code = TAFL_PLAY_OK;
            } else code = TAFL_PLAY_NO_PIECE;     // count and mask disagree: surfaced by the tests
        }
        status_to_effects(st, code, e);
        if (eff) *eff = e;
        if (out_play) *out_play = pl;
    }
    // playout dispatcher: the fast two-layout engine whenever the rules allow it (tafl_fast.hpp), else the generic one.
    // `force_generic` exists for the differential tests only.
    static TAFL_HD void playout(S& st, uint32_t sk, uint32_t max_plies, const K& C, tafl_rollout_result& r, bool force_generic = false) {
        if (fast_ok<NL>(C) && !force_generic) Fast<NL, W>::rollout(st, sk, max_plies, C, r);
        else E::rollout(st, sk, max_plies, C, r);
    }
    static TAFL_HD void rollout(S st, uint64_t seed, uint64_t game_id, uint32_t sim, uint32_t max_plies, const K& C, tafl_rollout_result& r,
                                bool force_generic = false) {
        playout(st, E::sim_key(E::game_key(seed, game_id), sim), max_plies, C, r, force_generic);
    }
    static TAFL_HD void random_advance(S& st, uint64_t seed, uint64_t game_id, uint32_t plies, const K& C, bool force_generic = false) {
        tafl_rollout_result r;
        playout(st, E::sim_key(E::game_key(seed, game_id), 0xFFFFFFFFu), plies, C, r, force_generic);
    }

    // ---- MCTS -------------------------------------------------------------------------------------------------
    static TAFL_HD uint8_t term_code(const S& st) {
        const uint32_t status = TAFL_F_STATUS(st.flags);
        if (status == TAFL_STATUS_ONGOING) return 0;
        if (status == TAFL_STATUS_DRAW) return 3;
        return (TAFL_F_WINNER(st.flags) == (st.flags & TAFL_F_SIDE)) ? 1 : 2;   // value for the player to move
    }
    static TAFL_HD double term_value(uint8_t t) { return t == 1 ? 1.0 : t == 2 ? -1.0 : TAFL_DRAW_VALUE; }

    static TAFL_HD void mcts_init_game(const MctsMem& M, uint32_t g, const S& root, const K& C) {
        NodeHdr h; h.parent = 0; h.edge_base = 0; h.ns = 0; h.pslot = 0; h.m = 0; h.cap = 0;
        Moves<NL> mv; E::movegen(root, root.flags & TAFL_F_SIDE, C, mv);
        h.n_legal = (uint16_t)mv.total; h.mv_from = 0; h.mv_dir = 0; h.mv_dist = 0;
        h.cur_from = (uint16_t)TAFL_NO_SQ; h.cur_dir = 0; h.cur_dist = 0;
        h.term = term_code(root); h.expanded = 0; h._pad[0] = h._pad[1] = 0;
        M.hdr[g] = h;
        IO::store_rec(M.node_state + (size_t)g * IO::QUADS, root);
        M.node_top[g] = 1; M.edge_top[g] = 0; M.leaf[g] = 0; M.kind[g] = 0; M.rvalue[g] = 0; M.fault[g] = 0;
        M.sim_next[g] = 0; M.spec_cool[g] = 0; M.spec_n[g] = 0; M.spec_parent[g] = 0; M.spec_o0[g] = 0; M.spec_first[g] = 0;
        for (uint32_t j = 0; j < M.spec_k; ++j) M.spec_kind[(size_t)j * M.G + g] = 0;
    }

    // backup of the pending simulation (mcts.py:127-136 unwound iteratively)
    static TAFL_HD void mcts_backup(const MctsMem& M, uint32_t g) {
        const uint8_t kind = M.kind[g];
        if (kind == 0) return;
        uint32_t cur = M.leaf[g];
        double v;
        if (kind == 1) {                                        // leaf was expanded by a playout: return -v (mcts.py:100-102)
            NodeHdr* lh = &M.hdr[(size_t)cur * M.G + g];
            lh->expanded = 1; lh->ns = 0;
            v = -(double)M.rvalue[g];
        } else {
            v = -term_value(M.hdr[(size_t)cur * M.G + g].term); // terminal: return -Es[s] (mcts.py:79-81)
        }
        while (cur != 0) {
            const NodeHdr ch = M.hdr[(size_t)cur * M.G + g];
            NodeHdr* ph = &M.hdr[(size_t)ch.parent * M.G + g];
            Edge* e = &M.edges[(size_t)g * M.edge_cap + ph->edge_base + ch.pslot];
            if (e->n > 0) { e->q = ((double)e->n * e->q + v) / (double)(e->n + 1); e->n += 1; }   // mcts.py:127-129
            else { e->q = v; e->n = 1; }                                                            // mcts.py:131-133
            ph->ns += 1;                                                                            // mcts.py:135
            v = -v;                                                                                 // mcts.py:136
            cur = ch.parent;
        }
        M.kind[g] = 0;
    }

    // select + expand of one simulation (mcts.py:77-123).  Leaves M.leaf/M.kind set for the rollout + backup.
    static TAFL_HD void mcts_select_expand(const MctsMem& M, uint32_t g, double c_puct, const K& C, LaneStats& ls) {
        uint32_t cur = 0;
        ls.sims += 1;
        for (uint32_t depth = 0; depth < M.node_cap + 1; ++depth) {
            NodeHdr* hp = &M.hdr[(size_t)cur * M.G + g];
            const NodeHdr h = *hp;
            if (h.term) { M.leaf[g] = cur; M.kind[g] = 2; ls.terminal_hits += 1; return; }
            if (!h.expanded) { M.leaf[g] = cur; M.kind[g] = 1; return; }
            // pick the action with the highest upper confidence bound (mcts.py:104-119)
            ls.depth += 1;
            const double p = 1.0 / (double)h.n_legal;
            const double cp = c_puct * p;
            const double sq = sqrt((double)h.ns);
            double cur_best = -__builtin_inf(); int best = -1;
            const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
            // the edge records are fetched eight at a time (independent loads in flight: this loop is bound by memory latency),
            // then evaluated in ascending order as mcts.py does
            for (uint32_t j0 = 0; j0 < h.m; j0 += 8) {
                Edge e[8];
                TAFL_UNROLL for (uint32_t t = 0; t < 8; ++t) e[t] = eb[(j0 + t < h.m) ? j0 + t : j0];
                TAFL_UNROLL for (uint32_t t = 0; t < 8; ++t) {
                    const double u = e[t].q + cp * sq / (double)(1 + e[t].n);
                    if (j0 + t < h.m && u > cur_best) { cur_best = u; best = (int)(j0 + t); }
                }
            }
            ls.scanned += h.m;
            if (h.m < h.n_legal) {
                const double u0 = cp * sqrt((double)h.ns + TAFL_MCTS_EPS);
                if (u0 > cur_best) { cur_best = u0; best = (int)h.m; }
            }
            if (best < 0) { M.fault[g] = 1; ls.faults += 1; M.leaf[g] = cur; M.kind[g] = 0; return; }
            if ((uint32_t)best < h.m) { cur = eb[best].child; continue; }
            // ---- expand edge h.m: getNextState (mcts.py:122-123) -------------------------------------------
            // If this child was prepared as a speculative slot of the previous issue (same parent, same ordinal), its state, play and
            // legal-play count are already there: no second canon_next / apply.
            S st; Move mv; Moves<NL> nx;
            const int32_t sj = (int32_t)h.m - M.spec_o0[g];
            const bool prepared = M.spec_n[g] > 1 && M.spec_parent[g] == cur && sj >= 1 && (uint32_t)sj < (uint32_t)M.spec_n[g]
                                  && M.spec_kind[(size_t)(sj > 0 ? sj : 0) * M.G + g] >= 2;
            if (prepared) {
                const size_t so = (size_t)sj * M.G + g;
                IO::load_rec(M.spec_state + so * IO::QUADS, st);
                const uint32_t meta = M.spec_meta[so];
                mv.from = meta & 0xFFu; mv.dir = (meta >> 8) & 3u; mv.dist = (meta >> 10) & 0x3Fu; mv.to = 0; nx.total = meta >> 16;
            } else {
                IO::load_rec(M.node_state + ((size_t)cur * M.G + g) * IO::QUADS, st);
                mv.from = h.cur_from; mv.to = 0; mv.dir = h.cur_dir; mv.dist = h.cur_dist;
                if (!E::canon_next(st, st.flags & TAFL_F_SIDE, C, mv)) { M.fault[g] = 1; ls.faults += 1; M.leaf[g] = cur; M.kind[g] = 0; return; }
                E::apply(st, mv, C, nullptr, nx);
            }
            const uint32_t id = M.node_top[g];
            uint32_t base = h.edge_base; uint32_t cap = h.cap;
            if (h.m == cap) {                                      // grow the edge array (amortised doubling)
                const uint32_t ncap = cap ? cap * 2u : 4u;
                const uint32_t nbase = M.edge_top[g];
                if (id >= M.node_cap || nbase + ncap > M.edge_cap) { M.fault[g] = 1; ls.faults += 1; M.leaf[g] = cur; M.kind[g] = 0; return; }
                Edge* dst = &M.edges[(size_t)g * M.edge_cap + nbase];
                for (uint32_t j = 0; j < h.m; ++j) dst[j] = eb[j];
                M.edge_top[g] = nbase + ncap; base = nbase; cap = ncap;
            } else if (id >= M.node_cap) { M.fault[g] = 1; ls.faults += 1; M.leaf[g] = cur; M.kind[g] = 0; return; }
            Edge ne; ne.q = 0.0; ne.n = 0; ne.child = id;
            M.edges[(size_t)g * M.edge_cap + base + h.m] = ne;
            hp->edge_base = base; hp->cap = (uint16_t)cap; hp->m = (uint16_t)(h.m + 1);
            hp->cur_from = (uint16_t)mv.from; hp->cur_dir = (uint8_t)mv.dir; hp->cur_dist = (uint8_t)mv.dist;
            NodeHdr nh; nh.parent = cur; nh.edge_base = 0; nh.ns = 0; nh.pslot = h.m; nh.m = 0; nh.n_legal = (uint16_t)nx.total; nh.cap = 0;
            nh.mv_from = (uint16_t)mv.from; nh.mv_dir = (uint8_t)mv.dir; nh.mv_dist = (uint8_t)mv.dist;
            nh.cur_from = (uint16_t)TAFL_NO_SQ; nh.cur_dir = 0; nh.cur_dist = 0;
            nh.term = term_code(st); nh.expanded = 0; nh._pad[0] = nh._pad[1] = 0;
            M.hdr[(size_t)id * M.G + g] = nh;
            IO::store_rec(M.node_state + ((size_t)id * M.G + g) * IO::QUADS, st);
            M.node_top[g] = id + 1;
            M.leaf[g] = id;
            if (nh.term) { M.kind[g] = 2; ls.terminal_hits += 1; } else M.kind[g] = 1;
            return;
        }
        M.fault[g] = 1; ls.faults += 1; M.kind[g] = 0;
    }

    // ---- simulation pipeline ----------------------------------------------------------------------------------------
    // One call advances game g by as many simulations as it can without waiting for a playout.  Every simulation does its
    // real selection on the committed tree (mcts_select_expand); when that expands child `ord` of node P, the value of its
    // playout is taken from a slot only if the slot was issued for exactly (P, ord, this simulation index) — the leaf
    // state and the RNG key (game, sim) are then identical, so the value is the one a fresh playout would return.
    // When no slot matches, the leaf becomes slot 0 and the next spec_k-1 slots are filled with the children
    // ord+1, ord+2, ... of P for the following simulation indices: unvisited edges tie in PUCT and the lowest index wins
    // (mcts.py:117-119), so unless a backed-up value lifts a visited child above them these are the next expansions.
    static TAFL_HD void consume_stats(const MctsMem& M, uint32_t g, uint32_t j, LaneStats& ls) {
        ls.rollouts += 1; ls.rollout_plies += M.spec_plies[(size_t)j * M.G + g]; ls.reason_hist4 += 1ull << (4u * (M.spec_reason[(size_t)j * M.G + g] & 15u));
    }
    static TAFL_HD void mcts_tree_step(const MctsMem& M, uint32_t g, double c_puct, uint32_t n_sims, const K& C, LaneStats& ls) {
        uint32_t sim = M.sim_next[g];
        if (M.kind[g] == 1) {                                    // slot 0 of the previous call: its playout has run
            M.rvalue[g] = M.spec_value[g];
            consume_stats(M, g, 0, ls);
            mcts_backup(M, g); ++sim;
        }
        const uint32_t had = M.spec_n[g], first = M.spec_first[g], sparent = M.spec_parent[g];
        const int32_t so0 = M.spec_o0[g];
        for (;;) {
            if (sim >= n_sims) { M.spec_n[g] = 0; break; }
            mcts_select_expand(M, g, c_puct, C, ls);
            const uint8_t kind = M.kind[g];
            if (kind == 0) { ++sim; continue; }                                     // fault (flagged per game): nothing to back up
            if (kind == 2) { mcts_backup(M, g); ++sim; continue; }                  // terminal node: value known at once
            const uint32_t L = M.leaf[g];
            uint32_t P = 0; int32_t ord = -1;
            if (L != 0) { const NodeHdr lh = M.hdr[(size_t)L * M.G + g]; P = lh.parent; ord = (int32_t)lh.pslot; }
            const uint32_t j = sim - first;
            if (had > 0 && sim > first && j < had && sparent == P && so0 + (int32_t)j == ord && M.spec_kind[(size_t)j * M.G + g] == 2) {
                M.rvalue[g] = M.spec_value[(size_t)j * M.G + g];                  // predicted expansion: reuse its playout
                consume_stats(M, g, j, ls); ls.spec_hits += 1;
                mcts_backup(M, g); ++sim;
                continue;
            }
            // issue new slots: slot 0 = this leaf, slots 1.. = the next unvisited children of P
            S lst; IO::load_rec(M.node_state + ((size_t)L * M.G + g) * IO::QUADS, lst);
            IO::store_rec(M.spec_state + (size_t)g * IO::QUADS, lst);
            M.spec_kind[g] = 1;
            uint32_t cnt = 1;
            const NodeHdr ph = M.hdr[(size_t)P * M.G + g];
            // adaptive gate: a misprediction (speculative slots left unconsumed) pauses speculation for spec_cooldown issues
            uint32_t cool = M.spec_cool[g];
            if (had > 1 && sim < first + had) cool = M.spec_cooldown;
            const bool speculate = cool == 0;
            M.spec_cool[g] = (uint8_t)(cool > 0 ? cool - 1 : 0);
            if (speculate && M.spec_k > 1 && sim + 1 < n_sims) {
                S pst; IO::load_rec(M.node_state + ((size_t)P * M.G + g) * IO::QUADS, pst);
                Move cur; cur.from = ph.cur_from; cur.to = 0; cur.dir = ph.cur_dir; cur.dist = ph.cur_dist;
                const uint32_t pside = pst.flags & TAFL_F_SIDE;
                for (uint32_t t = 1; t < M.spec_k && sim + t < n_sims && (uint32_t)(ord + (int32_t)t) < ph.n_legal; ++t) {
                    if (!E::canon_next(pst, pside, C, cur)) break;
                    S cst = pst; Moves<NL> nx;
                    E::apply(cst, cur, C, nullptr, nx);
                    const bool term = TAFL_F_STATUS(cst.flags) != TAFL_STATUS_ONGOING;
                    IO::store_rec(M.spec_state + ((size_t)t * M.G + g) * IO::QUADS, cst);
                    M.spec_meta[(size_t)t * M.G + g] = cur.from | (cur.dir << 8) | (cur.dist << 10) | (nx.total << 16);
                    M.spec_kind[(size_t)t * M.G + g] = term ? 3 : 1;
                    if (!term) ls.spec_issued += 1;
                    cnt = t + 1;
                }
            }
            for (uint32_t t = cnt; t < M.spec_k; ++t) M.spec_kind[(size_t)t * M.G + g] = 0;
            M.spec_n[g] = (uint8_t)cnt; M.spec_first[g] = sim; M.spec_parent[g] = P; M.spec_o0[g] = ord;
            break;                                                                  // wait for the playouts
        }
        M.sim_next[g] = sim;
    }

    // playout of slot j of game g (predict() of mcts.py:85 in random-rollout mode)
    static TAFL_HD void mcts_slot_rollout(const MctsMem& M, uint32_t j, uint32_t g, uint64_t seed, uint64_t game_id, uint32_t sim_offset,
                                          uint32_t max_plies, const K& C) {
        const size_t o = (size_t)j * M.G + g;
        if (j >= M.spec_n[g] || M.spec_kind[o] != 1) return;
        S st; IO::load_rec(M.spec_state + o * IO::QUADS, st);
        tafl_rollout_result r;
        playout(st, E::sim_key(E::game_key(seed, game_id), sim_offset + M.spec_first[g] + j), max_plies, C, r);
        M.spec_value[o] = r.value; M.spec_reason[o] = r.reason; M.spec_plies[o] = r.plies; M.spec_kind[o] = 2;
    }

    // root statistics (mcts.py:40-41): visited root children in canonical order
    static TAFL_HD uint32_t mcts_root_children(const MctsMem& M, uint32_t g, const K& C, tafl_root_child* out, uint32_t max_children) {
        const NodeHdr h = M.hdr[g];
        const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
        for (uint32_t j = 0; j < h.m && j < max_children; ++j) {
            const Edge e = eb[j];
            const NodeHdr ch = M.hdr[(size_t)e.child * M.G + g];
            Move m; m.from = ch.mv_from; m.dir = ch.mv_dir; m.dist = ch.mv_dist; m.to = 0;
            tafl_root_child rc; rc.play = to_play(m); rc.action = action_of(m, C); rc.visits = e.n; rc.q = e.q;
            out[j] = rc;
        }
        return h.m;
    }
};

}  // namespace tafl
