// tafl_guided.hpp — MCTS with an external evaluator: src/mcts.py:55-136 where nnet.predict (mcts.py:85) is the CALLER's
// network, evaluated for all games of the batch at once between two kernel launches.
//
// One lock-step round per game:   [expand the pending leaf with the priors / value just delivered, back the value up]
//                                 -> [run searches from the root until one reaches a node that is not in Ps (a leaf that
//                                    needs predict) — searches that end in a terminal node are completed on the way]
// so that every launch leaves at most one leaf per game waiting for the network.  The arithmetic is mcts.py's, in float64
// and in its order of operations:
//   Ps[s] = Ps[s] * valids                  priors arrive as float32, `* valids` widens them to float64       mcts.py:87
//   sum_Ps_s = np.sum(Ps[s])                numpy's PAIRWISE summation over the dense action vector           mcts.py:88
//   Ps[s] /= sum   |  (Ps[s] + valids) / np.sum(..)  when every valid move was masked                      mcts.py:89-98
//   u = Qsa + cpuct*Ps*sqrt(Ns)/(1+Nsa)  |  cpuct*Ps*sqrt(Ns + EPS),  strict >, ascending action index      mcts.py:109-119
//   Qsa = (Nsa*Qsa + v)/(Nsa+1), Nsa += 1, Ns += 1, return -v                                             mcts.py:127-136
// The value is taken as a Python float (float(v)); the tree is explicit (no transposition table), as in the rollout mode.
//
// Storage differs from the rollout mode (tafl_ops.hpp): with non-uniform priors the visited children are no longer a prefix
// of the legal list, so a node owns one edge per LEGAL move, in canonical (= ascending action index) order.
#pragma once
#include "tafl_ops.hpp"

namespace tafl {

struct GNode {                   // 16 bytes
    uint32_t parent;             // node id of the parent (0 for the root)
    uint32_t edge_base;          // first edge of this node inside the game's edge arena
    uint32_t ns;                 // Ns[s]
    uint16_t n_legal;            // |Vs[s]|
    uint8_t  term;               // 0 not ended, 1 Es=+1, 2 Es=-1, 3 draw
    uint8_t  expanded;           // s in Ps
};
struct GEdge {                   // 32 bytes
    double p, q;                 // Ps[s][a], Qsa
    uint32_t n, child;           // Nsa (0 = (s,a) not in Qsa), node of the next state (0 = not created yet)
    uint32_t action;             // dense action index
    uint16_t from; uint8_t dir, dist;
};
struct GuidedMem {
    Quad* node_state;            // [(k * G + g) * QUADS]
    GNode* hdr;                  // [k * G + g]
    uint32_t* pedge;             // [k * G + g] index (inside the game's edge arena) of the edge parent -> this node
    GEdge* edges;                // [g * edge_cap + e]
    uint32_t* node_top;          // [G]
    uint32_t* edge_top;          // [G]
    uint32_t* leaf;              // [G] node waiting for predict()
    uint8_t* kind;               // [G] 0 nothing pending, 1 leaf waits for predict, 3 all simulations done
    uint8_t* fault;              // [G] arena overflow / no selectable action: the game stops searching
    uint32_t* sims_done;         // [G]
    uint32_t G, node_cap, edge_cap;
};
struct GuidedStats { uint32_t sims, predicts, terminal_hits, faults, depth; };

template <int NL, int W>
struct Guided {
    using E = Engine<NL, W>;
    using O = Ops<NL, W>;
    using S = DState<NL>;
    using K = Consts<NL>;
    using IO = StateIO<NL>;

    static TAFL_HD void init_game(const GuidedMem& M, uint32_t g, const S& root) {
        GNode h; h.parent = 0; h.edge_base = 0; h.ns = 0; h.n_legal = 0; h.term = O::term_code(root); h.expanded = 0;
        M.hdr[g] = h; M.pedge[g] = 0;
        IO::store_rec(M.node_state + (size_t)g * IO::QUADS, root);
        M.node_top[g] = 1; M.edge_top[g] = 0; M.leaf[g] = 0; M.kind[g] = 0; M.fault[g] = 0; M.sims_done[g] = 0;
    }

    // np.sum over the dense action vector whose only non-zero entries are e[0..cnt) (ascending action index); `add1`
    // adds 1.0 to every entry first (the `Ps + valids` of mcts.py:97).  numpy: DOUBLE_pairwise_sum, blocks of <= 128
    // elements with 8 interleaved accumulators, halves split at a multiple of 8; adding the zeros in between is exact.
    static TAFL_HD double leaf_sum(const GEdge* e, uint32_t cnt, uint32_t& cur, uint32_t lo, uint32_t n, bool add1) {
        const uint32_t hi = lo + n;
        if (n < 8) {
            double res = 0.;
            while (cur < cnt && e[cur].action < hi) { res += add1 ? e[cur].p + 1.0 : e[cur].p; ++cur; }
            return res;
        }
        double r0 = 0., r1 = 0., r2 = 0., r3 = 0., r4 = 0., r5 = 0., r6 = 0., r7 = 0.;
        const uint32_t bulk = lo + (n - (n % 8u));
        while (cur < cnt && e[cur].action < bulk) {
            const double v = add1 ? e[cur].p + 1.0 : e[cur].p;
            const uint32_t j = (e[cur].action - lo) & 7u;
            r0 += j == 0 ? v : 0.; r1 += j == 1 ? v : 0.; r2 += j == 2 ? v : 0.; r3 += j == 3 ? v : 0.;
            r4 += j == 4 ? v : 0.; r5 += j == 5 ? v : 0.; r6 += j == 6 ? v : 0.; r7 += j == 7 ? v : 0.;
            ++cur;
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        while (cur < cnt && e[cur].action < hi) { res += add1 ? e[cur].p + 1.0 : e[cur].p; ++cur; }
        return res;
    }
    static TAFL_HD double np_sum_sparse(const GEdge* e, uint32_t cnt, uint32_t A, bool add1) {
        struct Frame { uint32_t lo, n; uint32_t stage; double left; };
        // the recursion's frames: a per-lane stack indexed at run time.  In registers that is scratch memory (it was 400 of this kernel's
        // 496 B per lane); on the device it lives in LDS, one column per lane of the 64-thread workgroup
#if defined(__HIP_DEVICE_COMPILE__)
        __shared__ Frame stk_lds[16 * 64];
        Frame* const stk = stk_lds + (threadIdx.x & 63u);
        constexpr int SS = 64;
#else
        Frame stk_host[16]; Frame* const stk = stk_host;
        constexpr int SS = 1;
#endif
        int sp = 0; uint32_t cur = 0; double ret = 0.;
        stk[sp * SS].lo = 0; stk[sp * SS].n = A; stk[sp * SS].stage = 0; stk[sp * SS].left = 0.; ++sp;
        while (sp > 0) {
            Frame& f = stk[(sp - 1) * SS];
            if (f.n <= 128u) { ret = leaf_sum(e, cnt, cur, f.lo, f.n, add1); --sp; continue; }
            uint32_t n2 = f.n / 2u; n2 -= n2 % 8u;
            if (f.stage == 0) { f.stage = 1; Frame& c = stk[sp * SS]; c.lo = f.lo; c.n = n2; c.stage = 0; c.left = 0.; ++sp; }
            else if (f.stage == 1) { f.left = ret; f.stage = 2; Frame& c = stk[sp * SS]; c.lo = f.lo + n2; c.n = f.n - n2; c.stage = 0; c.left = 0.; ++sp; }
            else { ret = f.left + ret; --sp; }
        }
        return 0.0 + ret;                                       // np.add.reduce: identity + pairwise sum
    }

    // mcts.py:127-136 unwound iteratively from node `cur` (whose search returned v) to the root
    static TAFL_HD void backup(const GuidedMem& M, uint32_t g, uint32_t cur, double v) {
        while (cur != 0) {
            const uint32_t par = M.hdr[(size_t)cur * M.G + g].parent;
            GEdge* e = &M.edges[(size_t)g * M.edge_cap + M.pedge[(size_t)cur * M.G + g]];
            if (e->n > 0) { e->q = ((double)e->n * e->q + v) / (double)(e->n + 1); e->n += 1; }
            else { e->q = v; e->n = 1; }
            M.hdr[(size_t)par * M.G + g].ns += 1;
            v = -v;
            cur = par;
        }
    }

    // mcts.py:83-102 for the pending leaf, with the network's answer
    static TAFL_HD bool expand(const GuidedMem& M, uint32_t g, const float* priors, uint32_t A, const K& C) {
        const uint32_t L = M.leaf[g];
        S st; IO::load_rec(M.node_state + ((size_t)L * M.G + g) * IO::QUADS, st);
        const uint32_t base = M.edge_top[g];
        GEdge* e = &M.edges[(size_t)g * M.edge_cap + base];
        const uint32_t side = st.flags & TAFL_F_SIDE;
        Move cur = E::canon_start();
        uint32_t cnt = 0;
        while (E::canon_next(st, side, C, cur)) {                 // getValidMoves, canonical = ascending action index
            if (base + cnt >= M.edge_cap) return false;
            const uint32_t a = O::action_of(cur, C);
            GEdge ne; ne.p = (double)priors[a] * 1.0; ne.q = 0.0; ne.n = 0; ne.child = 0; ne.action = a;
            ne.from = (uint16_t)cur.from; ne.dir = (uint8_t)cur.dir; ne.dist = (uint8_t)cur.dist;
            e[cnt++] = ne;
        }
        const double sum = np_sum_sparse(e, cnt, A, false);
        if (sum > 0) { for (uint32_t i = 0; i < cnt; ++i) e[i].p /= sum; }
        else {
            const double s2 = np_sum_sparse(e, cnt, A, true);
            for (uint32_t i = 0; i < cnt; ++i) e[i].p = (e[i].p + 1.0) / s2;
        }
        GNode* h = &M.hdr[(size_t)L * M.G + g];
        h->edge_base = base; h->n_legal = (uint16_t)cnt; h->ns = 0; h->expanded = 1;
        M.edge_top[g] = base + cnt;
        return true;
    }

    // One round for game g.  `priors` = this game's row of the network output (may be null when nothing is pending).
    static TAFL_HD void step(const GuidedMem& M, uint32_t g, const float* priors, float value, uint32_t A, double c_puct, uint32_t n_sims,
                             const K& C, GuidedStats& gs) {
        if (M.fault[g]) { M.kind[g] = 3; return; }
        uint32_t sims = M.sims_done[g];
        if (M.kind[g] == 1) {
            if (!priors || !expand(M, g, priors, A, C)) { M.fault[g] = 1; gs.faults += 1; M.kind[g] = 3; return; }
            gs.predicts += 1;
            backup(M, g, M.leaf[g], -(double)value);              // return -v (mcts.py:102) into the callers
            ++sims; gs.sims += 1;
        }
        M.kind[g] = 3;
        while (sims < n_sims) {
            uint32_t cur = 0; bool waiting = false;
            for (uint32_t depth = 0; depth <= M.node_cap; ++depth) {
                const GNode h = M.hdr[(size_t)cur * M.G + g];
                if (h.term) { backup(M, g, cur, -O::term_value(h.term)); gs.terminal_hits += 1; break; }   // mcts.py:79-81
                if (!h.expanded) { M.leaf[g] = cur; M.kind[g] = 1; waiting = true; break; }               // mcts.py:83-85
                gs.depth += 1;
                GEdge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
                const double sq = sqrt((double)h.ns), sq0 = sqrt((double)h.ns + TAFL_MCTS_EPS);
                double cur_best = -__builtin_inf(); int best = -1;
                // eight edge records in flight at a time (the scan is bound by memory latency), evaluated in ascending action order
                for (uint32_t j0 = 0; j0 < h.n_legal; j0 += 8) {
                    double ep[8], eq[8]; uint32_t en[8];
                    TAFL_UNROLL for (uint32_t t = 0; t < 8; ++t) { const GEdge* e = &eb[(j0 + t < h.n_legal) ? j0 + t : j0]; ep[t] = e->p; eq[t] = e->q; en[t] = e->n; }
                    TAFL_UNROLL for (uint32_t t = 0; t < 8; ++t) {
                        const double u = en[t] > 0 ? eq[t] + c_puct * ep[t] * sq / (double)(1 + en[t]) : c_puct * ep[t] * sq0;
                        if (j0 + t < h.n_legal && u > cur_best) { cur_best = u; best = (int)(j0 + t); }
                    }
                }
                if (best < 0) { M.fault[g] = 1; gs.faults += 1; M.sims_done[g] = sims; return; }
                uint32_t child = eb[best].child;
                if (child == 0) {                                     // getNextState (mcts.py:122-123): first visit of this edge
                    const uint32_t id = M.node_top[g];
                    if (id >= M.node_cap) { M.fault[g] = 1; gs.faults += 1; M.sims_done[g] = sims; return; }
                    S st; IO::load_rec(M.node_state + ((size_t)cur * M.G + g) * IO::QUADS, st);
                    Move mv; mv.from = eb[best].from; mv.dir = eb[best].dir; mv.dist = eb[best].dist;
                    mv.to = (uint32_t)((int)mv.from + E::delta(mv.dir) * (int)mv.dist);
                    Moves<NL> nx;
                    E::apply(st, mv, C, nullptr, nx);
                    GNode nh; nh.parent = cur; nh.edge_base = 0; nh.ns = 0; nh.n_legal = 0; nh.term = O::term_code(st); nh.expanded = 0;
                    M.hdr[(size_t)id * M.G + g] = nh;
                    M.pedge[(size_t)id * M.G + g] = h.edge_base + (uint32_t)best;
                    IO::store_rec(M.node_state + ((size_t)id * M.G + g) * IO::QUADS, st);
                    M.node_top[g] = id + 1;
                    eb[best].child = id; child = id;
                }
                cur = child;
            }
            if (waiting) break;
            ++sims; gs.sims += 1;
        }
        M.sims_done[g] = sims;
    }

    // visited root edges in ascending action order (mcts.py:40-41)
    static TAFL_HD uint32_t root_children(const GuidedMem& M, uint32_t g, tafl_root_child* out, uint32_t max_children) {
        const GNode h = M.hdr[g];
        if (!h.expanded) return 0;
        const GEdge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
        uint32_t k = 0;
        for (uint32_t j = 0; j < h.n_legal; ++j) {
            const GEdge e = eb[j];
            if (e.n == 0) continue;
            if (k < max_children) {
                Move m; m.from = e.from; m.dir = e.dir; m.dist = e.dist; m.to = 0;
                tafl_root_child rc; rc.play = O::to_play(m); rc.action = e.action; rc.visits = e.n; rc.q = e.q;
                out[k] = rc;
            }
            ++k;
        }
        return k;
    }
};

}  // namespace tafl
