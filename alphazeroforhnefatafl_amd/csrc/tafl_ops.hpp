// tafl_ops.hpp — per-game operations behind the C-ABI entry points, written once as
// __host__ __device__ functions: the HIP kernels (tafl_capi.hip) run them one game per lane,
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
struct alignas(16) NodeHdr {     // 32 bytes (two 16-byte accesses)
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
struct alignas(16) Edge { double q; uint32_t n; uint32_t child; };   // Qsa, Nsa (mcts.py:20-21), next state     16 bytes

// undo records of the speculation pass (mcts_speculate): the tree is modified in place by ASSUMED playout values to predict the next
// selections and then restored bit for bit
// The undo log lives in the caller's scratch - LDS on the device, one log per lane - as word-interleaved records: word k of record i of lane
// l at base[(i * WORDS + k) * stride + l] (stride = lanes that share the scratch: consecutive lanes hit consecutive LDS banks).
constexpr uint32_t kUndoEWords = 5;       // edge index inside the game's arena + the edge's old contents (q lo, q hi, n, child)
constexpr uint32_t kUndoHWords = 9;       // node id + the header's old contents (8 words)
struct LogMem { uint32_t* base; uint32_t stride, lane, cap; };       // cap records of each kind: cap * (kUndoEWords + kUndoHWords) * stride words

struct MctsMem {
    Quad* node_state;            // [(k * G + g) * QUADS]
    NodeHdr* hdr;                // [k * G + g]
    Edge* edges;                 // [g * edge_cap + e]
    uint32_t* node_top;          // [G]
    uint32_t* edge_top;          // [G]
    uint32_t* leaf;              // [G] leaf of the simulation whose playout value is pending
    uint8_t* kind;               // [G] 0 nothing pending, 1 rollout value pending, 2 terminal value pending
    uint8_t* fault;              // [G]
    // ---- simulation pipeline (DESIGN.md "playout slots") ---------------------------------------------------------------
    // Every game owns spec_k playout slots.  A slot holds one LEAF the search needs (or is predicted to need) a playout value for: child
    // number `ord` of node `node` (spec_ref), its prepared state, and - once the playout has run - its value.  The value of a leaf is a
    // function of the leaf's position alone (Engine::state_hash keys the playout), so a slot stays valid until the search really expands
    // that child, however many simulations later.  One slot is the leaf of the pending real simulation (spec_pend).
    uint32_t* sim_base;          // [G] added to the salt of the RNG key: 0 for a plain search, move * n_sims in a self-play run
    uint32_t* sim_next;          // [G] simulations completed so far
    Quad* spec_state;            // [(j * G + g) * QUADS] leaf state of slot j
    int8_t* spec_value;          // [j * G + g] playout value of slot j (kind 2); the child's terminal code (kind 3)
    uint8_t* spec_kind;          // [j * G + g] 0 free, 1 playout requested, 2 value ready, 3 no playout needed (terminal child)
    uint8_t* spec_reason;        // [j * G + g] playout termination reason
    uint8_t* spec_cls;           // [j * G + g] kind 1: priority class of the playout in the round's work lists (0 = the pending leaf; distinct per game)
    uint32_t* spec_meta;         // [j * G + g] the play that leads to the slot's leaf and the leaf's legal-play count:
                                 //     from | dir << 8 | dist << 10 | n_legal << 16 (the real expansion of that child reuses state and play)
    uint32_t* spec_plies;        // [j * G + g] plies of the playout
    uint32_t* spec_ref;          // [j * G + g] node | ord << 20: the slot's leaf is child number ord of node (its index in the node's edge list once it
                                 //     is really expanded); ord 0xFFF: the leaf is the node itself (a pending leaf that already exists in the tree)
    uint32_t* spec_pend;         // [G] slot of the pending leaf (valid while kind[g] == 1) | predicted playouts requested by the last step << 8
                                 //     | width << 16: predicted playouts this game may request next (grows by one when everything requested
                                 //     last time was consumed, falls back towards what was consumed otherwise)
    uint32_t* spec_bias;         // [G] playouts consumed by this search that the defender won | that the attacker won << 16 (steers the prediction
                                 //     pass only: which value a playout in flight is assumed to return)
    uint32_t G, node_cap, edge_cap, spec_k;      // spec_k: slots per game that exist (capacity of the arrays above)
    uint32_t flags;              // TAFL_MCTS_FLAG_* semantics bits of the running search
};

// Self-play run (tafl_selfplay_run): every game runs n_moves searches one after the other, each followed by its most visited root play on
// the batch state; a game starts its next search as soon as ITS OWN search is done, so the games of a batch are at different phases of
// their searches and every launch finds the device full (a batch of synchronous searches ends each of them in a tail of nearly empty
// rounds: DESIGN.md section 6).  Per game identical to the loop { tafl_mcts_run(sim_offset + move * n_sims); tafl_mcts_play_best }.
struct SelfPlay {
    uint32_t* moves_done;        // [G] searches + plays completed (n_moves: the game takes no further part)
    uint32_t* start_round;       // [G] launch index at which the game's current search began (its plan counts from there)
    tafl_play* plays;            // [n_moves * G] the plays made (all-zero play: the game was over)
    uint32_t n_moves;
};

struct LaneStats {
    uint32_t sims, rollouts, rollout_plies, depth, scanned, terminal_hits, faults;
    uint32_t reason;             // playout termination reason of this lane (valid when rollouts == 1)
    uint64_t reason_hist4;       // tree phase: 16 x 4-bit counters of the termination reasons of the playouts CONSUMED by this lane
                                 // (packed: a per-lane array indexed at run time would live in scratch)
    uint32_t spec_issued, spec_hits;
};

constexpr uint32_t kMctsMaxSlots = 8;     // playout slots per game that can exist (MctsMem::spec_k <= this): bound of the unrolled slot loops
constexpr uint32_t kMctsMaxPass = 12;     // simulations one prediction pass looks ahead at most (the undo log usually ends it earlier)
constexpr uint32_t kMctsExtraReq = 2;      // playouts the second scenario of a prediction pass may request beyond the round's share
constexpr uint32_t kMctsMaxNodes = 1u << 20;   // node ids fit the 20 bits of MctsMem::spec_ref
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

    // ---- tafl_movegen: count + dense action mask ------------------------------------------------------------------------------
    // The dense mask holds, per source tile, one field of 2(n-1) bits: [V+ distances 1..n-1-r][V- 1..r][H+ 1..n-1-c][H- 1..c]
    // (include/taflhip.h).  It is written one LINE of the board at a time: pass `i` serves the horizontal plays of the pieces in row i (its
    // occupancy is n contiguous bits of the position) and the vertical plays of the pieces in column i (n bits gathered at stride W).  On a
    // line the plays of one piece in one direction are the tiles between it and the first occupied tile (ValidPlayIterator, play.rs:189-225):
    //   run   = the empty tiles from the piece outwards up to the first occupied tile or the board's end,
    //           cut behind the first tile the piece may not pass (pass_forbid: NoPass / KingPass throne; the tile itself stays),
    //   plays = run minus the tiles it may not stop on (land_forbid: corners, throne by rule); a slow piece keeps distance 1 only
    // - a handful of 32-bit operations per piece and direction, no reach set of the whole position and no mirrored position (round 2 built
    // both: ~180 wave-instructions per game, now ~70).  On the device a workgroup serves 64 games with one WAVE PER LINE (lane = game): the
    // line index is wave-uniform, so every bit position below is a scalar, and the fields are OR-ed into the game's mask in LDS.
    // land_forbid / pass_forbid are made of corners and throne (make_consts_ct), sets that equal their mirror image: column i of them is row i.
#if defined(__HIP_DEVICE_COMPILE__)
#define TAFL_MASK_OR(p, v) atomicOr((p), (v))          /* the waves of a workgroup write one game's mask */
#else
#define TAFL_MASK_OR(p, v) (*(p) |= (v))
#endif
    // n (<= 15) bits of `a` from bit `base`
    static TAFL_HD uint32_t line_bits(const Bits<NL>& a, uint32_t base, uint32_t n) { return (uint32_t)field64<0>(a, base) & ((1u << n) - 1u); }
    // column `c` of a board word as an n-bit value (bit k = row k): the word moved c (< W <= 15) bits down, one funnel shift per limb, puts
    // tile (k, c) at the fixed bit k * W (off-board rows read as 0: board words hold no bits there)
    static TAFL_HD uint32_t col_bits(const Bits<NL>& a, uint32_t c, uint32_t n) {
        (void)n;
        Bits<NL> sft;
        TAFL_UNROLL for (int i = 0; i < NL; ++i) sft.w[i] = (a.w[i] >> c) | ((c != 0u && i + 1 < NL) ? (a.w[i + 1 < NL ? i + 1 : i] << ((32u - c) & 31u)) : 0u);
        uint32_t v = 0;
        TAFL_UNROLL for (int k = 0; k < W; ++k) v |= ((sft.w[(k * W) >> 5] >> ((k * W) & 31)) & 1u) << k;
        return v;
    }
    // one line as the rays see it: occupancy, land_forbid and pass_forbid bits of its n tiles, and the same three mirrored (tile n-1-k at bit
    // k), so that a ray towards lower positions is a ray towards higher positions of the mirror image
    struct LineView { uint32_t occ, lf, pf, occ_m, lf_m, pf_m; };
    static TAFL_HD uint32_t mirror_n(uint32_t v, uint32_t n) { return bitrev32(v) >> (32u - n); }
    // plays of the piece at position p of a line towards higher positions, bit k = distance k + 1
    static TAFL_HD uint32_t ray_up(uint32_t occ, uint32_t lf, uint32_t pf, uint32_t p, uint32_t n) {
        const uint32_t ext = n - 1u - p;                                 // tiles beyond p (<= 14)
        const uint32_t f = ~(occ >> (p + 1u)) & ((1u << ext) - 1u);       // empty tiles, nearest first
        const uint32_t t = ~f;                                           // (bit ext of t is set: the run ends at the board's end at the latest)
        uint32_t run = (t & (0u - t)) - 1u;
        const uint32_t pfb = (pf >> (p + 1u)) & run, cut = pfb & (0u - pfb);
        run = pfb ? (run & ((cut << 1) - 1u)) : run;
        return run & ~(lf >> (p + 1u));
    }
    // Pass `I` of the dense mask: horizontal plays of the pieces in row I and vertical plays of the pieces in column I.  Returns their number;
    // ORs them into `row` (the game's mask) unless it is null.  (The line index is a template parameter: every bit position is then a
    // literal; movegen_line dispatches on the run-time index, which is wave-uniform on the device.)  The pieces of the row and of the column
    // go through ONE loop: 64 games share an instruction stream, and the largest sum of two counts is smaller than the sum of the largest.
    template <int I> static TAFL_HD uint32_t movegen_line_ct(const S& st, const K& C, uint32_t* row) {
        constexpr uint32_t i = (uint32_t)I, base = (uint32_t)(I * W);
        if (TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING) return 0;                   // logic.rs:165-167
        const uint32_t side = st.flags & TAFL_F_SIDE, n = C.n, nm = n - 1u;
        const Bits<NL> occ = (st.att | st.def) & C.board, mine = sel(side != 0, st.def, st.att) & C.board;
        const int scls = side ? CLS_DEF : CLS_ATT;
        // rule masks of line I for the soldiers of the side and for the king (column I of them = row I: the sets equal their mirror image)
        const uint32_t s_lf = line_bits(C.land_forbid[scls], base, n), s_pf = line_bits(C.pass_forbid[scls], base, n);
        const uint32_t k_lf = line_bits(C.land_forbid[CLS_KING], base, n), k_pf = line_bits(C.pass_forbid[CLS_KING], base, n);
        const uint32_t s_lf_m = mirror_n(s_lf, n), s_pf_m = mirror_n(s_pf, n), k_lf_m = mirror_n(k_lf, n), k_pf_m = mirror_n(k_pf, n);
        const bool s_slow = C.slow[scls] != 0, k_slow = C.slow[CLS_KING] != 0;
        // the king: a defender on the tile the king nibble names (board/state.rs:24-26); his own rule masks only where they differ
        const uint32_t kr = TAFL_F_KROW(st.flags), kc = TAFL_F_KCOL(st.flags);
        const bool kingside = side != 0 && !C.king_like_soldier;
        const uint32_t ro = line_bits(occ, base, n), co = col_bits(occ, i, n);
        const uint32_t ro_m = mirror_n(ro, n), co_m = mirror_n(co, n);
        const uint32_t kpos_r = (kingside && kr == i) ? kc : 0xFFu, kpos_c = (kingside && kc == i) ? kr : 0xFFu;       // the king's place on the row / the column
        // items: bit p = piece at position p of the row, bit 16 + p = piece at position p of the column
        uint32_t items = line_bits(mine, base, n) | (col_bits(mine, i, n) << 16);
        uint32_t cnt = 0;
        while (items) {
            const uint32_t it = (uint32_t)__builtin_ctz(items);
            items &= items - 1u;
            const bool col = it >= 16u;
            const uint32_t p = it & 15u;
            const bool isk = p == (col ? kpos_c : kpos_r);
            const uint32_t lo = col ? co : ro, lo_m = col ? co_m : ro_m;
            const uint32_t lf = isk ? k_lf : s_lf, pf = isk ? k_pf : s_pf, lf_m = isk ? k_lf_m : s_lf_m, pf_m = isk ? k_pf_m : s_pf_m;
            uint32_t up = ray_up(lo, lf, pf, p, n), dn = ray_up(lo_m, lf_m, pf_m, nm - p, n);
            if (isk ? k_slow : s_slow) { up &= 1u; dn &= 1u; }
            const uint32_t both = up | (dn << (nm - p));                             // [+ distances 1..n-1-p][- distances 1..p]: n-1 bits
            if (both == 0u) continue;
            cnt += (uint32_t)__builtin_popcount(both);
            if (row) {
                // source tile (r, c) and the first slot of this axis inside its 2(n-1)-bit field: V at 0, H at n-1
                const uint32_t t_o = col ? mul24(p, n) + i : i * n + p;
                const uint32_t a0 = mul24(t_o, 2u * nm) + (col ? 0u : nm), w = a0 >> 5, sh2 = a0 & 31u;
                TAFL_MASK_OR(&row[w], both << sh2);
                if (sh2 != 0u && (both >> (32u - sh2)) != 0u) TAFL_MASK_OR(&row[w + 1u], both >> (32u - sh2));
            }
        }
        return cnt;
    }
    template <int I = 0> static TAFL_HD uint32_t movegen_line(const S& st, uint32_t i, const K& C, uint32_t* row) {
        if constexpr (I >= W) return 0;
        else return i == (uint32_t)I ? movegen_line_ct<I>(st, C, row) : movegen_line<I + 1>(st, i, C, row);
    }
    // count + dense action mask of one position (mask may be null; it must be zero-initialised by the caller)
    static TAFL_HD uint32_t movegen(const S& st, const K& C, uint32_t* mask) {
        if (!mask) { Moves<NL> mv; E::movegen(st, st.flags & TAFL_F_SIDE, C, mv); return mv.total; }       // counts alone: four reach sets, popcount
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < C.n; ++i) cnt += movegen_line(st, i, C, mask);
        return cnt;
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
    static constexpr uint32_t NO_ACTION = 0xFFFFFFFFu;
    // k-th (0-based) set bit of words [w0, w1) of the mask, k < their popcount; NO_ACTION if count and mask disagree (surfaced by the tests)
    static TAFL_HD uint32_t kth_set_bit(const uint32_t* mask, uint32_t w0, uint32_t w1, uint32_t k) {
        uint32_t a = NO_ACTION; bool found = false;
        for (uint32_t w = w0; w < w1; ++w) {
            const uint32_t v = mask[w], c = (uint32_t)__builtin_popcount(v);
            if (!found) { if (k < c) { a = w * 32u + nth_set_bit32(v, k); found = true; } else k -= c; }
        }
        return a;
    }
    static TAFL_HD void step_kth(S& st, uint32_t rank, const K& C, tafl_play* out_play, tafl_effects* eff, uint32_t* mask, uint32_t mask_words) {
        const uint32_t total = movegen(st, C, mask);
        step_action(st, total ? kth_set_bit(mask, 0, mask_words, rank % total) : NO_ACTION, total, C, out_play, eff);
    }
    // second half of step_kth: `action` = the chosen play as a dense action index (NO_ACTION: the side has no play), `total` its number of plays
    static TAFL_HD void step_action(S& st, uint32_t action, uint32_t total, const K& C, tafl_play* out_play, tafl_effects* eff) {
        tafl_effects e; caps_to_effects(bz<NL>(), 0, e);
        tafl_play pl; pl.from_row = pl.from_col = pl.axis = 0; pl.disp = 0;
        int code;
        if (total == 0) code = TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING ? TAFL_PLAY_GAME_OVER : TAFL_PLAY_NO_PIECE;
        else if (action != NO_ACTION) {
            const Move m = move_of_action(action, C);
            pl = to_play(m);
            StepOut<NL> so; Moves<NL> nx;
            E::apply(st, m, C, &so, nx);
            caps_to_effects(so.captures, so.n_captures, e);
            code = TAFL_PLAY_OK;
        } else code = TAFL_PLAY_NO_PIECE;         // count and mask disagree: surfaced by the tests
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
        M.node_top[g] = 1; M.edge_top[g] = 0; M.leaf[g] = 0; M.kind[g] = 0; M.fault[g] = 0;
        M.sim_next[g] = 0; M.sim_base[g] = 0; M.spec_pend[g] = (M.spec_k > 0 ? M.spec_k - 1 : 0u) << 16; M.spec_bias[g] = 0;
        for (uint32_t j = 0; j < M.spec_k; ++j) M.spec_kind[(size_t)j * M.G + g] = 0;
    }

    // Q / N update of one edge with the value v seen from the edge's owner (mcts.py:127-133)
    static TAFL_HD void edge_update(Edge& e, double v) {
        if (e.n > 0) { e.q = ((double)e.n * e.q + v) / (double)(e.n + 1); e.n += 1; }   // mcts.py:127-129
        else { e.q = v; e.n = 1; }                                                      // mcts.py:131-133
    }

    // ---- one tree step keeps what it touches again and again in registers ------------------------------------------------------------
    // The tree phase is a chain of dependent memory round trips (one wave per 64 games, nothing to hide them behind), so a value that is
    // already in a register is never fetched again: the per-game counters, the root header and the sign bits of the root's Qsa live in
    // StepCtx for the whole step (write-through: memory is always current), a simulation hands its leaf, parent and edge to its backup
    // (SimOut) instead of passing them through M.leaf / M.kind, and the slot a simulation may consume is fetched beside its selection
    // (SlotView).  Same arithmetic in the same order as before: results are unchanged.
    struct RootCache {
        uint32_t hw[8];              // header of node 0 as eight plain words (scalars: the cache must stay in registers)
        uint32_t pos[4];             // bit e: Qsa of root edge e (e < 128) is > 0 (puct_pick's shortcut only looks at those)
        bool pos_valid;
    };
    static_assert(sizeof(NodeHdr) == 32, "NodeHdr is eight words");
    static_assert(sizeof(Edge) == 16, "Edge is four words");
    struct StepCtx { uint32_t node_top, edge_top; RootCache rc; };
    struct SimOut {
        uint32_t leaf; uint8_t kind, term;    // kind: 0 fault, 1 playout needed, 2 terminal
        bool fresh;                  // the leaf was created by this simulation: its edge is new (Nsa = 0) and the fields below are valid
        uint32_t parent, pslot, eidx;
        NodeHdr ph;                  // the parent's header after the expansion
    };

    // (values, never addresses, are selected between the cache and memory: a pointer that may be private or global would put the
    // cache into scratch memory)
    static TAFL_HD NodeHdr hdr_get(const MctsMem& M, uint32_t g, uint32_t node, const StepCtx& X) {
        uint32_t w[8];
        TAFL_UNROLL for (int i = 0; i < 8; ++i) w[i] = X.rc.hw[i];
        if (node != 0) { const NodeHdr t = M.hdr[(size_t)node * M.G + g]; __builtin_memcpy(w, &t, sizeof t); }
        NodeHdr h; __builtin_memcpy(&h, w, sizeof h);
        return h;
    }
    static TAFL_HD void hdr_put(const MctsMem& M, uint32_t g, uint32_t node, const NodeHdr& h, StepCtx& X) {
        uint32_t w[8]; __builtin_memcpy(w, &h, sizeof h);
        M.hdr[(size_t)node * M.G + g] = h;
        TAFL_UNROLL for (int i = 0; i < 8; ++i) X.rc.hw[i] = node == 0 ? w[i] : X.rc.hw[i];
    }
    static TAFL_HD void root_load(const MctsMem& M, uint32_t g, StepCtx& X) {
        const NodeHdr t = M.hdr[g];
        uint32_t w[8]; __builtin_memcpy(w, &t, sizeof t);
        TAFL_UNROLL for (int i = 0; i < 8; ++i) X.rc.hw[i] = w[i];
    }
    // edge `slot` of node `node` now has Qsa q
    static TAFL_HD void pos_note(StepCtx& X, uint32_t node, uint32_t slot, double q) {
        if (node != 0 || !X.rc.pos_valid) return;
        if (slot >= 128u) { X.rc.pos_valid = false; return; }
        const uint32_t b = 1u << (slot & 31u), wi = slot >> 5;
        TAFL_UNROLL for (uint32_t i = 0; i < 4; ++i) { const uint32_t w = X.rc.pos[i]; X.rc.pos[i] = i == wi ? (q > 0.0 ? (w | b) : (w & ~b)) : w; }
    }

    // backup of one simulation (mcts.py:127-136 unwound iteratively); value: the playout's result for kind 1
    static TAFL_HD void mcts_backup(const MctsMem& M, uint32_t g, const SimOut& o, int value, StepCtx& X) {
        if (o.kind == 0) return;
        uint32_t cur = o.leaf;
        double v;
        if (o.kind == 1) {                                      // leaf was expanded by a playout: return -v (mcts.py:100-102)
            const uint32_t ns0 = (M.flags & TAFL_MCTS_FLAG_FPU_INF) ? 1u : 0u;                     // mcts.py:100-101 / mcts.rs:187
            if (cur == 0) { NodeHdr rh = hdr_get(M, g, 0, X); rh.expanded = 1; rh.ns = ns0; hdr_put(M, g, 0, rh, X); }
            else { NodeHdr* lh = &M.hdr[(size_t)cur * M.G + g]; lh->expanded = 1; lh->ns = ns0; }
            v = -(double)value;
        } else {
            v = -term_value(o.term);                            // terminal: return -Es[s] (mcts.py:79-81)
        }
        if (cur != 0 && o.fresh) {                              // the new edge: Nsa = 0, so Qsa = v, Nsa = 1 (mcts.py:131-133)
            Edge e; e.q = v; e.n = 1; e.child = cur;
            M.edges[(size_t)g * M.edge_cap + o.eidx] = e;
            pos_note(X, o.parent, o.pslot, v);
            NodeHdr ph = o.ph; ph.ns += 1;                                                          // mcts.py:135
            hdr_put(M, g, o.parent, ph, X);
            v = -v; cur = o.parent;                                                                 // mcts.py:136
        }
        while (cur != 0) {
            const NodeHdr ch = M.hdr[(size_t)cur * M.G + g];
            NodeHdr ph = hdr_get(M, g, ch.parent, X);
            Edge* ep = &M.edges[(size_t)g * M.edge_cap + ph.edge_base + ch.pslot];
            Edge e = *ep; edge_update(e, v); *ep = e;
            pos_note(X, ch.parent, ch.pslot, e.q);
            ph.ns += 1;                                                                             // mcts.py:135
            hdr_put(M, g, ch.parent, ph, X);
            v = -v;                                                                                 // mcts.py:136
            cur = ch.parent;
        }
    }

    // the action with the highest upper confidence bound (mcts.py:104-119) among the visited children (a prefix of the canonical
    // legal list) and the next unvisited one: returns its index in [0, h.m] (h.m = expand the next unvisited child), -1 if none.
    static TAFL_HD int puct_pick(const MctsMem& M, uint32_t g, uint32_t node, const NodeHdr& h, double c_puct, StepCtx& X) {
        // first-play urgency of src/mcts.rs:49-51: an unvisited action scores +infinity, the lowest index among them wins
        if ((M.flags & TAFL_MCTS_FLAG_FPU_INF) && h.m < h.n_legal) return (int)h.m;
        const double p = 1.0 / (double)h.n_legal;
        const double cp = c_puct * p;
        const double sq = sqrt((double)h.ns);
        double cur_best = -__builtin_inf(); int best = -1;
        const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
        if (h.m < h.n_legal && c_puct > 0.0) {
            // While an unvisited action exists it scores u0 = cp * sqrt(Ns + EPS) > cp * sqrt(Ns) / 2 >= cp * sqrt(Ns) / (1 + Nsa) for every
            // visited one (Nsa >= 1; strict only for cp > 0: at c_puct == 0 a visited action with Qsa == 0 ties u0 = 0 and, having the lower
            // index, wins - that case takes the full scan below), so a visited action with Qsa <= 0 can neither win nor tie: only the (few) actions with Qsa > 0 are
            // evaluated, in ascending order as mcts.py does - the float64 division is the expensive part of this loop and 64 games share
            // an instruction stream (-24 % VALU instructions in k_mcts_tree).  Same argmax, same tie rule.
            // At the root the sign bits come from the step's cache (scanned once per step, kept current by every backup).
            const double u0 = cp * sqrt((double)h.ns + TAFL_MCTS_EPS);
            const bool cacheable = node == 0 && h.m <= 128u;
            const bool cached = cacheable && X.rc.pos_valid;
            uint32_t seen[4] = {0u, 0u, 0u, 0u};
            for (uint32_t j0 = 0; j0 < h.m; j0 += 32) {
                uint32_t pos = 0;
                const uint32_t lim = h.m - j0 < 32u ? h.m - j0 : 32u;
                if (cached) {
                    const uint32_t wi = j0 >> 5;
                    pos = wi == 0 ? X.rc.pos[0] : wi == 1 ? X.rc.pos[1] : wi == 2 ? X.rc.pos[2] : X.rc.pos[3];
                } else {
                    // eight independent loads in flight per step (one load and one wait per edge made this scan the longest dependent
                    // chain of the tree phase: every wait is a trip to L2 / HBM)
                    for (uint32_t t0 = 0; t0 < lim; t0 += 8) {       // (named values, not an array: the array went to scratch memory)
#define TAFL_QLOAD(t) const double q##t = eb[j0 + (t0 + t < lim ? t0 + t : lim - 1u)].q
#define TAFL_QBIT(t) (((t0 + t < lim && q##t > 0.0) ? 1u : 0u) << (t0 + t))
                        TAFL_QLOAD(0u); TAFL_QLOAD(1u); TAFL_QLOAD(2u); TAFL_QLOAD(3u); TAFL_QLOAD(4u); TAFL_QLOAD(5u); TAFL_QLOAD(6u); TAFL_QLOAD(7u);
                        pos |= TAFL_QBIT(0u) | TAFL_QBIT(1u) | TAFL_QBIT(2u) | TAFL_QBIT(3u) | TAFL_QBIT(4u) | TAFL_QBIT(5u) | TAFL_QBIT(6u) | TAFL_QBIT(7u);
#undef TAFL_QLOAD
#undef TAFL_QBIT
                    }
                    if (cacheable) { const uint32_t wi = j0 >> 5; TAFL_UNROLL for (uint32_t i = 0; i < 4; ++i) seen[i] = i == wi ? pos : seen[i]; }
                }
                while (pos) {
                    const uint32_t t = (uint32_t)__builtin_ctz(pos);
                    pos &= pos - 1u;
                    const Edge e = eb[j0 + t];
                    const double u = e.q + cp * sq / (double)(1 + e.n);
                    if (u > cur_best) { cur_best = u; best = (int)(j0 + t); }
                }
            }
            if (cacheable && !cached) { TAFL_UNROLL for (uint32_t i = 0; i < 4; ++i) X.rc.pos[i] = seen[i]; X.rc.pos_valid = true; }
            if (u0 > cur_best) { cur_best = u0; best = (int)h.m; }
            return best;
        }
        // the edge records are fetched four at a time (independent loads in flight: this loop is bound by memory latency; eight at a time
        // cost the tree kernel 32 VGPRs it does not have: it spilled),
        // then evaluated in ascending order as mcts.py does
        for (uint32_t j0 = 0; j0 < h.m; j0 += 4) {
            Edge e[4];
            TAFL_UNROLL for (uint32_t t = 0; t < 4; ++t) e[t] = eb[(j0 + t < h.m) ? j0 + t : j0];
            TAFL_UNROLL for (uint32_t t = 0; t < 4; ++t) {
                const double u = e[t].q + cp * sq / (double)(1 + e[t].n);
                if (j0 + t < h.m && u > cur_best) { cur_best = u; best = (int)(j0 + t); }
            }
        }
        if (h.m < h.n_legal) {
            const double u0 = cp * sqrt((double)h.ns + TAFL_MCTS_EPS);
            if (u0 > cur_best) { cur_best = u0; best = (int)h.m; }
        }
        return best;
    }

    // edge array of a node moved to a larger allocation: four records in flight (a load, a wait and a store per record is one memory
    // round trip per record, and the whole wave waits for the game with the longest array)
    static TAFL_HD void copy_edges(Edge* dst, const Edge* src, uint32_t n) {
        for (uint32_t j = 0; j < n; j += 4) {          // (named values: an array of four records behind conditional stores cost 16 B more scratch)
            const Edge e0 = src[j], e1 = src[j + 1 < n ? j + 1 : n - 1u], e2 = src[j + 2 < n ? j + 2 : n - 1u], e3 = src[j + 3 < n ? j + 3 : n - 1u];
            dst[j] = e0;
            if (j + 1 < n) dst[j + 1] = e1;
            if (j + 2 < n) dst[j + 2] = e2;
            if (j + 3 < n) dst[j + 3] = e3;
        }
    }

    // selection of one simulation (mcts.py:77-119): walks down to a node that ends the simulation (o.kind 0 fault, 1 unexpanded node,
    // 2 terminal node) or to the node whose next unvisited child has to be created (o.kind 4: o.parent = that node, o.ph = its header,
    // o.pslot = the child's ordinal)
    static TAFL_HD void mcts_select(const MctsMem& M, uint32_t g, double c_puct, LaneStats& ls, StepCtx& X, SimOut& o) {
        uint32_t cur = 0;
        ls.sims += 1;
        o.fresh = false; o.term = 0; o.parent = 0; o.pslot = 0; o.eidx = 0;
        for (uint32_t depth = 0; depth < M.node_cap + 1; ++depth) {
            const NodeHdr h = hdr_get(M, g, cur, X);
            if (h.term) { o.leaf = cur; o.kind = 2; o.term = h.term; ls.terminal_hits += 1; return; }
            if (!h.expanded) { o.leaf = cur; o.kind = 1; return; }
            ls.depth += 1;
            const int best = puct_pick(M, g, cur, h, c_puct, X);
            ls.scanned += h.m;
            if (best < 0) { M.fault[g] = 1; ls.faults += 1; o.leaf = cur; o.kind = 0; return; }
            if ((uint32_t)best < h.m) { cur = M.edges[(size_t)g * M.edge_cap + h.edge_base + (uint32_t)best].child; continue; }
            o.leaf = cur; o.kind = 4; o.parent = cur; o.pslot = h.m; o.ph = h;
            return;
        }
        M.fault[g] = 1; ls.faults += 1; o.leaf = 0; o.kind = 0;
    }
    // child state of the expansion mcts_select asked for, computed from the parent's state: getNextState (mcts.py:122-123)
    static TAFL_HD bool mcts_child_state(const MctsMem& M, uint32_t g, const SimOut& o, const K& C, S& st, Move& mv, Moves<NL>& nx) {
        IO::load_rec(M.node_state + ((size_t)o.parent * M.G + g) * IO::QUADS, st);
        mv.from = o.ph.cur_from; mv.to = 0; mv.dir = o.ph.cur_dir; mv.dist = o.ph.cur_dist;
        if (!E::canon_next(st, st.flags & TAFL_F_SIDE, C, mv)) return false;
        E::apply(st, mv, C, nullptr, nx);
        return true;
    }
    // creates the child (state st reached by play mv, nx_total legal plays) the selection asked for: o becomes the simulation's leaf
    static TAFL_HD void mcts_expand(const MctsMem& M, uint32_t g, LaneStats& ls, StepCtx& X, SimOut& o, const S& st, const Move& mv, uint32_t nx_total) {
        const uint32_t cur = o.parent;
        NodeHdr h = o.ph;
        const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
        const uint32_t id = X.node_top;
        uint32_t base = h.edge_base; uint32_t cap = h.cap;
        if (h.m == cap) {                                      // grow the edge array (amortised doubling)
            const uint32_t ncap = cap ? cap * 2u : 4u;
            const uint32_t nbase = X.edge_top;
            if (id >= M.node_cap || nbase + ncap > M.edge_cap) { M.fault[g] = 1; ls.faults += 1; o.leaf = cur; o.kind = 0; return; }
            copy_edges(&M.edges[(size_t)g * M.edge_cap + nbase], eb, h.m);
            X.edge_top = nbase + ncap; base = nbase; cap = ncap;
        } else if (id >= M.node_cap) { M.fault[g] = 1; ls.faults += 1; o.leaf = cur; o.kind = 0; return; }
        Edge ne; ne.q = 0.0; ne.n = 0; ne.child = id;
        M.edges[(size_t)g * M.edge_cap + base + h.m] = ne;
        const uint32_t slot = h.m;
        h.edge_base = base; h.cap = (uint16_t)cap; h.m = (uint16_t)(slot + 1);
        h.cur_from = (uint16_t)mv.from; h.cur_dir = (uint8_t)mv.dir; h.cur_dist = (uint8_t)mv.dist;
        hdr_put(M, g, cur, h, X);
        pos_note(X, cur, slot, 0.0);
        NodeHdr nh; nh.parent = cur; nh.edge_base = 0; nh.ns = 0; nh.pslot = (uint16_t)slot; nh.m = 0; nh.n_legal = (uint16_t)nx_total; nh.cap = 0;
        nh.mv_from = (uint16_t)mv.from; nh.mv_dir = (uint8_t)mv.dir; nh.mv_dist = (uint8_t)mv.dist;
        nh.cur_from = (uint16_t)TAFL_NO_SQ; nh.cur_dir = 0; nh.cur_dist = 0;
        nh.term = term_code(st); nh.expanded = 0; nh._pad[0] = nh._pad[1] = 0;
        M.hdr[(size_t)id * M.G + g] = nh;
        IO::store_rec(M.node_state + ((size_t)id * M.G + g) * IO::QUADS, st);
        X.node_top = id + 1;
        o.leaf = id; o.term = nh.term; o.fresh = true; o.pslot = slot; o.eidx = base + slot; o.ph = h;
        if (nh.term) { o.kind = 2; ls.terminal_hits += 1; } else o.kind = 1;
    }

    // ---- simulation pipeline ----------------------------------------------------------------------------------------
    // Simulations of one game are sequential; the pipeline runs the playouts of several FUTURE simulations of a game beside the
    // pending one.  When a simulation has to wait for its playout, mcts_speculate predicts the expansions of the following
    // simulations: it assumes a value for every playout in flight (and takes the REAL value of every leaf whose playout has already
    // run), backs it up IN PLACE (undo log), runs the same PUCT selection the real simulation will run, notes which child of which node
    // that selection expands (a slot: node, ordinal, prepared leaf state) and finally restores the tree bit for bit.  The playouts of
    // all requested slots run in the same round.  Every simulation later performs its REAL selection on the committed tree and takes a
    // slot's value if a slot holds exactly the child it expands (node, ordinal): same leaf state, hence - the playout being keyed by
    // the leaf's position - the value a fresh playout would return.  A prediction that does not come true at once is not lost: its slot
    // waits until the search comes by that child.  Predictions therefore change timing only, never results.
    static constexpr uint32_t VIRT_CHILD = 0xFFFFFFFFu;       // child id of an edge that exists only during the speculation pass
    static constexpr uint32_t ORD_SELF = 0xFFFu;              // ordinal of a slot whose leaf is the node itself
    static constexpr uint32_t NO_SLOT = 0xFFu;
    static TAFL_HD uint32_t slot_ref(uint32_t node, uint32_t ord) { return node | (ord << 20); }
    // increment of MctsMem::spec_bias for a playout of value v (seen from the mover of a leaf whose state has these flags)
    static TAFL_HD uint32_t bias_of(uint32_t leaf_flags, int v) {
        const bool def_mover = (leaf_flags & TAFL_F_SIDE) != 0u;
        return v == 0 ? 0u : ((v > 0) == def_mover ? 1u : 0x10000u);
    }

    // the game's slot table in registers (named scalars behind unrolled selects: an array indexed at run time would live in scratch)
    struct Pool { uint32_t ref[kMctsMaxSlots]; uint8_t kind[kMctsMaxSlots]; };   // kind 0xFF: the slot does not exist (j >= spec_k)
    static TAFL_HD void pool_load(const MctsMem& M, uint32_t g, Pool& P) {
        TAFL_UNROLL for (uint32_t j = 0; j < kMctsMaxSlots; ++j) {
            const size_t o = (size_t)(j < M.spec_k ? j : 0u) * M.G + g;
            P.ref[j] = M.spec_ref[o]; P.kind[j] = j < M.spec_k ? M.spec_kind[o] : (uint8_t)0xFF;
        }
    }
    static TAFL_HD uint32_t pool_find(const Pool& P, uint32_t ref) {
        uint32_t f = NO_SLOT;
        TAFL_UNROLL for (uint32_t j = 0; j < kMctsMaxSlots; ++j) f = (P.kind[j] >= 1 && P.kind[j] <= 3 && P.ref[j] == ref) ? j : f;
        return f;
    }
    static TAFL_HD uint8_t pool_kind(const Pool& P, uint32_t f) {
        uint8_t k = 0;
        TAFL_UNROLL for (uint32_t j = 0; j < kMctsMaxSlots; ++j) k = j == f ? P.kind[j] : k;
        return k;
    }
    static TAFL_HD void pool_set(Pool& P, uint32_t f, uint8_t kind, uint32_t ref) {
        TAFL_UNROLL for (uint32_t j = 0; j < kMctsMaxSlots; ++j) { P.kind[j] = j == f ? kind : P.kind[j]; P.ref[j] = j == f ? ref : P.ref[j]; }
    }
    // a slot for a new leaf: a free one; else, outside `keep`, a terminal child (cheap to find again), a requested playout that has not run,
    // a value that is ready - in this order, the lowest index among equals.  NO_SLOT: none.
    static TAFL_HD uint32_t pool_alloc(const Pool& P, uint32_t keep) {
        uint32_t f = NO_SLOT, best = 0;
        TAFL_UNROLL for (uint32_t jj = 0; jj < kMctsMaxSlots; ++jj) {
            const uint32_t j = kMctsMaxSlots - 1u - jj;
            const uint32_t k = P.kind[j];
            const uint32_t score = k == 0u ? 4u : k == 3u ? 3u : k == 1u ? 2u : k == 2u ? 1u : 0u;
            const bool ok = score > 0u && !((keep >> j) & 1u) && score >= best;
            f = ok ? j : f; best = ok ? score : best;
        }
        return f;
    }

    struct SpecLog {                                          // undo log of one speculation pass (LogMem: the caller's scratch)
        uint32_t* eb; uint32_t* hb; uint32_t stride, ne, nh, cap; bool ok;
    };
    static TAFL_HD void log_edge(SpecLog& L, uint32_t eidx, const Edge& old) {
        if (L.ne >= L.cap) { L.ok = false; return; }
        uint32_t w[4]; __builtin_memcpy(w, &old, sizeof old);
        uint32_t* r = L.eb + (size_t)L.ne * kUndoEWords * L.stride;
        r[0] = eidx; TAFL_UNROLL for (uint32_t k = 0; k < 4; ++k) r[(k + 1u) * L.stride] = w[k];
        L.ne += 1;
    }
    static TAFL_HD void log_hdr(SpecLog& L, uint32_t node, const NodeHdr& old) {
        if (L.nh >= L.cap) { L.ok = false; return; }
        uint32_t w[8]; __builtin_memcpy(w, &old, sizeof old);
        uint32_t* r = L.hb + (size_t)L.nh * kUndoHWords * L.stride;
        r[0] = node; TAFL_UNROLL for (uint32_t k = 0; k < 8; ++k) r[(k + 1u) * L.stride] = w[k];
        L.nh += 1;
    }
    // assumed backup: edge `eidx` of node `cur` receives v, then the path to the root as in mcts_backup; every touched record is logged.
    // new_edge: the edge was created by this pass (Qsa = 0, Nsa = 0, virtual child): nothing to fetch.
    static TAFL_HD void spec_backup(const MctsMem& M, uint32_t g, SpecLog& L, uint32_t cur, uint32_t eidx, double v, bool new_edge, StepCtx& X) {
        for (uint32_t guard = 0; guard < M.node_cap + 1 && L.ok; ++guard) {
            NodeHdr ph = hdr_get(M, g, cur, X);
            Edge* ep = &M.edges[(size_t)g * M.edge_cap + eidx];
            Edge e;
            if (new_edge) { e.q = 0.0; e.n = 0; e.child = VIRT_CHILD; } else e = *ep;
            log_edge(L, eidx, e); log_hdr(L, cur, ph);
            if (!L.ok) return;
            edge_update(e, v); *ep = e;
            pos_note(X, cur, eidx - ph.edge_base, e.q);
            ph.ns += 1;
            hdr_put(M, g, cur, ph, X);
            if (cur == 0) return;
            const uint32_t parent = ph.parent, pslot = ph.pslot;
            eidx = hdr_get(M, g, parent, X).edge_base + pslot;
            cur = parent; v = -v; new_edge = false;
        }
    }
    // Predicts the expansions of the simulations that follow simulation `first`, whose real leaf `leaf` waits for its playout.
    // want: playouts this game may have requested in this round, the pending one included.  P: the slot table (kept current), keep: slots
    // the pass has used (bit j), ncls: next free priority class.  Assumed value of a playout in flight: 0.
    static TAFL_HD void mcts_speculate(const MctsMem& M, uint32_t g, uint32_t leaf, uint32_t first, uint32_t want, double c_puct,
                                       uint32_t n_sims, const K& C, LaneStats& ls, StepCtx& X, const LogMem& lm, Pool& P, uint32_t& keep, uint32_t& ncls, uint32_t& req, double a_pend) {
        SpecLog L; L.eb = lm.base + lm.lane; L.hb = lm.base + (size_t)lm.cap * kUndoEWords * lm.stride + lm.lane; L.stride = lm.stride; L.ne = L.nh = 0; L.cap = lm.cap; L.ok = true;
        uint32_t vtop = X.edge_top;                                // edge arrays that grow during the pass take free arena space, not committed
        // the pending leaf as it will be once its playout value arrives: expanded, its path updated with the assumed value
        {
            NodeHdr lh = hdr_get(M, g, leaf, X);
            log_hdr(L, leaf, lh);
            if (L.ok) { lh.expanded = 1; lh.ns = (M.flags & TAFL_MCTS_FLAG_FPU_INF) ? 1u : 0u; hdr_put(M, g, leaf, lh, X); }
            if (leaf != 0 && L.ok) {
                const uint32_t parent = lh.parent;
                spec_backup(M, g, L, parent, hdr_get(M, g, parent, X).edge_base + lh.pslot, -a_pend, false, X);
            }
        }
        for (uint32_t t = 1; t < kMctsMaxPass && first + t < n_sims && L.ok; ++t) {
            // the selection walks down in a loop of its own: games of a wave stop at different depths, and the expansion below (the
            // expensive part) must run once for all of them, not once per depth
            uint32_t cur = 0; bool stop = false, placed = false, expand = false;
            NodeHdr h;
            for (uint32_t depth = 0; depth < M.node_cap + 1; ++depth) {
                h = hdr_get(M, g, cur, X);
                if (h.term) {                                     // the simulation ends on a terminal node: its value is exact, no playout
                    if (cur == 0) { stop = true; break; }
                    spec_backup(M, g, L, h.parent, hdr_get(M, g, h.parent, X).edge_base + h.pslot, -term_value(h.term), false, X);
                    placed = true;
                    break;
                }
                if (!h.expanded) { stop = true; break; }
                const int best = puct_pick(M, g, cur, h, c_puct, X);
                if (best < 0) { stop = true; break; }
                if ((uint32_t)best < h.m) {
                    const uint32_t child = M.edges[(size_t)g * M.edge_cap + h.edge_base + (uint32_t)best].child;
                    if (child == VIRT_CHILD) { stop = true; break; }   // would descend into a leaf that exists only as a slot
                    cur = child; continue;
                }
                expand = true;
                break;
            }
            if (expand) {
                // predicted expansion: child number h.m of node cur
                const uint32_t ref = slot_ref(cur, h.m);
                uint32_t f = pool_find(P, ref);
                const bool known = f != NO_SLOT;
                const uint8_t fk = pool_kind(P, f);
                bool ok = h.m < ORD_SELF;                         // (the ordinal must fit the slot reference)
                if ((!known || (fk == 1 && !((keep >> f) & 1u))) && req >= want) ok = false;      // this round's share of playouts is used up
                if (ok && !known) { f = pool_alloc(P, keep); ok = f != NO_SLOT; }
                const size_t so = (size_t)(ok ? f : 0u) * M.G + g;
                S cst{}; Move mv; Moves<NL> nx; nx.total = 0; uint8_t tc = 0; double val = -0.0;
                if (ok && known) {                                // the leaf is in a slot already: its play from there, its value if it has one
                    const uint32_t meta = M.spec_meta[so]; const int8_t v = M.spec_value[so];
                    mv.from = meta & 0xFFu; mv.dir = (meta >> 8) & 3u; mv.dist = (meta >> 10) & 0x3Fu; mv.to = 0;
                    if (fk == 2) val = -(double)v; else if (fk == 3) val = -term_value((uint8_t)v);
                } else if (ok) {
                    IO::load_rec(M.node_state + ((size_t)cur * M.G + g) * IO::QUADS, cst);
                    mv.from = h.cur_from; mv.to = 0; mv.dir = h.cur_dir; mv.dist = h.cur_dist;
                    ok = E::canon_next(cst, cst.flags & TAFL_F_SIDE, C, mv);
                    if (ok) { E::apply(cst, mv, C, nullptr, nx); tc = term_code(cst); if (tc) val = -term_value(tc); }
                }
                if (ok) { log_hdr(L, cur, h); ok = L.ok; }
                uint32_t base = h.edge_base, cap = h.cap;
                if (ok && h.m == cap) {                           // uncommitted growth into free arena space
                    const uint32_t ncap = cap ? cap * 2u : 4u;
                    if (vtop + ncap > M.edge_cap) ok = false;
                    else {
                        copy_edges(&M.edges[(size_t)g * M.edge_cap + vtop], &M.edges[(size_t)g * M.edge_cap + base], h.m);
                        base = vtop; cap = ncap; vtop += ncap;
                    }
                }
                if (!ok) stop = true;
                else {
                    Edge ne; ne.q = 0.0; ne.n = 0; ne.child = VIRT_CHILD;
                    M.edges[(size_t)g * M.edge_cap + base + h.m] = ne;
                    const uint32_t slot = h.m;
                    h.edge_base = base; h.cap = (uint16_t)cap; h.m = (uint16_t)(slot + 1);
                    h.cur_from = (uint16_t)mv.from; h.cur_dir = (uint8_t)mv.dir; h.cur_dist = (uint8_t)mv.dist;
                    hdr_put(M, g, cur, h, X);
                    pos_note(X, cur, slot, 0.0);
                    if (!known) {
                        IO::store_rec(M.spec_state + so * IO::QUADS, cst);
                        M.spec_meta[so] = mv.from | (mv.dir << 8) | (mv.dist << 10) | (nx.total << 16);
                        M.spec_ref[so] = ref; M.spec_kind[so] = tc ? 3 : 1; M.spec_value[so] = (int8_t)tc;
                        pool_set(P, f, tc ? 3 : 1, ref);
                        if (!tc) ls.spec_issued += 1;
                    }
                    if ((known ? fk == 1 : !tc) && !((keep >> f) & 1u)) { M.spec_cls[so] = (uint8_t)ncls; ++ncls; ++req; }
                    keep |= 1u << f;
                    placed = true;
                    spec_backup(M, g, L, cur, base + slot, val, true, X);
                }
            }
            if (stop || !placed) break;
        }
        // restore the tree (reverse order: a record may have been logged more than once)
        for (uint32_t i = L.ne; i > 0; --i) {
            const uint32_t* r = L.eb + (size_t)(i - 1u) * kUndoEWords * L.stride;
            uint32_t w[4]; TAFL_UNROLL for (uint32_t k = 0; k < 4; ++k) w[k] = r[(k + 1u) * L.stride];
            Edge e; __builtin_memcpy(&e, w, sizeof e);
            M.edges[(size_t)g * M.edge_cap + r[0]] = e;
        }
        for (uint32_t i = L.nh; i > 0; --i) {
            const uint32_t* r = L.hb + (size_t)(i - 1u) * kUndoHWords * L.stride;
            uint32_t w[8]; TAFL_UNROLL for (uint32_t k = 0; k < 8; ++k) w[k] = r[(k + 1u) * L.stride];
            NodeHdr h; __builtin_memcpy(&h, w, sizeof h);
            M.hdr[(size_t)r[0] * M.G + g] = h;
        }
        // (the step's register copy of the root - X.rc - still shows the pass's last state: the pass is the last thing a step does with it)
    }

    // One call advances game g by as many simulations as it can without waiting for a playout.
    // rounds_left: rounds the host still plans for this search (0: no plan): a game requests ceil(remaining / rounds_left) playouts, so
    // that a game that lost a round catches up instead of trailing.  scen: scenario passes of the prediction (1 or 2, mcts_scenarios).
    // wcap: most predicted playouts a game may request beside the pending one (lowered while few predictions come true: a prediction
    // costs a child expansion in the tree phase whether it is consumed or not).
    static TAFL_HD void mcts_tree_step(const MctsMem& M, uint32_t g, double c_puct, uint32_t n_sims, uint32_t rounds_left, uint32_t scen, uint32_t wcap, const K& C, LaneStats& ls, const LogMem& lm) {
        // everything the step needs from the per-game arrays, fetched side by side
        uint32_t sim = M.sim_next[g];
        const uint8_t kind0 = M.kind[g];
        const uint32_t leaf0 = M.leaf[g];
        const uint32_t pw = M.spec_pend[g];
        const size_t po = (size_t)(pw & 0xFFu) * M.G + g;
        StepCtx X; X.node_top = M.node_top[g]; X.edge_top = M.edge_top[g];
        root_load(M, g, X); X.rc.pos_valid = false; TAFL_UNROLL for (uint32_t i = 0; i < 4; ++i) X.rc.pos[i] = 0;
        const uint8_t sk0 = M.spec_kind[po], sr0 = M.spec_reason[po]; const int8_t sv0 = M.spec_value[po]; const uint32_t sp0 = M.spec_plies[po];
        const uint32_t sf0 = M.spec_state[po * IO::QUADS + (IO::QUADS - 1)].w;      // flags word of the pending leaf's state (its mover)
        uint32_t bias = M.spec_bias[g];
        Pool P; pool_load(M, g, P);                              // the slot table: registers for the whole step, memory kept current
        if (kind0 == 1) {                                        // the pending leaf of the previous call
            if (sk0 != 2) return;                                // its playout has not run yet (the round was full): the slots stay requested
            ls.rollouts += 1; ls.rollout_plies += sp0; ls.reason_hist4 += 1ull << (4u * (sr0 & 15u));
            SimOut o; o.leaf = leaf0; o.kind = 1; o.term = 0; o.fresh = false; o.parent = 0; o.pslot = 0; o.eidx = 0;
            mcts_backup(M, g, o, (int)sv0, X); ++sim;
            bias += bias_of(sf0, sv0);
            M.spec_kind[po] = 0; pool_set(P, pw & 0xFFu, 0, 0u);
        }
        bool pending = false; uint32_t pend_leaf = leaf0, pend_slot = NO_SLOT, hits = 0;
        // The games of a wave take different numbers of turns through the inner loop, so it holds only what is cheap (selection, an
        // expansion whose state a slot has prepared, backup).  An expansion that needs canon_next + apply (no slot holds that child)
        // leaves it and is computed behind it, once per wave; the game then waits for that leaf's playout, and only a terminal
        // child sends it round again.
        for (;;) {
            bool deferred = false;
            SimOut o; o.parent = 0; o.pslot = 0;
            for (;;) {
                if (sim >= n_sims) break;
                mcts_select(M, g, c_puct, ls, X, o);
                uint32_t f = NO_SLOT; uint8_t fk = 0, fr = 0; int8_t fv = 0; uint32_t fp = 0, fflags = 0;
                if (o.kind == 4) {
                    // a slot holds exactly this child (same node, same ordinal): state, play and legal-play count are there, no canon_next / apply
                    f = o.pslot < ORD_SELF ? pool_find(P, slot_ref(o.parent, o.pslot)) : NO_SLOT;       // (an ordinal beyond the reference's 12 bits: no slot ever holds it)
                    if (f == NO_SLOT) { deferred = true; break; }
                    fk = pool_kind(P, f);
                    const size_t so = (size_t)f * M.G + g;
                    const uint32_t meta = M.spec_meta[so]; fv = M.spec_value[so]; fp = M.spec_plies[so]; fr = M.spec_reason[so];
                    S st; IO::load_rec(M.spec_state + so * IO::QUADS, st);
                    fflags = st.flags;
                    Move mv; mv.from = meta & 0xFFu; mv.dir = (meta >> 8) & 3u; mv.dist = (meta >> 10) & 0x3Fu; mv.to = 0;
                    mcts_expand(M, g, ls, X, o, st, mv, meta >> 16);
                }
                const size_t fo = (size_t)(f != NO_SLOT ? f : 0u) * M.G + g;
                if (o.kind == 0) { if (f != NO_SLOT) { M.spec_kind[fo] = 0; pool_set(P, f, 0, 0u); } ++sim; continue; }                                    // fault (flagged per game): nothing to back up
                if (o.kind == 2) { if (f != NO_SLOT) { M.spec_kind[fo] = 0; pool_set(P, f, 0, 0u); } mcts_backup(M, g, o, 0, X); ++sim; continue; }       // terminal node: value known at once
                if (f != NO_SLOT && fk == 2) {                                          // its playout has run: the value a fresh playout would return
                    ls.rollouts += 1; ls.rollout_plies += fp; ls.reason_hist4 += 1ull << (4u * (fr & 15u));
                    ls.spec_hits += 1; ++hits; bias += bias_of(fflags, fv);
                    M.spec_kind[fo] = 0; pool_set(P, f, 0, 0u);
                    mcts_backup(M, g, o, (int)fv, X); ++sim;
                    continue;
                }
                pending = true; pend_leaf = o.leaf; pend_slot = f;                      // (f: requested, not run yet - that slot is now the pending one)
                break;                                                                  // wait for the playouts
            }
            if (!deferred) break;
            S st; Move mv; Moves<NL> nx;
            if (!mcts_child_state(M, g, o, C, st, mv, nx)) { M.fault[g] = 1; ls.faults += 1; ++sim; continue; }
            mcts_expand(M, g, ls, X, o, st, mv, nx.total);
            if (o.kind == 1) { pending = true; pend_leaf = o.leaf; pend_slot = NO_SLOT; break; }
            if (o.kind == 2) mcts_backup(M, g, o, 0, X);
            ++sim;
        }
        // Request playouts: the waiting leaf's and those of the predicted expansions.  This sits BEHIND the loop on purpose: the games of a
        // wave leave the loop after different numbers of consumed slots, and inside the loop the wave would run the prediction pass (the
        // most expensive code of the step) once per distinct count instead of once.
        if (pending) {
            const uint32_t L = pend_leaf;
            if (pend_slot == NO_SLOT) {                          // the leaf was not in a slot: it takes one (nothing is kept: there always is one)
                pend_slot = pool_alloc(P, 0u);
                const size_t so = (size_t)pend_slot * M.G + g;
                S lst; IO::load_rec(M.node_state + ((size_t)L * M.G + g) * IO::QUADS, lst);
                IO::store_rec(M.spec_state + so * IO::QUADS, lst);
                M.spec_kind[so] = 1; M.spec_ref[so] = slot_ref(L, ORD_SELF);
                pool_set(P, pend_slot, 1, slot_ref(L, ORD_SELF));
            }
            M.spec_cls[(size_t)pend_slot * M.G + g] = 0;
            // width: what the last request showed to be predictable (+1); after a miss half-way back to what came true (+1)
            const uint32_t issued = (pw >> 8) & 0xFFu;
            uint32_t w = pw >> 16;
            if (issued > 0) w = (hits >= issued) ? w + 1u : ((w + hits + 1u) / 2u > hits + 1u ? (w + hits + 1u) / 2u : hits + 1u);
            if (w > M.spec_k - 1) w = M.spec_k - 1;
            const uint32_t w_own = w;
            if (w > wcap) w = wcap;
            // rounds_left == 1: the plan is through and the device is emptying (the host says so): every slot that exists is used.
            uint32_t want = M.spec_k;
            if (rounds_left > 1) {
                const uint32_t rem = n_sims - sim;
                want = (rem + rounds_left - 1) / rounds_left;
                if (want > w + 1) want = w + 1;
            } else if (rounds_left == 0 && want > w + 1) want = w + 1;
            if (want > M.spec_k) want = M.spec_k;
            if (want > n_sims - sim) want = n_sims - sim;        // (never more playouts than simulations are left)
            uint32_t keep = 1u << pend_slot, ncls = 1;
            const uint32_t issued0 = ls.spec_issued;
            uint32_t req = 1;                                    // playouts requested for this round so far (the pending leaf's)
            if (want > 1 && sim + 1 < n_sims && lm.cap > 0) {
                // Scenarios for the pending playout's value: "no decision" (0: ply cap or draw) and "the side that has won more of this
                // search's playouts wins".  "No decision" goes first (its requests get the better classes) unless three quarters of the
                // search's playouts were decided: it also predicts right whenever a decision does not change the next selection, which is
                // the rule in a wide tree (measured at 65 536 games, S = 64: mid-game Copenhagen positions, 55 % decided, 89.9 M sims/s with
                // 0 first, 79 M with the decisive value first; Brandubh, nearly always decided, 112 M against 123 M).
                // The second pass may request up to kMctsExtraReq playouts beyond the round's share; what both passes ask for is requested
                // once (the slot table knows it).
                const uint32_t dw = bias & 0xFFFFu, aw = bias >> 16;
                const uint32_t lf = reinterpret_cast<const uint32_t*>(M.spec_state + ((size_t)pend_slot * M.G + g) * IO::QUADS)[IO::WORDS - 1];
                const double dec = (((lf & TAFL_F_SIDE) != 0u) == (dw >= aw)) ? 1.0 : -1.0;        // seen from the leaf's mover
                const bool dec_first = scen >= 2u && 4u * (dw + aw) > 3u * sim;
                mcts_speculate(M, g, L, sim, want, c_puct, n_sims, C, ls, X, lm, P, keep, ncls, req, dec_first ? dec : 0.0);
                if (scen >= 2u) {
                    root_load(M, g, X); X.rc.pos_valid = false;
                    uint32_t w2 = req + kMctsExtraReq; if (w2 > M.spec_k) w2 = M.spec_k;
                    mcts_speculate(M, g, L, sim, w2, c_puct, n_sims, C, ls, X, lm, P, keep, ncls, req, dec_first ? 0.0 : dec);
                }
            }
            M.spec_pend[g] = pend_slot | ((ls.spec_issued - issued0) << 8) | (w_own << 16);
            // requested playouts the pass did not come by (left over from a round that was full) run last
            TAFL_UNROLL for (uint32_t j = 0; j < kMctsMaxSlots; ++j)
                if (P.kind[j] == 1 && !((keep >> j) & 1u)) { M.spec_cls[(size_t)j * M.G + g] = (uint8_t)ncls; ++ncls; }
        } else {
            for (uint32_t j = 0; j < M.spec_k; ++j) M.spec_kind[(size_t)j * M.G + g] = 0;      // the search is over: what is left in the slots is not needed
        }
        M.sim_next[g] = sim; M.node_top[g] = X.node_top; M.edge_top[g] = X.edge_top; M.spec_bias[g] = bias;
        M.leaf[g] = pend_leaf; M.kind[g] = pending ? 1 : 0;
    }

    // Scenario passes of a step's prediction (a policy: it steers which playouts run when, never a result).  Inside the plan of a short
    // search the device is full and the rounds are counted: one pass (a second one costs the tree phase 10 % and gains nothing there:
    // 88 - 90 M sims/s against 95.7 M at S = 64).  A long search (deep trees: the next selection depends on the pending value) takes two
    // (S = 256: 91.6 -> 94.7 M, S = 1000: 67.5 -> 71 M), and so does every step past the plan or without one (S = 64: 95.7 -> 97.2 M;
    // a third scenario, "the other side wins", was measured too: more playouts, no fewer rounds).
    static TAFL_HD uint32_t mcts_scenarios(uint32_t rounds_left, uint32_t planned) { return (rounds_left > 1u && planned < 32u) ? 1u : 2u; }

    // playout of slot j of game g (predict() of mcts.py:85 in random-rollout mode), keyed by the leaf's position
    static TAFL_HD void mcts_slot_rollout(const MctsMem& M, uint32_t j, uint32_t g, uint64_t seed, uint64_t game_id, uint32_t sim_offset,
                                          uint32_t max_plies, const K& C) {
        const size_t o = (size_t)j * M.G + g;
        if (j >= M.spec_k || M.spec_kind[o] != 1) return;
        S st; IO::load_rec(M.spec_state + o * IO::QUADS, st);
        tafl_rollout_result r;
        playout(st, E::sim_key(E::game_key(seed, game_id), sim_offset + M.sim_base[g] + E::state_hash(st, C)), max_plies, C, r);
        M.spec_value[o] = r.value; M.spec_reason[o] = r.reason; M.spec_plies[o] = r.plies; M.spec_kind[o] = 2;
    }

    // Self-play: if game g has finished its search, play the most visited root play (first maximum, src/mcts.rs:216-227, = tafl_mcts_play_best)
    // on its batch state and start its next search.  The batch holds the position in the reference layout <NLS, WS>, the search arena in
    // <NL, W> (the same, or the dense 13-column layout).  Returns 0 nothing to do, 1 a new search begins, 2 the game has made its last play.
    template <int NLS, int WS>
    static TAFL_HD int selfplay_advance(const MctsMem& M, uint32_t g, Quad* soa, const SelfPlay& sp, uint32_t n_sims, uint32_t round, const K& C) {
        const uint32_t md = sp.moves_done[g];
        if (md >= sp.n_moves) return 0;
        if (M.sim_next[g] < n_sims || M.kind[g] == 1) return 0;              // its search is still running
        const NodeHdr h = M.hdr[g];
        const Edge* eb = &M.edges[(size_t)g * M.edge_cap + h.edge_base];
        uint32_t best = 0, child = 0;
        for (uint32_t j = 0; j < h.m; ++j) { const Edge e = eb[j]; if (e.n > best) { best = e.n; child = e.child; } }
        S st;
        if constexpr (NLS == NL && WS == W) StateIO<NL>::load_soa(soa, M.G, g, st);
        else { DState<NLS> t; StateIO<NLS>::load_soa(soa, M.G, g, t); restride<NLS, WS, NL, W>(t, C.n, st); }
        tafl_play p; p.from_row = p.from_col = p.axis = 0; p.disp = 0;
        if (best > 0 && TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING) {
            const NodeHdr ch = M.hdr[(size_t)child * M.G + g];
            Move bm; bm.from = ch.mv_from; bm.dir = ch.mv_dir; bm.dist = ch.mv_dist;
            bm.to = (uint32_t)((int)bm.from + E::delta(bm.dir) * (int)bm.dist);
            p = to_play(bm);
            Moves<NL> nx;
            E::apply(st, bm, C, nullptr, nx);
            if constexpr (NLS == NL && WS == W) StateIO<NL>::store_soa(soa, M.G, g, st);
            else { DState<NLS> t; restride<NL, W, NLS, WS>(st, C.n, t); StateIO<NLS>::store_soa(soa, M.G, g, t); }
        }
        sp.plays[(size_t)md * M.G + g] = p;
        if (md + 1 < sp.n_moves && TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING) {
            mcts_init_game(M, g, st, C);
            M.sim_base[g] = (md + 1u) * n_sims; sp.moves_done[g] = md + 1u; sp.start_round[g] = round;
            return 1;
        }
        sp.moves_done[g] = sp.n_moves;                                       // (the plays of the moves it does not make stay all-zero)
        return 2;
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
