// tafl_core.hpp — the rules engine as bit-parallel device functions (one game per lane).
//
// Re-expression (NOT a translation) of the reference's Tile-at-a-time logic:
//   move generation   game/play.rs:139-226 + game/game/logic.rs:119-214  -> occluded Kogge-Stone ray fills
//   captures          game/game/logic.rs:604-699                           -> shifted-mask custodial test
//   shieldwall        game/game/logic.rs:471-569                           -> edge walk (rare, branchy)
//   enclosure         game/game/logic.rs:309-401 (span fill)               -> bitboard flood to fixpoint
//   enclosure_secure  game/game/logic.rs:408-463                           -> threat masks per axis
//   exit fort         game/game/logic.rs:572-601
//   outcome           game/game/logic.rs:702-771
//   do_valid_play     game/game/logic.rs:782-820
//   repetition        game/game/state.rs:41-114
// Equality with the literal oracle (oracle/tafl_oracle.c) is checked by tests/test_hostsim_parity.py
// (this header compiled for the host) and tests/test_gpu_parity.py (the HIP kernels).
//
// Rules are general over the reference `Ruleset` (game/rules.rs:83-117): all presets of
// game/preset.rs run here, not only Copenhagen.
#pragma once
#include "tafl_bits.hpp"
#include "../../include/taflhip.h"

namespace tafl {

enum : int { CLS_ATT = 0, CLS_DEF = 1, CLS_KING = 2 };
enum : int { DIR_VP = 0, DIR_VM = 1, DIR_HP = 2, DIR_HM = 3 };

// ---- constants derived on the host from (rules, side_len, word) ---------------------------------
template <int NL>
struct Consts {
    Bits<NL> board, col0, coln, row0, rown, edge, corners, throne, throne_nb;
    Bits<NL> land_forbid[3];     // tiles class c may not stop on (corners / throne by rule), logic.rs:190-206
    Bits<NL> pass_forbid[3];     // empty tiles class c may not slide across,                  logic.rs:129-148,194-200
    Bits<NL> hostile_special[3]; // special tiles hostile to class c (special_tile_hostile),   logic.rs:76-82
    uint32_t n, w, throne_sq;
    uint32_t slow[3], edge_hostile[3];
    uint32_t king_like_soldier;  // king's movement masks equal the defender soldier's
    tafl_rules rules;
};

// ---- per-game state in registers -------------------------------------------------------------------
// flags: bit0 side(1=defender) bit1 attacker_mid_pair bit2 defender_mid_pair bits3-4 status
//        bits5-8 reason bit9 winner(1=defender) bits16-19 king_row bits20-23 king_col
template <int NL>
struct DState {
    Bits<NL> att, def;
    uint32_t rep[4];   // oldest first; 0 = None; TAFL_REP_PACK layout
    uint32_t turn, psc;
    uint32_t reps;     // attacker_reps | defender_reps << 16
    uint32_t flags;
};
#define TAFL_F_SIDE 0x1u
#define TAFL_F_AMID 0x2u
#define TAFL_F_DMID 0x4u
#define TAFL_F_STATUS(f) (((f) >> 3) & 3u)
#define TAFL_F_REASON(f) (((f) >> 5) & 15u)
#define TAFL_F_WINNER(f) (((f) >> 9) & 1u)
#define TAFL_F_KROW(f) (((f) >> 16) & 15u)
#define TAFL_F_KCOL(f) (((f) >> 20) & 15u)
#define TAFL_NO_SQ 0xFFFFu

template <int NL>
struct Moves {
    Bits<NL> reach[4];   // destination sets per direction V+,V-,H+,H- (rays of distinct pieces are disjoint)
    uint32_t cnt[4];
    uint32_t total;
};

struct Move { uint32_t from, to, dir, dist; };

template <int NL>
struct StepOut {
    Bits<NL> captures;
    uint32_t n_captures;
};

template <int NL, int W>
struct Engine {
    using B = Bits<NL>;
    using S = DState<NL>;
    using K = Consts<NL>;

    // ---- shifts along the four directions ----------------------------------------------------------
    template <int DIR, int STEP> static TAFL_HD B shd(const B& a) {
        if constexpr (DIR == DIR_VP) return shl<STEP * W>(a);
        else if constexpr (DIR == DIR_VM) return shr<STEP * W>(a);
        else if constexpr (DIR == DIR_HP) return shl<STEP>(a);
        else return shr<STEP>(a);
    }
    // destinations that are NOT the product of a row wrap for a one-step move in DIR
    template <int DIR> static TAFL_HD B nowrap(const K& C) {
        if constexpr (DIR == DIR_HP) return andn(C.board, C.col0);
        else if constexpr (DIR == DIR_HM) return andn(C.board, C.coln);
        else return C.board;
    }
    // one step in DIR, on-board, no wrap
    template <int DIR> static TAFL_HD B step1(const B& a, const K& C) { return shd<DIR, 1>(a) & nowrap<DIR>(C); }
    static TAFL_HD B dilate(const B& a, const K& C) {
        return step1<DIR_VP>(a, C) | step1<DIR_VM>(a, C) | step1<DIR_HP>(a, C) | step1<DIR_HM>(a, C);
    }
    // tiles whose neighbour in DIR is off the board
    template <int DIR> static TAFL_HD B lastline(const K& C) {
        if constexpr (DIR == DIR_VP) return C.rown; else if constexpr (DIR == DIR_VM) return C.row0;
        else if constexpr (DIR == DIR_HP) return C.coln; else return C.col0;
    }

    static TAFL_HD uint32_t king_sq(const S& st, const K& C) {
        const uint32_t r = TAFL_F_KROW(st.flags), c = TAFL_F_KCOL(st.flags);
        return (r < C.n && c < C.n) ? mul24(r, (uint32_t)W) + c : TAFL_NO_SQ;
    }
    // the tile that get_piece() reports as King: a defender standing on the nibble position
    // (game/board/state.rs:173-187, :24-26)
    static TAFL_HD B king_bit(const S& st, const K& C) {
        const uint32_t k = king_sq(st, C);
        return k == TAFL_NO_SQ ? bz<NL>() : (bit_at<NL>(k) & st.def);
    }
    static TAFL_HD bool king_armed_as_anvil(const K& C) { return C.rules.king_attack == TAFL_KING_ARMED || C.rules.king_attack == TAFL_KING_ANVIL; }

    // ---- move generation: occluded ray fill (Kogge-Stone), replaces ValidPlayIterator -------------------
    template <int DIR> static TAFL_HD B ray_reach(const B& gen0, const B& pass, const B& land, bool slow, const K& C) {
        const B nw = nowrap<DIR>(C);
        B g = gen0;
        if (!slow) {
            B p = pass & nw;
            g |= p & shd<DIR, 1>(g); p &= shd<DIR, 1>(p);
            g |= p & shd<DIR, 2>(g); p &= shd<DIR, 2>(p);
            g |= p & shd<DIR, 4>(g);
            if constexpr (W > 7) { p &= shd<DIR, 4>(p); g |= p & shd<DIR, 8>(g); }
        }
        return shd<DIR, 1>(g) & land & nw;
    }
    static TAFL_HD void class_reach(const B& gen, int cls, const B& empty, const K& C, B out[4]) {
        const B pass = andn(empty, C.pass_forbid[cls]);
        const B land = andn(empty, C.land_forbid[cls]);
        const bool slow = C.slow[cls] != 0;
        out[0] = ray_reach<DIR_VP>(gen, pass, land, slow, C);
        out[1] = ray_reach<DIR_VM>(gen, pass, land, slow, C);
        out[2] = ray_reach<DIR_HP>(gen, pass, land, slow, C);
        out[3] = ray_reach<DIR_HM>(gen, pass, land, slow, C);
    }
    // get_all_possible_moves as destination sets (game/main.rs:33-43).  `side`: 0 attacker, 1 defender.
    static TAFL_HD void movegen(const S& st, uint32_t side, const K& C, Moves<NL>& mv) {
        TAFL_UNROLL for (int d = 0; d < 4; ++d) { mv.reach[d] = bz<NL>(); mv.cnt[d] = 0; }
        mv.total = 0;
        if (TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING) return;         // logic.rs:165-167
        const B occ = st.att | st.def;
        const B empty = andn(C.board, occ);
        if (side == 0) {
            class_reach(st.att & C.board, CLS_ATT, empty, C, mv.reach);
        } else {
            const B kb = king_bit(st, C);
            if (C.king_like_soldier) {
                class_reach(st.def & C.board, CLS_DEF, empty, C, mv.reach);
            } else {
                class_reach(andn(st.def & C.board, kb), CLS_DEF, empty, C, mv.reach);
                if (any(kb)) {
                    B kr[4];
                    class_reach(kb, CLS_KING, empty, C, kr);
                    TAFL_UNROLL for (int d = 0; d < 4; ++d) mv.reach[d] |= kr[d];
                }
            }
        }
        TAFL_UNROLL for (int d = 0; d < 4; ++d) { mv.cnt[d] = popc(mv.reach[d]); mv.total += mv.cnt[d]; }
    }

    static TAFL_HD int delta(uint32_t dir) { return dir == 0 ? W : dir == 1 ? -W : dir == 2 ? 1 : -1; }

    // column `c` of the board as a mask (col0 pattern moved c < W <= 15 bits up: at most one limb of carry)
    static TAFL_HD B col_mask(uint32_t c, const K& C) {
        B o;
        TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = (C.col0.w[i] << c) | ((c != 0 && i > 0) ? (C.col0.w[i > 0 ? i - 1 : 0] >> ((32 - c) & 31)) : 0u);
        return o;
    }
    // the piece that reaches `to` moving in `dir`: the nearest occupied tile behind `to` on its line (no loop: highest /
    // lowest occupied bit of the line segment behind the destination)
    static TAFL_HD Move resolve(const S& st, uint32_t dir, uint32_t to, const K& C) {
        const B occ = (st.att | st.def) & C.board;
        const B lo = below<NL>(to), hi = andn(C.board, below<NL>(to + 1));
        uint32_t from;
        if (dir >= 2) from = (dir == 2) ? msb(occ & lo) : lsb(occ & hi);                      // same row: bits are contiguous
        else { const B col = occ & col_mask(to % (uint32_t)W, C); from = (dir == 0) ? msb(col & lo) : lsb(col & hi); }
        Move m; m.from = from; m.to = to; m.dir = dir;
        const uint32_t d = to > from ? to - from : from - to;
        m.dist = dir >= 2 ? d : div_w<W>(d);
        return m;
    }
    // idx-th play in ROLLOUT ORDER [build-defined, DESIGN.md]: direction-major V+,V-,H+,H-; inside a direction the
    // destinations in the natural order of the layout where that direction is "+1": V+ ascending (col,row), V- descending
    // (col,row), H+ ascending (row,col), H- descending (row,col).  This generic version works from the row-major sets;
    // the fast playout engine (tafl_fast.hpp) produces the same order directly from its four layouts.
    static TAFL_HD uint32_t nth_colmajor(const B& r, uint32_t j, const K& C) {
        uint32_t res = 0; bool found = false;
        for (uint32_t c = 0; c < C.n; ++c) {
            const B colm = col_mask(c, C);
            const B rc = r & colm;
            const uint32_t k = popc(rc);
            if (!found) { if (j < k) { res = nth_set_bit(rc, j); found = true; } else j -= k; }
        }
        return res;
    }
    static TAFL_HD Move pick_rollout(const S& st, const Moves<NL>& mv, uint32_t idx, const K& C) {
        uint32_t dir = 0;
        if (idx >= mv.cnt[0]) { idx -= mv.cnt[0]; dir = 1;
            if (idx >= mv.cnt[1]) { idx -= mv.cnt[1]; dir = 2;
                if (idx >= mv.cnt[2]) { idx -= mv.cnt[2]; dir = 3; } } }
        uint32_t to;
        if (dir == 0) to = nth_colmajor(mv.reach[0], idx, C);
        else if (dir == 1) to = nth_colmajor(mv.reach[1], mv.cnt[1] - 1 - idx, C);
        else if (dir == 2) to = nth_set_bit(mv.reach[2], idx);
        else to = nth_set_bit(mv.reach[3], mv.cnt[3] - 1 - idx);
        return resolve(st, dir, to, C);
    }

    // ---- canonical iteration (ValidPlayIterator order: play.rs:157,166-183 over iter_occupied) -------------
    // piece class of an occupied tile of `side`
    static TAFL_HD int cls_of(const S& st, uint32_t side, uint32_t sq, const K& C) {
        if (side == 0) return CLS_ATT;
        return (sq == king_sq(st, C)) ? CLS_KING : CLS_DEF;
    }
    // Next legal play after `cur` in canonical order (cur = canon_start() to begin).  Returns false when the
    // side has no further play.  Literal stepping of ValidPlayIterator::next (play.rs:189-225): used once per
    // simulation (tree expansion) and by the dense-mask writer, never per rollout ply.
    static TAFL_HD bool canon_next(const S& st, uint32_t side, const K& C, Move& cur) {
        if (TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING) return false;
        const B occ = st.att | st.def;
        const B mine = sel(side == 0, st.att, st.def) & C.board;      // (sel: a `?:` between two objects becomes a pointer select + a trip through scratch memory)
        uint32_t sq = cur.from, dir = cur.dir, dist = cur.dist;
        if (sq == TAFL_NO_SQ) {                       // start: first piece of the side (iter_occupied order)
            if (!any(mine)) return false;
            sq = lsb(mine); dir = 0; dist = 0;
        }
        for (;;) {
            const int cls = cls_of(st, side, sq, C);
            const int r0 = (int)(sq / (uint32_t)W), c0 = (int)(sq % (uint32_t)W);
            if (dist > 0) {
                // resuming after a yield: a landable tile that cannot be passed (NoPass/KingPass throne) ends the ray
                const int dl = delta(dir);
                const uint32_t xp = (uint32_t)((int)sq + dl * (int)dist);
                if (test(C.pass_forbid[cls], xp)) { ++dir; dist = 0; }
            }
            while (dir < 4) {
                const uint32_t nd = dist + 1;
                int rr = r0, cc = c0;
                if (dir == 0) rr += (int)nd; else if (dir == 1) rr -= (int)nd; else if (dir == 2) cc += (int)nd; else cc -= (int)nd;
                bool end_dir = (rr < 0 || cc < 0 || rr >= (int)C.n || cc >= (int)C.n);          // off board
                uint32_t x = 0;
                if (!end_dir) {
                    x = (uint32_t)rr * (uint32_t)W + (uint32_t)cc;
                    if (test(occ, x)) end_dir = true;                                         // BlockedByPiece
                    else if (C.slow[cls] && nd > 1) end_dir = true;                           // TooFar (never yields)
                }
                if (!end_dir) {
                    if (!test(C.land_forbid[cls], x)) {
                        cur.from = sq; cur.to = x; cur.dir = dir; cur.dist = nd;
                        return true;
                    }
                    if (test(C.pass_forbid[cls], x)) end_dir = true;                          // neither occupy nor pass
                    else { dist = nd; continue; }                                             // pass (empty throne)
                }
                ++dir; dist = 0;
            }
            // all four directions exhausted: next piece in ascending bit order
            if (sq + 1 >= (uint32_t)(NL * 32)) return false;
            const B rest = andn(mine, below<NL>(sq + 1));
            if (!any(rest)) return false;
            sq = lsb(rest); dir = 0; dist = 0;
        }
    }
    static TAFL_HD Move canon_start() { Move m; m.from = TAFL_NO_SQ; m.to = 0; m.dir = 0; m.dist = 0; return m; }

    // ---- validate_play_for_side (logic.rs:159-214) --------------------------------------------------------
    static TAFL_HD int validate(const S& st, tafl_play p, uint32_t side, const K& C, Move* out) {
        if (TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING) return TAFL_PLAY_GAME_OVER;
        const uint32_t fr = p.from_row, fc = p.from_col;
        const bool from_in = fr < C.n && fc < C.n;
        // get_piece(from) masks bit row*ROW_WIDTH+col of the raw reference words with no bounds check
        // (bitfield.rs:72-74, board/state.rs:173-187): an off-board `from` can alias another tile, or the king nibble
        // in the top four bits.  Reproduced so that the error code matches the reference for ANY play.
        const uint32_t from = fr * (uint32_t)W + fc;
        const B occ = st.att | st.def;
        auto raw_test = [&](const B& word, uint32_t nib, uint32_t idx) -> bool {
            if (idx >= (uint32_t)(NL * 32)) return false;
            if (idx >= (uint32_t)(NL * 32 - 4)) return (nib >> (idx - (uint32_t)(NL * 32 - 4))) & 1u;
            return test(word, idx);
        };
        const bool has_def = raw_test(st.def, TAFL_F_KROW(st.flags), from), has_att = raw_test(st.att, TAFL_F_KCOL(st.flags), from);
        if (!has_def && !has_att) return TAFL_PLAY_NO_PIECE;
        const uint32_t pside = has_def ? 1u : 0u;
        if (pside != side) return TAFL_PLAY_WRONG_PLAYER;
        int tr = (int)fr, tc = (int)fc;
        const bool horiz = p.axis != 0;
        if (horiz) tc += p.disp; else tr += p.disp;
        tr &= 0xFF; tc &= 0xFF;                                                    // Play::to casts to u8 (play.rs:59-62)
        if (!(from_in && (uint32_t)tr < C.n && (uint32_t)tc < C.n)) return TAFL_PLAY_OUT_OF_BOUNDS;
        // NoCommonAxis cannot arise from (axis, displacement) plays
        const uint32_t to = (uint32_t)tr * (uint32_t)W + (uint32_t)tc;
        if (test(occ, to)) return TAFL_PLAY_BLOCKED_BY_PIECE;
        const int cls = has_def ? ((fr == TAFL_F_KROW(st.flags) && fc == TAFL_F_KCOL(st.flags)) ? CLS_KING : CLS_DEF) : CLS_ATT;
        const int dl = horiz ? 1 : W;
        const uint32_t lo = from < to ? from : to, hi = from < to ? to : from;
        bool through_throne = false;
        for (uint32_t x = lo + (uint32_t)dl; x < hi; x += (uint32_t)dl) {
            if (test(occ, x)) return TAFL_PLAY_BLOCKED_BY_PIECE;
            if (x == C.throne_sq) through_throne = true;
        }
        if (test(C.corners, to) && !((C.rules.may_enter_corners >> (cls == CLS_ATT ? 1 : cls == CLS_DEF ? 9 : 8)) & 1u))
            return TAFL_PLAY_MOVE_ONTO_BLOCKED_TILE;
        const bool is_king = cls == CLS_KING;
        if ((C.rules.throne_movement == TAFL_THRONE_NOPASS || (C.rules.throne_movement == TAFL_THRONE_KINGPASS && !is_king)) && through_throne)
            return TAFL_PLAY_MOVE_THROUGH_BLOCKED_TILE;
        if ((C.rules.throne_movement == TAFL_THRONE_NOENTRY || (C.rules.throne_movement == TAFL_THRONE_KINGENTRY && !is_king)) && to == C.throne_sq)
            return TAFL_PLAY_MOVE_ONTO_BLOCKED_TILE;
        const uint32_t dist = (uint32_t)(p.disp < 0 ? -(int)p.disp : (int)p.disp);
        if (C.slow[cls] && dist > 1) return TAFL_PLAY_TOO_FAR;
        if (out) { out->from = from; out->to = to; out->dist = dist; out->dir = horiz ? (p.disp > 0 ? 2u : 3u) : (p.disp > 0 ? 0u : 1u); }
        return TAFL_PLAY_OK;
    }

    // ---- enclosure: bitboard flood (replaces the span fill of logic.rs:309-401) -----------------------------
    // inside: tiles that can be filled (empty + enclosed piece types); neither: pieces that abort the search.
    // Returns false for `None`.  fill = occupied ∪ unoccupied; boundary = enclosing pieces adjacent to the fill.
    static TAFL_HD bool flood_from(const B& start, const B& inside, const B& neither, bool abort_edge, bool abort_corner,
                                   const K& C, B& fill) {
        B f = start;
        B stopmask = bz<NL>();
        if (abort_edge) stopmask |= C.edge;
        if (abort_corner) stopmask |= C.corners;
        for (int it = 0; it < NL * 32; ++it) {
            if (any(f & stopmask)) return false;
            const B d = dilate(f, C);
            if (any(d & neither)) return false;
            const B nf = f | (d & inside);
            if (eq(nf, f)) break;
            f = nf;
        }
        fill = f;
        return true;
    }
    static TAFL_HD bool flood(uint32_t start_sq, const B& inside, const B& neither, bool abort_edge, bool abort_corner,
                              const K& C, B& fill) {
        if (start_sq == TAFL_NO_SQ) return false;
        const B f = bit_at<NL>(start_sq) & inside;
        if (!any(f)) return false;
        return flood_from(f, inside, neither, abort_edge, abort_corner, C, fill);
    }

    // enclosure_secure (logic.rs:408-463) for a boundary made of soldiers of class `bcls`
    static TAFL_HD bool secure(const S& st, const B& fill, const B& boundary, int bcls, bool inside_safe, bool outside_safe, const K& C) {
        if (inside_safe && outside_safe) return true;
        const B occ = st.att | st.def;
        const B empty = andn(C.board, occ);
        const int hcls = bcls == CLS_ATT ? CLS_DEF : CLS_ATT;                       // hostile soldier
        B enemy = sel(bcls == CLS_ATT, st.def, st.att);
        if (bcls == CLS_ATT && !king_armed_as_anvil(C)) enemy = andn(enemy, king_bit(st, C));
        const B th = (enemy & C.board) | (empty & C.hostile_special[bcls]);         // tile_hostile
        B safe = bz<NL>();
        if (inside_safe) safe |= fill;
        if (outside_safe) safe |= andn(C.board, fill);
        const B a = andn(safe, C.hostile_special[bcls]);
        const B b = andn(occ | C.land_forbid[hcls], th);
        const B thr = andn(C.board, a | b);                                         // threatening tiles
        const bool eh = C.edge_hostile[bcls] != 0;
        const B up = step1<DIR_VP>(thr, C) | gate(C.row0, eh);     // tile above is threatening
        const B dn = step1<DIR_VM>(thr, C) | gate(C.rown, eh);
        const B lf = step1<DIR_HP>(thr, C) | gate(C.col0, eh);
        const B rt = step1<DIR_HM>(thr, C) | gate(C.coln, eh);
        return !any(boundary & ((up & dn) | (lf & rt)));
    }

    // detect_exit_fort (logic.rs:572-601)
    static TAFL_HD bool exit_fort(const S& st, const K& C) {
        const uint32_t k = king_sq(st, C);
        if (k == TAFL_NO_SQ) return false;
        const B kt = bit_at<NL>(k);
        TAFL_STAT_HIT(0);
        if (!any(kt & C.edge)) return false;
        TAFL_STAT_HIT(1);
        const B occ = st.att | st.def;
        const B empty = andn(C.board, occ);
        const B attb = st.att & C.board;
        // start tile: a defender there is the king (enclosed type), empty is enclosed, an attacker aborts (logic.rs:320-327)
        if (any(kt & attb)) return false;
        // first two rings of the flood unrolled: they decide almost every case (an attacker next to the growing region
        // aborts, logic.rs:288-291), and the first ring is also the "king has space to move" test (logic.rs:590)
        const B d1 = dilate(kt, C);
        if (any(d1 & attb)) return false;
        TAFL_STAT_HIT(2);
        if (!any(d1 & empty)) return false;
        const B f1 = kt | (d1 & empty);
        if (any(f1 & C.corners)) return false;
        TAFL_STAT_HIT(3);
        const B d2 = dilate(f1, C);
        if (any(d2 & attb)) return false;
        TAFL_STAT_HIT(4);
        const B inside = empty | (kt & st.def);
        B fill;
        if (!flood_from(f1 | (d2 & inside), inside, attb, false, true, C, fill)) return false;
        TAFL_STAT_HIT(5);
        const B boundary = dilate(fill, C) & andn(st.def, fill);
        return secure(st, fill, boundary, CLS_DEF, true, false, C);
    }

    // ---- shieldwall (logic.rs:471-569): literal walk along the edge, rare --------------------------------------
    static TAFL_HD bool sw_search(const S& st, uint32_t to, bool horiz, int away, int dir, uint32_t mover, const K& C, B& wall) {
        const B occ = st.att | st.def;
        const B mine = sel(mover != 0, st.def, st.att), theirs = sel(mover != 0, st.att, st.def);
        int r = (int)(to / (uint32_t)W), c = (int)(to % (uint32_t)W);
        wall = bz<NL>(); uint32_t n_wall = 0;
        for (int guard = 0; guard < 32; ++guard) {
            if (horiz) c += dir; else r += dir;
            if (r < 0 || c < 0 || r >= (int)C.n || c >= (int)C.n) return false;
            const uint32_t t = (uint32_t)r * (uint32_t)W + (uint32_t)c;
            const bool occd = test(occ, t), corner = test(C.corners, t);
            if (!(occd || (C.rules.sw_corners_may_close && corner))) return false;
            if (!occd) return n_wall >= 2;
            if (test(theirs, t)) {
                const int pr = horiz ? r + away : r, pc = horiz ? c : c + away;
                if (pr < 0 || pc < 0 || pr >= (int)C.n || pc >= (int)C.n) return false;
                const uint32_t pin = (uint32_t)pr * (uint32_t)W + (uint32_t)pc;
                if (test(mine, pin)) { wall |= bit_at<NL>(t); ++n_wall; } else return false;
            }
            if (test(mine, t) || (corner && C.rules.sw_corners_may_close)) return n_wall >= 2;
        }
        return false;
    }
    static TAFL_HD B shieldwall(const S& st, uint32_t to, uint32_t mover, const K& C) {
        if (!C.rules.has_shieldwall) return bz<NL>();
        const uint32_t r = div_w<W>(to), c = mod_w<W>(to);
        bool horiz; int away;
        if (r == 0) { horiz = true; away = 1; } else if (r == C.n - 1) { horiz = true; away = -1; }
        else if (c == 0) { horiz = false; away = 1; } else if (c == C.n - 1) { horiz = false; away = -1; }
        else return bz<NL>();
        B wall;
        bool found = sw_search(st, to, horiz, away, -1, mover, C, wall);
        if (!found) found = sw_search(st, to, horiz, away, 1, mover, C, wall);
        if (!found) return bz<NL>();
        // filter by sw_rule.captures (logic.rs:562-565); wall tiles hold pieces of the non-moving side
        const B kb = king_bit(st, C);
        const uint32_t sold_bit = mover ? 1u : 9u, king_bitpos = 8u;
        B keep = bz<NL>();
        if ((C.rules.sw_captures >> sold_bit) & 1u) keep |= andn(wall, kb);
        if ((C.rules.sw_captures >> king_bitpos) & 1u) keep |= wall & kb;
        return keep;
    }

    // ---- get_captures (logic.rs:604-699) on the post-move board, side_to_play still the mover ------------------
    // Custodial captures, the king capture and the shieldwall pre-filter are evaluated on 64-bit fields around the destination
    // (field64, tafl_bits.hpp): with the field starting BK = 2W+1 bits below `to`, the four neighbours sit at the fixed bits
    // BK-1, BK+1, BK-W, BK+W, the tiles behind them at BK-2, BK+2, BK-2W, BK+2W, and every tile the king test or the
    // shieldwall filter looks at lies within BK-2W-1 .. BK+2W+1 = 0 .. 4W+2 <= 62; validity (board limits, row wrap) comes
    // from (row, col).  Straight-line on purpose: 64 games share an instruction stream, so a branch that 1 % of the games
    // take is taken by half of the waves.
    // cust_out: which directions captured custodially — bit0 V+ (row+1), bit1 V-, bit2 H+ (col+1), bit3 H-.
    static TAFL_HD B captures(const S& st, const Move& m, uint32_t mover, bool mover_is_king, const K& C, uint32_t* cust_out = nullptr) {
        constexpr int BK = 2 * W + 1;
        static_assert(BK <= 32 && 4 * W + 2 < 64, "field layout needs W <= 15");
        B caps = bz<NL>();
        uint32_t cust = 0;
        const uint32_t to = m.to, r = div_w<W>(to), c = mod_w<W>(to), n = C.n;
        const uint64_t fa = field64<BK>(st.att, to), fd = field64<BK>(st.def, to);
        const uint64_t fm = mover ? fd : fa, ft = mover ? fa : fd;
        const uint32_t kq = king_sq(st, C);
        const uint32_t krel = kq - to + (uint32_t)BK;                    // field position of the king's tile (>= 64: outside / no king)
        const uint64_t kf = krel < 64u ? ((1ull << krel) & fd) : 0ull;   // the king, if he stands inside the field
        auto bit = [](uint64_t f, int pos) -> uint32_t { return (uint32_t)(f >> pos) & 1u; };
        auto vbit = [](uint64_t f, uint32_t pos) -> uint32_t { return (uint32_t)(f >> (pos & 63u)) & 1u; };
        if (!mover_is_king || C.rules.king_attack == TAFL_KING_ARMED || C.rules.king_attack == TAFL_KING_HAMMER) {
            // hostile-to-victim occupied tiles: mover-side pieces, an unarmed king excluded (tile_hostile :85-93);
            // victim class = the other side's soldiers
            const uint64_t friends = (mover && !king_armed_as_anvil(C)) ? (fm & ~kf) : fm;
            const uint64_t victims = mover ? ft : (ft & ~kf);
            const uint64_t hs = field64<BK>(blend(mover != 0, C.hostile_special[CLS_ATT], C.hostile_special[CLS_DEF]), to);
            const uint64_t hostile = friends | (hs & ~(fa | fd));
            const uint32_t mm = mover ? 0xFFFFFFFFu : 0u;
            const uint32_t eh = (((C.edge_hostile[CLS_ATT] & mm) | (C.edge_hostile[CLS_DEF] & ~mm)) != 0) ? 1u : 0u;
            const uint32_t vp = (uint32_t)(r + 1 < n) & bit(victims, BK + W) & ((r + 2 < n) ? bit(hostile, BK + 2 * W) : eh);
            const uint32_t vm = (uint32_t)(r >= 1) & bit(victims, BK - W) & ((r >= 2) ? bit(hostile, BK - 2 * W) : eh);
            const uint32_t hp = (uint32_t)(c + 1 < n) & bit(victims, BK + 1) & ((c + 2 < n) ? bit(hostile, BK + 2) : eh);
            const uint32_t hm = (uint32_t)(c >= 1) & bit(victims, BK - 1) & ((c >= 2) ? bit(hostile, BK - 2) : eh);
            cust = vp | (vm << 1) | (hp << 2) | (hm << 3);
            const uint64_t cf = ((uint64_t)vp << (BK + W)) | ((uint64_t)vm << (BK - W)) | ((uint64_t)hp << (BK + 1)) | ((uint64_t)hm << (BK - 1));
            B cs = deposit64<BK, NL>(cf, to);
            // Linnaean capture (logic.rs:676-685, :859-879): only for victims whose far tile was not hostile
            if (C.rules.linnaean_capture && mover == 0) {
                const B kb = king_bit(st, C);
                if (any(kb & C.throne)) {
                    const B empty = andn(C.board, st.att | st.def);
                    const B hk = (st.att & C.board) | (empty & C.hostile_special[CLS_KING]);
                    if (popc(C.throne_nb & hk) == 3) {
                        // victim n adjacent to `to` with far == throne: n is a throne neighbour in line with `to`
                        const B tbit = bit_at<NL>(to);
                        const B lin = (step1<DIR_VP>(tbit, C) & step1<DIR_VM>(C.throne, C)) | (step1<DIR_VM>(tbit, C) & step1<DIR_VP>(C.throne, C))
                                    | (step1<DIR_HP>(tbit, C) & step1<DIR_HM>(C.throne, C)) | (step1<DIR_HM>(tbit, C) & step1<DIR_HP>(C.throne, C));
                        cs |= lin & andn(st.def & C.board, kb);
                    }
                }
            }
            caps |= cs;
            TAFL_PROF_SPLIT(1);
            // enemy king next to the destination (only an attacker can face it), logic.rs:612-675
#ifndef TAFL_ABLATE_KINGCAP
            const uint32_t kr = TAFL_F_KROW(st.flags), kc = TAFL_F_KCOL(st.flags);
            const uint32_t kd = to > kq ? to - kq : kq - to;
            const bool king_adjacent = mover == 0 && kf != 0ull && (kd == (uint32_t)W || (kd == 1u && r == kr));
            if (king_adjacent) TAFL_STAT_HIT(9);
            if (wave_any(king_adjacent)) {
                // tile_hostile(., king) on the field; the king's other three neighbours: `far` opposite the attacker, and the
                // two perpendicular ones
                const uint64_t hk = fa | (field64<BK>(C.hostile_special[CLS_KING], to) & ~(fa | fd));
                const uint32_t ehk = C.edge_hostile[CLS_KING] != 0 ? 1u : 0u;
                const bool same_row = kr == r;
                const uint32_t pstep = same_row ? (uint32_t)W : 1u;
                const uint32_t far_pos = 2u * krel - (uint32_t)BK, p1_pos = krel + pstep, p2_pos = krel - pstep;
                const uint32_t fr = 2u * kr - r, fc = 2u * kc - c;                               // wraps to >= n when off the board
                const bool far_on = fr < n && fc < n;
                const bool p1_on = same_row ? (kr + 1 < n) : (kc + 1 < n), p2_on = same_row ? (kr >= 1) : (kc >= 1);
                const uint32_t far_h = vbit(hk, far_pos), p1_h = vbit(hk, p1_pos), p2_h = vbit(hk, p2_pos);
                const uint32_t far_sq = 2u * kq - to, p1_sq = kq + pstep, p2_sq = kq - pstep;
                const bool beside = test(C.throne_nb, kq & 0xFFu);
                bool captured = false;
                // (i) strong-by-throne king beside his throne: every on-board neighbour hostile or the throne (logic.rs:621-632)
                if (C.rules.king_strength == TAFL_KING_STRONG_BY_THRONE
                    && (C.rules.throne_movement == TAFL_THRONE_NOENTRY || C.rules.throne_movement == TAFL_THRONE_KINGENTRY)) {
                    const bool ok_far = !far_on || far_h || far_sq == C.throne_sq;
                    const bool ok_p1 = !p1_on || p1_h || p1_sq == C.throne_sq;
                    const bool ok_p2 = !p2_on || p2_h || p2_sq == C.throne_sq;
                    captured = beside && ok_far && ok_p1 && ok_p2;
                }
                // (ii) far tile hostile, and for a strong king both perpendicular tiles too (logic.rs:634-675)
                const bool host_far = far_on ? far_h != 0 : ehk != 0;
                const bool host_p1 = p1_on ? p1_h != 0 : ehk != 0, host_p2 = p2_on ? p2_h != 0 : ehk != 0;
                bool strong;
                switch (C.rules.king_strength) {
                    case TAFL_KING_STRONG: strong = true; break;
                    case TAFL_KING_WEAK: strong = false; break;
                    default: strong = beside || kq == C.throne_sq; break;
                }
                captured = captured || (host_far && (!strong || (host_p1 && host_p2)));
                caps |= gate(bit_at<NL>(kq & 0xFFu), king_adjacent && captured);
            }
#endif
        }
        TAFL_PROF_SPLIT(2);
#ifndef TAFL_ABLATE_SW
        if (C.rules.has_shieldwall) {
            // A wall captures only if >= 2 enemy pieces stand in a row next to `to` along its edge, each pinned by a piece of
            // the mover on its inner side (logic.rs:507-530,556): test those four tiles before paying for the edge walk.
            const uint64_t pe_row = ft & ((r == 0) ? (fm >> W) : (fm << W));      // enemy with a mover piece one row inwards
            const uint64_t pe_col = ft & ((c == 0) ? (fm >> 1) : (fm << 1));      //                   ... one column inwards
            const uint32_t hp = bit(pe_row, BK + 1) & bit(pe_row, BK + 2) & (uint32_t)(c + 2 < n);
            const uint32_t hm = bit(pe_row, BK - 1) & bit(pe_row, BK - 2) & (uint32_t)(c >= 2);
            const uint32_t vp = bit(pe_col, BK + W) & bit(pe_col, BK + 2 * W) & (uint32_t)(r + 2 < n);
            const uint32_t vm = bit(pe_col, BK - W) & bit(pe_col, BK - 2 * W) & (uint32_t)(r >= 2);
            const uint32_t row_edge = (uint32_t)(r == 0) | (uint32_t)(r == n - 1), col_edge = (uint32_t)(c == 0) | (uint32_t)(c == n - 1);
            if (((row_edge & (hp | hm)) | (col_edge & (vp | vm))) != 0) { TAFL_STAT_HIT(8); caps |= shieldwall(st, to, mover, C); }
        }
#endif
        TAFL_PROF_SPLIT(3);
        if (cust_out) *cust_out = cust;
        return caps;
    }

    // ---- RepetitionTracker::track_play (game/game/state.rs:92-113) ------------------------------------------------
    static TAFL_HD void track(S& st, uint32_t mover, const Move& m, bool captured) {
        const uint32_t fr = div_w<W>(m.from), fc = mod_w<W>(m.from);
        const bool horiz = m.dir >= 2; const int disp = (m.dir & 1) ? -(int)m.dist : (int)m.dist;
        const uint32_t rec = TAFL_REP_PACK(mover, fr, fc, horiz, disp, captured);
        // branch-free: `hit` = this play repeats the one four plies back; every second hit counts (the mid-pair flag toggles);
        // a miss clears the mover's counter and flag
        const uint32_t sh = mover ? 16u : 0u;
        const uint32_t midbit = mover ? TAFL_F_DMID : TAFL_F_AMID;
        const bool hit = !captured && rec == st.rep[0];
        const bool is_rep = hit && !(st.flags & midbit);
        const uint32_t cur = (st.reps >> sh) & 0xFFFFu;
        const uint32_t nxt = hit ? (cur + ((is_rep && cur < 0xFFFFu) ? 1u : 0u)) : 0u;
        st.flags = hit ? (st.flags ^ midbit) : (st.flags & ~midbit);
        const uint32_t ar = mover ? (st.reps & 0xFFFFu) : nxt, dr = mover ? nxt : (st.reps >> 16);
        st.reps = ar | (dr << 16);
        st.rep[0] = st.rep[1]; st.rep[1] = st.rep[2]; st.rep[2] = st.rep[3]; st.rep[3] = rec;
    }

    // ---- do_valid_play (logic.rs:782-820) + get_game_outcome (:702-771) ---------------------------------------------
    // Split in three so that the fast playout engine (tafl_fast.hpp) can slot its own move generator in between:
    //   apply_pre      move the piece, captures, removal, repetition tracker             (logic.rs:787-799)
    //   outcome_early  every outcome test that precedes the no-plays test              (logic.rs:709-758)
    //   apply_finish   no-plays test from the opponent's move count, turn/side/status  (logic.rs:760-816)
    struct ApplyCtx { uint32_t mover; bool mover_is_king, king_captured; B tbit, caps; uint32_t ncap, cust; };
    struct Outcome { bool over; uint32_t status, reason, winner; };

    // `m` must be a valid play of the side to move (destination empty).
    static TAFL_HD void apply_pre(S& st, const Move& m, const K& C, ApplyCtx& ax) { apply_pre(st, m, C, ax, st.flags & TAFL_F_SIDE); }
    // the same with the mover handed in: `mover` must equal st.flags & TAFL_F_SIDE.  The playout loop of tafl_fast.hpp passes a
    // compile-time constant (its two half-iterations each serve one side), which folds every `mover ? a : b` below and in captures().
    static TAFL_HD void apply_pre(S& st, const Move& m, const K& C, ApplyCtx& ax, const uint32_t mover) {
        const B fbit = bit_at<NL>(m.from), tbit = bit_at<NL>(m.to);
        const bool mover_is_king = mover && m.from == king_sq(st, C);
        // board.move_piece (board/state.rs:218-223): clear `from`, set `to` on the mover's side — branch-free
        {
            const B mv = fbit | tbit;
            st.def = st.def ^ gate(mv, mover != 0);
            st.att = st.att ^ gate(mv, mover == 0);
            const uint32_t r = div_w<W>(m.to), c = mod_w<W>(m.to);
            const uint32_t kf = (st.flags & ~0x00FF0000u) | (r << 16) | (c << 20);
            st.flags = mover_is_king ? kf : st.flags;
        }
        const B caps = captures(st, m, mover, mover_is_king, C, &ax.cust);
        ax.king_captured = mover == 0 && king_sq(st, C) != TAFL_NO_SQ && test(caps, king_sq(st, C));
        st.att = andn(st.att, caps); st.def = andn(st.def, caps);
        ax.ncap = popc(caps); ax.caps = caps; ax.mover = mover; ax.mover_is_king = mover_is_king; ax.tbit = tbit;
#ifndef TAFL_ABLATE_TRACK
        track(st, mover, m, ax.ncap != 0);
#endif
        if (ax.ncap == 0) st.psc += 1;
    }
    // skip_enclosure / skip_fort: the caller has PROVED that the enclosure win / the exit fort cannot apply (see tafl_fast.hpp); never
    // set otherwise.
    static TAFL_HD Outcome outcome_early(const S& st, const ApplyCtx& ax, const K& C, bool skip_enclosure, bool skip_fort = false) {
        Outcome o; o.over = false; o.status = TAFL_STATUS_ONGOING; o.reason = 0; o.winner = 0;
        const uint32_t mover = ax.mover;
        const B other = sel(mover != 0, st.att, st.def) & C.board;
        if (!any(other)) { o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_ALL_CAPTURED; o.winner = mover; o.over = true; }
        if (!o.over && mover == 0) {
            if (ax.king_captured) { o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_KING_CAPTURED; o.winner = 0; o.over = true; }
#ifndef TAFL_ABLATE_FLOOD
            else if (C.rules.enclosure_win != TAFL_ENCL_NONE && !skip_enclosure) {
                TAFL_PROF_BEGIN(8);
                TAFL_STAT_HIT(10);
                B fill;
                const B inside = andn(C.board, st.att);
                if (flood(king_sq(st, C), inside, bz<NL>(), C.rules.enclosure_win == TAFL_ENCL_WITHOUT_EDGE_ACCESS, true, C, fill)) {
                    if (popc(fill & st.def) == popc(st.def & C.board)) {
                        const B boundary = dilate(fill, C) & st.att;
                        if (secure(st, fill, boundary, CLS_ATT, false, true, C)) {
                            o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_ENCLOSED; o.winner = 0; o.over = true;
                        }
                    }
                }
                TAFL_PROF_END(8);
            }
#endif
        } else if (!o.over) {
            if (ax.mover_is_king && any(ax.tbit & (C.rules.edge_escape ? C.edge : C.corners))) {
                o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_KING_ESCAPED; o.winner = 1; o.over = true;
            }
#ifndef TAFL_ABLATE_FORT
            else if (C.rules.exit_fort && !skip_fort) {
                TAFL_PROF_BEGIN(9);
                const bool fort = exit_fort(st, C);
                TAFL_PROF_END(9);
                if (fort) { o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_EXIT_FORT; o.winner = 1; o.over = true; }
            }
#endif
        }
        if (!o.over && C.rules.has_repetition_rule) {
            const uint32_t reps = mover ? (st.reps >> 16) : (st.reps & 0xFFFFu);
            if (reps >= C.rules.n_repetitions) {
                if (C.rules.rep_is_loss) { o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_REPETITION; o.winner = mover ^ 1u; }
                else { o.status = TAFL_STATUS_DRAW; o.reason = TAFL_DRAW_REPETITION; }
                o.over = true;
            }
        }
        return o;
    }
    // next_total: number of plays of the opponent on the post-move board (side_can_play, logic.rs:760-768, :837-846)
    static TAFL_HD void apply_finish(S& st, const ApplyCtx& ax, Outcome o, uint32_t next_total, const K& C) {
        const uint32_t mover = ax.mover;
        if (!o.over && next_total == 0) {
            if (C.rules.draw_on_no_plays) { o.status = TAFL_STATUS_DRAW; o.reason = TAFL_DRAW_NO_PLAYS; }
            else { o.status = TAFL_STATUS_WIN; o.reason = TAFL_WIN_NO_PLAYS; o.winner = mover; }
            o.over = true;
        }
        st.turn += 1;
        st.flags = (st.flags & ~(TAFL_F_SIDE | (0x7Fu << 3))) | (mover ^ 1u) | (o.status << 3) | (o.reason << 5) | (o.winner << 9);
    }
    // `next` receives the opponent's move set (computed for the NoPlays test, reused by rollouts).
    static TAFL_HD void apply(S& st, const Move& m, const K& C, StepOut<NL>* out, Moves<NL>& next) {
        ApplyCtx ax;
        apply_pre(st, m, C, ax);
        if (out) { out->captures = ax.caps; out->n_captures = ax.ncap; }
        const Outcome o = outcome_early(st, ax, C, false);
        // side_can_play(other): status is still Ongoing while it is evaluated
        if (!o.over) movegen(st, ax.mover ^ 1u, C, next);
        else { TAFL_UNROLL for (int d = 0; d < 4; ++d) { next.reach[d] = bz<NL>(); next.cnt[d] = 0; } next.total = 0; }
        apply_finish(st, ax, o, o.over ? 1u : next.total, C);
        if (TAFL_F_STATUS(st.flags) != TAFL_STATUS_ONGOING) { TAFL_UNROLL for (int d = 0; d < 4; ++d) { next.reach[d] = bz<NL>(); next.cnt[d] = 0; } next.total = 0; }
    }

    // ---- taflmix32 RNG (build-defined) -----------------------------------------------------------------------
    static TAFL_HD uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
    // game key: 64 bits (two independently mixed words), so that no two of the 524 288 games of BASELINE configs[3] share their whole
    // stream of simulation keys (a 32-bit key would collide for ~30 pairs); simulation and ply words are 32 bits
    static TAFL_HD uint64_t game_key(uint64_t seed, uint64_t game_id) {
        const uint32_t h0 = fmix32((uint32_t)seed ^ fmix32((uint32_t)(seed >> 32) + 0x9E3779B9u));
        const uint32_t h1 = fmix32((uint32_t)(seed >> 32) ^ fmix32((uint32_t)seed + 0x7F4A7C15u));
        const uint32_t lo = fmix32(fmix32(h0 ^ (uint32_t)game_id) + (uint32_t)(game_id >> 32));
        const uint32_t hi = fmix32(fmix32(h1 ^ (uint32_t)(game_id >> 32)) + (uint32_t)game_id * 0x9E3779B1u);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
    static TAFL_HD uint32_t sim_key(uint64_t gk, uint32_t sim) {
        return fmix32((uint32_t)gk ^ (sim * 0x9E3779B1u + 0x7F4A7C15u)) ^ fmix32((uint32_t)(gk >> 32) + sim * 0x85EBCA77u + 0x165667B1u);
    }
    // Leaf key of the search's playouts (include/taflhip.h "leaf key"): MurmurHash3_x86_32 over the position as layout-independent words -
    // per row (attacker bits) | (defender bits) << 16, the repetition records, turn, repetition counts, side / mid-pair flags / king tile.
    // The same value in every board layout (reference stride or dense 13 columns).      oracle: orc_state_hash
    static TAFL_HD uint32_t mm3_word(uint32_t h, uint32_t k) {
        k *= 0xCC9E2D51u; k = (k << 15) | (k >> 17); k *= 0x1B873593u;
        h ^= k; h = (h << 13) | (h >> 19); return h * 5u + 0xE6546B64u;
    }
    template <int R>
    static TAFL_HD uint32_t row_of(const Bits<NL>& a) {
        constexpr int p = R * W;
        uint32_t v = a.w[p >> 5] >> (p & 31);
        if constexpr ((p & 31) + W > 32 && (p >> 5) + 1 < NL) v |= a.w[(p >> 5) + 1] << (32 - (p & 31));
        return v & ((1u << W) - 1u);
    }
    template <int R>
    static TAFL_HD uint32_t hash_rows(const S& st, uint32_t n, uint32_t h) {
        if constexpr (R < W) {
            if ((uint32_t)R < n) h = mm3_word(h, (row_of<R>(st.att) & ((1u << n) - 1u)) | ((row_of<R>(st.def) & ((1u << n) - 1u)) << 16));
            return hash_rows<R + 1>(st, n, h);
        } else return h;
    }
    static TAFL_HD uint32_t state_hash(const S& st, const K& C) {
        uint32_t h = hash_rows<0>(st, C.n, C.n);
        TAFL_UNROLL for (int i = 0; i < 4; ++i) h = mm3_word(h, st.rep[i]);
        h = mm3_word(h, st.turn);
        h = mm3_word(h, st.reps);
        h = mm3_word(h, st.flags & 0x00FF0007u);
        h ^= (C.n + 7u) * 4u;
        return fmix32(h);
    }
    static TAFL_HD uint32_t ply_rand(uint32_t sk, uint32_t ply) { return fmix32(sk + ply * 0x85EBCA77u); }
    static TAFL_HD uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

    // one seeded uniform-random playout, state modified in place.  value for `start_side`.
    static TAFL_HD void rollout(S& st, uint32_t sk, uint32_t max_plies, const K& C, tafl_rollout_result& res) {
        const uint32_t start_side = st.flags & TAFL_F_SIDE;
        Moves<NL> mv;
        movegen(st, start_side, C, mv);
        uint32_t ply = 0; bool stuck = false;
        while (ply < max_plies && TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING) {
            if (mv.total == 0) { stuck = true; break; }
            const uint32_t idx = mulhi(ply_rand(sk, ply), mv.total);
            const Move m = pick_rollout(st, mv, idx, C);
            Moves<NL> nx;
            apply(st, m, C, nullptr, nx);
            mv = nx;
            ++ply;
        }
        finish_rollout(st, start_side, ply, stuck, res);
    }
    static TAFL_HD void finish_rollout(const S& st, uint32_t start_side, uint32_t ply, bool stuck, tafl_rollout_result& res) {
        const uint32_t status = TAFL_F_STATUS(st.flags);
        res.plies = ply; res.status = (uint8_t)status; res.winner = (uint8_t)(TAFL_F_WINNER(st.flags) ? TAFL_DEFENDER : TAFL_ATTACKER);
        if (status == TAFL_STATUS_WIN) { res.value = (int8_t)(TAFL_F_WINNER(st.flags) == start_side ? 1 : -1); res.reason = (uint8_t)TAFL_F_REASON(st.flags); }
        else if (status == TAFL_STATUS_DRAW) { res.value = 0; res.reason = (uint8_t)(8 + TAFL_F_REASON(st.flags)); res.winner = 0; }
        else { res.value = 0; res.reason = (uint8_t)(stuck ? TAFL_ROLLOUT_REASON_STUCK : TAFL_ROLLOUT_REASON_PLY_CAP); res.winner = 0; }
    }
};

// ---- construction of Consts: constexpr, so that kernels specialised for a preset get every mask as a literal ------
template <int NL, int W>
constexpr Consts<NL> make_consts_ct(const tafl_rules& r, uint32_t n) {
    Consts<NL> C{};
    for (uint32_t rr = 0; rr < n; ++rr) for (uint32_t cc = 0; cc < n; ++cc) {
        const uint32_t i = rr * (uint32_t)W + cc; const uint32_t wi = i >> 5, b = 1u << (i & 31);
        C.board.w[wi] |= b;
        if (cc == 0) C.col0.w[wi] |= b;
        if (cc == n - 1) C.coln.w[wi] |= b;
        if (rr == 0) C.row0.w[wi] |= b;
        if (rr == n - 1) C.rown.w[wi] |= b;
        if (cc == 0 || rr == 0 || cc == n - 1 || rr == n - 1) C.edge.w[wi] |= b;
        if ((rr == 0 || rr == n - 1) && (cc == 0 || cc == n - 1)) C.corners.w[wi] |= b;
    }
    const uint32_t t = n / 2;                                         // SpecialTiles::from, geometry.rs:13-25
    C.throne_sq = t * (uint32_t)W + t;
    C.throne.w[C.throne_sq >> 5] |= 1u << (C.throne_sq & 31);
    const uint32_t nb[4] = {(t - 1) * W + t, t * W + t - 1, (t + 1) * W + t, t * W + t + 1};
    const bool nbok[4] = {t >= 1, t >= 1, t + 1 < n, t + 1 < n};
    for (int k = 0; k < 4; ++k) if (nbok[k]) C.throne_nb.w[nb[k] >> 5] |= 1u << (nb[k] & 31);
    C.n = n; C.w = (uint32_t)W; C.rules = r;
    const uint32_t psbit[3] = {1u /*Soldier<<0*/, 9u /*Soldier<<8*/, 8u /*King<<8*/};
    for (int c = 0; c < 3; ++c) {
        const bool is_king = c == CLS_KING;
        const bool enter = ((r.may_enter_corners >> psbit[c]) & 1u) != 0;
        const bool hthr = ((r.hostility_throne >> psbit[c]) & 1u) != 0, hcor = ((r.hostility_corners >> psbit[c]) & 1u) != 0;
        const bool no_land_throne = r.throne_movement == TAFL_THRONE_NOENTRY || (r.throne_movement == TAFL_THRONE_KINGENTRY && !is_king);
        const bool no_pass_throne = r.throne_movement == TAFL_THRONE_NOPASS || (r.throne_movement == TAFL_THRONE_KINGPASS && !is_king);
        for (int i = 0; i < NL; ++i) {
            uint32_t lf = 0, pf = 0, hs = 0;
            if (!enter) { lf |= C.corners.w[i]; pf |= C.corners.w[i]; }
            if (no_land_throne) lf |= C.throne.w[i];
            if (no_pass_throne) pf |= C.throne.w[i];
            if (hthr) hs |= C.throne.w[i];
            if (hcor) hs |= C.corners.w[i];
            C.land_forbid[c].w[i] = lf; C.pass_forbid[c].w[i] = pf; C.hostile_special[c].w[i] = hs;
        }
        C.slow[c] = ((r.slow_pieces >> psbit[c]) & 1u) ? 1u : 0u;
        C.edge_hostile[c] = ((r.hostility_edge >> psbit[c]) & 1u) ? 1u : 0u;
    }
    bool same = C.slow[CLS_KING] == C.slow[CLS_DEF];
    for (int i = 0; i < NL; ++i) same = same && C.land_forbid[CLS_KING].w[i] == C.land_forbid[CLS_DEF].w[i] && C.pass_forbid[CLS_KING].w[i] == C.pass_forbid[CLS_DEF].w[i];
    C.king_like_soldier = same ? 1u : 0u;
    return C;
}
template <int NL, int W>
inline int make_consts(const tafl_rules& r, uint32_t n, Consts<NL>& C) {
    if (n < 3 || n > (uint32_t)W || n > 15) return -1;
    C = make_consts_ct<NL, W>(r, n);
    return 0;
}

// ---- presets known at compile time (game/preset.rs:12-56): kernels specialised on these fold all rule tests -------
constexpr tafl_rules rules_copenhagen_ct() {
    tafl_rules r{};
    r.king_strength = TAFL_KING_STRONG; r.king_attack = TAFL_KING_ARMED; r.has_shieldwall = 1; r.sw_corners_may_close = 1;
    r.sw_captures = TAFL_PS_TYPE(TAFL_PT_SOLDIER); r.exit_fort = 1; r.throne_movement = TAFL_THRONE_KINGENTRY;
    r.may_enter_corners = TAFL_PS_TYPE(TAFL_PT_KING); r.hostility_throne = TAFL_PS_ALL; r.hostility_corners = TAFL_PS_TYPE(TAFL_PT_SOLDIER);
    r.starting_side = TAFL_ATTACKER; r.enclosure_win = TAFL_ENCL_WITHOUT_EDGE_ACCESS; r.has_repetition_rule = 1; r.n_repetitions = 3; r.rep_is_loss = 1;
    return r;
}
constexpr tafl_rules rules_brandubh_ct() {
    tafl_rules r{};
    r.king_strength = TAFL_KING_STRONG_BY_THRONE; r.king_attack = TAFL_KING_ARMED; r.throne_movement = TAFL_THRONE_KINGENTRY;
    r.may_enter_corners = TAFL_PS_TYPE(TAFL_PT_KING); r.hostility_throne = TAFL_PS_TYPE(TAFL_PT_SOLDIER); r.hostility_corners = TAFL_PS_ALL;
    r.starting_side = TAFL_ATTACKER; r.enclosure_win = TAFL_ENCL_WITHOUT_EDGE_ACCESS; r.has_repetition_rule = 1; r.n_repetitions = 3; r.rep_is_loss = 1;
    return r;
}
// PRESET: 0 = run-time rules (Consts passed as a kernel argument), 1 = Copenhagen 11x11 / u128, 2 = Brandubh 7x7 / u64,
//         3 = Copenhagen 13x13 / U256
enum : int { PRESET_NONE = 0, PRESET_COPENHAGEN11 = 1, PRESET_BRANDUBH7 = 2, PRESET_COPENHAGEN13 = 3 };
template <int NL, int W, int PRESET>
constexpr Consts<NL> preset_consts() {
    if constexpr (PRESET == PRESET_COPENHAGEN11 && NL == 4 && W == 11) return make_consts_ct<NL, W>(rules_copenhagen_ct(), 11);
    else if constexpr (PRESET == PRESET_BRANDUBH7 && NL == 2 && W == 7) return make_consts_ct<NL, W>(rules_brandubh_ct(), 7);
    else if constexpr (PRESET == PRESET_COPENHAGEN13 && NL == 8 && W == 15) return make_consts_ct<NL, W>(rules_copenhagen_ct(), 13);
    else if constexpr (PRESET == PRESET_COPENHAGEN13 && NL == 6 && W == 13) return make_consts_ct<NL, W>(rules_copenhagen_ct(), 13);   // dense search layout
    else return Consts<NL>{};
}
inline bool rules_equal(const tafl_rules& a, const tafl_rules& b) {
    return a.edge_escape == b.edge_escape && a.king_strength == b.king_strength && a.king_attack == b.king_attack
        && a.has_shieldwall == b.has_shieldwall && (!a.has_shieldwall || (a.sw_corners_may_close == b.sw_corners_may_close && a.sw_captures == b.sw_captures))
        && a.exit_fort == b.exit_fort && a.throne_movement == b.throne_movement && a.starting_side == b.starting_side
        && a.enclosure_win == b.enclosure_win && a.has_repetition_rule == b.has_repetition_rule
        && (!a.has_repetition_rule || (a.rep_is_loss == b.rep_is_loss && a.n_repetitions == b.n_repetitions))
        && a.draw_on_no_plays == b.draw_on_no_plays && a.linnaean_capture == b.linnaean_capture
        && a.may_enter_corners == b.may_enter_corners && a.hostility_throne == b.hostility_throne
        && a.hostility_corners == b.hostility_corners && a.hostility_edge == b.hostility_edge && a.slow_pieces == b.slow_pieces;
}
inline int detect_preset(const tafl_rules& r, uint32_t n, uint32_t word_bits) {
    if (n == 11 && word_bits == 128 && rules_equal(r, rules_copenhagen_ct())) return PRESET_COPENHAGEN11;
    if (n == 7 && word_bits == 64 && rules_equal(r, rules_brandubh_ct())) return PRESET_BRANDUBH7;
    if (n == 13 && word_bits == 256 && rules_equal(r, rules_copenhagen_ct())) return PRESET_COPENHAGEN13;
    return PRESET_NONE;
}

// ---- the same position with another row stride ----------------------------------------------------------------------------------
// The reference stores a 13x13 board in a U256 with 15 columns per row (8 limbs of 32 bits).  Searches and playouts of that preset run
// on a dense 13-column layout (169 bits = 6 limbs): a quarter less work in every multi-limb operation and a register file that holds
// the playout loop without spilling.  Tile (row, col), the king's (row, col) in `flags` and the repetition ring (it records row / col)
// do not depend on the stride; only the two board words are re-packed, row by row, when a position enters or leaves that layout.
template <int NLS, int WS, int NLD, int WD, int R0 = 0>
TAFL_HD void restride_rows(const Bits<NLS>& a, uint32_t n, Bits<NLD>& d) {
    constexpr int WMIN = WS < WD ? WS : WD;
    if constexpr (R0 < WMIN) {
        if ((uint32_t)R0 < n) {
            constexpr int ps = R0 * WS, pd = R0 * WD;
            uint32_t v = a.w[ps >> 5] >> (ps & 31);
            if constexpr ((ps & 31) + WMIN > 32 && (ps >> 5) + 1 < NLS) v |= a.w[(ps >> 5) + 1] << (32 - (ps & 31));
            v &= (1u << WMIN) - 1u;
            d.w[pd >> 5] |= v << (pd & 31);
            if constexpr ((pd & 31) + WMIN > 32 && (pd >> 5) + 1 < NLD) d.w[(pd >> 5) + 1] |= v >> (32 - (pd & 31));
        }
        restride_rows<NLS, WS, NLD, WD, R0 + 1>(a, n, d);
    }
}
template <int NLS, int WS, int NLD, int WD>
TAFL_HD void restride(const DState<NLS>& s, uint32_t n, DState<NLD>& d) {
    d.att = bz<NLD>(); d.def = bz<NLD>();
    restride_rows<NLS, WS, NLD, WD>(s.att, n, d.att); restride_rows<NLS, WS, NLD, WD>(s.def, n, d.def);
    TAFL_UNROLL for (int i = 0; i < 4; ++i) d.rep[i] = s.rep[i];
    d.turn = s.turn; d.psc = s.psc; d.reps = s.reps; d.flags = s.flags;
}
template <int WS, int WD> TAFL_HD uint32_t restride_sq(uint32_t sq) { const uint32_t r = div_w<WS>(sq); return mul24(r, (uint32_t)WD) + (sq - mul24(r, (uint32_t)WS)); }

// ---- ABI <-> device state conversion (host side) -------------------------------------------------------------------------
template <int NL>
inline void state_from_abi(const tafl_state& a, DState<NL>& s) {
    constexpr int L64 = NL / 2;
    for (int i = 0; i < L64; ++i) {
        s.att.w[2 * i] = (uint32_t)a.att[i]; s.att.w[2 * i + 1] = (uint32_t)(a.att[i] >> 32);
        s.def.w[2 * i] = (uint32_t)a.def[i]; s.def.w[2 * i + 1] = (uint32_t)(a.def[i] >> 32);
    }
    const uint32_t krow = s.def.w[NL - 1] >> 28, kcol = s.att.w[NL - 1] >> 28;     // get_king, board/state.rs:127-131
    s.att.w[NL - 1] &= 0x0FFFFFFFu; s.def.w[NL - 1] &= 0x0FFFFFFFu;
    for (int i = 0; i < 4; ++i) s.rep[i] = a.rep_ring[i];
    s.turn = a.turn; s.psc = a.plays_since_capture;
    s.reps = (uint32_t)a.attacker_reps | ((uint32_t)a.defender_reps << 16);
    s.flags = (a.side_to_play ? TAFL_F_SIDE : 0u) | (a.attacker_mid_pair ? TAFL_F_AMID : 0u) | (a.defender_mid_pair ? TAFL_F_DMID : 0u)
            | ((uint32_t)(a.status & 3) << 3) | ((uint32_t)(a.reason & 15) << 5) | ((a.winner ? 1u : 0u) << 9) | (krow << 16) | (kcol << 20);
}
template <int NL>
inline void state_to_abi(const DState<NL>& s, uint8_t side_len, tafl_state& a) {
    constexpr int L64 = NL / 2;
    a = tafl_state{};
    uint32_t aw[NL], dw[NL];
    for (int i = 0; i < NL; ++i) { aw[i] = s.att.w[i]; dw[i] = s.def.w[i]; }
    aw[NL - 1] = (aw[NL - 1] & 0x0FFFFFFFu) | (TAFL_F_KCOL(s.flags) << 28);
    dw[NL - 1] = (dw[NL - 1] & 0x0FFFFFFFu) | (TAFL_F_KROW(s.flags) << 28);
    for (int i = 0; i < L64; ++i) {
        a.att[i] = (uint64_t)aw[2 * i] | ((uint64_t)aw[2 * i + 1] << 32);
        a.def[i] = (uint64_t)dw[2 * i] | ((uint64_t)dw[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; ++i) a.rep_ring[i] = s.rep[i];
    a.turn = s.turn; a.plays_since_capture = s.psc;
    a.attacker_reps = (uint16_t)(s.reps & 0xFFFFu); a.defender_reps = (uint16_t)(s.reps >> 16);
    a.attacker_mid_pair = (s.flags & TAFL_F_AMID) ? 1 : 0; a.defender_mid_pair = (s.flags & TAFL_F_DMID) ? 1 : 0;
    a.side_to_play = (s.flags & TAFL_F_SIDE) ? TAFL_DEFENDER : TAFL_ATTACKER;
    a.status = (uint8_t)TAFL_F_STATUS(s.flags); a.reason = (uint8_t)TAFL_F_REASON(s.flags);
    a.winner = (uint8_t)((a.status == TAFL_STATUS_WIN && TAFL_F_WINNER(s.flags)) ? TAFL_DEFENDER : TAFL_ATTACKER);
    a.side_len = side_len;
}

}  // namespace tafl
