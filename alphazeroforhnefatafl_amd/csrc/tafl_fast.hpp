// tafl_fast.hpp — the playout (random rollout) engine: same results as Engine::rollout, far fewer instructions.
//
// Idea: a rook ray in the "+1" bit direction of a packed board falls out of ONE multi-word subtraction
//        attacks = ((occ | line_starts) - 2*sliders) XOR (occ | line_starts)
// for all sliders at once (the borrow of each slider ripples through the empty tiles up to its first blocker; the
// first tile of the next line is made a virtual blocker so that a borrow never leaves its line).  So the board is kept
// in two layouts — N (bit = row*W+col, the reference's) and T (bit = col*W+row) — and their bit-reversals are formed
// on the fly (v_bfrev_b32): in each of the four layouts one ray direction is "+1":
//        T : V+ (row+)      rev(T) : V- (row-)      N : H+ (col+)      rev(N) : H- (col-)
// which also defines the ROLLOUT ORDER (ascending bit index inside each direction's own layout; oracle: rollout_key()).
// Captures / shieldwall / king logic / enclosure / exit fort / repetition are the Engine's code on the N layout, so the
// only new logic here is move generation, move picking and the upkeep of the T layout.
//
// The enclosure flood (logic.rs:720-734) is skipped when it provably cannot succeed: if any defender stands on an edge
// tile or has a play that lands on one, then either that piece shares the king's region — which then touches an edge and
// `find_enclosure(.., abort_on_edge = true, ..)` is None — or it is outside it and `occupied.len() == count(Defender)`
// fails.  The opponent's plays are generated anyway (no-plays test + next ply), so the filter is free.
//
// Preconditions on the rules (checked by fast_ok(), else the generic Engine::rollout runs): no slow pieces, and an empty
// throne may be crossed by every piece (throne_movement in {NoThrone, NoEntry, KingEntry}).
#pragma once
#include "tafl_core.hpp"

namespace tafl {

template <int NL> constexpr uint32_t bitrev32_ct(uint32_t v) {
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
TAFL_HD uint32_t bitrev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
// whole-word reversal: bit i -> bit NL*32-1-i
template <int NL> TAFL_HD Bits<NL> rev(const Bits<NL>& a) { Bits<NL> o; TAFL_UNROLL for (int i = 0; i < NL; ++i) o.w[i] = bitrev32(a.w[NL - 1 - i]); return o; }
template <int NL> constexpr Bits<NL> rev_ct(const Bits<NL>& a) { Bits<NL> o{}; for (int i = 0; i < NL; ++i) o.w[i] = bitrev32_ct<NL>(a.w[NL - 1 - i]); return o; }
// a - b over NL limbs
template <int NL> TAFL_HD Bits<NL> sub(const Bits<NL>& a, const Bits<NL>& b) {
    Bits<NL> o; uint32_t borrow = 0;
    TAFL_UNROLL for (int i = 0; i < NL; ++i) {
        const uint64_t d = (uint64_t)a.w[i] - (uint64_t)b.w[i] - (uint64_t)borrow;
        o.w[i] = (uint32_t)d; borrow = (uint32_t)(d >> 63);
    }
    return o;
}

// constants of the fast engine, all derived from Consts (literals when Consts is constexpr)
template <int NL>
struct FastConsts {
    Bits<NL> vb, vbr;        // virtual blockers: first tile of every line (N/T layouts) / reversed layouts
    Bits<NL> em, emr;        // legal "+1" destinations: board minus the line starts
    Bits<NL> lfs[2], lfsr[2];  // landing forbidden: attacker soldier, defender soldier
    Bits<NL> lfk, lfkr;      // landing forbidden: king
    Bits<NL> edge, edger;
    // the king's plays from the soldiers' fill (Fast::gen): possible when every tile open to a defender soldier is open to the king too;
    // ks_corner / ks_throne: ... and the king may, unlike the soldiers, stop on the corners / on the throne
    bool king_lines, ks_corner, ks_throne;
};
template <int NL>
constexpr FastConsts<NL> make_fast_consts(const Consts<NL>& C) {
    FastConsts<NL> F{};
    Bits<NL> emr_n{};
    for (int i = 0; i < NL; ++i) {
        F.vb.w[i] = C.col0.w[i]; F.em.w[i] = C.board.w[i] & ~C.col0.w[i];
        emr_n.w[i] = C.board.w[i] & ~C.coln.w[i];
        F.lfs[0].w[i] = C.land_forbid[CLS_ATT].w[i]; F.lfs[1].w[i] = C.land_forbid[CLS_DEF].w[i]; F.lfk.w[i] = C.land_forbid[CLS_KING].w[i];
        F.edge.w[i] = C.edge.w[i];
    }
    F.vbr = rev_ct<NL>(C.coln); F.emr = rev_ct<NL>(emr_n);
    F.lfsr[0] = rev_ct<NL>(F.lfs[0]); F.lfsr[1] = rev_ct<NL>(F.lfs[1]); F.lfkr = rev_ct<NL>(F.lfk); F.edger = rev_ct<NL>(F.edge);
    F.king_lines = true; F.ks_corner = false; F.ks_throne = false;
    for (int i = 0; i < NL; ++i) {
        if (F.lfk.w[i] & ~F.lfs[1].w[i]) F.king_lines = false;
        const uint32_t only_king = F.lfs[1].w[i] & ~F.lfk.w[i];
        if (only_king & C.corners.w[i]) F.ks_corner = true;
        if (only_king & C.throne.w[i]) F.ks_throne = true;
        if (only_king & ~(C.corners.w[i] | C.throne.w[i])) F.king_lines = false;       // (cannot happen: land_forbid is made of corners and throne)
    }
    return F;
}
template <int NL>
constexpr bool fast_ok(const Consts<NL>& C) {
    if (C.slow[0] || C.slow[1] || C.slow[2]) return false;
    const uint8_t t = C.rules.throne_movement;
    return t == TAFL_THRONE_NOTHRONE || t == TAFL_THRONE_NOENTRY || t == TAFL_THRONE_KINGENTRY;
}

template <int NL, int W>
struct Fast {
    using E = Engine<NL, W>;
    using B = Bits<NL>;
    using S = DState<NL>;
    using K = Consts<NL>;
    using F = FastConsts<NL>;
    static constexpr uint32_t Z = NL * 32 - 1;

    struct Gen {                 // destination sets, each in the layout of its own direction
        B r[4];
        uint32_t cnt[4], total;
        bool edge_hit;           // some play of the side lands on an edge tile
    };

    static TAFL_HD uint32_t n_to_t(uint32_t idx) { const uint32_t q = div_w<W>(idx), m = idx - mul24(q, (uint32_t)W); return mul24(m, (uint32_t)W) + q; }

    // all rays in the +1 direction: attack set (first blocker included) of every slider at once
    static TAFL_HD B fill(const B& occ, const B& sliders, const B& vblock) {
        const B op = occ | vblock;
        return sub(op, shl<1>(sliders)) ^ op;
    }
    // plays of `side` (0 attacker / 1 defender) in one layout: occ/mine/kbit in that layout
    static TAFL_HD B reach1(const B& occ, const B& mine, const B& kbit, const B& vblock, const B& em, const B& lfs, const B& lfk, bool with_king) {
        const B open = andn(em, occ);
        if (!with_king) return andn(fill(occ, mine, vblock) & open, lfs);
        // soldiers and king separately: they differ only in the tiles they may stop on (pieces of either kind block alike)
        return andn(fill(occ, andn(mine, kbit), vblock) & open, lfs) | andn(fill(occ, kbit, vblock) & open, lfk);
    }
    // n (<= 15) bits of `a` from bit `base`
    static TAFL_HD uint32_t line_bits(const B& a, uint32_t base, uint32_t n) { return (uint32_t)field64<0>(a, base) & ((1u << n) - 1u); }

    // The king's plays onto the tiles only he may stop on (corners, throne): FastConsts::king_lines.  Everywhere else the king moves like a
    // defender soldier (fast_ok: every piece may cross the empty throne), so gen() lets him slide with the soldiers in ONE fill per direction
    // - with `~lfs` taking the special tiles out of the result - and adds here what only he may do: a special tile lies at the end of an
    // edge line (corner) or in the middle of the centre line (throne), the king reaches it iff he stands on that line and every tile from
    // his neighbour up to and including the target is empty.  Lines are read from the layout in which they are contiguous (rows from N,
    // columns from T), at the three fixed positions 0, n/2, n-1.  At most twelve (tile, direction) pairs, each one bit of one limb for a
    // preset (the indices are literals then) - instead of four more 128-bit subtraction fills for a single piece.
    static TAFL_HD void king_specials(const S& st, const B& occN, const B& occT, const K& C, const F& fc, B r[4]) {
        const uint32_t kr = TAFL_F_KROW(st.flags), kc = TAFL_F_KCOL(st.flags), n = C.n, L = n - 1u, c = div_w<W>(C.throne_sq);
        const uint32_t lw = mul24(L, (uint32_t)W), cw = mul24(c, (uint32_t)W);
        const bool kp = kr < n && kc < n && test(st.def, mul24(kr, (uint32_t)W) + kc);      // the king is on the board
        const uint32_t rowocc = kr == 0u ? line_bits(occN, 0u, n) : kr == c ? line_bits(occN, cw, n) : line_bits(occN, lw, n);
        const uint32_t colocc = kc == 0u ? line_bits(occT, 0u, n) : kc == c ? line_bits(occT, cw, n) : line_bits(occT, lw, n);
        const uint32_t lo_r = rowocc & ((1u << kc) - 1u), hi_r = rowocc >> (kc + 1u);      // the row below / above the king's column
        const uint32_t lo_c = colocc & ((1u << kr) - 1u), hi_c = colocc >> (kr + 1u);      // the column below / above the king's row
        auto put = [](B& set, uint32_t idx, bool cond) { set |= gate(bit_at<NL>(idx), cond); };
        if (fc.ks_corner) {
            const bool r0 = kp && kr == 0u, rL = kp && kr == L, c0 = kp && kc == 0u, cL = kp && kc == L;
            const bool hm = kc > 0u && lo_r == 0u, hp = kc < L && hi_r == 0u, vm = kr > 0u && lo_c == 0u, vp = kr < L && hi_c == 0u;
            put(r[3], Z, r0 && hm);             put(r[3], Z - lw, rL && hm);          // H-: (kr, 0), reversed N index
            put(r[2], L, r0 && hp);             put(r[2], lw + L, rL && hp);          // H+: (kr, n-1)
            put(r[1], Z, c0 && vm);             put(r[1], Z - lw, cL && vm);          // V-: (0, kc), reversed T index (bit = col * W + row)
            put(r[0], L, c0 && vp);             put(r[0], lw + L, cL && vp);          // V+: (n-1, kc)
        }
        if (fc.ks_throne) {
            const bool rc = kp && kr == c, cc = kp && kc == c;
            put(r[2], cw + c, rc && kc < c && (hi_r & ((1u << ((c - kc) & 31u)) - 1u)) == 0u);     // columns kc+1 .. c empty
            put(r[3], Z - (cw + c), rc && kc > c && (lo_r >> c) == 0u);                            // columns c .. kc-1 empty
            put(r[0], cw + c, cc && kr < c && (hi_c & ((1u << ((c - kr) & 31u)) - 1u)) == 0u);
            put(r[1], Z - (cw + c), cc && kr > c && (lo_c >> c) == 0u);
        }
    }

    static TAFL_HD void gen(const S& st, const B& attT, const B& defT, uint32_t side, const K& C, const F& fc, Gen& g) {
        const B occN = (st.att | st.def) & C.board, occT = (attT | defT) & C.board;
        const B mineN = sel(side != 0, st.def, st.att) & C.board, mineT = sel(side != 0, defT, attT) & C.board;
        const uint32_t k = E::king_sq(st, C);
        B kN = bz<NL>(), kT = bz<NL>();
        // the king's own rays are needed only where a defender is to move (and only under rules that do not allow king_specials):
        // skipped when no game of the wave is in that case
        const bool any_king_lane = !fc.king_lines && wave_any(side != 0 && k != TAFL_NO_SQ);
        if (any_king_lane) {
            const bool kalive = side && k != TAFL_NO_SQ;
            const uint32_t ks = kalive ? k : 0u;
            kN = gate(bit_at<NL>(ks) & st.def, kalive); kT = gate(bit_at<NL>(n_to_t(ks)) & defT, kalive);
        }
        const B lfs = blend(side != 0, fc.lfs[1], fc.lfs[0]), lfsr = blend(side != 0, fc.lfsr[1], fc.lfsr[0]);
        g.r[0] = reach1(occT, mineT, kT, fc.vb, fc.em, lfs, fc.lfk, any_king_lane);
        g.r[1] = reach1(rev(occT), rev(mineT), rev(kT), fc.vbr, fc.emr, lfsr, fc.lfkr, any_king_lane);
        g.r[2] = reach1(occN, mineN, kN, fc.vb, fc.em, lfs, fc.lfk, any_king_lane);
        g.r[3] = reach1(rev(occN), rev(mineN), rev(kN), fc.vbr, fc.emr, lfsr, fc.lfkr, any_king_lane);
        if (side != 0 && fc.king_lines && (fc.ks_corner || fc.ks_throne)) king_specials(st, occN, occT, C, fc, g.r);
        g.total = 0;
        TAFL_UNROLL for (int d = 0; d < 4; ++d) { g.cnt[d] = popc(g.r[d]); g.total += g.cnt[d]; }
        // (only the enclosure filter reads it: behind an attacker's play, i.e. for a defender's move set)
        g.edge_hit = side != 0 && any(((g.r[0] | g.r[2]) & fc.edge) | ((g.r[1] | g.r[3]) & fc.edger));
    }

    // idx-th play in ROLLOUT ORDER
    static TAFL_HD Move pick(const S& st, const B& attT, const B& defT, const Gen& g, uint32_t idx, const K& C) {
        // direction by prefix sums, branch-free
        const uint32_t c0 = g.cnt[0], c1 = c0 + g.cnt[1], c2 = c1 + g.cnt[2];
        const uint32_t d = (uint32_t)(idx >= c0) + (uint32_t)(idx >= c1) + (uint32_t)(idx >= c2);
        idx -= d == 0 ? 0u : d == 1 ? c0 : d == 2 ? c1 : c2;
        const bool odd = (d & 1u) != 0, horiz = d >= 2;
        const B rsel = blend(horiz, blend(odd, g.r[3], g.r[2]), blend(odd, g.r[1], g.r[0]));
        const B occb = blend(horiz, st.att | st.def, attT | defT) & C.board;
        const B occ = blend(odd, rev(occb), occb);
        const uint32_t x = nth_set_bit(rsel, idx);                       // destination, layout index
        const uint32_t s = msb(occ & below<NL>(x));                      // nearest piece behind it = the mover
        const uint32_t ux = odd ? Z - x : x, us = odd ? Z - s : s;       // un-reverse
        const uint32_t qx = div_w<W>(ux), mx = ux - mul24(qx, (uint32_t)W), qs = div_w<W>(us), ms = us - mul24(qs, (uint32_t)W);
        Move m;
        m.to = horiz ? ux : mul24(mx, (uint32_t)W) + qx;                       // T layout (col,row) -> N index
        m.from = horiz ? us : mul24(ms, (uint32_t)W) + qs;
        m.dir = d; m.dist = odd ? us - ux : ux - us;
        return m;
    }

    static TAFL_HD void transpose_in(const B& n, B& t) {
        t = bz<NL>();
        B r = n;
        while (any(r)) { const uint32_t i = lsb(r); r = andn(r, bit_at<NL>(i)); t |= bit_at<NL>(n_to_t(i)); }
    }

    // ---- exit-fort candidate test (exact pre-filter of Engine::exit_fort, logic.rs:572-601) ---------------------------------------------
    // detect_exit_fort floods {king tile} U {empty tiles} from the king's tile, which must lie on an edge, and gives up (None) as soon as
    // the region touches an attacker (row_col_enclosed, logic.rs:288-291) or holds a corner (abort_on_corner, :344-346,:368-370).  The
    // tiles of the king's EDGE LINE next to him belong to that region as long as they are empty (4-connected), so walking from the king
    // along the edge in either direction the flood meets, first of all, either
    //   - an attacker                          -> None,
    //   - the end of the line = a corner tile that is empty (the walk only passes empty tiles) -> None, or
    //   - a defender                           -> a boundary piece: the only case in which a fort is still possible.
    // Hence: "on BOTH sides of the king the nearest occupied tile of his edge line exists and is a defender" is necessary for a fort
    // (a king on a corner has an empty side: no fort, as in Engine::exit_fort where the start tile is a corner).  In random play the test
    // holds for one defender ply in ~10^3 (tools/playout_event_stats.py), so that the rings and the flood of Engine::exit_fort - paid by
    // every wave on every defender ply before - are entered by a wave only now and then.  The line is read from the layout in which it is
    // contiguous: rows 0 / n-1 from N, columns 0 / n-1 from T (= rows 0 / n-1 of T).
    static TAFL_HD bool fort_candidate(const S& st, const B& attT, const B& defT, const K& C) {
        const uint32_t kr = TAFL_F_KROW(st.flags), kc = TAFL_F_KCOL(st.flags), n = C.n, last = n - 1u, lb = mul24(last, (uint32_t)W);
        const bool top = kr == 0u, bot = kr == last, lef = kc == 0u, rig = kc == last;
        const bool rowline = top || bot;
        // the four edge lines at fixed positions (literals for a preset), then 2 x 3 selects of n-bit values instead of selects of whole words
        const uint32_t a0 = line_bits(st.att, 0u, n), a1 = line_bits(st.att, lb, n), a2 = line_bits(attT, 0u, n), a3 = line_bits(attT, lb, n);
        const uint32_t d0 = line_bits(st.def, 0u, n), d1 = line_bits(st.def, lb, n), d2 = line_bits(defT, 0u, n), d3 = line_bits(defT, lb, n);
        const uint32_t la = top ? a0 : bot ? a1 : lef ? a2 : a3, ld = top ? d0 : bot ? d1 : lef ? d2 : d3;
        const uint32_t pos = rowline ? kc : kr;                                       // the king's place on his line (< n <= 15)
        const uint32_t occ = la | ld;
        const uint32_t above = occ >> (pos + 1u), below = occ & ((1u << pos) - 1u);
        const uint32_t first_above = above & (0u - above);                           // nearest occupied tile on the high side
        const bool ok_hi = (first_above & (ld >> (pos + 1u))) != 0u;                 // ... exists and is a defender
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(below | 1u);                // nearest occupied tile on the low side (below != 0)
        const bool ok_lo = below != 0u && ((ld >> hb) & 1u) != 0u;
        return (top || bot || lef || rig) && kr < n && kc < n && ok_hi && ok_lo;
    }

    // ---- one ply of a playout with the mover known at compile time -----------------------------------------------------------------------
    // Every ply flips the side to move, so the playout loop below runs as pairs of half-iterations, the first for an attacker's play, the
    // second for a defender's: inside a half MOVER is a literal, and every `mover ? a : b` of apply_pre / captures / track / gen /
    // outcome_early folds away - with it the code of the other side: the king's own rays and the enclosure test exist only behind an
    // attacker's play, the king-escape and exit-fort tests only behind a defender's, the king-capture block only in the attacker's half.
    // 64 games share an instruction stream: before, a wave paid for both sides' code on every ply.
    // g: plays of MOVER on entry, of the other side on exit; rk: the ply's RNG word before mixing (sk + ply * 0x85EBCA77, kept incrementally).
    template <uint32_t MOVER>
    static TAFL_HD void ply_of(S& st, B& attT, B& defT, Gen& g, uint32_t& rk, const K& C, const F& fc) {
        TAFL_PROF_COUNT(31); TAFL_STAT_HIT(31);
        TAFL_PROF_BEGIN(10); TAFL_PROF_SPLIT(10); TAFL_PROF_END(11);      // two empty sections: the cost of a mark
        TAFL_PROF_BEGIN(0);
        const uint32_t idx = E::mulhi(E::fmix32(rk), g.total);
        rk += 0x85EBCA77u;
        const Move m = pick(st, attT, defT, g, idx, C);
        TAFL_PROF_SPLIT(0);
        typename E::ApplyCtx ax;
        E::apply_pre(st, m, C, ax, MOVER);
        TAFL_PROF_SPLIT(4);
        // T layout upkeep: the move, the custodial captures (V+- are +-1 here, H+- are +-W), then whatever else was
        // captured (shieldwall, king, Linnaean: rare)
        {
            constexpr int BK = 2 * W;
            const uint32_t tT = n_to_t(m.to);
            const B mvT = bit_at<NL>(n_to_t(m.from)) | bit_at<NL>(tT);
            if constexpr (MOVER != 0) defT = defT ^ mvT; else attT = attT ^ mvT;
            const uint32_t cu = ax.cust;
            const uint64_t cf = ((uint64_t)(cu & 1u) << (BK + 1)) | ((uint64_t)((cu >> 1) & 1u) << (BK - 1))
                              | ((uint64_t)((cu >> 2) & 1u) << (BK + W)) | ((uint64_t)((cu >> 3) & 1u) << (BK - W));
            const B cT = deposit64<BK, NL>(cf, tT);
            if constexpr (MOVER != 0) attT = andn(attT, cT); else defT = andn(defT, cT);       // custodial victims are the other side's
            if (ax.ncap != 0) TAFL_STAT_HIT(11);
            if (ax.ncap != (uint32_t)__builtin_popcount(cu)) {
                TAFL_STAT_HIT(12);
                B c = ax.caps;
                while (any(c)) { const uint32_t i = lsb(c); c = andn(c, bit_at<NL>(i)); const B cb = bit_at<NL>(n_to_t(i)); attT = andn(attT, cb); defT = andn(defT, cb); }
            }
        }
        // opponent's plays on the post-move board: no-plays test, enclosure filter, and the next ply's move set
        TAFL_PROF_SPLIT(5);
        gen(st, attT, defT, MOVER ^ 1u, C, fc, g);
        TAFL_PROF_SPLIT(6);
        bool skip_encl = false, skip_fort = false;
        if constexpr (MOVER == 0) skip_encl = C.rules.enclosure_win == TAFL_ENCL_WITHOUT_EDGE_ACCESS && (g.edge_hit || any(st.def & C.edge));
        else if (C.rules.exit_fort) { skip_fort = !fort_candidate(st, attT, defT, C); if (!skip_fort) TAFL_STAT_HIT(13); }
        const typename E::Outcome o = E::outcome_early(st, ax, C, skip_encl, skip_fort);
        E::apply_finish(st, ax, o, o.over ? 1u : g.total, C);
        TAFL_PROF_END(7);
    }

    // one seeded uniform-random playout; identical results to Engine::rollout
    static TAFL_HD void rollout(S& st, uint32_t sk, uint32_t max_plies, const K& C, tafl_rollout_result& res) {
        const F fc = make_fast_consts<NL>(C);
        const uint32_t start_side = st.flags & TAFL_F_SIDE;
        B attT, defT;
        transpose_in(st.att & C.board, attT); transpose_in(st.def & C.board, defT);
        Gen g;
        if (TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING) gen(st, attT, defT, start_side, C, fc, g);
        else { TAFL_UNROLL for (int d = 0; d < 4; ++d) { g.r[d] = bz<NL>(); g.cnt[d] = 0; } g.total = 0; g.edge_hit = false; }
        uint32_t ply = 0, rk = sk; bool stuck = false;
        // a playout goes on while the game is on, the cap is not reached and the side to move has a play (else: stuck)
        auto goes_on = [&]() -> bool {
            if (!(ply < max_plies && TAFL_F_STATUS(st.flags) == TAFL_STATUS_ONGOING)) return false;
            if (g.total == 0) { stuck = true; return false; }
            return true;
        };
        bool active = goes_on();
        bool skip_first = start_side != 0;     // a playout that starts with a defender's play sits out the first (attacker's) half
        while (wave_any(active)) {
            if (active && !skip_first) { ply_of<0>(st, attT, defT, g, rk, C, fc); ++ply; active = goes_on(); }
            skip_first = false;
            if (active) { ply_of<1>(st, attT, defT, g, rk, C, fc); ++ply; active = goes_on(); }
        }
        E::finish_rollout(st, start_side, ply, stuck, res);
    }
};

}  // namespace tafl
