// tafl_host.hpp — host-side helpers of the library: FEN parsing into the ABI state, presets.
// FEN grammar and piece letters follow game/board/state.rs:225-250 and game/pieces.rs:117-141.
#pragma once
#include <string.h>
#include <string>
#include "../../include/taflhip.h"

namespace tafl {

inline int word_params(uint32_t word_bits, int* limbs64, int* row_width) {
    switch (word_bits) {
        case 64:  *limbs64 = 1; *row_width = 7;  return 0;    // game/bitfield.rs:178
        case 128: *limbs64 = 2; *row_width = 11; return 0;    // game/bitfield.rs:179
        case 256: *limbs64 = 4; *row_width = 15; return 0;    // game/bitfield.rs:180
        default: return -1;
    }
}

// GameState::new(fen, side) (game/game/state.rs:136-145) into the ABI struct.
// Returns 0, or a negative code: -1 bad word size, -2 BadChar, -3 BadLineLen, -4 board does not fit the word.
inline int fen_to_state(const char* fen, uint8_t side_to_play, uint32_t word_bits, tafl_state* out, std::string* err) {
    int l64, rw;
    if (word_params(word_bits, &l64, &rw)) { if (err) *err = "word_bits must be 64, 128 or 256"; return -1; }
    memset(out, 0, sizeof *out);
    uint64_t att[TAFL_MAX_LIMBS] = {0, 0, 0, 0}, def[TAFL_MAX_LIMBS] = {0, 0, 0, 0};
    unsigned krow = 0, kcol = 0;
    unsigned side_len = 0, r = 0;
    const char* p = fen;
    auto setbit = [&](uint64_t* w, unsigned idx) { w[idx / 64] |= 1ull << (idx % 64); };
    auto clrbit = [&](uint64_t* w, unsigned idx) { w[idx / 64] &= ~(1ull << (idx % 64)); };
    for (;;) {
        unsigned n_empty = 0, c = 0;
        while (*p && *p != '/') {
            const char chr = *p++;
            if (chr >= '0' && chr <= '9') { n_empty = n_empty * 10 + (unsigned)(chr - '0'); continue; }
            c += n_empty; n_empty = 0;
            const bool upper = chr >= 'A' && chr <= 'Z';
            const char low = upper ? (char)(chr - 'A' + 'a') : chr;
            if (!(low == 't' || low == 'k' || low == 'n' || low == 'c' || low == 'g' || low == 'm')) {
                if (err) *err = std::string("BadChar('") + chr + "')";
                return -2;
            }
            if ((int)c >= rw || r * (unsigned)rw + c >= (unsigned)(word_bits - 4)) { if (err) *err = "board does not fit the word"; return -4; }
            const unsigned idx = r * (unsigned)rw + c;
            if (upper) { setbit(def, idx); clrbit(att, idx); } else { setbit(att, idx); clrbit(def, idx); }   // set_piece :149-165
            if (low == 'k') { krow = r; kcol = c; }                                                            // set_king :136-147
            c += 1;
        }
        if (n_empty > 0) c += n_empty;
        if (side_len == 0) side_len = c;
        else if (side_len != c) { if (err) *err = "BadLineLen(" + std::to_string(c) + ")"; return -3; }
        r += 1;
        if (*p == '/') { ++p; continue; }
        break;
    }
    if (side_len > (unsigned)rw || side_len > 15) { if (err) *err = "board does not fit the word"; return -4; }
    const int top = l64 - 1;
    att[top] = (att[top] & 0x0FFFFFFFFFFFFFFFull) | ((uint64_t)(kcol & 15) << 60);
    def[top] = (def[top] & 0x0FFFFFFFFFFFFFFFull) | ((uint64_t)(krow & 15) << 60);
    for (int i = 0; i < TAFL_MAX_LIMBS; ++i) { out->att[i] = att[i]; out->def[i] = def[i]; }
    out->side_to_play = side_to_play ? TAFL_DEFENDER : TAFL_ATTACKER;
    out->status = TAFL_STATUS_ONGOING;
    out->side_len = (uint8_t)side_len;
    return 0;
}

// BoardState::to_fen (game/board/state.rs:271-295) from the ABI struct: 't' attacker, 'T' defender, 'K' the defender standing on the
// king nibble's tile (get_piece, board/state.rs:173-187); runs of empty tiles as decimal numbers, rows joined by '/'.
inline int state_to_fen(const tafl_state* st, uint32_t word_bits, std::string* out) {
    int l64, rw;
    if (word_params(word_bits, &l64, &rw) || st->side_len == 0 || st->side_len > rw) return -1;
    const int top = l64 - 1;
    const unsigned krow = (unsigned)(st->def[top] >> 60) & 15u, kcol = (unsigned)(st->att[top] >> 60) & 15u;
    auto bit = [&](const uint64_t* w, unsigned idx) -> bool { return (w[idx / 64] >> (idx % 64)) & 1ull; };
    out->clear();
    for (unsigned r = 0; r < st->side_len; ++r) {
        unsigned n_empty = 0;
        for (unsigned c = 0; c < st->side_len; ++c) {
            const unsigned idx = r * (unsigned)rw + c;
            char ch = 0;
            if (bit(st->def, idx)) ch = (r == krow && c == kcol) ? 'K' : 'T';
            else if (bit(st->att, idx)) ch = 't';
            if (ch) { if (n_empty) { *out += std::to_string(n_empty); n_empty = 0; } out->push_back(ch); } else ++n_empty;
        }
        if (n_empty) *out += std::to_string(n_empty);
        if (r + 1 < st->side_len) out->push_back('/');
    }
    return 0;
}

// game/preset.rs:12-124
inline int preset_rules(const char* name, tafl_rules* r) {
    memset(r, 0, sizeof *r);
    const std::string s(name ? name : "");
    const tafl_pieceset KING = TAFL_PS_TYPE(TAFL_PT_KING), SOLD = TAFL_PS_TYPE(TAFL_PT_SOLDIER);
    r->king_attack = TAFL_KING_ARMED; r->starting_side = TAFL_ATTACKER;
    if (s == "copenhagen") {
        r->king_strength = TAFL_KING_STRONG; r->has_shieldwall = 1; r->sw_corners_may_close = 1; r->sw_captures = SOLD;
        r->exit_fort = 1; r->throne_movement = TAFL_THRONE_KINGENTRY; r->may_enter_corners = KING;
        r->hostility_throne = TAFL_PS_ALL; r->hostility_corners = SOLD; r->enclosure_win = TAFL_ENCL_WITHOUT_EDGE_ACCESS;
        r->has_repetition_rule = 1; r->n_repetitions = 3; r->rep_is_loss = 1;
    } else if (s == "brandubh") {
        r->king_strength = TAFL_KING_STRONG_BY_THRONE; r->throne_movement = TAFL_THRONE_KINGENTRY; r->may_enter_corners = KING;
        r->hostility_throne = SOLD; r->hostility_corners = TAFL_PS_ALL; r->enclosure_win = TAFL_ENCL_WITHOUT_EDGE_ACCESS;
        r->has_repetition_rule = 1; r->n_repetitions = 3; r->rep_is_loss = 1;
    } else if (s == "magpie") {
        r->king_strength = TAFL_KING_STRONG; r->throne_movement = TAFL_THRONE_KINGENTRY; r->may_enter_corners = KING;
        r->hostility_throne = TAFL_PS_ALL; r->hostility_corners = TAFL_PS_ALL; r->slow_pieces = KING;
    } else if (s == "tablut") {
        r->edge_escape = 1; r->king_strength = TAFL_KING_STRONG_BY_THRONE; r->throne_movement = TAFL_THRONE_NOENTRY;
        r->may_enter_corners = TAFL_PS_ALL; r->hostility_throne = TAFL_PS_ALL;
        r->has_repetition_rule = 1; r->n_repetitions = 3; r->rep_is_loss = 0; r->draw_on_no_plays = 1; r->linnaean_capture = 1;
    } else if (s == "koch") {
        r->king_strength = TAFL_KING_STRONG_BY_THRONE; r->throne_movement = TAFL_THRONE_KINGENTRY; r->may_enter_corners = KING;
        r->hostility_throne = TAFL_PS_ALL; r->hostility_corners = SOLD; r->enclosure_win = TAFL_ENCL_WITHOUT_EDGE_ACCESS;
        r->has_repetition_rule = 1; r->n_repetitions = 3; r->rep_is_loss = 1;
    } else return -1;
    return 0;
}

// game/preset.rs:126-135 (+ build-defined 13x13)
inline const char* preset_board(const char* name) {
    const std::string s(name ? name : "");
    if (s == "copenhagen") return "3ttttt3/5t5/11/t4T4t/t3TTT3t/tt1TTKTT1tt/t3TTT3t/t4T4t/11/5t5/3ttttt3";
    if (s == "brandubh") return "3t3/3t3/3T3/ttTKTtt/3T3/3t3/3t3";
    if (s == "magpie") return "3t3/1t3t1/3T3/t1TKT1t/3T3/1t3t1/3t3";
    if (s == "tablut") return "3ttt3/4t4/4T4/t3T3t/ttTTKTTtt/t3T3t/4T4/4t4/3ttt3";
    if (s == "copenhagen13") return "4ttttt4/6t6/13/13/t5T5t/t4TTT4t/tt2TTKTT2tt/t4TTT4t/t5T5t/13/13/6t6/4ttttt4";
    return nullptr;
}

}  // namespace tafl
