/*
 * tafl_oracle.h — CPU oracle (TEST INFRASTRUCTURE, never shipped, never on the product path).
 *
 * A literal, Tile-at-a-time C restatement of the reference's Rust rules crate under
 * /root/reference/game (which cannot be compiled here: no rustc/cargo, broken manifest —
 * SURVEY.md §8c) plus a C restatement of the arithmetic of /root/reference/src/mcts.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Pinning: every reference unit test is transcribed as a data fixture under
 * tests/golden/ and replayed against this oracle (tests/test_oracle_kat.py); the MCTS arithmetic
 * is checked against the reference's own src/mcts.py imported in the build container
 * (tests/golden/make_mcts_golden.py -> tests/golden/mcts_golden.json).
 */
#ifndef TAFL_ORACLE_H
#define TAFL_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/taflhip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LIMBS 8   /* u64 x8 = U512 (HugeBasicBoardState, game/board/state.rs:340) */
#define ORC_MAX_SIDE 24

typedef struct { uint64_t w[ORC_MAX_LIMBS]; } obits;

typedef struct { uint8_t row, col; } otile;
typedef struct { int8_t row, col; } ocoords;
typedef struct { uint8_t piece_type; uint8_t side; } opiece;   /* piece_type 0 = no piece */
typedef struct { otile from; uint8_t axis; int8_t disp; } oplay;

typedef struct {
    obits attackers, defenders;
    uint8_t side_len;
    uint8_t nl;   /* 64-bit limbs of T: 1 (u64), 2 (u128), 4 (U256), 8 (U512) */
    uint8_t rw;   /* BitField::ROW_WIDTH: 7 / 11 / 15 / 21 */
} oboard;

typedef struct { uint8_t some, side; oplay play; uint8_t captures; } oshortrec;

typedef struct {
    size_t attacker_reps, defender_reps;
    uint8_t attacker_mid_pair, defender_mid_pair;
    oshortrec queue[4];
    size_t first_i;
} otracker;

typedef struct {
    oboard board;
    uint8_t side_to_play;
    otracker repetitions;
    size_t plays_since_capture;
    uint8_t status, reason, winner;   /* TAFL_STATUS_*, WinReason/DrawReason, Side */
    size_t turn;
} ostate;

typedef struct { uint8_t m[ORC_MAX_SIDE][ORC_MAX_SIDE]; int len; } otileset;

typedef struct { otileset occupied, unoccupied, boundary; } oenclosure;

typedef struct {
    tafl_rules rules;
    uint8_t side_len;
    otile throne;
    otile corners[4];
} ologic;

/* --- construction ------------------------------------------------------------------------- */
size_t orc_sizeof_state(void);
size_t orc_sizeof_logic(void);
size_t orc_sizeof_enclosure(void);
size_t orc_sizeof_tracker(void);
int orc_logic_init(ologic* lg, const tafl_rules* rules, uint8_t side_len);          /* logic.rs:70 */
int orc_state_init(ostate* st, const char* fen, uint8_t side, uint32_t word_bits);  /* state.rs:136 */
int orc_state_export(const ostate* st, tafl_state* out);   /* <= 256-bit words only */
int orc_state_import(ostate* st, const tafl_state* in, uint32_t word_bits);
int orc_preset_rules(const char* name, tafl_rules* out);   /* game/preset.rs:12-124 */
const char* orc_preset_board(const char* name);            /* game/preset.rs:126-135 */

/* --- board (game/board/state.rs) ------------------------------------------------------------ */
int  orc_get_piece(const ostate* st, uint8_t row, uint8_t col);   /* 0 none else type | side<<8 */
void orc_set_piece(ostate* st, uint8_t row, uint8_t col, uint8_t piece_type, uint8_t side);
void orc_clear_tile(ostate* st, uint8_t row, uint8_t col);
int  orc_move_piece(ostate* st, uint8_t fr, uint8_t fc, uint8_t tr, uint8_t tc);
void orc_swap_pieces(ostate* st, uint8_t r1, uint8_t c1, uint8_t r2, uint8_t c2);
int  orc_get_king(const ostate* st);                              /* row<<8 | col */
int  orc_tile_occupied(const ostate* st, uint8_t row, uint8_t col);
int  orc_count_pieces(const ostate* st, uint8_t side);
int  orc_iter_occupied(const ostate* st, uint8_t side, uint8_t* out_rc, int cap);
int  orc_to_fen(const ostate* st, char* out, int cap);
int  orc_from_display_str(ostate* st, const char* s, uint32_t word_bits);
int  orc_board_to_matrix(const ostate* st, uint8_t* out);                 /* game/main.rs:55-83 */

/* --- geometry (game/board/geometry.rs) -------------------------------------------------------- */
int orc_neighbors(const ologic* lg, uint8_t row, uint8_t col, uint8_t* out_rc);
int orc_tiles_between(const ologic* lg, uint8_t r1, uint8_t c1, uint8_t r2, uint8_t c2, uint8_t* out_rc);

/* --- logic (game/game/logic.rs, game/play.rs) ----------------------------------------------------- */
int orc_validate_play(const ologic* lg, const ostate* st, tafl_play play);           /* :219 */
int orc_validate_play_for_side(const ologic* lg, const ostate* st, tafl_play play, uint8_t side); /* :159 */
int orc_iter_plays(const ologic* lg, const ostate* st, uint8_t row, uint8_t col, tafl_play* out, int cap); /* :850 */
int orc_all_plays(const ologic* lg, const ostate* st, tafl_play* out, int cap);      /* game/main.rs:33-43 */
int orc_side_can_play(const ologic* lg, const ostate* st, uint8_t side);             /* :837 */
int orc_get_captures(const ologic* lg, const ostate* st, tafl_play play, uint8_t mp_type,
                     uint8_t mp_side, uint8_t* out_rc, int cap);                     /* :604 */
int orc_detect_shieldwall(const ologic* lg, const ostate* st, tafl_play play, uint8_t* out_rc, int cap); /* :535; -1 = None */
int orc_find_enclosure(const ologic* lg, const ostate* st, uint8_t row, uint8_t col,
                       uint16_t enclosed, uint16_t enclosing, int abort_on_edge,
                       int abort_on_corner, oenclosure* out);                        /* :309; 0 = None */
int orc_enclosure_tiles(const oenclosure* e, int which, uint8_t* out_rc, int cap);   /* 0 occ 1 unocc 2 boundary */
int orc_enclosure_secure(const ologic* lg, const ostate* st, const oenclosure* e,
                         int inside_safe, int outside_safe);                         /* :408 */
int orc_detect_exit_fort(const ologic* lg, const ostate* st);                        /* :572 */
int orc_do_play(const ologic* lg, ostate* st, tafl_play play, tafl_effects* eff);    /* :827 */
int orc_do_valid_play(const ologic* lg, ostate* st, tafl_play play, tafl_effects* eff); /* :782 */

/* --- repetition tracker (game/game/state.rs:41-114) -------------------------------------------------- */
void   orc_tracker_init(otracker* t);
void   orc_tracker_track_play(otracker* t, uint8_t side, tafl_play play, int captures);
size_t orc_tracker_get_repetitions(const otracker* t, uint8_t side);

/* --- dense action index (include/taflhip.h) ------------------------------------------------------------ */
uint32_t orc_action_size(uint8_t side_len);
uint32_t orc_action_encode(uint8_t side_len, tafl_play p);
tafl_play orc_action_decode(uint8_t side_len, uint32_t a);

/* --- build-defined rollout policy ------------------------------------------------------------ */
uint32_t orc_rng(uint64_t seed, uint64_t game_id, uint32_t sim, uint32_t ply);
int orc_rollout_order_plays(const ologic* lg, const ostate* st, tafl_play* out, int cap);
/* leaf key of the search's playouts: predict(s) = the playout with simulation word sim_offset + orc_state_hash(s) (tafl_oracle.c) */
uint32_t orc_state_hash(const ostate* st);
int orc_rollout(const ologic* lg, const ostate* st, uint64_t seed, uint64_t game_id, uint32_t sim,
                uint32_t max_plies, tafl_rollout_result* out);
int orc_random_advance(const ologic* lg, ostate* st, uint64_t seed, uint64_t game_id, uint32_t plies);

/* --- MCTS (arithmetic of src/mcts.py:55-136 on an explicit per-game tree) ----------------------------------- */
typedef struct orc_mcts orc_mcts;
orc_mcts* orc_mcts_new(const ologic* lg, const ostate* root, const tafl_mcts_params* p, uint64_t game_id);
void orc_mcts_free(orc_mcts* m);
int  orc_mcts_run(orc_mcts* m);                 /* runs p->n_sims searches */
int  orc_mcts_root_children(const orc_mcts* m, tafl_root_child* out, int cap);
uint32_t orc_mcts_root_ns(const orc_mcts* m);
void orc_mcts_get_stats(const orc_mcts* m, tafl_mcts_stats* out);

/* --- guided MCTS: src/mcts.py:55-136 with an external predict() (nnet.predict, mcts.py:85) ------------------------------ */
typedef void (*orc_predict_fn)(void* ctx, const ostate* st, float* priors_out /* [action_size] */, float* value_out);
typedef struct orc_gmcts orc_gmcts;
orc_gmcts* orc_gmcts_new(const ologic* lg, const ostate* root, double c_puct, orc_predict_fn predict, void* ctx);
void orc_gmcts_free(orc_gmcts* m);
int  orc_gmcts_run(orc_gmcts* m, uint32_t n_sims);
int  orc_gmcts_root_children(const orc_gmcts* m, tafl_root_child* out, int cap);
uint32_t orc_gmcts_root_ns(const orc_gmcts* m);
int  orc_gmcts_root_priors(const orc_gmcts* m, double* out);
void orc_gmcts_counts(const orc_gmcts* m, uint64_t* out4);   /* sims, predicts, terminal hits, selection depth sum */
double orc_np_sum(const double* a, long n);                  /* numpy's pairwise np.sum of a contiguous float64 array */

/* --- batch drivers (differential tests at scale, cpu_baseline) ------------------------------------------------- */
int orc_batch_movegen(const ologic* lg, const tafl_state* states, uint32_t n, uint32_t word_bits,
                      uint32_t* out_counts, uint32_t* out_masks, uint32_t mask_words);
int orc_batch_step(const ologic* lg, tafl_state* states, uint32_t n, uint32_t word_bits,
                   const tafl_play* plays, tafl_effects* eff);
int orc_batch_step_kth(const ologic* lg, tafl_state* states, uint32_t n, uint32_t word_bits,
                       const uint32_t* ranks, tafl_play* out_plays, tafl_effects* eff);
int orc_batch_rollout(const ologic* lg, const tafl_state* states, uint32_t n, uint32_t word_bits,
                      uint64_t seed, uint32_t sim, uint32_t max_plies, uint64_t game_id_base,
                      tafl_rollout_result* out);
int orc_batch_random_advance(const ologic* lg, tafl_state* states, uint32_t n, uint32_t word_bits,
                             uint64_t seed, const uint32_t* plies, uint64_t game_id_base);
int orc_batch_mcts(const ologic* lg, const tafl_state* states, uint32_t n, uint32_t word_bits,
                   const tafl_mcts_params* p, uint64_t game_id_base, tafl_root_child* out_children,
                   uint32_t max_children, uint32_t* out_n, tafl_mcts_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
