/*
 * taflhip.h — C-ABI of the MI355X-native batched Hnefatafl engine (libtaflhip.so).
 *
 * This is the drop-in boundary for ONE hot path of payelmuk91/AlphaZeroForHnefatafl:
 * batched move generation + env step + random rollout + MCTS over many concurrent games.
 * The reference has no FFI; the surface a host uses there is the `GameLogic` method set over
 * `GameState<T>` values (game/game/logic.rs:62-880, game/game/state.rs:119-146) and, for search,
 * `MCTS.getActionProb` (src/mcts.py:28-53).  Every entry point below cites the reference
 * interface it replaces.  Plain C: opaque handles, POD structs, pointers + sizes; no C++ or
 * torch types.  All functions return 0 on success and a negative `tafl_status` on failure
 * (message via tafl_last_error()).  Per-game rule errors are returned as per-game codes, never
 * as a failed call (reference: `Result<_, PlayInvalid>`, game/error.rs:49-70).
 *
 * There is NO CPU fallback in the library: every compute entry point runs HIP kernels on a
 * gfx950 device and fails with TAFL_ERR_NO_DEVICE when none is usable.
 */
#ifndef TAFLHIP_H
#define TAFLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAFLHIP_ABI_VERSION 1

/* ---- vocabulary (numeric values mirror the reference enums) ------------------------------ */

/* Side — game/pieces.rs:13-16 (`Attacker = 0, Defender = 8`). */
#define TAFL_ATTACKER 0
#define TAFL_DEFENDER 8

/* PieceType one-hot — game/pieces.rs:31-38.  Only King/Soldier exist on a bitfield board. */
#define TAFL_PT_KING      0x01
#define TAFL_PT_SOLDIER   0x02
#define TAFL_PT_KNIGHT    0x04
#define TAFL_PT_COMMANDER 0x08
#define TAFL_PT_GUARD     0x10
#define TAFL_PT_MERCENARY 0x20

/* PieceSet(u16) — game/pieces.rs:157-273: bit = piece_type << side (attackers low byte,
 * defenders high byte). */
typedef uint16_t tafl_pieceset;
#define TAFL_PS_NONE 0x0000u
#define TAFL_PS_ALL  0xFFFFu
#define TAFL_PS_TYPE(pt) ((tafl_pieceset)((pt) | ((pt) << 8)))   /* PieceSet::from_piece_type */
#define TAFL_PS_PIECE(pt, side) ((tafl_pieceset)((pt) << (side))) /* PieceSet::from_piece */
#define TAFL_PS_SIDE(side) ((tafl_pieceset)(0xFFu << (side)))     /* PieceSet::from(Side) */

/* Axis — game/tiles.rs:167-170 (`Vertical = 0, Horizontal = 0x80`). */
#define TAFL_AXIS_VERTICAL   0x00
#define TAFL_AXIS_HORIZONTAL 0x80

/* ThroneRule — game/rules.rs:5-17 (declaration order). */
enum { TAFL_THRONE_NOTHRONE = 0, TAFL_THRONE_NOPASS = 1, TAFL_THRONE_KINGPASS = 2,
       TAFL_THRONE_NOENTRY = 3, TAFL_THRONE_KINGENTRY = 4 };
/* KingStrength — game/rules.rs:21-30. */
enum { TAFL_KING_STRONG = 0, TAFL_KING_STRONG_BY_THRONE = 1, TAFL_KING_WEAK = 2 };
/* KingAttack — game/rules.rs:33-42. */
enum { TAFL_KING_ARMED = 0, TAFL_KING_ANVIL = 1, TAFL_KING_HAMMER = 2 };
/* EnclosureWinRules — game/rules.rs:64-70; 0 = `None`. */
enum { TAFL_ENCL_NONE = 0, TAFL_ENCL_WITH_EDGE_ACCESS = 1, TAFL_ENCL_WITHOUT_EDGE_ACCESS = 2 };

/* PlayInvalid — game/error.rs:49-70, declaration order, shifted by one so that 0 = valid. */
enum { TAFL_PLAY_OK = 0, TAFL_PLAY_WRONG_PLAYER = 1, TAFL_PLAY_NO_PIECE = 2,
       TAFL_PLAY_OUT_OF_BOUNDS = 3, TAFL_PLAY_NO_COMMON_AXIS = 4, TAFL_PLAY_BLOCKED_BY_PIECE = 5,
       TAFL_PLAY_MOVE_THROUGH_BLOCKED_TILE = 6, TAFL_PLAY_MOVE_ONTO_BLOCKED_TILE = 7,
       TAFL_PLAY_TOO_FAR = 8, TAFL_PLAY_GAME_OVER = 9 };

/* GameStatus / GameOutcome — game/game/mod.rs:16-70. */
enum { TAFL_STATUS_ONGOING = 0, TAFL_STATUS_WIN = 1, TAFL_STATUS_DRAW = 2 };
/* WinReason — game/game/mod.rs:17-33 (declaration order). */
enum { TAFL_WIN_KING_ESCAPED = 0, TAFL_WIN_EXIT_FORT = 1, TAFL_WIN_KING_CAPTURED = 2,
       TAFL_WIN_ALL_CAPTURED = 3, TAFL_WIN_ENCLOSED = 4, TAFL_WIN_NO_PLAYS = 5,
       TAFL_WIN_REPETITION = 6 };
/* DrawReason — game/game/mod.rs:36-42. */
enum { TAFL_DRAW_REPETITION = 0, TAFL_DRAW_NO_PLAYS = 1 };
/* Build-defined rollout terminations (the reference has no rollout). */
enum { TAFL_ROLLOUT_REASON_PLY_CAP = 14, TAFL_ROLLOUT_REASON_STUCK = 15 };

/* library status codes */
typedef enum {
    TAFL_OK = 0,
    TAFL_ERR_INVALID_ARG = -1,
    TAFL_ERR_NO_DEVICE = -2,
    TAFL_ERR_HIP = -3,
    TAFL_ERR_PARSE = -4,
    TAFL_ERR_UNSUPPORTED = -5,
    TAFL_ERR_OOM = -6,
    TAFL_ERR_CAPACITY = -7
} tafl_status;

/* ---- POD structs --------------------------------------------------------------------------- */

/* Ruleset — game/rules.rs:83-117, field for field (Option<..> flattened with a has_ flag). */
typedef struct tafl_rules {
    uint8_t  edge_escape;             /* bool                                   rules.rs:86  */
    uint8_t  king_strength;           /* TAFL_KING_STRONG...                    rules.rs:89  */
    uint8_t  king_attack;             /* TAFL_KING_ARMED...                     rules.rs:91  */
    uint8_t  has_shieldwall;          /* Option<ShieldwallRules>::is_some       rules.rs:93  */
    uint8_t  sw_corners_may_close;    /* ShieldwallRules.corners_may_close      rules.rs:56  */
    uint8_t  exit_fort;               /* bool                                   rules.rs:95  */
    uint8_t  throne_movement;         /* TAFL_THRONE_...                        rules.rs:97  */
    uint8_t  starting_side;           /* TAFL_ATTACKER / TAFL_DEFENDER          rules.rs:105 */
    uint8_t  enclosure_win;           /* TAFL_ENCL_... (0 = None)               rules.rs:107 */
    uint8_t  has_repetition_rule;     /* Option<RepetitionRule>::is_some        rules.rs:109 */
    uint8_t  rep_is_loss;             /* RepetitionRule.is_loss                 rules.rs:79  */
    uint8_t  draw_on_no_plays;        /* bool                                   rules.rs:112 */
    uint8_t  linnaean_capture;        /* bool                                   rules.rs:116 */
    uint8_t  _pad0[3];
    tafl_pieceset sw_captures;        /* ShieldwallRules.captures               rules.rs:58  */
    tafl_pieceset may_enter_corners;  /*                                        rules.rs:99  */
    tafl_pieceset hostility_throne;   /* HostilityRules.throne                  rules.rs:47  */
    tafl_pieceset hostility_corners;  /* HostilityRules.corners                 rules.rs:48  */
    tafl_pieceset hostility_edge;     /* HostilityRules.edge                    rules.rs:49  */
    tafl_pieceset slow_pieces;        /*                                        rules.rs:103 */
    uint32_t n_repetitions;           /* RepetitionRule.n_repetitions           rules.rs:76  */
} tafl_rules;                         /* 32 bytes */

/* Play — game/play.rs:22-27 (`from: Tile{row,col}`, `movement: AxisOffset{axis,displacement}`,
 * game/tiles.rs:11-17,108-111). 4 bytes. */
typedef struct tafl_play {
    uint8_t from_row;
    uint8_t from_col;
    uint8_t axis;        /* TAFL_AXIS_VERTICAL (0) or TAFL_AXIS_HORIZONTAL (0x80) */
    int8_t  disp;        /* signed displacement along the axis */
} tafl_play;

/* GameState<T> — game/game/state.rs:119-132 with BitfieldBoardState<T> (game/board/state.rs:116-121)
 * flattened.  `att`/`def` are the reference integers as little-endian 64-bit limbs, SAME bit
 * positions: tile bit = row*ROW_WIDTH + col (game/bitfield.rs:72-74; ROW_WIDTH 7/11/15 for
 * 64/128/256-bit words, :178-180) and the king's (row, col) in the top nibble of the most
 * significant byte of `def` / `att` (game/board/state.rs:127-147) — a Rust host can transmute a
 * u64/u128/U256.  Only the first word_bits/64 limbs are used. */
#define TAFL_MAX_LIMBS 4
typedef struct tafl_state {
    uint64_t att[TAFL_MAX_LIMBS];
    uint64_t def[TAFL_MAX_LIMBS];
    uint32_t turn;                 /* GameState.turn                       state.rs:131 */
    uint32_t plays_since_capture;  /* GameState.plays_since_capture        state.rs:127 */
    uint32_t rep_ring[4];          /* RepetitionTracker.recent_plays, OLDEST FIRST (state.rs:46,
                                      utils.rs:30-81); 0 = None, else TAFL_REP_PACK(...) */
    uint16_t attacker_reps;        /* state.rs:42 */
    uint16_t defender_reps;        /* state.rs:43 */
    uint8_t  attacker_mid_pair;    /* state.rs:44 */
    uint8_t  defender_mid_pair;    /* state.rs:45 */
    uint8_t  side_to_play;         /* TAFL_ATTACKER / TAFL_DEFENDER        state.rs:123 */
    uint8_t  status;               /* TAFL_STATUS_*                        state.rs:129 */
    uint8_t  reason;               /* WinReason / DrawReason when over */
    uint8_t  winner;               /* Side when status == WIN */
    uint8_t  side_len;             /* BitfieldBoardState.side_len          board/state.rs:120 */
    uint8_t  _pad;
} tafl_state;                      /* 104 bytes */

/* ShortPlayRecord (game/game/state.rs:15-19) packed into 32 bits, bit 31 = Some. */
#define TAFL_REP_PACK(side, from_row, from_col, axis, disp, captured)                        \
    (0x80000000u | ((uint32_t)((side) ? 1u : 0u) << 30) | ((uint32_t)((captured) ? 1u : 0u) << 29) | \
     ((uint32_t)((axis) ? 1u : 0u) << 28) | ((uint32_t)((from_row) & 0xFF) << 16) |                  \
     ((uint32_t)((from_col) & 0xFF) << 8) | ((uint32_t)((uint8_t)(disp))))

/* PlayEffects — game/game/mod.rs:56-61 — as a capture mask in board-word layout plus the
 * outcome; `code` is the PlayInvalid code when the play was rejected (state left unchanged). */
typedef struct tafl_effects {
    uint64_t captures[TAFL_MAX_LIMBS]; /* bit row*ROW_WIDTH+col set for each captured tile */
    uint8_t  code;                     /* TAFL_PLAY_* (0 = executed) */
    uint8_t  status;                   /* TAFL_STATUS_* after the play */
    uint8_t  reason;
    uint8_t  winner;
    uint8_t  n_captures;
    uint8_t  _pad[3];
} tafl_effects;                        /* 40 bytes */

/* Random-rollout result (build-defined; see DESIGN.md "Rollout policy"). */
typedef struct tafl_rollout_result {
    int8_t   value;     /* +1 / -1 / 0 from the perspective of the side to move at the start */
    uint8_t  status;    /* TAFL_STATUS_* at the end (ONGOING when capped/stuck) */
    uint8_t  reason;    /* WinReason/DrawReason or TAFL_ROLLOUT_REASON_* */
    uint8_t  winner;
    uint32_t plies;     /* plies played */
} tafl_rollout_result;  /* 8 bytes */

/* One root child after MCTS: the visited edges (s_root, a) of src/mcts.py:41 (`Nsa`), :128-133 (`Qsa`). */
typedef struct tafl_root_child {
    tafl_play play;
    uint32_t  action;   /* dense action index, see tafl_action_size() */
    uint32_t  visits;   /* Nsa */
    double    q;        /* Qsa */
} tafl_root_child;      /* 24 bytes */

typedef struct tafl_mcts_params {
    uint32_t n_sims;           /* args.numMCTSSims   src/mcts.py:37  */
    uint32_t max_rollout_plies;
    double   c_puct;           /* args.cpuct         src/mcts.py:112 */
    uint64_t seed;
    uint32_t sim_offset;       /* salt of the playouts' RNG key (normally 0): predict(s) of the random-rollout search is ONE playout from s with
                                  simulation word sim_offset + leaf key(s) - "leaf key" = MurmurHash3_x86_32, seeded with the side length, over
                                  the words { per row r: attacker bits | defender bits << 16 (king included) }, rep_ring[0..3], turn,
                                  attacker_reps | defender_reps << 16, side | attacker_mid_pair << 1 | defender_mid_pair << 2 | king row << 16 |
                                  king col << 20 - so the value of a leaf is a function of (seed, global game id, sim_offset, position) alone,
                                  as nnet.predict(canonicalBoard) of src/mcts.py:85 is a function of the board alone */
    uint32_t flags;            /* TAFL_MCTS_FLAG_* | tuning fields; 0 = src/mcts.py semantics, default pipeline */
} tafl_mcts_params;
/* semantics bits of tafl_mcts_params.flags.
 *   TAFL_MCTS_FLAG_FPU_INF  first-play urgency of the reference's Rust sketch, src/mcts.rs: an action that was never taken scores
 *     f64::INFINITY in the selection (mcts.rs:49-51), so every legal child of a node is expanded before any child is revisited, and a
 *     newly expanded node starts with visits = 1.0 instead of 0 (mcts.rs:187).  Ties keep src/mcts.py's rule (first maximum = lowest
 *     action index; the sketch does not compile and leaves the order of `valid_actions` undefined).  Everything else is src/mcts.py.
 *     This mode is pinned by the oracle only (no reference implementation can run it). */
#define TAFL_MCTS_FLAG_FPU_INF 0x1u
/* tuning fields of tafl_mcts_params.flags: they choose HOW the same search is executed and never change its results
 * (tests/test_gpu_parity.py::test_mcts_pipelines_agree).
 *   bits 4-7   pipeline: 0 default (64-bit boards: fused; wider boards: two kernels per round, tree phase + playouts over dense work
 *              lists), 1 fused (one kernel per chunk of rounds, at most 2 playout slots per game; 64-bit boards only), 2 two-kernel
 *   bits 8-11  playout slots per game (one holds the pending simulation's leaf, the rest predicted leaves, DESIGN.md
 *              section 4.5); 0 = chosen from the batch size, at most 8
 *   bits 12-15 partitions of the batch that run the pipeline on their own streams (two-kernel pipeline); 0 = from the batch size, at most 8 */
#define TAFL_MCTS_PIPELINE_DEFAULT 0u
#define TAFL_MCTS_PIPELINE_FUSED 1u
#define TAFL_MCTS_PIPELINE_TWO_KERNEL 2u
#define TAFL_MCTS_TUNE_PIPELINE(x) (((uint32_t)(x) & 15u) << 4)
#define TAFL_MCTS_TUNE_SLOTS(x) (((uint32_t)(x) & 15u) << 8)
#define TAFL_MCTS_TUNE_PARTS(x) (((uint32_t)(x) & 15u) << 12)
#define TAFL_MCTS_TUNE_PARTS_OF(f) (((f) >> 12) & 15u)
#define TAFL_MCTS_TUNE_SHARE(x) (((uint32_t)(x) & 15u) << 16)     /* this search may fill 1/x of the device (0 = all of it): batches searched side by side */
#define TAFL_MCTS_TUNE_SHARE_OF(f) (((f) >> 16) & 15u)
#define TAFL_MCTS_TUNE_PIPELINE_OF(f) (((f) >> 4) & 15u)
#define TAFL_MCTS_TUNE_SLOTS_OF(f) (((f) >> 8) & 15u)
#define TAFL_MCTS_FLAGS_KNOWN 0x000FFFF1u

typedef struct tafl_mcts_stats {
    uint64_t sims;             /* simulations executed (all games) */
    uint64_t rollouts;         /* random playouts consumed by simulations (mispredicted speculative ones excluded) */
    uint64_t rollout_plies;    /* env steps inside playouts */
    uint64_t tree_depth_sum;   /* sum over sims of nodes on the selection path (d) */
    uint64_t children_scanned; /* sum over sims of visited children examined (for c-bar) */
    uint64_t terminal_hits;    /* sims that ended on a terminal tree node */
    uint64_t reason_hist[16];  /* playout terminations by reason (win reasons 0-6, draws 8-9, cap 14, stuck 15) */
    uint64_t faults;           /* games that raised a device-side fault flag */
    uint64_t spec_issued;      /* speculative playouts launched ahead of their simulation */
    uint64_t spec_hits;        /* ... of which were consumed by the simulation they were predicted for */
} tafl_mcts_stats;

typedef struct tafl_ctx   tafl_ctx;    /* rules + geometry + device + stream */
typedef struct tafl_batch tafl_batch;  /* n game states resident in HBM (+ optional MCTS arena) */

/* ---- context ---------------------------------------------------------------------------------
 * tafl_ctx_create replaces GameLogic::new(rules, board_length) (game/game/logic.rs:70-72).
 * word_bits selects the reference board word: 64 (SmallBasic, ROW_WIDTH 7), 128 (MediumBasic,
 * ROW_WIDTH 11) or 256 (LargeBasic, ROW_WIDTH 15) — game/board/state.rs:332-337.
 * `stream` is a hipStream_t to enqueue on, or NULL to let the library create one. */
int tafl_ctx_create(const tafl_rules* rules, uint8_t side_len, uint32_t word_bits, int device,
                    void* stream, tafl_ctx** out);
int tafl_ctx_destroy(tafl_ctx* ctx);
const char* tafl_last_error(void);
int tafl_abi_version(void);

/* Rule/board presets — game/preset.rs:12-124 (rules), :126-135 (boards).
 * names: "copenhagen", "brandubh", "magpie", "tablut", "koch". */
int tafl_preset_rules(const char* name, tafl_rules* out);
const char* tafl_preset_board(const char* name); /* start FEN, NULL if unknown; "copenhagen13" is build-defined */

/* dense action space (the `getActionSize()` of src/mcts.py:41): side_len^2 * 2*(side_len-1) actions — every
 * (from tile, destination on its row or column) pair.  For from = (r, c), n = side_len, dist >= 1:
 *   action = (r*n + c) * 2*(n-1) + slot,   slot = V+ (row+): dist-1            [n-1-r slots]
 *                                                 V- (row-): (n-1-r) + dist-1    [r slots]
 *                                                 H+ (col+): (n-1) + dist-1      [n-1-c slots]
 *                                                 H- (col-): (n-1) + (n-1-c) + dist-1   [c slots]
 * i.e. the iteration order of ValidPlayIterator (game/play.rs:157,166-183: V+,V-,H+,H-, distance ascending) over
 * iter_occupied (game/board/state.rs:202-216, row-major), so ascending action index == get_all_possible_moves
 * order (game/main.rs:33-43).  Dense mask: 2420 bits = 76 uint32 words for 11x11 (SURVEY.md §8d, M = 304 B). */
uint32_t tafl_action_size(const tafl_ctx* ctx);
uint32_t tafl_action_mask_words(const tafl_ctx* ctx);  /* uint32 words per game in a dense mask */
int tafl_action_encode(const tafl_ctx* ctx, tafl_play play, uint32_t* action);
int tafl_action_decode(const tafl_ctx* ctx, uint32_t action, tafl_play* play);

/* ---- batch of game states ---------------------------------------------------------------------
 * tafl_batch_reset_fen replaces GameState::new(fen, side) for every game (game/game/state.rs:136-145,
 * FEN grammar game/board/state.rs:225-250).  upload/download move whole GameState values. */
int tafl_batch_create(tafl_ctx* ctx, uint32_t n_games, tafl_batch** out);
int tafl_batch_destroy(tafl_batch* b);
uint32_t tafl_batch_size(const tafl_batch* b);
int tafl_batch_reset_fen(tafl_batch* b, const char* fen, uint8_t side_to_play);
int tafl_batch_upload(tafl_batch* b, const tafl_state* states, uint32_t first, uint32_t count);
int tafl_batch_download(tafl_batch* b, tafl_state* states, uint32_t first, uint32_t count);
int tafl_state_from_fen(const tafl_ctx* ctx, const char* fen, uint8_t side_to_play, tafl_state* out); /* host-only helper */
/* BoardState::to_fen (game/board/state.rs:271-295) of one state; returns the string length or a negative error (host only) */
int tafl_state_to_fen(const tafl_state* st, uint32_t word_bits, char* out, uint32_t cap);
int tafl_sync(tafl_ctx* ctx);

/* ---- hot path -----------------------------------------------------------------------------------
 * tafl_movegen: get_all_possible_moves (game/main.rs:33-43) = iter_occupied(side_to_play) x
 *   GameLogic::iter_plays (logic.rs:850-856, play.rs:139-226) for every game.
 *   out_counts[n] (host, may be NULL), out_masks[n * tafl_action_mask_words] (host, may be NULL):
 *   bit `action` set iff the play is legal.  Games that are over yield 0 plays (logic.rs:165-167).
 * tafl_validate: GameLogic::validate_play (logic.rs:219-222); out_codes[n] = TAFL_PLAY_*.
 * tafl_step: GameLogic::do_play (logic.rs:827-834) = validate + do_valid_play (:782-820);
 *   plays[n]; effects[n] may be NULL.  Invalid plays leave that game unchanged and set effects.code.
 * tafl_step_kth: same, but game i plays its (ranks[i] mod count)-th legal move in canonical order
 *   (BASELINE config 2); games with no legal move or already over are left unchanged (code GAME_OVER / NO_PIECE).
 * tafl_side_can_play: GameLogic::side_can_play (logic.rs:837-846) for `side`; out[n] = 0/1.
 * tafl_rollout: one seeded uniform-random playout per game from its current state WITHOUT
 *   modifying the batch (build-defined policy); results[n].
 * tafl_random_advance: game i plays plies[i] seeded random plies in place (config-2 input maker).
 */
int tafl_movegen(tafl_batch* b, uint32_t* out_counts, uint32_t* out_masks);
int tafl_validate(tafl_batch* b, const tafl_play* plays, uint8_t* out_codes);
int tafl_step(tafl_batch* b, const tafl_play* plays, tafl_effects* out_effects);
int tafl_step_kth(tafl_batch* b, const uint32_t* ranks, tafl_play* out_plays, tafl_effects* out_effects);
int tafl_side_can_play(tafl_batch* b, uint8_t side, uint8_t* out);
int tafl_rollout(tafl_batch* b, uint64_t seed, uint32_t sim, uint32_t max_plies,
                 uint64_t game_id_base, tafl_rollout_result* out_results);
int tafl_random_advance(tafl_batch* b, uint64_t seed, const uint32_t* plies, uint64_t game_id_base);

/* ---- MCTS -----------------------------------------------------------------------------------------
 * tafl_mcts_run replaces `for i in range(numMCTSSims): self.search(canonicalBoard)` of
 * MCTS.getActionProb (src/mcts.py:37-38) for every game of the batch, rooted at its current state,
 * with predict() = uniform priors + one seeded random playout (SURVEY.md §8a resolution):
 * select (mcts.py:104-123) / expand (:83-102) / rollout / backup (:127-136) as lock-step kernels.
 * The batch's states are not modified.  game_id_base is the global id of game 0 (RNG key), so
 * shards of a larger job reproduce the single-device results.
 * Results stay on the device until fetched:
 *   tafl_mcts_root_children: visited root edges in canonical order; out[n * max_children],
 *     out_n[n] = number of visited root children (<= max_children else TAFL_ERR_CAPACITY).
 *   tafl_mcts_root_visits: dense `counts` vector of mcts.py:41, out[n * tafl_action_size] uint32.
 *   tafl_mcts_policy: probs of mcts.py:40-53 (temp > 0: counts^(1/temp) normalised; temp == 0:
 *     one-hot on the FIRST maximum — the reference draws uniformly among maxima with np.random,
 *     the deterministic choice is documented in DESIGN.md); out[n * tafl_action_size] float64.
 *   tafl_mcts_best_play: max-visit root child (src/mcts.rs:216-227), first maximum.
 */
int tafl_mcts_reserve(tafl_batch* b, uint32_t max_sims);
int tafl_mcts_run(tafl_batch* b, const tafl_mcts_params* params, uint64_t game_id_base);
/* The same search without blocking the host (SURVEY 8b "calls enqueue ... asynchronous until tafl_sync"): tafl_mcts_run_async enqueues the
 * whole search on streams of the BATCH and returns; nothing is read back while it runs (plan and prediction width are steered on the
 * device).  tafl_mcts_wait joins it - and, while games are unfinished, runs the rounds its slowest games still need - so a search
 * always completes or the call fails.  tafl_mcts_run == tafl_mcts_run_async + tafl_mcts_wait; per-game results are identical.
 * Every reader of the results (root_children, root_visits, policy*, best_play, play_best, get_stats, round_trace) joins a search in
 * flight by itself.  One search per batch at a time (a second run_async first joins the first); DIFFERENT batches of one context search
 * side by side: the last rounds of a search are nearly empty (a few games whose simulations could not be predicted), and the device is
 * kept busy by another batch's full rounds.  tafl_mcts_run_async_after(b, ..., other) additionally holds b's search back until `other`'s
 * search in flight is half-way through its planned rounds (no effect if `other` has none): two half-size batches started this way stay
 * half a search apart, the steady state of a self-play loop `wait(A); play(A); run_async(A); wait(B); play(B); run_async(B)`.
 * The batch must not be modified (upload, step, reset) between run_async and wait. */
int tafl_mcts_run_async(tafl_batch* b, const tafl_mcts_params* params, uint64_t game_id_base);
int tafl_mcts_run_async_after(tafl_batch* b, const tafl_mcts_params* params, uint64_t game_id_base, tafl_batch* other);
int tafl_mcts_wait(tafl_batch* b);
int tafl_mcts_get_stats(tafl_batch* b, tafl_mcts_stats* out);
int tafl_mcts_root_children(tafl_batch* b, tafl_root_child* out, uint32_t max_children, uint32_t* out_n);
int tafl_mcts_root_visits(tafl_batch* b, uint32_t* out);
int tafl_mcts_policy(tafl_batch* b, double temp, double* out);
int tafl_mcts_best_play(tafl_batch* b, tafl_play* out_plays, uint32_t* out_visits);
/* self-play step without leaving the device: every game plays the most visited root play of its last tafl_mcts_run (first maximum)
 * on its batch state (do_valid_play); finished games are left alone (effects.code = TAFL_PLAY_GAME_OVER).  out_* may be NULL
 * (then nothing is copied back and the call only enqueues). */
int tafl_mcts_play_best(tafl_batch* b, tafl_play* out_plays, tafl_effects* out_effects);
/* Self-play without leaving the device and without a barrier between the moves: for every game, n_moves times, a search of
 * params->n_sims simulations from its current state followed by its most visited root play (first maximum) on the batch state - per game
 * exactly `for m in 0..n_moves: tafl_mcts_run(sim_offset + m * n_sims); tafl_mcts_play_best` - but a game starts its next search as soon
 * as ITS OWN search is done, so the games of the batch are at different phases of their searches and the device stays full (a batch of
 * synchronous searches ends every search in a tail of nearly empty rounds).  Games that end stop searching; out_plays[m * n + g] (may be
 * NULL) is the play game g made at move m, all-zero once its game was over.  tafl_mcts_get_stats afterwards covers all searches. */
int tafl_selfplay_run(tafl_batch* b, const tafl_mcts_params* params, uint32_t n_moves, uint64_t game_id_base, tafl_play* out_plays);

/* ---- training-tensor writers (the step right after the hot path) ------------------------------
 * tafl_encode_boards: board_to_matrix (game/main.rs:55-83) for every game: uint8 [n * side_len * side_len], row-major;
 *   corner tiles 20, throne 30, soldier +1, king +5 (no side distinction, as in the reference).
 * tafl_mcts_policy_device: the probs of src/mcts.py:40-53 written by a kernel, for any temp >= 0 (temp == 1: counts / sum, exact;
 *   temp == 0: one-hot on the first maximum); float64 [n * tafl_action_size].
 * `out` may be a host pointer (out_is_device = 0) or a device pointer of this ctx's device (out_is_device = 1, e.g. a
 * torch tensor's data_ptr): the second form never crosses PCIe. */
int tafl_encode_boards(tafl_batch* b, uint8_t* out, int out_is_device);
int tafl_mcts_policy_device(tafl_batch* b, double temp, double* out, int out_is_device);
/* the same with the choices mcts.py:44-45 leaves to np.random: temp == 0 puts the 1 on the floor(r * ties / 2^32)-th maximum in ascending action
 * order, r = taflmix32 word of (tie_seed, game_id_base + game); tie_seed == 0 = the first maximum.  Any temp >= 0: counts ** (1 / temp) is the
 * device math library's float64 pow (exact for temp == 1; within 4 ulp of the host's pow otherwise, tests/test_gpu_parity.py). */
int tafl_mcts_policy_device_ex(tafl_batch* b, double temp, uint64_t tie_seed, uint64_t game_id_base, double* out, int out_is_device);

/* ---- guided MCTS: src/mcts.py:55-136 with the CALLER's network as nnet.predict (mcts.py:85), SURVEY.md section 8f rank 3 ---
 * Lock-step over the batch: each tafl_gmcts_step (i) expands every waiting leaf with the priors / value the caller computed
 * for it (mask by the legal moves, renormalise with numpy's pairwise np.sum, all-masked workaround: mcts.py:86-98) and backs
 * the value up (mcts.py:127-136), then (ii) runs searches from the root until one reaches a state that needs predict();
 * searches that end in a terminal state are completed on the way.  A game is done after n_sims searches.
 *   priors  float32 [n * tafl_action_size], row g = network policy for game g's waiting leaf (rows of games that are not
 *           waiting are ignored); widened to float64 by the masking multiply as in mcts.py:87.
 *   values  float32 [n], used as Python floats (float(v)).
 *   first call after tafl_gmcts_begin: priors = values = NULL.  out_waiting (may be NULL) = games now waiting for predict().
 * tafl_gmcts_leaves: the network input for the waiting leaves: board_to_matrix planes uint8 [n * side_len * side_len]
 *   (game/main.rs:55-83), side to move [n] (TAFL_ATTACKER / TAFL_DEFENDER), waiting flag [n].
 * Device pointers (in_is_device / out_is_device = 1, e.g. torch tensors) keep the whole loop off PCIe.
 * tafl_gmcts_begin sizes the arena: max_sims + 1 nodes and (max_sims + 1) * edges_per_node edges per game (one edge per LEGAL
 * move of every expanded node); a game that outgrows it raises its fault flag and stops searching (stats.faults). */
typedef struct tafl_gmcts_stats {
    uint64_t sims, predicts, terminal_hits, faults, select_depth_sum, waiting, _reserved[2];
} tafl_gmcts_stats;                /* 64 bytes */
int tafl_gmcts_begin(tafl_batch* b, uint32_t max_sims, uint32_t edges_per_node);
int tafl_gmcts_step(tafl_batch* b, const float* priors, const float* values, int in_is_device, double c_puct, uint32_t n_sims,
                    uint32_t* out_waiting);
int tafl_gmcts_leaves(tafl_batch* b, uint8_t* boards, uint8_t* sides, uint8_t* waiting, int out_is_device);
int tafl_gmcts_root_children(tafl_batch* b, tafl_root_child* out, uint32_t max_children, uint32_t* out_n);
int tafl_gmcts_root_visits(tafl_batch* b, uint32_t* out, int out_is_device);
int tafl_gmcts_policy(tafl_batch* b, double temp, double* out, int out_is_device);
int tafl_gmcts_policy_ex(tafl_batch* b, double temp, uint64_t tie_seed, uint64_t game_id_base, double* out, int out_is_device);
int tafl_gmcts_get_stats(tafl_batch* b, tafl_gmcts_stats* out);

/* ---- replay buffer on disk: write_to_file, game/main.rs:86-132 ------------------------
 * Host-only, byte-exact text format of the reference: per record `side_len` lines of comma-separated matrix values, one line
 * with the comma-separated vector, one line value1, one line value2; every line ends in '\n'.
 * FIFO rule exactly as the reference implements it: the existing file is split into LINES, and if their number is
 * >= max_entries ONE line (the first) is dropped before the new record is appended (main.rs:98-106) — the cap counts lines,
 * not records.
 *   tafl_replay_append        one record (one write_to_file call).
 *   tafl_replay_append_batch  n records, identical to n consecutive tafl_replay_append calls in order, with one read and one
 *                             write of the file.  matrices[n*side_len*side_len]; record g's vector is
 *                             vectors[vector_offsets[g] .. vector_offsets[g+1]).
 *   tafl_replay_read          (the reference has no reader) the newest <= max_records complete records, oldest first, parsed
 *                             from the end of the file; vectors[k*vector_cap ..], vector_lens[k].  Any output may be NULL. */
int tafl_replay_append(const char* path, const uint8_t* matrix, uint8_t side_len, const uint8_t* vector, uint32_t vector_len,
                       uint8_t value1, uint8_t value2, uint64_t max_entries);
int tafl_replay_append_batch(const char* path, const uint8_t* matrices, uint8_t side_len, uint32_t n, const uint8_t* vectors,
                             const uint32_t* vector_offsets, const uint8_t* values1, const uint8_t* values2, uint64_t max_entries);
int tafl_replay_read(const char* path, uint8_t side_len, uint32_t max_records, uint8_t* matrices, uint8_t* vectors, uint32_t vector_cap,
                     uint32_t* vector_lens, uint8_t* values1, uint8_t* values2, uint32_t* out_n);

/* ---- measurement helpers (bench.py) -----------------------------------------------------------------
 * HIP-event timing on the ctx stream: average duration of the named kernel class since the last reset.
 * classes: 0 movegen, 1 step, 2 rollout, 3 mcts tree step (k_mcts_tree), 4 mcts playouts (k_mcts_rollout / k_mcts_fused) */
int tafl_mcts_round_trace(tafl_batch* b, uint32_t* requested, uint32_t* run, uint32_t cap, uint32_t* n_rounds); /* playouts requested /
    run in each round of the last tafl_mcts_run (two-kernel pipeline; 0 rounds after a fused search); arrays may be NULL */
int tafl_timing_enable(tafl_ctx* ctx, int enable);
int tafl_timing_reset(tafl_ctx* ctx);
int tafl_timing_get(tafl_ctx* ctx, int kernel_class, double* total_ms, uint64_t* launches);
int tafl_timing_get_union(tafl_ctx* ctx, int kernel_class, double* union_ms, double* sum_ms);   /* time with >= 1 launch of the class
    in flight (launches on different streams overlap) and the plain sum of the launch durations, since the last reset */
void* tafl_ctx_stream(tafl_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* TAFLHIP_H */
