/* azg.h -- C-ABI of the MI355X-native batched self-play engine (libazg_hip.so).
 *
 * The reference (cestpasphoto/alpha-zero-general) is 100% Python and has NO FFI; its boundary for this path is duck
 * typing.  This header is the C boundary a binding for the reference would call: every entry point
 * names the reference interface it replaces (file:line under the reference root).  The Python host side in
 * alpha-zero-general_amd/ mirrors Game.py / MCTS.py / Coach.py on top of exactly these functions through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; azg_last_error() gives the message (thread-local).
 *   - `*_dev` pointers are DEVICE pointers (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (NULL = default stream).  Kernels are enqueued on that stream and the call returns without synchronising, so
 *     they can be captured in a HIP graph.  Functions without a stream argument synchronise the device.
 *   - states are int8, C-contiguous, byte-identical to the reference's board.tobytes()
 *     (splendor/SplendorLogicNumba.py:207-219, santorini/SantoriniLogicNumba.py:658-665).
 *   - no callbacks into the host language; one host thread per forest handle.
 *
 * RNG contract (shared with oracle/rng.c):
 *     mix64(x): x^=x>>30; x*=0xBF58476D1CE4E5B9; x^=x>>27; x*=0x94D049BB133111EB; x^=x>>31
 *     raw(seed, stream, counter) = mix64(mix64(mix64(seed ^ 0x9E3779B97F4A7C15) + stream) + counter)
 *     u01 = (raw >> 11) * 2^-53;   stream = global game index, counter = per-game draw count.
 *   Per ply a self-play game draws: u_full (MCTS.py:58), u_pick (Coach.py:289-292), then whatever
 *   make_move(random_seed=0) consumes (SplendorLogicNumba.py:311-315).  u_pick drives np.random.choice(len(p), p=p) as NumPy does it
 *   (searchsorted of the normalised cumulative sum).  With temperature == 0 (Coach.py:278-283) the reference calls np.random.choice
 *   twice -- the unweighted choice among the maxima, k = floor(u_pick * n_maxima), then the weighted choice on the one-hot result --
 *   so a SECOND uniform is consumed after u_pick (fixture tests/golden/episode_splendor2_temp0.npz, played by the reference).
 */
#ifndef AZG_H
#define AZG_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { AZG_SPLENDOR = 0, AZG_SANTORINI = 1, AZG_AZUL = 2, AZG_MINIVILLES = 3, AZG_ABALONE = 4, AZG_TLP = 5, AZG_BOTANIK = 6, AZG_AKROPOLIS = 7, AZG_SMALLWORLD = 8 };
#define AZG_MAX_PLAYERS 5
#define AZG_MAX_UNIVERSES 8

const char* azg_last_error(void);
const char* azg_version(void);
int azg_device_count(void);
int azg_set_device(int device);

/* GameSwitcher.import_game + Game.getBoardSize/getActionSize/getNumberOfPlayers (GameSwitcher.py:15-24, Game.py:27-42).
   variant: Splendor = NUMBER_PLAYERS (2..4, splendor/SplendorGame.py:9); Santorini = NB_GODS (1 or 11,
   santorini/SantoriniConstants.py:19); Azul = 2 players (azul/AzulGame.py:9); Minivilles = NUMBER_PLAYERS (2..4,
   minivilles/MinivillesGame.py:9); Abalone = 2 players, Belgian-Daisy layout (abalone/AbaloneLogicNumba.py:5); The Little Prince =
   NUMBER_PLAYERS (3..5, thelittleprince/TLPGame.py:9); Botanik = 2 players; Akropolis = N_PLAYERS (2..4,
   akropolis/AkropolisConstants.py:3); Smallworld = NUMBER_PLAYERS (2..4, smallworld/SmallworldConstants.py).  0 = the shipped constant. */
int azg_game_info(int game, int variant, int* state_bytes, int* action_size, int* num_players, int* rows, int* cols);

/* ---- batched env step (one wavefront per state) ------------------------------------------------------------------
   Replace the per-call <G>Game adaptor methods (splendor/SplendorGame.py:28-48, santorini/SantoriniGame.py:28-48). */
/* Game.getValidMoves(board, player) -> bool[A]            out_valid_dev: u8[n][A] */
int azg_env_valid_moves(int game, int variant, const int8_t* states_dev, const int32_t* players_dev, int n,
                        uint8_t* out_valid_dev, void* stream);
/* Game.getNextState(board, player, action, random_seed) -> (board', next_player).  random_seed==0 draws from the RNG
   contract with (rng_seed, stream = stream0 + i, counter = counters_dev[i]) and advances counters_dev[i]. */
int azg_env_next_state(int game, int variant, const int8_t* states_dev, const int32_t* players_dev,
                       const int32_t* actions_dev, const int64_t* random_seeds_dev, int n, int8_t* out_states_dev,
                       int32_t* out_next_players_dev, uint64_t rng_seed, uint64_t stream0, uint64_t* counters_dev,
                       void* stream);
/* Game.getGameEnded(board, next_player) -> f32[P] ; getScore -> i32[P] ; getRound -> i32 */
int azg_env_game_ended(int game, int variant, const int8_t* states_dev, const int32_t* next_players_dev, int n,
                       float* out_ended_dev, int32_t* out_scores_dev, int32_t* out_round_dev, void* stream);
/* Game.getCanonicalForm(board, player) */
int azg_env_canonical(int game, int variant, const int8_t* states_dev, const int32_t* players_dev, int n,
                      int8_t* out_states_dev, void* stream);
/* Game.getInitBoard() for n games (Board.init_game); RNG contract with stream = stream0 + i, counter from 0;
   out_counters_dev (optional) receives the number of draws consumed. */
int azg_env_init_boards(int game, int variant, int n, int8_t* out_states_dev, uint64_t rng_seed, uint64_t stream0,
                        uint64_t* out_counters_dev, void* stream);
/* Game.getSymmetries (Game.py:96-109; Board.get_symmetries SplendorLogicNumba.py:255-301, SantoriniLogicNumba.py:578-653,
   AzulLogicNumba.py:310-331) for n (state, pi, valids) triples: form k of triple t goes to row [t][k] of
   out_states int8[n][max_sym][S], out_pi f32[n][max_sym][A], out_valids u8[n][max_sym][A]; out_count i32[n] = forms
   written (the reference's order: identity first; <= 10 + 2P for Splendor, 8 for Santorini, 120 for Azul).
   Coach.executeEpisode records every form of every full-search ply (Coach.py:66-69). */
int azg_env_symmetries(int game, int variant, const int8_t* states_dev, const float* pi_dev, const uint8_t* valids_dev, int n,
                       int max_sym, int8_t* out_states_dev, float* out_pi_dev, uint8_t* out_valids_dev,
                       int32_t* out_count_dev, void* stream);
/* The same with a random source, for games whose get_symmetries is itself random (The Little Prince, TLPLogicNumba.py:177-272:
   np.random.shuffle of players / market cards / planet slots, duplicate states dropped; <= 2P + 1 forms): triple t draws from
   the RNG contract's stream (rng_seed, stream0 + t), counter from 0, with shuffle(x) = Fisher-Yates from the top
   (j = floor(u (i + 1)) for i = len-1 .. 1).  Other games ignore the two arguments; azg_env_symmetries = (0, 0). */
int azg_env_symmetries_ex(int game, int variant, const int8_t* states_dev, const float* pi_dev, const uint8_t* valids_dev, int n,
                          int max_sym, int8_t* out_states_dev, float* out_pi_dev, uint8_t* out_valids_dev,
                          int32_t* out_count_dev, uint64_t rng_seed, uint64_t stream0, void* stream);

/* ---- forest: T independent MCTS trees, one wavefront per tree ----------------------------------------------------
   Replaces MCTS (MCTS.py:19-261) for a batch of trees and, in self-play mode, Coach.executeEpisode (Coach.py:37-84). */
typedef struct azg_forest_cfg {
    int game, variant;
    int n_trees;                 /* concurrent games on this GPU */
    int node_capacity;           /* nodes per tree (arena); GC keeps it bounded in self-play */
    int row_capacity_bytes;      /* per-tree row heap; 0 = auto */
    /* args.* of the reference (main.py:120-156, pit.py:49-57) */
    int numMCTSSims;
    double cpuct, fpu;
    int universes;
    double prob_fullMCTS;
    int ratio_fullMCTS;
    int forced_playouts;
    double dirichletAlpha;       /* 0 = no root noise; noise samples are supplied by the caller (torch Dirichlet) */
    double temperature[3];       /* [begin, end, root-softmax] (Coach.py:266-271, MCTS.py:148) */
    double tempThreshold;
    uint64_t rng_seed;
    uint64_t stream0;            /* global index of tree 0 (rank * n_trees in multi-GPU runs) */
    int max_examples;            /* capacity of the on-device example ring (self-play mode) */
    int level_budget;            /* max descent levels per tree per azg_forest_select launch; a deeper simulation is parked
                                    and resumed by the next launch (0 = unlimited).  Pure scheduling: results identical. */
    int work_budget;             /* cap on the work of one tree in one azg_forest_select launch, in level units (one descent
                                    level = 1, one frontier-edge resolution = 5; 0 = unlimited): a launch lasts as long as
                                    its slowest tree, so the few trees with a very deep or transposition-heavy simulation
                                    are parked at a level boundary and resume in the next launch.  Pure scheduling. */
    int gc_high_water_pct;       /* self-play clean-up (the reference's MCTS.py:86-91, every > 20 rounds): a tree whose arena holds more
                                    than this share of node_capacity when its next search begins is cleaned up first -- not only when
                                    the arena could not take another search -- so the arena keeps headroom and can be sized for the
                                    LIVE tree instead of live + many plies of dead nodes.  0 = only at exhaustion (rounds 1-3).  Which
                                    nodes survive does not depend on when the clean-up runs: results identical. */
} azg_forest_cfg;

typedef struct azg_forest azg_forest;

int azg_forest_create(const azg_forest_cfg* cfg, azg_forest** out);           /* MCTS.__init__ MCTS.py:24-47 */
int azg_forest_destroy(azg_forest* f);
size_t azg_forest_device_bytes(const azg_forest* f);

/* MCTS.reset_all_search_trees (MCTS.py:199-203) */
int azg_forest_reset(azg_forest* f, void* stream);

/* --- host-driven mode: the pieces of MCTS.getActionProb (MCTS.py:49-103) --- */
/* begin a search on every tree: canonical roots int8[T][S]; full_dev u8[T] (1 = full search, 0 = fast search, 2 = this
   tree sits the search out and keeps its contents -- e.g. the opponent's turn in an Arena) or NULL (all full).
   The tree is reused if the root state is already a node (MCTS.py:125-126). */
int azg_forest_begin_search(azg_forest* f, const int8_t* roots_dev, const uint8_t* full_dev, void* stream);
/* one lock-step round, part 1: every tree runs simulations until it needs the net (one leaf per tree) or has finished
   its numMCTSSims.  Writes the leaf batch for NeuralNet.predict (NeuralNet.py:32-43):
   leaf_states int8[T][S], leaf_valid u8[T][A], needs_eval u8[T].   MCTS.search :105-175 */
int azg_forest_select(azg_forest* f, int8_t* leaf_states_dev, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev,
                      const double* root_noise_dev /* f64[T][|noise_stride|] or NULL */, int noise_stride, void* stream);
/* root noise (MCTS.py:64,187-197): applied on simulation 0 of full searches when root_noise_dev != NULL.
   noise_stride > 0: rows hold iid Gamma(dirichletAlpha,1) variates, normalised on device over the root's n_valid
   first entries (== rng.dirichlet([alpha]*n_valid)); noise_stride < 0: rows already hold a Dirichlet sample over the
   valid actions (parity tests inject the reference's own sample), stride = -noise_stride.
   root_noise_dev == NULL with noise_stride == -1: the engine draws the Gamma variates itself (counter-based stream keyed
   by rng_seed / game stream / simulation count; cfg.dirichletAlpha > 0, or < 0 for the automatic 10/n_valid).
   root_noise_dev == NULL with noise_stride == -2 (self-play): same device sampler, but run by the next
   azg_selfplay_advance launch instead of a launch of its own; a tree whose root noise is pending sits out the selects
   in between (same per-tree event sequence, so results do not depend on how often advance is launched). */
/* part 2 + part 1 in one launch (self-play): first the expansion + backup of the leaves the PREVIOUS select handed out
   (pi / v as for azg_forest_expand_backup; leaf_valid must still hold that select's masks), then the next descent.  Saves
   a launch and its cold header reads per round; results are identical to expand_backup followed by select.
   noise_stride: 0 = no root noise, -2 = device sampler run by azg_selfplay_advance (see azg_forest_select). */
int azg_forest_select_fused(azg_forest* f, int8_t* leaf_states_dev, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev,
                            const float* pi_dev, const float* v_dev, int noise_stride, void* stream);
/* The per-CU form of the self-play round for Splendor 2 players + the V80 net (csrc/azg_fused.hip.h): ONE launch runs `rounds` rounds of
       azg_forest_select_fused  ->  azg_nn_v80_forward_h2 on the leaf batch
   with one 16-wave workgroup per 16 trees doing both -- every wave descends one tree, then the same waves evaluate the workgroup's 16
   leaves -- so that a round of a CU waits for the slowest of ITS 16 trees instead of the slowest of all T, and nothing is launched
   between the rounds (what the reference does with a lock ring of N game threads around one inference batch, Coach.py:117-144).
   Arguments as for those two calls (same buffers, same 43-pointer weight table + 16 descale factors); per-tree results are identical
   bit for bit to `rounds` times the two calls (tests/test_gpu_selfplay.py).  azg_selfplay_advance is launched by the caller between
   launches, as between rounds of the two-kernel form. */
int azg_forest_rounds_v80_h2(azg_forest* f, int8_t* leaf_states_dev, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev, float* pi_dev,
                             float* v_dev, int noise_stride, const void* const* w, const float* descale_host, int rounds, void* stream);
/* measurement: phase times of that kernel since the last reset, from the 100 MHz wall clock read inside the kernel, averaged over the
   workgroups -- out[0] = select phase of a round (a workgroup waits for the slowest of its 16 trees), out[1] = net phase, out[2] = a
   wave's own descent, all in microseconds per round; out[3] = rounds measured.  (bench.py's roofline uses out[0]: there is no launch
   of the descent alone to put HIP events around.) */
int azg_forest_rounds_profile(azg_forest* f, double* out4, int reset);

/* The ASYNCHRONOUS TREE PIPELINE form of the self-play round for Splendor 2 players + the V80 net (csrc/azg_async.hip.h): games search
   while other games sit in the net -- what the reference's lock ring of N game threads around one inference server does
   (Coach.py:117-144, GenericNNetWrapper.py:122-157).  ONE call launches two persistent kernels that run concurrently: `n_sel` descent
   workgroups (16 waves; each owns a fixed share of the trees, a free wave takes whichever of them has its pi / v, runs the expansion +
   backup + next descent of azg_forest_select_fused and queues the leaf) and `n_net` net workgroups (each takes up to 16 queued leaves --
   fewer after `batch_wait_ticks` x 10 ns of waiting --, runs the forward of azg_nn_v80_forward_h2 on them and hands the trees back).
   A "call" of a tree is what one round of the two-kernel form does for it (expansion + backup of its evaluated leaf, then descents until a
   leaf needs the net, the search ends or the work budget parks it).  shared_budget == 0: every tree has exactly `rounds` calls, then the
   kernels end -- results are a function of `rounds` alone.  shared_budget != 0: the kernels end when the trees TOGETHER have had rounds x T
   calls; a fast tree gets more of them, no tree waits for the slowest at the end of the launch (per-game results are the same, how far
   each game gets is not fixed).  What azg_selfplay_advance does between rounds -- the move, the example record, the restart, the
   clean-up, the next search, the root noise -- happens inside, per tree, the moment its search is finished: no call of
   azg_selfplay_advance is needed (or harmful) between launches.  n_net + n_sel must not exceed the CUs of the device (every workgroup has to
   be resident; <= 0: a default split); the split may change from call to call.  Per-tree results are identical bit for bit to
   `rounds` x (azg_forest_select_fused -> azg_selfplay_advance -> azg_nn_v80_forward_h2) with shared_budget == 0 (tests/test_gpu_selfplay.py).
   leaf_valid_dev u8[T][A], needs_eval_dev u8[T], pi_dev f32[T][A], v_dev f32[T][P]: as for azg_forest_select_fused (the leaf states
   travel through a buffer the forest owns).
   Determinism: with shared_budget != 0 the GAMES played are the same as with per-tree budgets (a tree's sequence of calls is its own),
   but how many of them are finished after N launches depends on GPU timing; callers that need "the examples after N runs" to be a function
   of the seed use shared_budget == 0 (Python: SelfPlayEngine(deterministic=True) or AZG_DETERMINISTIC=1; 3-4 % slower).
   Exclusive use: every workgroup of both kernels must be resident, so the device's CUs must not be held by other kernels for long (two
   engines on one GPU split the CUs with n_net / n_sel).
   Time-outs and recovery (csrc/azg_async.hip.h "Recovery"): a wave that finds nothing to do for AZG_ASYNC_TIMEOUT_MS (default 50) ends
   the launch.  That is a launch that ENDS EARLY, not an error: on this platform a few workgroups are occasionally not scheduled for
   about a second; the leaves they held are re-queued by the next launch (a small kernel in front of it), nothing is evaluated twice or
   lost, and the early end is counted (azg_forest_async_profile out[17] / [18]: ended by a descent / a net wave, [26] ticket ranges
   abandoned, [27] leaves re-queued).  With per-tree budgets every tree keeps the calls it has not run yet and a call of this function
   launches the pipeline TWICE -- the second launch grants nothing and only runs what the first left over -- so the "exactly `rounds` calls
   per tree" contract holds at every return whether or not the first launch ended early (both ending early: the next call catches up).
   Eight early ends in a row set error bit 128 (azg_selfplay_stats.errors): a pipeline that cannot make progress fails loudly.
   rounds: 1 .. 2^24 - 1 with a shared budget, 1 .. 2^22 - 1 with per-tree budgets; rounds == 0 with per-tree budgets: a catch-up launch only
   (Python's SelfPlayEngine(deterministic=True).run() issues these until the early-end counters [17] / [18] stop moving); < 0, or 0 with
   a shared budget: the call does nothing.  Early ends are counted by the net kernel's workgroup 0, which leaves only when the launch is over. */
int azg_forest_async_rounds_v80_h2(azg_forest* f, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev, float* pi_dev, float* v_dev,
                                   int noise_stride, const void* const* w, const float* descale_host, int rounds, int n_net, int n_sel,
                                   int batch_wait_ticks, int shared_budget, void* stream);
/* The same pipeline for a Santorini no-gods forest with the V89 net: w = the 14 device pointers and descale of azg_nn_conv5_forward_h2, 8
   leaves per forward; default split 49 / 64 of the CUs for the net.  Everything else as above. */
int azg_forest_async_rounds_conv5_h2(azg_forest* f, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev, float* pi_dev, float* v_dev,
                                     int noise_stride, const float* const* w, float descale, int rounds, int n_net, int n_sel,
                                     int batch_wait_ticks, int shared_budget, void* stream);
/* ... and for Splendor 3 / 4 players and Azul with their MobileNet-1d nets: geometry, w (43 device pointers) and descale (16 host floats) as
   for azg_nn_mb1d_forward_h2; 8 (Splendor) / 16 (Azul) leaves per forward. */
int azg_forest_async_rounds_mb1d_h2(azg_forest* f, int geometry, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev, float* pi_dev, float* v_dev,
                                    int noise_stride, const void* const* w, const float* descale_host, int rounds, int n_net, int n_sel,
                                    int batch_wait_ticks, int shared_budget, void* stream);
/* measurement: counters of the pipeline since the last reset (ticks = 10 ns of the 100 MHz wall clock read inside the kernels):
   out[0] descents (select_tree calls), [1] ticks inside them, [2] ticks descent waves spent looking for a ready tree, [3] net batches,
   [4] leaves in them, [5] ticks inside the forward, [6] ticks net workgroups waited for leaves, [7] sum over leaves of (claimed by a net
   workgroup - queued), [8] sum over descents of (tree claimed - tree handed back by the net), [9] launches, [10] / [11] resident ticks
   summed over the select / net workgroups, [12] n_sel, [13] n_net, [17] / [18] launches ended early by a descent / net wave's time-out,
   [26] ticket ranges abandoned, [27] leaves re-queued at the start of a launch, [32..63] histogram of [7]'s waits in microseconds (last bucket: >= 31), [64..95] of [8]'s. */
#define AZG_ASYNC_NPROF 96
int azg_forest_async_profile(azg_forest* f, double* out96, int reset);

/* part 2: store (Ps, v) on the pending leaves and back the values up (MCTS.py:147-154,176-183).
   pi f32[T][A] are PROBABILITIES (exp of the net's log-softmax, GenericNNetWrapper.py:107,119), v f32[T][P].
   The leaf_valid buffer handed to the preceding azg_forest_select must still hold what that call wrote (the kernel maps
   entry j of a leaf to its j-th valid action through it). */
int azg_forest_expand_backup(azg_forest* f, const float* pi_dev, const float* v_dev, const double* root_noise_dev,
                             int noise_stride, void* stream);
/* number of trees that still have simulations to run (synchronises) */
int azg_forest_active(azg_forest* f, int* n_active);
/* MCTS.getActionProb epilogue (:67-103): probs f64[T][A] for temperature temp_dev f64[T] (or scalar temp if NULL),
   q f32[T][P], is_full u8[T].  Any output may be NULL. */
int azg_forest_action_probs(azg_forest* f, double temp, double* probs_dev, float* q_dev, uint8_t* is_full_dev,
                            void* stream);
/* root statistics for parity checks: Ns i32[T], Qs f32[T], Nsa i32[T][A], Qsa f64[T][A] (sentinel -42 where unvisited),
   Ps f32[T][A], n_nodes i32[T] */
int azg_forest_root_stats(azg_forest* f, int32_t* Ns_dev, float* Qs_dev, int32_t* Nsa_dev, double* Qsa_dev,
                          float* Ps_dev, int32_t* n_nodes_dev, void* stream);
/* whole-tree dump of one tree to HOST memory for parity tests (synchronises): for every node: state[S], flags,
   Ns, Qs, Es[P], and dense Nsa/Qsa/Ps[A].  Returns node count; arrays sized for max_nodes. */
int azg_forest_dump_tree(azg_forest* f, int tree, int max_nodes, int8_t* states, int32_t* Ns, float* Qs, float* Es,
                         int32_t* Nsa, double* Qsa, float* Ps, uint8_t* has_policy);

/* XCD-pinned streams (no reference counterpart: the reference time-slices N game threads on one core, Coach.py:117-144; here groups
   of games can run as independent select -> predict pipelines, selfplay.py): a HIP stream whose queue ASKS for the CUs of XCDs
   [xcd_first, xcd_first + xcd_count) of the current device only (hipExtStreamCreateWithCUMask; bit b of the mask = CU b / 8 of XCD
   b % 8).  Whether the mask is honoured is up to the platform: on the round-4 MI355X box (one partition over 8 XCDs) it was NOT -- the
   hardware deals a queue's workgroups round-robin over all XCDs; azg_testaids.h has a probe that shows where a launch really ran.  Pass the
   handle as `stream` to any call of this header. */
int azg_stream_create_xcd(int xcd_first, int xcd_count, void** out_stream);
int azg_stream_destroy(void* stream);
/* debug / tests: check the structural invariants of every tree on the host; returns the number of violations */
int azg_forest_validate(azg_forest* f, int verbose);

/* --- self-play mode: Coach.executeEpisode on device (Coach.py:37-84) --- */
/* start one game per tree (Board.init_game or the given init boards int8[T][S]) */
int azg_selfplay_start(azg_forest* f, const int8_t* init_boards_dev /* or NULL */, void* stream);
/* the same with (a) an RNG epoch: epoch 0 == azg_selfplay_start; another epoch re-keys every random stream of the forest, so
   the next wave of episodes of one Coach.learn run (Coach.py:150-215 draws fresh randomness every iteration) does not replay
   the previous one; (b) an episode quota = Coach.executeEpisodes' numEps (Coach.py:86-148): tree t plays quota / T (+1 for
   t < quota % T) games to their end and then idles -- every started game is finished and kept; 0 = restart forever; -1 = this
   forest plays no game (every tree idle: a group of a larger engine whose share of numEps is empty).
   Seed and quota are kernel arguments: HIP graphs that captured this forest's launches under another epoch / quota must be
   captured again. */
int azg_selfplay_start_ex(azg_forest* f, const int8_t* init_boards_dev /* or NULL */, uint64_t epoch, int64_t episode_quota,
                          void* stream);
/* sizeof(azg_forest_cfg) of this build: a binding compares it with its own declaration before it hands a cfg over */
int azg_forest_cfg_size(void);
/* to be called after expand_backup, every round or every few rounds: trees whose search finished sample the move
   (Coach.py:63,278-292), record the example (Coach.py:65-69), play it (Coach.py:71), detect the end (Coach.py:73-82),
   restart finished games, re-root and begin the next search -- all on device. */
int azg_selfplay_advance(azg_forest* f, void* stream);
/* trees that are still playing (synchronises): 0 once every tree has used up its episode quota (or stopped on an error) */
int azg_selfplay_active(azg_forest* f, int* n_active);
/* counters (synchronises): plies executed, games finished, simulations run, examples stored, error flags.
   errors = OR over trees: 1 node arena full, 2 record heap full, 4 descent deeper than 256, 8 bad state, 16 / 32 example ring / per-game
   record buffer full, 64 every pruned root count was 0 (numMCTSSims too small for the game's number of valid actions with
   forced_playouts: the reference divides 0 / 0 at MCTS.py:77-80,100-102 and raises).  A tree with an error is parked. */
typedef struct azg_selfplay_stats {
    uint64_t plies, games, sims, levels, expansions, sum_valid_visited, terminal_hits, examples, gc_runs, max_nodes,
        errors, sum_depth_at_expand,
        cyc_select, cyc_levels, cyc_edge, cyc_leaf,   /* (library built with -DAZG_CYC_COUNTERS only, else 0) shader-clock cycles summed over trees: whole k_select, descent levels,
                                                         frontier edge resolution (incl. leaf creation), leaf creation */
        cyc_seg[4],                                   /* frontier edge split: parent-state load, env step, canonical form +
                                                         hash, table probe */
        max_live_after_gc,                            /* most nodes of one tree that survived a clean-up: node_capacity must stay
                                                         above this + numMCTSSims */
        examples_dropped;                             /* records of finished games that did not fit the example ring (a game is
                                                         stored whole or not at all); != 0 also sets error bit 16 */
} azg_selfplay_stats;
int azg_selfplay_stats_get(azg_forest* f, azg_selfplay_stats* out);
/* drain finished-game examples: (board int8[S], pi f32[A], z f32[P], valids u8[A], q f32[P]) per record
   (Coach.py:76-82).  Returns count copied (<= max_records), device->device on `stream` then synchronises. */
int azg_selfplay_drain_examples(azg_forest* f, int max_records, int8_t* boards_dev, float* pi_dev, float* z_dev,
                                uint8_t* valids_dev, float* q_dev, int32_t* meta_dev /* i32[n][4] = (global game
                                stream, game index on it, ply, player) or NULL */, int* n_out, void* stream);

/* ---- policy/value net building blocks: NeuralNet.predict for a whole leaf batch (NeuralNet.py:32-43,
   GenericNNetWrapper.py:94-120; V80 network splendor/SplendorNNet.py:148-202,262-283,397-440), fp32 ----
   out[M][N] = act(A'[M][K] @ W + bias) (+ R);  A' = A * rowscale[row / rows_per_group][k] when rowscale != NULL (fuses the
   SqueezeExcitation multiply into the project GEMM).  Wp is the weight pre-padded with zeros to [Kp][NP], Kp = K rounded
   up to 16, NP/16 in {1,4,6,11}.  act: 0 none, 1 ReLU, 2 Hardswish, 3 Hardsigmoid.  K, lda % 4 == 0.  ksplit = 1 selects the variant for
   M ~ batch and long K (flatten->Linear heads).  MFMA v_mfma_f32_16x16x4_f32 (exact f32 == an fmaf chain). */
int azg_nn_linear(const float* A_dev, int lda, const float* Wp_dev, int Kp, int NP, const float* bias_dev,
                  const float* R_dev, int ldr, const float* rowscale_dev, int rows_per_group, float* out_dev, int ldc,
                  int M, int K, int N, int act, int ksplit, void* stream);
/* Same contract, weight-stationary kernel (weights in registers as the MFMA A operand, activations streamed as float4,
   float4 epilogue, no LDS): the variant the V80 forward uses for every layer.  bias_padded: NP floats (zero padded) or
   NULL; ldc (and ldr) % 4 == 0; Kp/16 in {1,3,4,6,11,25}. */
int azg_nn_linear_ws(const float* A_dev, int lda, const float* Wp_dev, int Kp, int NP, const float* bias_padded_dev,
                     const float* R_dev, int ldr, const float* rowscale_dev, int rows_per_group, float* out_dev, int ldc,
                     int M, int K, int N, int act, void* stream);
/* depthwise Linear(7->7) over the token axis + folded BatchNorm + activation, in place on H[B*7][ldh], and the SE squeeze
   pooled[B][E] (mean or max over the 7 tokens)  (SplendorNNet.py:148-187).  The SE excitation
   hardsigmoid(relu(pooled @ W1 + b1) @ W2 + b2) is two azg_nn_linear calls (act 1, then act 3 = Hardsigmoid). */
int azg_nn_dw_pool(float* H_dev, int ldh, const float* Wd_dev /*[7][7] out,in*/, const float* bn_scale_dev,
                   const float* bn_bias_dev, float* pooled_dev, int B, int E, int act, int pool_max, void* stream);
/* the same for L tokens per sample (7: Splendor boards [C][7]; 6: Azul boards [23][6], AzulNNet.py:91-113) */
int azg_nn_dw_pool_l(float* H_dev, int ldh, const float* Wd_dev /*[L][L] out,in*/, const float* bn_scale_dev,
                     const float* bn_bias_dev, float* pooled_dev, int B, int E, int L, int act, int pool_max, void* stream);
/* One fused InvertedResidual1d block of the V80 net (SplendorNNet.py:189-202: expand + depthwise + SE + project +
   residual) for x[B*7][56] -> out[B*7][56]; the 168-wide expanded activations stay in LDS.  w = 11 HOST-array device
   pointers {We[64][176], be[176], Wd[7][7], bn_scale[168], bn_bias[168], W1[176][48], b1[48], W2[48][176], b2[176],
   Wp[176][64], bp[64]} (zero padded, BatchNorm folded).  The four matrices are stored in MFMA fragment order:
   frag[n/16][k/16][lane 0..63][j 0..3] = W[16*(k/16) + 4*(lane>>4) + j][16*(n/16) + (lane&15)] (one 1 KiB load per wave and
   K chunk).  act 1 ReLU / 2 Hardswish; pool_max 0 mean / 1 max. */
int azg_nn_v80_block(const float* xin_dev, float* xout_dev, const float* const* w, int B, int act, int pool_max,
                     void* stream);
/* The whole V80 forward (NeuralNet.predict for a leaf batch, SplendorNNet.py:397-440 / GenericNNetWrapper.py:94-110) in
   ONE launch, one workgroup pass per 16 samples: boards int8[B][56][7] -> first_layer + trunk block (output kept in LDS,
   two copies) -> policy head block + Flatten + Linear + ReLU + Linear + masked softmax -> pi f32[B][81]; value head block
   + Flatten + Linear + ReLU + Linear + tanh -> v f32[B][P].  x_trunk (f32 [B*7][56]) is only written by the three-launch
   variant (environment AZG_NN_THREE_LAUNCHES=1, kept for A/B measurements).  w = 43 device pointers: {W0[64][64], b0[64]}, 3 x the 11 block tensors of
   azg_nn_v80_block (trunk, policy head, value head), {Wpi1[432][96], bpi1[96], Wpi2[96][96], bpi2[96]},
   {Wv1[432][16], bv1[16], Wv2[P][P], bv2[P]}; the flatten index of Wpi1 / Wv1 rows is k = l*60 + c (c < 56), zero
   padded; every matrix except Wv2 is in the fragment order described at azg_nn_v80_block.  Fixed to the V80 activations (trunk ReLU + mean squeeze, heads Hardswish + max squeeze). */
int azg_nn_v80_forward(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int B, int P,
                       float* x_trunk_dev, float* pi_dev, float* v_dev, void* stream);
/* The same forward with the expand GEMM of the three blocks on split-precision operands (every f32 value as three bf16 numbers
   hi + mid + lo, a product = the six bf16 MFMAs of relative weight >= 2^-24; f32-input MFMA runs at 1/16 of the bf16 rate on
   gfx950): the tile the blocks read lives in LDS as three bf16 planes, written split once by the producing epilogue; one copy of
   the trunk output serves both heads.  Only We (w[2], w[13], w[24]) differs: [11 column tiles][2 K chunks of 32][3 planes]
   [64 lanes][8] bf16 with element = W_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)], K zero padded 56 -> 64. */
int azg_nn_v80_forward_split(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int B, int P,
                             float* x_trunk_dev, float* pi_dev, float* v_dev, void* stream);
/* The same forward (NeuralNet.predict for a leaf batch, SplendorNNet.py:397-440 / GenericNNetWrapper.py:94-110) on fp16 hi+lo
   split operands (every f32 weight / activation as hi = rn16(x), lo = rn16(x - hi): 22 significant bits; a product = three
   v_mfma_f32_16x16x32_f16) with TOKEN-MAJOR tiles, so that the depthwise token mix, its BN + activation, the squeeze and the SE
   scale run on MFMA accumulators in registers (csrc/nn_v80_h2.hip.h).  Same 1e-5 contract as azg_nn_v80_forward.
   w = the 43 slots of azg_nn_v80_forward with every matrix except Wd / Wv2 as h2 fragments: zero padded to K % 32 == 0,
   N % 16 == 0 -- W0[64][64], We[64][176], W1[192][48], W2[64][176], Wp[192][64], Wpi1[448][96] and Wv1[448][16] (row k =
   token*64 + c), Wpi2[96][96] -- scaled by a power of two 2^k per matrix and stored as
   [N/16 tiles][K/32 chunks][2 planes hi, lo][64 lanes][8] f16, element = W_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)];
   vectors zero padded f32: b0[64], be / sd / bd / b2[176], b1[48], bp[64], bpi1 / bpi2[96], bv1[16].
   The token-mix matrices Wd[7][7] of the two Hardswish blocks (policy, value head) are passed DIVIDED BY 6: the kernel computes
   6 * Hardswish and leaves the 1/6 to the next linear step.
   descale (HOST array of 16 floats) = 2^-k / 64 for W0, {We, W1, W2, Wp} x (trunk, policy, value), Wpi1, Wpi2, Wv1. */
/* Activation range of the f16 x 2 ("h2") forwards -- azg_nn_v80_forward_h2, azg_nn_mb1d_forward_h2, azg_nn_conv5_forward_h2,
   azg_nn_s78_forward_h2: their LDS planes hold 64 * x as f16 (hi) + f16 (lo), so the 1e-5 contract holds for activations and
   residual-stream values |x| < 1023 (the reference's trained nets stay below ~50).  Beyond that the value SATURATES: the kernels run
   with the FP16_OVFL bit of the wave MODE register set, so an f16 conversion that overflows gives +-65504 instead of inf -- pi / v stay
   finite, never NaN (tests/test_nnet.py::test_h2_kernels_saturate_out_of_range_activations_gpu).  A net that may leave the range (e.g.
   a diverging training run) is evaluated with the f32-operand kernels (azg_nn_v80_forward / _mb1d_forward / _conv5_forward /
   _s78_forward; Python: h2=False), which have the full f32 range at 1/3 .. 1/2 of the speed. */
int azg_nn_v80_forward_h2(const int8_t* boards_dev, const uint8_t* valid_dev, const void* const* w, const float* descale_host,
                          int B, int P, float* pi_dev, float* v_dev, void* stream);
/* The whole MobileNetV3-1d forward (first layer, trunk block, policy block + head, value block + head) in one launch for
   the geometries of the reference's Splendor (SplendorNNet.py:259-283, n players: C = 32 + 10n + n^2 channels x 7 tokens)
   and Azul (AzulNNet.py:91-113: 23 channels x 6 tokens) nets, and of the shipped nets of Minivilles (MinivillesNNet.py:101-123: 58 x 2)
   and The Little Prince (TLPNNet.py:175-196: 55 x 15) -- the generic sibling of azg_nn_v80_forward.
   boards int8 [B][C][L], valid u8 [B][A] -> pi f32 [B][A] (probabilities), v f32 [B][P].  w = 43 device pointers:
   {W0, b0}, 3 x {We, be, Wd[L][L], bn_scale_d, bn_bias_d, W1, b1, W2, b2, Wp, bp} (trunk, policy head, value head),
   {Wpi1, bpi1, Wpi2, bpi2, Wv1, bv1, Wv2[P][P], bv2}; every matrix but Wd / Wv2 is zero-padded to multiples of 16 in both
   dimensions and stored in MFMA fragment order frag[N/16][K/16][64][4] = W[16c + 4*(lane>>4) + j][16nt + (lane&15)],
   every vector zero-padded to a multiple of 16; the rows of Wpi1 / Wv1 are indexed l*OS + c with OS = 16*ceil(max(C,
   policy-block channels)/16) + 4. */
enum { AZG_NET_SPLENDOR2 = 0, AZG_NET_SPLENDOR3 = 1, AZG_NET_SPLENDOR4 = 2, AZG_NET_AZUL = 3,
       AZG_NET_MINIVILLES2 = 4,   /* minivilles/MinivillesNNet.py:101-123 nn_version 82, 2 players: C = 58, L = 2, A = 21 */
       AZG_NET_TLP3 = 5 };        /* thelittleprince/TLPNNet.py:175-196 nn_version 83, 3 players: C = 55, L = 15, A = 9 */
int azg_nn_mb1d_forward(int geometry, const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int B,
                        float* pi_dev, float* v_dev, void* stream);
/* The same forward with every GEMM phase on f16 x 2 split-precision operands (hi + lo halves, three v_mfma_f32_16x16x32_f16 per
   K chunk of 32, fp32 accumulation: <= 1e-5 of the fp32 forward); the activations stay fp32 in the LDS and are split as they
   are read.  w: the same 43 pointers, every matrix but Wd / Wv2 as an azg_nn_v80_forward_h2 fragment array (K zero-padded to a
   multiple of 32: 16*ceil(K/16) rounded up; N to a multiple of 16); descale_host[16] as there. */
int azg_nn_mb1d_forward_h2(int geometry, const int8_t* boards_dev, const uint8_t* valid_dev, const void* const* w,
                           const float* descale_host, int B, float* pi_dev, float* v_dev, void* stream);
/* The Santorini ResNet (santorini/SantoriniNNet.py nn_version 88/89 :194-219,273-281: conv3x3(2->64)+BN+ReLU, n_blocks
   SimpleResBlocks :71-84, SimpleHead heads :17-40) in one launch.  boards int8 [B][5][5][3] (planes 0, 1 are the net's
   input), valid u8 [B][A] -> pi f32 [B][A], v f32 [B][P].  w = 14 device pointers {W0, b0, Wc, bc, Wp, bp, Wfp, bfp, Wv, bv,
   Wf1, bf1, Wf2, bf2}: BatchNorm folded; W0 [9*16][64] and the 2*n_blocks trunk matrices Wc [9*64][64] (row = tap*Cin + ci,
   tap = ky*3 + kx, column = co) in the MFMA fragment order of azg_nn_mb1d_forward, back to back; bc [2*n_blocks][64];
   Wp [64][2], Wfp [50][A] (row = c*25 + cell), Wv [64], Wf1 [25][64], Wf2 [64][P] plain row-major.
   Built for the no-gods geometry (n_blocks = 5, A = 162, P = 2). */
int azg_nn_conv5_forward(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int n_blocks, int A, int P,
                         int B, float* pi_dev, float* v_dev, void* stream);
/* The same forward with the 2*n_blocks trunk convolutions on split-precision operands: every f32 weight / activation is carried
   as three bf16 numbers (hi + mid + lo = 24 significant bits) and a product is the six bf16 MFMAs of weight >= 2^-24, accumulated
   in f32 (f32-input MFMA runs at 1/16 of the bf16 rate on gfx950; outputs stay within the 1e-5 contract).  Only w[2] differs:
   Wc = [2*n_blocks][4 column tiles][18 K chunks of 32][3 planes hi, mid, lo][64 lanes][8] bf16 with
   element = W_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)], row index K = tap*64 + ci. */
int azg_nn_conv5_forward_split(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int n_blocks, int A,
                               int P, int B, float* pi_dev, float* v_dev, void* stream);
/* The same net with the trunk on f16 x 2 split-precision operands (hi = rn16(x), lo = rn16(x - hi): 22 significant bits, three
   v_mfma_f32_16x16x32_f16 per product instead of the six of bf16 x 3; two activation planes per tile holding 64 * x).
   w[2] = the ten 64 -> 64 convolutions as [4 ct][18 chunks][2 planes hi, lo][64 lanes][8] f16 of W * 2^k (one k for the whole net),
   element = W_plane[32*chunk + 8*(lane>>4) + j][16*ct + (lane&15)], K = tap*64 + ci; descale = 2^-k / 64.  Same 1e-5 contract. */
int azg_nn_conv5_forward_h2(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, float descale, int n_blocks,
                            int A, int P, int B, float* pi_dev, float* v_dev, void* stream);
/* The Santorini-with-gods net (nn_version 78, SantoriniNNet.py:167-192,264-271, HeadWithMeta :42-69): two launches on `stream`
   (trunk + value head; policy FC + masked softmax -- the pi rows carry the 132 policy features in between).
   boards int8 [B][5][5][3] (planes 0, 1 = workers / levels, plane 2 = gods and metadata), valid u8 [B][A] -> pi, v.
   w = 19 device pointers {W0, We, be, Wd, bd, Wp, bp, Wm, bm, Whp, bhp, Wfp, bfp, Whv, bhv, Wf1, bf1, Wf2, bf2}: BatchNorm
   folded; W0 [9*16][64], the n_blocks expand matrices We [64][192] and project matrices Wp [192][64] in MFMA fragment order,
   back to back; Wd [n_blocks][192][9] (channel, tap = ky*3 + kx); biases [n_blocks][192 / 192 / 64]; Wm [25][32];
   Whp [64][4], Whv [64][2], Wf1 [82][64], Wf2 [64][P] plain row-major; Wfp [132][A] (rows: channel*25 + cell, then the 32
   metadata features) zero padded to [144][1792] in MFMA fragment order, bfp padded to 1792.  Built for n_blocks = 10, A = 1782, P = 2. */
int azg_nn_s78_forward(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int n_blocks, int A, int P,
                       int B, float* pi_dev, float* v_dev, void* stream);
/* The same net with the 1x1 convolutions of the trunk on split-precision operands (bf16 x 3, as azg_nn_conv5_forward_split): 8
   samples per workgroup, the expanded tile processed in thirds of 64 channels.  Only w[1] and w[5] differ:
   We / Wp = [n_blocks][3 thirds][4 column tiles][2 K chunks of 32][3 planes hi, mid, lo][64 lanes][8] bf16 with
   element = M_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)], M = We[:, 64 t .. 64 t + 63] resp. Wp[64 t .. 64 t + 63, :]. */
int azg_nn_s78_forward_split(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, int n_blocks, int A, int P,
                             int B, float* pi_dev, float* v_dev, void* stream);
/* The same with the trunk's 1x1 convolutions and the depthwise pass on f16 x 2 split-precision operands (hi + lo, three
   v_mfma_f32_16x16x32_f16 per product, two planes per tile holding 64 * x): We / Wp = [n_blocks][3 thirds][4 ct][2 chunks]
   [2 planes hi, lo][64 lanes][8] f16 of W * 2^k (one k per family), ds_e / ds_p = 2^-k / 64.  The policy FC runs on the same operands
   (k_s78_policy_h2): w[11] = [112 column tiles][5 K chunks of 32][2 planes hi, lo][64 lanes][8] f16 of Wfp * 2^k (K 132 -> 160,
   N 1782 -> 1792, zero padded; element = W_plane[32*chunk + 8*(lane>>4) + j][16*tile + (lane&15)]) followed by ONE float, its descale
   2^-k / 64 (the 16-byte tail of the buffer).  Three launches: trunk + value head; the FC as a GEMM (a workgroup = 64 samples x a quarter
   of the column tiles: every weight fragment serves four sample groups; raw logits go to a per-device workspace of B x 1792 floats that
   the library keeps and grows on demand); masked softmax into pi.  AZG_S78_POLICY2=0: FC + softmax in one launch (k_s78_policy_h2,
   16 samples per workgroup, no workspace).  Same 1e-5 contract. */
int azg_nn_s78_forward_h2(const int8_t* boards_dev, const uint8_t* valid_dev, const float* const* w, float ds_e, float ds_p, int n_blocks,
                          int A, int P, int B, float* pi_dev, float* v_dev, void* stream);
/* boards int8 [B][C][7] (reference board layout) -> x f32 [B][7][C] */
int azg_nn_board_to_x(const int8_t* boards_dev, float* x_dev, int B, int C, void* stream);
/* boards int8 [B][C][L] -> x f32 [B][L][ldx], columns C..ldx-1 zeroed (row stride padded to a multiple of 4 floats) */
int azg_nn_board_to_x_ld(const int8_t* boards_dev, float* x_dev, int B, int C, int L, int ldx, void* stream);
/* pi = softmax(where(valid, logits, -1e8)) (== exp(log_softmax), GenericNNetWrapper.py:107); v = tanh(relu(vhid) @ Wv2 + bv2) */
int azg_nn_heads_out(const float* logits_dev, int ldl, const uint8_t* valid_dev, const float* vhid_dev, int ldv,
                     const float* Wv2_dev, const float* bv2_dev, float* pi_dev, float* v_dev, int B, int A, int P,
                     void* stream);

/* ---- measurement helpers ---- */
/* time `iters` back-to-back launches of kernel `which` (0 = select, 1 = expand_backup) with hipEvents on `stream`
   (state is mutated like normal rounds); returns average milliseconds per launch. */
int azg_forest_last_kernel_ms(azg_forest* f, int which, double* avg_ms, uint64_t* launches);
int azg_forest_enable_timing(azg_forest* f, int enable);
/* args.numMCTSSims / args.prob_fullMCTS for the searches that BEGIN after this call (MCTS.py:58-59 reads both at every
   getActionProb call); searches in flight keep their size.  Both are kernel arguments: HIP graphs that captured this forest's
   launches must be captured again (SelfPlayEngine.set_search_params does). */
int azg_forest_set_search_params(azg_forest* f, int numMCTSSims, double prob_fullMCTS);

#ifdef __cplusplus
}
#endif
#endif
