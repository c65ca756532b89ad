/* azg_testaids.h -- TEST AND DEBUGGING AIDS exported by libazg_hip.so that have NO counterpart in the reference and are NOT part of the
   drop-in surface (include/azg.h is that surface: every entry point there replaces a call of the reference).  Used by tests/, tools/ and
   the plugin benches only; nothing under alpha-zero-general_amd/ needs them to play, search or train. */
#ifndef AZG_TESTAIDS_H
#define AZG_TESTAIDS_H
#include "azg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* debug: fill the LDS of every CU and the queue's scratch memory with `pattern` (a kernel that reads on-chip memory it never wrote then
   depends on the pattern, not on what ran before) */
int azg_debug_poison_onchip(uint32_t pattern, void* stream);

/* debug / tests: launch n_workgroups one-wave workgroups on `stream`; out_dev[i] = XCC_ID | cu_id << 8 | se_id << 16 | sh_id << 24 of
   the CU that ran workgroup i */
int azg_debug_placement(int n_workgroups, uint32_t* out_dev, void* stream);

/* debug / tests / plugin benches: the integer hash-net of SURVEY.md Appendix C.3 as a leaf evaluator on the device (the deterministic stand-in
   for NeuralNet.predict, NeuralNet.py:32-43, that the MCTS parity tests run on both sides): boards int8[T][S], valid u8[T][A] ->
   pi f32[T][A], v f32[T][P], bit-identical to tests/hashnet.py.  Not a product net. */
int azg_eval_hashnet(const int8_t* boards, const uint8_t* valid, int T, int S, int A, int P, float* pi, float* v, void* stream);

/* tests: the ASYNCHRONOUS TREE PIPELINE (include/azg.h azg_forest_async_rounds_v80_h2) with that hash-net as its evaluator instead of an
   engine net -- the persistent descent kernel of the forest's game + a persistent evaluator kernel, same queues, same in-kernel advance --
   so that the pipeline itself can play the oracle's episodes and the episodes the reference's Coach.executeEpisode played
   (Coach.py:37-84,117-144).  For Splendor 2 - 4 players, Santorini without gods and Azul.  Arguments as for azg_forest_async_rounds_v80_h2
   without the weights.  Not a product path. */
int azg_forest_async_rounds_hashnet(azg_forest* f, uint8_t* leaf_valid_dev, uint8_t* needs_eval_dev, float* pi_dev, float* v_dev, int noise_stride,
                                    int rounds, int n_net, int n_sel, int batch_wait_ticks, int shared_budget, void* stream);

/* placement study: one row of four u64 per workgroup of the pipeline (the n_sel descent workgroups first): where it ran (XCC id | cu_id
   << 8 | se_id << 16 | sh_id << 24), role (1 descent, 2 net), calls (descents / forwards) and the shader cycles spent in them since
   the last reset.  Returns the number of rows written (<= max_wg). */
int azg_forest_async_wginfo(azg_forest* f, unsigned long long* out_host, int max_wg, int reset);
/* post-mortem of a pipeline time-out (error bit 128).  out280: what the descent wave that gave up saw of its workgroup (libraries built with
   -DAZG_ASYNC_POSTMORTEM only, zeros otherwise): [0] claimed mask (trees 0..63), [1] seen mask, [2] stop | scout << 32, [3] workgroup | wave << 16 |
   trees << 24 | retired << 32, [4 + k] snapshot ready word | last consumed word << 32 of its tree k (k < 128), [132 + k] ready word as the kernel
   read it | (status | err << 8) << 32; [280 + 2 b], [281 + 2 b]: the ticket range net workgroup b held when it left (first ticket | taken mask << 32,
   valid | last batch size << 32).  ready_out (or null): the ready words [workgroup][128] as they are in memory after the launch; ring_out (or
   null): the leaf ring.  Returns the ring's log2 size (0: the forest never ran the pipeline). */
int azg_forest_async_debug(azg_forest* f, unsigned long long* out792_host, uint32_t* ready_out_host, int max_wg, unsigned long long* ring_out_host, int max_ring);

#ifdef __cplusplus
}
#endif
#endif
