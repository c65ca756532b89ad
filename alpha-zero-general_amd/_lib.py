"""ctypes loader for libazg_hip.so (the C-ABI declared in include/azg.h).

Fails loudly when the HIP extension is missing: there is NO CPU / eager fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AZG_LIB: A/B-test another build of the same library (tools/archive/ab_bench.sh); never a different implementation
LIB_PATH = os.environ.get('AZG_LIB') or os.path.join(HERE, 'libazg_hip.so')

SPLENDOR, SANTORINI, AZUL, MINIVILLES, ABALONE, TLP, BOTANIK, AKROPOLIS, SMALLWORLD = 0, 1, 2, 3, 4, 5, 6, 7, 8


class ForestCfg(C.Structure):
    _fields_ = [('game', C.c_int), ('variant', C.c_int), ('n_trees', C.c_int), ('node_capacity', C.c_int),
                ('row_capacity_bytes', C.c_int), ('numMCTSSims', C.c_int), ('cpuct', C.c_double), ('fpu', C.c_double),
                ('universes', C.c_int), ('prob_fullMCTS', C.c_double), ('ratio_fullMCTS', C.c_int),
                ('forced_playouts', C.c_int), ('dirichletAlpha', C.c_double), ('temperature', C.c_double * 3),
                ('tempThreshold', C.c_double), ('rng_seed', C.c_uint64), ('stream0', C.c_uint64),
                ('max_examples', C.c_int), ('level_budget', C.c_int), ('work_budget', C.c_int), ('gc_high_water_pct', C.c_int)]


class SelfplayStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ('plies', 'games', 'sims', 'levels', 'expansions', 'sum_valid_visited',
                                          'terminal_hits', 'examples', 'gc_runs', 'max_nodes', 'errors',
                                          'sum_depth_at_expand', 'cyc_select', 'cyc_levels', 'cyc_edge', 'cyc_leaf')] + [
                                          ('cyc_seg', C.c_uint64 * 4), ('max_live_after_gc', C.c_uint64), ('examples_dropped', C.c_uint64)]


class AzgError(RuntimeError):
    pass


_lib = None

EXPORTS = [
    'azg_last_error', 'azg_version', 'azg_device_count', 'azg_set_device', 'azg_game_info', 'azg_env_valid_moves',
    'azg_env_next_state', 'azg_env_game_ended', 'azg_env_canonical', 'azg_env_init_boards', 'azg_env_symmetries', 'azg_env_symmetries_ex', 'azg_debug_poison_onchip', 'azg_stream_create_xcd', 'azg_stream_destroy', 'azg_debug_placement', 'azg_eval_hashnet', 'azg_forest_create',
    'azg_forest_destroy', 'azg_forest_device_bytes', 'azg_forest_reset', 'azg_forest_begin_search',
    'azg_forest_select', 'azg_forest_select_fused', 'azg_forest_rounds_v80_h2', 'azg_forest_rounds_profile', 'azg_forest_expand_backup', 'azg_forest_active', 'azg_forest_action_probs',
    'azg_forest_root_stats', 'azg_forest_dump_tree', 'azg_forest_validate', 'azg_selfplay_start', 'azg_selfplay_start_ex', 'azg_selfplay_advance', 'azg_selfplay_active',
    'azg_selfplay_stats_get', 'azg_selfplay_drain_examples', 'azg_forest_last_kernel_ms', 'azg_forest_enable_timing', 'azg_forest_set_search_params', 'azg_nn_linear', 'azg_nn_linear_ws', 'azg_nn_dw_pool', 'azg_nn_v80_block', 'azg_nn_v80_forward', 'azg_nn_v80_forward_split', 'azg_nn_v80_forward_h2',
    'azg_nn_board_to_x', 'azg_nn_heads_out', 'azg_nn_dw_pool_l', 'azg_nn_board_to_x_ld', 'azg_nn_mb1d_forward', 'azg_nn_mb1d_forward_h2', 'azg_nn_conv5_forward', 'azg_nn_conv5_forward_split', 'azg_nn_conv5_forward_h2', 'azg_nn_s78_forward', 'azg_nn_s78_forward_split', 'azg_nn_s78_forward_h2',
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AzgError('HIP extension %s is missing: run `python __graft_entry__.py build` (hipcc, gfx950). '
                       'There is no CPU fallback.' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, u64, dbl = C.c_void_p, C.c_int, C.c_uint64, C.c_double
    ip = C.POINTER(C.c_int)
    L.azg_last_error.restype = C.c_char_p
    L.azg_version.restype = C.c_char_p
    # ABI check: azg_forest_cfg grew over the rounds; a library whose struct differs from this binding's would read garbage past the end
    # (an older build loaded through AZG_LIB for an A/B run has no azg_forest_cfg_size at all: it must then be a build of this tree's layout)
    if hasattr(L, 'azg_forest_cfg_size') and L.azg_forest_cfg_size() != C.sizeof(ForestCfg):
        raise AzgError('%s (%s) has a %d-byte azg_forest_cfg, this binding a %d-byte one: rebuild the library'
                       % (LIB_PATH, L.azg_version().decode(), L.azg_forest_cfg_size(), C.sizeof(ForestCfg)))
    L.azg_game_info.argtypes = [i, i, ip, ip, ip, ip, ip]
    L.azg_env_valid_moves.argtypes = [i, i, vp, vp, i, vp, vp]
    L.azg_env_next_state.argtypes = [i, i, vp, vp, vp, vp, i, vp, vp, u64, u64, vp, vp]
    L.azg_env_game_ended.argtypes = [i, i, vp, vp, i, vp, vp, vp, vp]
    L.azg_env_canonical.argtypes = [i, i, vp, vp, i, vp, vp]
    L.azg_env_init_boards.argtypes = [i, i, i, vp, u64, u64, vp, vp]
    L.azg_env_symmetries.argtypes = [i, i, vp, vp, vp, i, i, vp, vp, vp, vp, vp]
    L.azg_env_symmetries_ex.argtypes = [i, i, vp, vp, vp, i, i, vp, vp, vp, vp, u64, u64, vp]
    L.azg_forest_create.argtypes = [C.POINTER(ForestCfg), C.POINTER(vp)]
    L.azg_forest_destroy.argtypes = [vp]
    L.azg_forest_device_bytes.restype = C.c_size_t
    L.azg_forest_device_bytes.argtypes = [vp]
    L.azg_forest_reset.argtypes = [vp, vp]
    L.azg_forest_begin_search.argtypes = [vp, vp, vp, vp]
    L.azg_forest_select.argtypes = [vp, vp, vp, vp, vp, i, vp]
    L.azg_forest_select_fused.argtypes = [vp, vp, vp, vp, vp, vp, i, vp]
    L.azg_forest_expand_backup.argtypes = [vp, vp, vp, vp, i, vp]
    L.azg_forest_active.argtypes = [vp, ip]
    L.azg_forest_action_probs.argtypes = [vp, dbl, vp, vp, vp, vp]
    L.azg_forest_root_stats.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.azg_forest_dump_tree.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp]
    L.azg_forest_validate.argtypes = [vp, i]
    L.azg_selfplay_start.argtypes = [vp, vp, vp]
    L.azg_selfplay_start_ex.argtypes = [vp, vp, u64, C.c_int64, vp]
    L.azg_selfplay_advance.argtypes = [vp, vp]
    L.azg_selfplay_active.argtypes = [vp, ip]
    L.azg_selfplay_stats_get.argtypes = [vp, C.POINTER(SelfplayStats)]
    L.azg_selfplay_drain_examples.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, ip, vp]
    L.azg_forest_last_kernel_ms.argtypes = [vp, i, C.POINTER(dbl), C.POINTER(u64)]
    L.azg_forest_enable_timing.argtypes = [vp, i]
    L.azg_forest_rounds_v80_h2.argtypes = [vp, vp, vp, vp, vp, vp, i, vp, vp, i, vp]
    L.azg_forest_rounds_profile.argtypes = [vp, C.POINTER(C.c_double), i]
    if hasattr(L, 'azg_forest_async_rounds_v80_h2'):        # (absent from older builds loaded through AZG_LIB for A/B runs)
        L.azg_forest_async_rounds_v80_h2.argtypes = [vp, vp, vp, vp, vp, i, vp, vp, i, i, i, i, i, vp]
        if hasattr(L, 'azg_forest_async_rounds_mb1d_h2'):
            L.azg_forest_async_rounds_mb1d_h2.argtypes = [vp, i, vp, vp, vp, vp, i, vp, vp, i, i, i, i, i, vp]
        if hasattr(L, 'azg_forest_async_rounds_conv5_h2'):
            L.azg_forest_async_rounds_conv5_h2.argtypes = [vp, vp, vp, vp, vp, i, vp, C.c_float, i, i, i, i, i, vp]
        if hasattr(L, 'azg_forest_async_rounds_hashnet'):    # (test aid, include/azg_testaids.h)
            L.azg_forest_async_rounds_hashnet.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        L.azg_forest_async_profile.argtypes = [vp, C.POINTER(C.c_double), i]
        L.azg_forest_async_wginfo.argtypes = [vp, vp, i, i]
        if hasattr(L, 'azg_forest_async_debug'):
            L.azg_forest_async_debug.argtypes = [vp, vp, vp, i, vp, i]
    L.azg_stream_create_xcd.argtypes = [i, i, C.POINTER(vp)]
    L.azg_stream_destroy.argtypes = [vp]
    L.azg_debug_placement.argtypes = [i, vp, vp]
    if hasattr(L, 'azg_eval_hashnet'):
        L.azg_eval_hashnet.argtypes = [vp, vp, i, i, i, i, vp, vp, vp]
    if hasattr(L, 'azg_forest_set_search_params'):          # (absent from older builds loaded through AZG_LIB for A/B runs)
        L.azg_forest_set_search_params.argtypes = [vp, i, dbl]
    L.azg_nn_linear.argtypes = [vp, i, vp, i, i, vp, vp, i, vp, i, vp, i, i, i, i, i, i, vp]
    L.azg_nn_linear_ws.argtypes = [vp, i, vp, i, i, vp, vp, i, vp, i, vp, i, i, i, i, i, vp]
    L.azg_nn_v80_block.argtypes = [vp, vp, vp, i, i, i, vp]
    L.azg_nn_v80_forward.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp]
    L.azg_nn_v80_forward_split.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp]
    L.azg_nn_v80_forward_h2.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp]
    L.azg_nn_dw_pool.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, i, vp]
    L.azg_nn_board_to_x.argtypes = [vp, vp, i, i, vp]
    L.azg_nn_s78_forward.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp]
    L.azg_nn_s78_forward_split.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp]
    L.azg_nn_s78_forward_h2.argtypes = [vp, vp, vp, C.c_float, C.c_float, i, i, i, i, vp, vp, vp]
    L.azg_nn_conv5_forward.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp]
    L.azg_nn_conv5_forward_split.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp]
    L.azg_nn_conv5_forward_h2.argtypes = [vp, vp, vp, C.c_float, i, i, i, i, vp, vp, vp]
    L.azg_nn_mb1d_forward.argtypes = [i, vp, vp, vp, i, vp, vp, vp]
    L.azg_nn_mb1d_forward_h2.argtypes = [i, vp, vp, vp, vp, i, vp, vp, vp]
    L.azg_nn_board_to_x_ld.argtypes = [vp, vp, i, i, i, i, vp]
    L.azg_nn_dw_pool_l.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, i, i, vp]
    L.azg_nn_heads_out.argtypes = [vp, i, vp, vp, i, vp, vp, vp, vp, i, i, i, vp]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise AzgError(lib().azg_last_error().decode())
    return rc


def game_info(game, variant):
    S, A, P, rows, cols = (C.c_int() for _ in range(5))
    check(lib().azg_game_info(game, variant, C.byref(S), C.byref(A), C.byref(P), C.byref(rows), C.byref(cols)))
    return S.value, A.value, P.value, rows.value, cols.value
