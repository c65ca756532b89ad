// azg.hip -- C-ABI (include/azg.h) of the MI355X self-play engine: host-side launch code for the gfx950 kernels.
// Forest, env and self-play kernels + their C-ABI; the net kernels live in azg_nn.hip (second translation unit of the same
// library, see build.py for the flags of each).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/azg.h"
#include "../../include/azg_testaids.h"
#include "game_splendor.hip.h"
#include "game_santorini.hip.h"
#include "game_azul.hip.h"
#include "game_minivilles.hip.h"
#include "game_abalone.hip.h"
#include "game_tlp.hip.h"
#include "game_botanik.hip.h"
#include "game_akropolis.hip.h"
#include "game_smallworld.hip.h"
#include "selfplay.hip.h"
#include "azg_host.h"

using namespace azg;

static thread_local std::string g_err;
int azg_fail(const std::string& m) { g_err = m; return -1; }

extern "C" const char* azg_last_error(void) { return g_err.c_str(); }
extern "C" const char* azg_version(void) { return "azg-hip r6 (gfx950)"; }
extern "C" int azg_forest_cfg_size(void) { return (int)sizeof(azg_forest_cfg); }      // ABI check of a binding against this build
extern "C" int azg_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int azg_set_device(int d) { HIPCHK(hipSetDevice(d)); return 0; }

// ---- game dispatch --------------------------------------------------------------------------------------------------
#define AZG_DISPATCH(game, variant, ...)                                                         \
    do {                                                                                           \
        if ((game) == AZG_SPLENDOR && (variant) == 2) { using G = SplendorDev<2>; __VA_ARGS__; }          \
        else if ((game) == AZG_SPLENDOR && (variant) == 3) { using G = SplendorDev<3>; __VA_ARGS__; }     \
        else if ((game) == AZG_SPLENDOR && (variant) == 4) { using G = SplendorDev<4>; __VA_ARGS__; }     \
        else if ((game) == AZG_SANTORINI && (variant) == 1) { using G = SantoriniDev<1>; __VA_ARGS__; }   \
        else if ((game) == AZG_SANTORINI && (variant) == 11) { using G = SantoriniDev<11>; __VA_ARGS__; } \
        else if ((game) == AZG_AZUL) { using G = AzulDev; __VA_ARGS__; }                                   \
        else if ((game) == AZG_ABALONE) { using G = AbaloneDev; __VA_ARGS__; }                             \
        else if ((game) == AZG_MINIVILLES && (variant) == 2) { using G = MinivillesDev<2>; __VA_ARGS__; } \
        else if ((game) == AZG_MINIVILLES && (variant) == 3) { using G = MinivillesDev<3>; __VA_ARGS__; } \
        else if ((game) == AZG_MINIVILLES && (variant) == 4) { using G = MinivillesDev<4>; __VA_ARGS__; } \
        else if ((game) == AZG_TLP && (variant) == 3) { using G = TLPDev<3>; __VA_ARGS__; }               \
        else if ((game) == AZG_TLP && (variant) == 4) { using G = TLPDev<4>; __VA_ARGS__; }               \
        else if ((game) == AZG_TLP && (variant) == 5) { using G = TLPDev<5>; __VA_ARGS__; }               \
        else if ((game) == AZG_BOTANIK) { using G = BotanikDev; __VA_ARGS__; }                             \
        else if ((game) == AZG_AKROPOLIS && (variant) == 2) { using G = AkropolisDev<2>; __VA_ARGS__; }   \
        else if ((game) == AZG_AKROPOLIS && (variant) == 3) { using G = AkropolisDev<3>; __VA_ARGS__; }   \
        else if ((game) == AZG_AKROPOLIS && (variant) == 4) { using G = AkropolisDev<4>; __VA_ARGS__; }   \
        else if ((game) == AZG_SMALLWORLD && (variant) == 2) { using G = SmallworldDev<2>; __VA_ARGS__; } \
        else if ((game) == AZG_SMALLWORLD && (variant) == 3) { using G = SmallworldDev<3>; __VA_ARGS__; } \
        else if ((game) == AZG_SMALLWORLD && (variant) == 4) { using G = SmallworldDev<4>; __VA_ARGS__; } \
        else return fail("unsupported game/variant");                                              \
    } while (0)

// games whose expanded nodes carry many more entries than the default heap sizing assumes name their typical count (REC_NV_HINT)
// ... or the AVERAGE record size in bytes over the nodes of a tree (REC_BYTES_HINT, measured), when the default -- room for 64 entries in
// every record, plus a quarter -- would make the heap twice what the records take (Azul: three nodes in four fit one 32-entry page)
template <class G, class = void> struct RecBytesHint { static constexpr int value = 0; };
template <class G> struct RecBytesHint<G, std::void_t<decltype(G::REC_BYTES_HINT)>> { static constexpr int value = G::REC_BYTES_HINT; };
template <class G, class = void> struct RecNvHint { static constexpr int value = 0; };
template <class G> struct RecNvHint<G, std::void_t<decltype(G::REC_NV_HINT)>> { static constexpr int value = G::REC_NV_HINT; };

static int norm_variant(int game, int variant) {
    if (game == AZG_SPLENDOR) return variant ? variant : 2;
    if (game == AZG_SANTORINI) return variant ? variant : 11;
    if (game == AZG_AZUL) return 2;
    if (game == AZG_MINIVILLES) return variant ? variant : 2;
    if (game == AZG_ABALONE) return 1;
    if (game == AZG_TLP) return variant ? variant : 3;
    if (game == AZG_BOTANIK) return 2;
    if (game == AZG_AKROPOLIS) return variant ? variant : 2;
    if (game == AZG_SMALLWORLD) return variant ? variant : 2;
    return variant;
}

extern "C" int azg_game_info(int game, int variant, int* S, int* A, int* P, int* rows, int* cols) {
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, {
        if (S) *S = G::S;
        if (A) *A = G::A;
        if (P) *P = G::P;
        if (rows) *rows = G::ROWS;
        if (cols) *cols = G::COLS;
    });
    return 0;
}

// ---- env kernels ----------------------------------------------------------------------------------------------------
extern "C" int azg_env_valid_moves(int game, int variant, const int8_t* states, const int32_t* players, int n,
                                   uint8_t* out, void* stream) {
    if (n <= 0) return 0;
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, k_env_valid_moves<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, players, n, out));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_next_state(int game, int variant, const int8_t* states, const int32_t* players,
                                  const int32_t* actions, const int64_t* seeds, int n, int8_t* out_states,
                                  int32_t* out_next, uint64_t rng_seed, uint64_t stream0, uint64_t* counters,
                                  void* stream) {
    if (n <= 0) return 0;
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant,
                 k_env_next_state<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, players,
                                     actions, (const int64_t*)seeds, n, out_states, out_next, rng_seed, stream0,
                                     counters));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_game_ended(int game, int variant, const int8_t* states, const int32_t* next_players, int n,
                                  float* out_ended, int32_t* out_scores, int32_t* out_round, void* stream) {
    if (n <= 0) return 0;
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, k_env_game_ended<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, next_players, n, out_ended, out_scores, out_round));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_canonical(int game, int variant, const int8_t* states, const int32_t* players, int n,
                                 int8_t* out_states, void* stream) {
    if (n <= 0) return 0;
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, k_env_canonical<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, players, n, out_states));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_init_boards(int game, int variant, int n, int8_t* out_states, uint64_t rng_seed,
                                   uint64_t stream0, uint64_t* out_counters, void* stream) {
    if (n <= 0) return 0;
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, k_env_init_boards<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(n,
                                                    out_states, rng_seed, stream0, out_counters));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_symmetries_ex(int game, int variant, const int8_t* states, const float* pi, const uint8_t* valids, int n,
                                     int max_sym, int8_t* out_states, float* out_pi, uint8_t* out_valids, int32_t* out_count,
                                     uint64_t rng_seed, uint64_t stream0, void* stream) {
    if (n <= 0) return 0;
    if (!states || !pi || !valids || !out_states || !out_pi || !out_valids || !out_count || max_sym <= 0)
        return fail("azg_env_symmetries: null/empty argument");
    variant = norm_variant(game, variant);
    AZG_DISPATCH(game, variant, {
        if constexpr (G::RANDOM_SYM)
            k_env_symmetries_built<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, pi, valids, n, max_sym, out_states,
                                                                                     out_pi, out_valids, out_count, rng_seed, stream0);
        else
            k_env_symmetries<G><<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(states, pi, valids, n, max_sym, out_states, out_pi,
                                                                              out_valids, out_count);
    });
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_env_symmetries(int game, int variant, const int8_t* states, const float* pi, const uint8_t* valids, int n,
                                  int max_sym, int8_t* out_states, float* out_pi, uint8_t* out_valids, int32_t* out_count,
                                  void* stream) {
    return azg_env_symmetries_ex(game, variant, states, pi, valids, n, max_sym, out_states, out_pi, out_valids, out_count, 0, 0, stream);
}

// debugging aid: leave a known pattern in the LDS of every CU and in the scratch (private) memory of the queue, so that a kernel that reads
// LDS or a local variable it never wrote changes its results with the pattern instead of with whatever ran before it
__global__ __launch_bounds__(1024) void k_debug_poison(uint32_t pattern, uint32_t* sink, int mode) {
    __shared__ uint32_t lds[16000];
    volatile uint32_t priv[512];
    if (mode & 1) for (int i = threadIdx.x; i < 16000; i += 1024) lds[i] = pattern;
    if (mode & 2) for (int i = 0; i < 512; i++) priv[i] = pattern;
    __syncthreads();
    uint32_t acc = lds[(threadIdx.x * 7) % 16000];
    for (int i = 0; i < 512; i += 37) acc ^= priv[i];
    if (acc == 0x12345u) sink[0] = acc;
}
extern "C" int azg_debug_poison_onchip(uint32_t pattern, void* stream) {
    static uint32_t* sink = nullptr;
    if (!sink) HIPCHK(hipMalloc(&sink, 64));
    const int mode = getenv("AZG_POISON_MODE") ? atoi(getenv("AZG_POISON_MODE")) : 3;      // 1 LDS, 2 scratch
    k_debug_poison<<<dim3(4096), dim3(1024), 0, (hipStream_t)stream>>>(pattern, sink, mode);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ---- XCD-pinned streams ---------------------------------------------------------------------------------------------
// MI355X = 8 XCDs x 32 CUs, one L2 per XCD.  A stream whose hardware queue is masked to the CUs of one XCD (or a few) keeps a group
// of trees, its leaf batch and its pi / v on one L2 and lets the groups run as independent pipelines (selfplay.py).  The CU mask of a
// queue is a bit vector over the device's CUs in the driver's enumeration, where consecutive bits go round-robin over the XCDs: bit b
// is CU b / n_xcd of XCD b % n_xcd (checked on the box by azg_debug_placement: every workgroup reports its XCC_ID).
extern "C" int azg_stream_create_xcd(int xcd_first, int xcd_count, void** out_stream) {
    if (!out_stream || xcd_count <= 0) return fail("azg_stream_create_xcd: bad arguments");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, dev));
    const int n_cu = pr.multiProcessorCount;
    const int n_xcd = getenv("AZG_N_XCD") ? atoi(getenv("AZG_N_XCD")) : 8;
    if (n_cu % n_xcd || xcd_first < 0 || xcd_first + xcd_count > n_xcd) return fail("azg_stream_create_xcd: XCD range does not fit the device");
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int b = 0; b < n_cu; b++) {
        const int x = b % n_xcd;
        if (x >= xcd_first && x < xcd_first + xcd_count) mask[(size_t)b / 32] |= 1u << (b % 32);
    }
    hipStream_t st = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *out_stream = (void*)st;
    return 0;
}
extern "C" int azg_stream_destroy(void* stream) {
    if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
// where did the workgroups of a launch on `stream` run?  out[i] = XCC_ID | HW_ID cu_id << 8 | se_id << 16 of workgroup i
__global__ __launch_bounds__(64) void k_debug_placement(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));              // HW_REG_XCC_ID
    asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw));               // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}                                           // keep the wave resident so that the launch spreads out
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xFu) | (((hw >> 8) & 0xFu) << 8) | (((hw >> 13) & 0x7u) << 16) | (((hw >> 12) & 1u) << 24);
}
extern "C" int azg_debug_placement(int n_workgroups, uint32_t* out_dev, void* stream) {
    if (n_workgroups <= 0) return 0;
    k_debug_placement<<<dim3(n_workgroups), dim3(64), 0, (hipStream_t)stream>>>(out_dev, 20000);
    HIPCHK(hipGetLastError());
    return 0;
}

// The integer hash-net of SURVEY.md Appendix C.3 on the device: the deterministic stand-in for NeuralNet.predict that the parity tests use
// on both sides (tests/hashnet.py is the same function as torch ops / numpy).  One wave per sample.  Not a product net: it lets the plugin
// benches time the tree + env side of a game without ~35 tiny torch kernels per round.
__global__ __launch_bounds__(64) void k_eval_hashnet(const int8_t* __restrict__ boards, const uint8_t* __restrict__ valid, int T, int S, int A, int P,
                                                     float* __restrict__ pi, float* __restrict__ v) {
    const int t = blockIdx.x, l = lane_id();
    if (t >= T) return;
    const int8_t* b = boards + (size_t)t * S;
    long long acc = 0;
    for (int i = l; i < S; i += 64) acc += (long long)b[i] * (long long)(i + 1);
    const uint64_t s = wave_sum_u64((uint64_t)acc);                               // (two's complement: the low 32 bits of the product are
    const uint32_t h = (uint32_t)(s * 2654435761ull);                              // torch.remainder(s * 2654435761, 2^32) for negative s too)
    const uint8_t* va = valid + (size_t)t * A;
    int wsum = 0;
    for (int a = l; a < A; a += 64) wsum += va[a] ? 1 + (int)(((h >> 8) + 2654435761u * (uint32_t)a) % 13u) : 0;
    wsum = wave_sum_i32(wsum);
    for (int a = l; a < A; a += 64) {
        const int w = va[a] ? 1 + (int)(((h >> 8) + 2654435761u * (uint32_t)a) % 13u) : 0;
        pi[(size_t)t * A + a] = (float)((double)w / (double)wsum);
    }
    if (l < P) {
        const float v0 = (float)((double)h / 2147483648.0 - 1.0);
        v[(size_t)t * P + l] = l == 0 ? v0 : (float)(-(double)v0 / (double)(P - 1));
    }
}
extern "C" int azg_eval_hashnet(const int8_t* boards, const uint8_t* valid, int T, int S, int A, int P, float* pi, float* v, void* stream) {
    if (!boards || !valid || !pi || !v || T <= 0 || S <= 0 || A <= 0 || P < 2) return fail("azg_eval_hashnet: bad argument");
    k_eval_hashnet<<<dim3(T), dim3(64), 0, (hipStream_t)stream>>>(boards, valid, T, S, A, P, pi, v);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- forest ---------------------------------------------------------------------------------------------------------
struct azg_forest {
    azg_forest_cfg cfg;
    ForestDev dev;
    int S, SP, A, P;
    size_t bytes;
    std::vector<void*> allocs;
    const uint8_t* last_leaf_valid = nullptr;   // valid masks written by the last azg_forest_select (read by expand_backup)
    struct Attached { std::string key; void* obj; void (*deleter)(void*); };
    std::vector<Attached> attached;             // azg_forest_attach (azg_host.h): round-kernel state owned by this forest
    // timing
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];
    size_t ev_used[2];
    double ms_total[2];
    uint64_t launches[2];
};

template <class T>
static int dalloc(azg_forest* f, T** p, size_t count) {
    size_t b = count * sizeof(T);
    if (b == 0) b = 16;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, b);
    if (e != hipSuccess) {
        (void)hipGetLastError();        // clear the sticky error: the caller may retry with a smaller forest
        return fail(std::string("hipMalloc(") + std::to_string(b) + "): " + hipGetErrorString(e));
    }
    // debugging aid: AZG_DEBUG_POISON = bit mask over the allocations in order (hdr 0, node_hdr 1, node_state 2, heap 3, htab 4, free_ids 5,
    // rec_free 6, path 7, root_state 8, board 9, rec_* 10..15, ex_* 16..21, ex_count 22) -> filled with 0xA5 before first use, so that a
    // read of memory the engine never wrote changes results instead of depending on what the allocator hands out
    static const unsigned long long poison = getenv("AZG_DEBUG_POISON") ? strtoull(getenv("AZG_DEBUG_POISON"), nullptr, 0) : 0ull;
    if ((poison >> f->allocs.size()) & 1ull) (void)hipMemset(q, 0xA5, b);
    f->allocs.push_back(q);
    f->bytes += b;
    *p = (T*)q;
    return 0;
}

extern "C" int azg_forest_create(const azg_forest_cfg* cfg, azg_forest** out) {
    if (!cfg || !out) return fail("null argument");
    azg_forest* f = new azg_forest();
    f->cfg = *cfg;
    f->cfg.variant = norm_variant(cfg->game, cfg->variant);
    f->bytes = 0;
    f->timing = false;
    f->ev_used[0] = f->ev_used[1] = 0;
    f->ms_total[0] = f->ms_total[1] = 0;
    f->launches[0] = f->launches[1] = 0;
    int game = f->cfg.game, variant = f->cfg.variant;
    int nv_hint = 0;                     // typical valid actions per expanded node (sizes the record heap of large action spaces)
    int rec_bytes_hint = 0;              // average record bytes per node (overrides the nv_hint sizing)
    int seed_actions = 1 << 30;          // RecGeom SE: leading action ids whose env step can read random_seed (forest.hip.h SeedActions)
    AZG_DISPATCH(game, variant, { f->S = G::S; f->SP = G::SP; f->A = G::A; f->P = G::P; nv_hint = RecNvHint<G>::value; rec_bytes_hint = RecBytesHint<G>::value; seed_actions = SeedActions<G>::value; });
    if (cfg->n_trees <= 0 || cfg->node_capacity < 16) { delete f; return fail("bad n_trees / node_capacity"); }
    if (cfg->node_capacity > (1 << AZG_IDX_BITS) - 2) { delete f; return fail("node_capacity too large"); }
    if (cfg->universes < 0 || cfg->universes > AZG_MAX_UNIVERSES) { delete f; return fail("universes out of range"); }
    if (cfg->numMCTSSims <= 0 || cfg->numMCTSSims >= (1 << 24)) { delete f; return fail("numMCTSSims must be in 1 .. 2^24 - 1"); }   // (kernels.hip.h: the forced-playout test)
    ForestDev& D = f->dev;
    memset(&D, 0, sizeof(D));
    D.T = cfg->n_trees;
    D.cap = cfg->node_capacity;
    int ht = 64;
    while (ht < 2 * D.cap) ht <<= 1;
    D.HT = ht;
    D.U = cfg->universes > 0 ? cfg->universes : 1;
    // record size classes (forest.hip.h ForestDev::cls_q): a small action space (Splendor, A = 81) gets ONE class for all
    // expanded nodes -- the heap is sized for cap records of the maximum size (+ cap entry-less records) and can never
    // fragment; larger action spaces use classes of 32 entries and a heap sized for the typical record
    D.cls_q = f->A <= 96 ? f->A : AZG_CLS_Q_MULTI;
    D.seed_entries = seed_actions < D.cls_q ? seed_actions : D.cls_q;
    D.id_bytes = f->A <= 256 ? 1 : 2;
    const RecGeom RG0(D.cls_q, D.U, D.seed_entries, D.id_bytes);
    size_t heap_bytes = cfg->row_capacity_bytes > 0
                            ? (size_t)cfg->row_capacity_bytes
                            : (D.cls_q == f->A ? (size_t)D.cap * RG0.total(f->A) + 4096
                                               : (rec_bytes_hint && D.cap >= 8192 && cfg->gc_high_water_pct > 0)
                                                     ? (size_t)D.cap * (size_t)rec_bytes_hint   // (an average: large arenas that are cleaned at a high-water mark only)
                                               : (size_t)D.cap * RG0.total(nv_hint ? nv_hint : (f->A <= 256 ? 64 : 160)) * 5 / 4) + 8192;
    heap_bytes = (heap_bytes + 15) / 16 * 16;
    if (heap_bytes / 16 > (size_t)AZG_CHILD_IDX_MASK) { delete f; return fail("record heap per tree exceeds the 29-bit record offset (8 GiB)"); }
    D.heap_units = (uint32_t)(heap_bytes / 16);
    auto skew = [](size_t bytes) { size_t r = (bytes + 255) / 256 * 256; return ((r >> 8) & 1) ? r : r + 256; };
    D.s_heap = skew(heap_bytes);
    D.s_nstate = skew((size_t)D.cap * f->SP);
    D.s_nhdr = skew((size_t)D.cap * sizeof(NodeHdr)) / sizeof(NodeHdr);
    D.s_htab = skew((size_t)D.HT * 4) / 4;
    D.s_free = skew((size_t)D.cap * 4) / 4;
    D.s_recfree = skew((size_t)(f->A + 2) * 4) / 4;
    D.universes = cfg->universes;
    D.numMCTSSims = cfg->numMCTSSims;
    D.ratio_fullMCTS = cfg->ratio_fullMCTS > 0 ? cfg->ratio_fullMCTS : 1;
    D.forced_playouts = cfg->forced_playouts;
    D.level_budget = cfg->level_budget;
    D.work_budget = cfg->work_budget;
    D.gc_high_water = cfg->gc_high_water_pct > 0 ? (uint32_t)((long long)cfg->node_capacity * cfg->gc_high_water_pct / 100) : 0u;
    { const char* e = getenv("AZG_SPEC_STATE"); D.spec_state = e ? atoi(e) : 8; }
    D.cpuct = cfg->cpuct; D.fpu = cfg->fpu; D.prob_fullMCTS = cfg->prob_fullMCTS;
    D.dirichletAlpha = cfg->dirichletAlpha;
    D.temp_begin = cfg->temperature[0]; D.temp_end = cfg->temperature[1]; D.temp_root = cfg->temperature[2];
    D.tempThreshold = cfg->tempThreshold;
    D.rng_seed = cfg->rng_seed; D.stream0 = cfg->stream0;
    D.max_examples = cfg->max_examples > 0 ? cfg->max_examples : 0;
    D.max_rec = f->cfg.game == AZG_SPLENDOR ? 64 * f->P + 8 : (f->cfg.game == AZG_MINIVILLES ? 320 : 256);   // Azul / Santorini: plies per
                                                                                     // game < 256 in practice; Minivilles: < 126 rounds + re-rolls
    const size_t T = D.T;
    int rc = 0;
    rc |= dalloc(f, &D.hdr, T);
    rc |= dalloc(f, &D.node_hdr, T * D.s_nhdr);
    rc |= dalloc(f, &D.node_state, T * D.s_nstate);
    rc |= dalloc(f, &D.heap, T * D.s_heap);
    rc |= dalloc(f, &D.htab, T * D.s_htab);
    rc |= dalloc(f, &D.free_ids, T * D.s_free);
    rc |= dalloc(f, &D.rec_free, T * D.s_recfree);
    rc |= dalloc(f, &D.path, T * AZG_MAXD);
    rc |= dalloc(f, &D.root_state, T * f->SP);
    rc |= dalloc(f, &D.board, T * f->SP);
    if (D.max_examples > 0) {
        const size_t R = T * D.max_rec, E = D.max_examples;
        rc |= dalloc(f, &D.rec_board, R * f->S);
        rc |= dalloc(f, &D.rec_pi, R * f->A);
        rc |= dalloc(f, &D.rec_valid, R * f->A);
        rc |= dalloc(f, &D.rec_q, R * f->P);
        rc |= dalloc(f, &D.rec_player, R);
        rc |= dalloc(f, &D.rec_ply, R);
        rc |= dalloc(f, &D.ex_board, E * f->S);
        rc |= dalloc(f, &D.ex_pi, E * f->A);
        rc |= dalloc(f, &D.ex_z, E * f->P);
        rc |= dalloc(f, &D.ex_valid, E * f->A);
        rc |= dalloc(f, &D.ex_q, E * f->P);
        rc |= dalloc(f, &D.ex_meta, E * 4);
    }
    rc |= dalloc(f, &D.ex_count, 4);
    if (rc) { azg_forest_destroy(f); return -1; }
    if (getenv("AZG_DEBUG_PTRS"))
        fprintf(stderr, "azg forest: hdr %p node_hdr %p node_state %p heap %p htab %p free_ids %p rec_free %p path %p root_state %p (strides: nhdr %zu nstate %zu heap %zu htab %zu)\n",
                (void*)D.hdr, (void*)D.node_hdr, (void*)D.node_state, (void*)D.heap, (void*)D.htab, (void*)D.free_ids, (void*)D.rec_free, (void*)D.path,
                (void*)D.root_state, (size_t)D.s_nhdr * sizeof(*D.node_hdr), (size_t)D.s_nstate, (size_t)D.s_heap, (size_t)D.s_htab * sizeof(*D.htab));
    hipError_t e = hipMemset(D.hdr, 0, T * sizeof(TreeHdr));
    if (e == hipSuccess) {
        const unsigned long long init[4] = {0ull, 0ull, 0ull, 0ull};
        e = hipMemcpy(D.ex_count, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(D.root_state, 0, T * f->SP);
    if (e == hipSuccess) e = hipMemset(D.board, 0, T * f->SP);
    if (e != hipSuccess) { azg_forest_destroy(f); return fail(hipGetErrorString(e)); }
    *out = f;
    return azg_forest_reset(f, nullptr);
}

void azg_forest_attach(azg_forest* f, const char* key, void* obj, void (*deleter)(void*)) { f->attached.push_back({key, obj, deleter}); }
void* azg_forest_attached(azg_forest* f, const char* key) {
    for (auto& a : f->attached) if (a.key == key) return a.obj;
    return nullptr;
}

extern "C" int azg_forest_destroy(azg_forest* f) {
    if (!f) return 0;
    if (!f->attached.empty()) (void)hipDeviceSynchronize();      // (a round kernel may still be reading its argument block)
    for (auto& a : f->attached) if (a.deleter) a.deleter(a.obj);
    for (void* p : f->allocs) (void)hipFree(p);
    for (int k = 0; k < 2; k++)
        for (auto& pr : f->ev[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete f;
    return 0;
}

extern "C" size_t azg_forest_device_bytes(const azg_forest* f) { return f ? f->bytes : 0; }

#define FDISPATCH(f, ...) AZG_DISPATCH((f)->cfg.game, (f)->cfg.variant, __VA_ARGS__)

const azg::ForestDev* azg_forest_dev_internal(azg_forest* f, int* game, int* variant, double* dirichlet_alpha) {
    if (game) *game = f->cfg.game;
    if (variant) *variant = f->cfg.variant;
    if (dirichlet_alpha) *dirichlet_alpha = f->cfg.dirichletAlpha;
    return &f->dev;
}

extern "C" int azg_forest_reset(azg_forest* f, void* stream) {
    if (!f) return fail("null forest");
    FDISPATCH(f, k_forest_reset<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_begin_search(azg_forest* f, const int8_t* roots, const uint8_t* full, void* stream) {
    if (!f || !roots) return fail("null argument");
    FDISPATCH(f, k_begin_search<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, roots,
                                     full));
    HIPCHK(hipGetLastError());
    return 0;
}

static void ev_begin(azg_forest* f, int which, hipStream_t s) {
    if (!f->timing) return;
    if (f->ev_used[which] >= f->ev[which].size()) return;
    (void)hipEventRecord(f->ev[which][f->ev_used[which]].first, s);
}
static void ev_end(azg_forest* f, int which, hipStream_t s) {
    if (!f->timing) return;
    if (f->ev_used[which] >= f->ev[which].size()) return;
    (void)hipEventRecord(f->ev[which][f->ev_used[which]].second, s);
    f->ev_used[which]++;
}

extern "C" int azg_forest_select(azg_forest* f, int8_t* leaf_states, uint8_t* leaf_valid, uint8_t* needs_eval,
                                 const double* root_noise, int noise_stride, void* stream) {
    if (!f || !leaf_states || !leaf_valid || !needs_eval) return fail("null argument");
    if (f->cfg.dirichletAlpha != 0.0 && (root_noise || noise_stride == -1))
        FDISPATCH(f, k_root_noise<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, root_noise, noise_stride));
    const int wait_noise = (f->cfg.dirichletAlpha != 0.0 && !root_noise && noise_stride == -2) ? 1 : 0;
    f->last_leaf_valid = leaf_valid;
    ev_begin(f, 0, (hipStream_t)stream);
    FDISPATCH(f, k_select<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, leaf_states,
                                     leaf_valid, needs_eval, wait_noise, nullptr, nullptr, 0));
    ev_end(f, 0, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_select_fused(azg_forest* f, int8_t* leaf_states, uint8_t* leaf_valid, uint8_t* needs_eval,
                                       const float* pi, const float* v, int noise_stride, void* stream) {
    if (!f || !leaf_states || !leaf_valid || !needs_eval || !pi || !v) return fail("null argument");
    if (noise_stride != 0 && noise_stride != -2) return fail("azg_forest_select_fused: noise_stride must be 0 or -2");
    const int noise = (f->cfg.dirichletAlpha != 0.0 && noise_stride == -2) ? 1 : 0;
    f->last_leaf_valid = leaf_valid;
    ev_begin(f, 0, (hipStream_t)stream);
    FDISPATCH(f, k_select<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, leaf_states,
                                     leaf_valid, needs_eval, noise, pi, v, noise));
    ev_end(f, 0, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_expand_backup(azg_forest* f, const float* pi, const float* v, const double* root_noise,
                                        int noise_stride, void* stream) {
    if (!f || !pi || !v) return fail("null argument");
    if (!f->last_leaf_valid) return fail("azg_forest_expand_backup: no preceding azg_forest_select");
    ev_begin(f, 1, (hipStream_t)stream);
    FDISPATCH(f, k_expand_backup<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, pi,
                                     v, f->last_leaf_valid, (root_noise || noise_stride == -1 || noise_stride == -2) ? 1 : 0));
    ev_end(f, 1, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_active(azg_forest* f, int* n_active) {
    if (!f || !n_active) return fail("null argument");
    std::vector<TreeHdr> h(f->dev.T);
    HIPCHK(hipMemcpy(h.data(), f->dev.hdr, sizeof(TreeHdr) * h.size(), hipMemcpyDeviceToHost));
    int n = 0;
    uint32_t err = 0;
    for (auto& x : h) { n += (x.status == ST_SEARCHING || x.status == ST_WAIT_NN); err |= x.err; }
    *n_active = n;
    if (err) return fail("forest error flags: " + std::to_string(err) +
                         " (1=node overflow 2=row-heap overflow 4=depth overflow 16/32=example overflow 64=all pruned root counts are 0)");
    return 0;
}

extern "C" int azg_forest_action_probs(azg_forest* f, double temp, double* probs, float* q, uint8_t* is_full,
                                       void* stream) {
    if (!f) return fail("null forest");
    FDISPATCH(f, k_action_probs<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, temp,
                                     probs, q, is_full));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_root_stats(azg_forest* f, int32_t* Ns, float* Qs, int32_t* Nsa, double* Qsa, float* Ps,
                                     int32_t* n_nodes, void* stream) {
    if (!f) return fail("null forest");
    FDISPATCH(f, k_root_stats<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, Ns, Qs,
                                     Nsa, Qsa, Ps, n_nodes));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_forest_dump_tree(azg_forest* f, int tree, int max_nodes, int8_t* states, int32_t* Ns, float* Qs,
                                    float* Es, int32_t* Nsa, double* Qsa, float* Ps, uint8_t* has_policy) {
    if (!f || tree < 0 || tree >= f->dev.T) return fail("bad tree index");
    HIPCHK(hipDeviceSynchronize());
    const ForestDev& D = f->dev;
    TreeHdr H;
    HIPCHK(hipMemcpy(&H, D.hdr + tree, sizeof(H), hipMemcpyDeviceToHost));
    const int n = (int)H.n_nodes, top = (int)H.id_top;
    if (n > max_nodes) return n;
    std::vector<NodeHdr> nh(top);
    std::vector<int8_t> st((size_t)top * f->SP);
    const RecGeom RG(D.cls_q, D.U, D.seed_entries, D.id_bytes);
    const size_t heap_used = D.cls_q == f->A ? (size_t)top * (RG.total(f->A) / 16u) : (size_t)H.heap_top;
    std::vector<uint8_t> hp(heap_used * 16);
    if (top) {
        HIPCHK(hipMemcpy(nh.data(), D.node_hdr + (size_t)tree * D.s_nhdr, sizeof(NodeHdr) * top, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(st.data(), D.node_state + (size_t)tree * D.s_nstate, (size_t)top * f->SP, hipMemcpyDeviceToHost));
    }
    if (!hp.empty()) HIPCHK(hipMemcpy(hp.data(), D.heap + (size_t)tree * D.s_heap, hp.size(), hipMemcpyDeviceToHost));
    const int A = f->A, P = f->P, S = f->S;
    int i = 0;
    for (int id = 0; id < top; id++) {                    // live nodes in id order (ids of dropped nodes are skipped)
        if (nh[id].flags & NF_FREE) continue;
        if (i >= n) return fail("dump_tree: more live headers than n_nodes");
        memcpy(states + (size_t)i * S, st.data() + (size_t)id * f->SP, S);
        const uint8_t* rec = hp.data() + (size_t)nh[id].rec_off * 16;
        const RecHdr* rh = (const RecHdr*)rec;
        Ns[i] = (int32_t)rh->Ns;
        Qs[i] = rh->Qs;
        for (int p = 0; p < P; p++) Es[(size_t)i * P + p] = (rh->flags & NF_TERMINAL) ? rec_es(*rh, p) : 0.f;
        has_policy[i] = (rh->flags & NF_EXPANDED) ? 1 : 0;
        for (int a = 0; a < A; a++) { Nsa[(size_t)i * A + a] = 0; Qsa[(size_t)i * A + a] = AZG_NANQ; Ps[(size_t)i * A + a] = 0.f; }
        if (has_policy[i]) {
            const RecIds ids(rec, RG);
            for (int j = 0; j < rh->nv; j++) {
                const uint8_t* ent = rec + RG.hot((uint32_t)j);
                const int a = ids[j];
                Nsa[(size_t)i * A + a] = (int32_t)*(const uint32_t*)(ent + AZG_H_N);
                Qsa[(size_t)i * A + a] = *(const double*)(ent + AZG_H_Q);
                Ps[(size_t)i * A + a] = *(const float*)(ent + AZG_H_P);
            }
        }
        i++;
    }
    return n;
}

// Structural validation of every tree on the HOST (debug / tests): returns the number of violated invariants.
extern "C" int azg_forest_validate(azg_forest* f, int verbose) {
    if (!f) return fail("null forest");
    HIPCHK(hipDeviceSynchronize());
    const ForestDev& D = f->dev;
    const size_t heap_bytes = (size_t)D.heap_units * 16;
    int bad = 0;
    std::vector<NodeHdr> nh(D.cap);
    std::vector<uint8_t> hp(heap_bytes);
    std::vector<uint32_t> tab(D.HT);
#define VBAD(...) do { bad++; if (verbose) fprintf(stderr, __VA_ARGS__); } while (0)
    for (int t = 0; t < D.T; t++) {
        TreeHdr H;
        HIPCHK(hipMemcpy(&H, D.hdr + t, sizeof(H), hipMemcpyDeviceToHost));
        const uint32_t n = H.n_nodes, top = H.id_top;
        if (top > (uint32_t)D.cap || n > top || H.heap_top > D.heap_units) { VBAD("[validate] t=%d n=%u id_top=%u heap_top=%u\n", t, n, top, H.heap_top); continue; }
        if (H.root != AZG_NONE && H.root >= top) VBAD("[validate] t=%d root=%u id_top=%u\n", t, H.root, top);
        HIPCHK(hipMemcpy(nh.data(), D.node_hdr + (size_t)t * D.s_nhdr, sizeof(NodeHdr) * top, hipMemcpyDeviceToHost));
        const RecGeom RG(D.cls_q, D.U, D.seed_entries, D.id_bytes);
        const uint32_t heap_used = D.cls_q == f->A ? (uint32_t)(top * (RG.total(f->A) / 16u)) : H.heap_top;
        HIPCHK(hipMemcpy(hp.data(), D.heap + (size_t)t * D.s_heap, (size_t)heap_used * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tab.data(), D.htab + (size_t)t * D.s_htab, sizeof(uint32_t) * D.HT, hipMemcpyDeviceToHost));
        if (H.root != AZG_NONE && H.root < top && nh[H.root].rec_off != H.root_rec)
            VBAD("[validate] t=%d root_rec=%u but node %u has rec_off=%u\n", t, H.root_rec, H.root, nh[H.root].rec_off);
        uint32_t live = 0, n_free = 0;
        for (uint32_t i = 0; i < top; i++) {
            if (nh[i].flags & NF_FREE) { n_free++; continue; }
            live++;
            const int cap_nv = (int)nh[i].nv * D.cls_q;                      // NodeHdr.nv = size class = pages of cls_q entries
            if (nh[i].rec_off + (AZG_REC_HDR + (uint32_t)nh[i].nv * RG.PAGE) / 16u > heap_used) { VBAD("[validate] t=%d node=%u record beyond heap_top %u (gc=%u)\n", t, i, H.heap_top, H.gc_runs); continue; }
            const uint8_t* rec = hp.data() + (size_t)nh[i].rec_off * 16;
            const RecHdr* rh = (const RecHdr*)rec;
            if (rh->node_id != i || (int)rh->nv > cap_nv || rh->round != nh[i].round)
                VBAD("[validate] t=%d node=%u header mismatch (rec node_id=%u nv=%u)\n", t, i, rh->node_id, rh->nv);
            if (!(rh->flags & NF_EXPANDED)) continue;
            const RecIds ids(rec, RG);
            for (int j = 0; j < rh->nv; j++) {
                for (int u = 0; u < D.U; u++) {
                    const uint32_t c = *(const uint32_t*)(rec + RG.child((uint32_t)j, (uint32_t)u));
                    if (c == AZG_NONE) continue;
                    const uint32_t cr = c & AZG_CHILD_IDX_MASK;
                    if (cr >= heap_used) { VBAD("[validate] t=%d node=%u child[%d][%d]=%08x beyond heap\n", t, i, j, u, c); continue; }
                    const RecHdr* ch = (const RecHdr*)(hp.data() + (size_t)cr * 16);
                    if (ch->node_id >= top || (nh[ch->node_id].flags & NF_FREE) || nh[ch->node_id].rec_off != cr)
                        VBAD("[validate] t=%d node=%u child[%d][%d] -> bad / dropped record %u\n", t, i, j, u, cr);
                }
                if (ids[j] >= f->A || (j && ids[j] <= ids[j - 1])) { VBAD("[validate] t=%d node=%u ids[%d]=%u\n", t, i, j, ids[j]); break; }
            }
        }
        if (live != n) VBAD("[validate] t=%d live headers %u != n_nodes %u\n", t, live, n);
        if (n_free != H.n_free_ids) VBAD("[validate] t=%d free headers %u != n_free_ids %u\n", t, n_free, H.n_free_ids);
        uint32_t cnt = 0;
        for (int k = 0; k < D.HT; k++)
            if (tab[k] != AZG_NONE) {
                cnt++;
                const uint32_t id = tab[k] & AZG_IDX_MASK;
                if (id >= top || (nh[id].flags & NF_FREE)) VBAD("[validate] t=%d htab[%d]=%08x -> free / out of range id\n", t, k, tab[k]);
                else if ((uint32_t)(nh[id].hash >> 54) != (tab[k] >> AZG_IDX_BITS)) VBAD("[validate] t=%d htab tag mismatch id=%u\n", t, id);
            }
        if (cnt != n) VBAD("[validate] t=%d htab entries %u != n %u\n", t, cnt, n);
    }
#undef VBAD
    return bad;
}

// ---- self-play ------------------------------------------------------------------------------------------------------
extern "C" int azg_selfplay_start_ex(azg_forest* f, const int8_t* init_boards, uint64_t epoch, int64_t episode_quota,
                                     void* stream) {
    if (!f) return fail("null forest");
    if (f->dev.max_examples <= 0) return fail("forest created with max_examples == 0");
    if (episode_quota < -1) return fail("azg_selfplay_start_ex: episode_quota < -1");
    // epoch 0 keeps the streams of the RNG contract; any other epoch re-keys every stream of this forest (boards, playout-cap
    // draws, root noise, move picks), so that successive self-play / arena waves of one run do not replay the same games
    // (seed and quota are kernel arguments: HIP graphs captured under other values must be captured again)
    f->dev.rng_seed = epoch ? f->cfg.rng_seed ^ (0x9E3779B97F4A7C15ULL * (epoch + 0x632BE59BD9B4E019ULL)) : f->cfg.rng_seed;
    if (episode_quota > 0xFFFFFFFEll) return fail("azg_selfplay_start_ex: episode_quota too large");
    f->dev.episode_quota = episode_quota < 0 ? 0xFFFFFFFFu : (uint32_t)episode_quota;      // -1: this forest plays no game at all (every tree idle)
    HIPCHK(hipMemsetAsync(f->dev.ex_count, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));
    FDISPATCH(f, k_selfplay_start<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev,
                                     init_boards));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int azg_selfplay_start(azg_forest* f, const int8_t* init_boards, void* stream) {
    return azg_selfplay_start_ex(f, init_boards, 0, 0, stream);
}

extern "C" int azg_selfplay_active(azg_forest* f, int* n_active) {
    if (!f || !n_active) return fail("null argument");
    std::vector<TreeHdr> h(f->dev.T);
    HIPCHK(hipMemcpy(h.data(), f->dev.hdr, sizeof(TreeHdr) * h.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (auto& x : h) n += (x.status != ST_IDLE && !x.err);
    *n_active = n;
    return 0;
}

extern "C" int azg_selfplay_advance(azg_forest* f, void* stream) {
    if (!f) return fail("null forest");
    static const bool dbg = getenv("AZG_DEBUG_SYNC") != nullptr;       // debugging aid: attribute a device fault to one kernel
    if (dbg) { fprintf(stderr, "[azg] k_selfplay_advance\n"); fflush(stderr); }
    FDISPATCH(f, k_selfplay_advance<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev));
    if (dbg) { (void)hipDeviceSynchronize(); fprintf(stderr, "[azg] k_gc\n"); fflush(stderr); }
    // trees whose arena ran short asked for the clean-up (16 waves per tree), then begin their search
    FDISPATCH(f, k_gc<G><<<dim3(f->dev.T), dim3(1024), 0, (hipStream_t)stream>>>(f->dev));
    if (dbg) { (void)hipDeviceSynchronize(); fprintf(stderr, "[azg] k_after_gc\n"); fflush(stderr); }
    FDISPATCH(f, k_after_gc<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev));
    // root Dirichlet noise (device sampler) of the searches that just began on an existing root and of the roots that simulation 0
    // expanded since the last advance (MCTS.py:64,147-149,156-160): ONE piece of code applies it, in its own kernel, so that the f64
    // pow / log / cos of the Gamma sampler weigh on no other kernel's registers
    if (f->cfg.dirichletAlpha != 0.0)
        FDISPATCH(f, k_root_noise<G><<<dim3(f->dev.T), dim3(64), 0, (hipStream_t)stream>>>(f->dev, nullptr, -1));
    if (dbg) { (void)hipDeviceSynchronize(); fprintf(stderr, "[azg] advance done\n"); fflush(stderr); }
    HIPCHK(hipGetLastError());
    return 0;
}

#ifdef AZG_CYC_COUNTERS
// debug builds only: one 64-bit cycle counter of every tree (which: 0 cyc_select, 1 cyc_levels, 2 cyc_edge, 3 cyc_leaf)
extern "C" int azg_debug_tree_cycles(azg_forest* f, int which, uint64_t* out /* host [T] */) {
    HIPCHK(hipDeviceSynchronize());
    std::vector<TreeHdr> h(f->dev.T);
    HIPCHK(hipMemcpy(h.data(), f->dev.hdr, sizeof(TreeHdr) * h.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i++)
        out[i] = which == 0 ? h[i].cyc_select : which == 1 ? h[i].cyc_levels : which == 2 ? h[i].cyc_edge :
                 which == 3 ? h[i].cyc_leaf : which == 4 ? h[i].c_levels : which == 5 ? h[i].pad0_ : which == 10 ? h[i].pad1_ :
                 h[i].cyc_seg[which - 6];
    return 0;
}
#endif

extern "C" int azg_selfplay_stats_get(azg_forest* f, azg_selfplay_stats* out) {
    if (!f || !out) return fail("null argument");
    HIPCHK(hipDeviceSynchronize());
    std::vector<TreeHdr> h(f->dev.T);
    HIPCHK(hipMemcpy(h.data(), f->dev.hdr, sizeof(TreeHdr) * h.size(), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    for (auto& x : h) {
        out->plies += x.c_plies; out->games += x.games_done; out->sims += x.c_sims; out->levels += x.c_levels;
        out->expansions += x.c_exp; out->sum_valid_visited += x.c_sumvalid; out->terminal_hits += x.c_term;
        out->examples += x.c_examples; out->gc_runs += x.gc_runs; out->errors |= x.err;
        out->sum_depth_at_expand += x.c_depth;
        out->cyc_select += x.cyc_select; out->cyc_levels += x.cyc_levels; out->cyc_edge += x.cyc_edge; out->cyc_leaf += x.cyc_leaf;
        for (int k = 0; k < 4; k++) out->cyc_seg[k] += x.cyc_seg[k];
        if (x.max_nodes_seen > out->max_nodes) out->max_nodes = x.max_nodes_seen;
        if (x.max_live > out->max_live_after_gc) out->max_live_after_gc = x.max_live;
    }
    unsigned long long cnt[4];
    HIPCHK(hipMemcpy(cnt, f->dev.ex_count, sizeof(cnt), hipMemcpyDeviceToHost));
    out->examples_dropped = cnt[1];
    if (cnt[1]) out->errors |= ERR_EXAMPLE_OVERFLOW;              // finished games that did not fit the example ring
    return 0;
}

extern "C" int azg_selfplay_drain_examples(azg_forest* f, int max_records, int8_t* boards, float* pi, float* z,
                                           uint8_t* valids, float* q, int32_t* meta, int* n_out, void* stream) {
    if (!f || !n_out) return fail("null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));
    unsigned long long cnt[2];
    HIPCHK(hipMemcpy(cnt, f->dev.ex_count, sizeof(cnt), hipMemcpyDeviceToHost));
    size_t n = cnt[0] < (unsigned long long)f->dev.max_examples ? (size_t)cnt[0] : (size_t)f->dev.max_examples;
    if ((size_t)max_records < n) n = (size_t)max_records;
    const ForestDev& D = f->dev;
    if (n) {
        if (boards) HIPCHK(hipMemcpyAsync(boards, D.ex_board, n * f->S, hipMemcpyDeviceToDevice, s));
        if (pi) HIPCHK(hipMemcpyAsync(pi, D.ex_pi, n * f->A * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (z) HIPCHK(hipMemcpyAsync(z, D.ex_z, n * f->P * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (valids) HIPCHK(hipMemcpyAsync(valids, D.ex_valid, n * f->A, hipMemcpyDeviceToDevice, s));
        if (q) HIPCHK(hipMemcpyAsync(q, D.ex_q, n * f->P * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (meta) HIPCHK(hipMemcpyAsync(meta, D.ex_meta, n * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipMemsetAsync(D.ex_count, 0, 2 * sizeof(unsigned long long), s));
    HIPCHK(hipStreamSynchronize(s));
    *n_out = (int)n;
    return 0;
}

// ---- measurement ----------------------------------------------------------------------------------------------------
// Change the search size / playout-cap probability of a forest between launches (MCTS.py:58-59 reads args.numMCTSSims and
// args.prob_fullMCTS at every getActionProb call, so the reference's args may change between moves as well).  Both are kernel
// arguments: HIP graphs that captured this forest's launches must be captured again.  Searches in flight keep their n_sims.
extern "C" int azg_forest_set_search_params(azg_forest* f, int numMCTSSims, double prob_fullMCTS) {
    if (!f) return fail("null forest");
    if (numMCTSSims <= 0 || numMCTSSims >= (1 << 24) || !(prob_fullMCTS >= 0.0 && prob_fullMCTS <= 1.0)) return fail("azg_forest_set_search_params: bad argument");
    f->dev.numMCTSSims = numMCTSSims;
    f->dev.prob_fullMCTS = prob_fullMCTS;
    return 0;
}

extern "C" int azg_forest_enable_timing(azg_forest* f, int enable) {
    if (!f) return fail("null forest");
    if (enable) {
        for (int k = 0; k < 2; k++) {
            while (f->ev[k].size() < 4096) {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a));
                HIPCHK(hipEventCreate(&b));
                f->ev[k].push_back({a, b});
            }
            f->ev_used[k] = 0;
            f->ms_total[k] = 0;
            f->launches[k] = 0;
        }
    }
    f->timing = enable != 0;
    return 0;
}

extern "C" int azg_forest_last_kernel_ms(azg_forest* f, int which, double* avg_ms, uint64_t* launches) {
    if (!f || which < 0 || which > 1) return fail("bad argument");
    HIPCHK(hipDeviceSynchronize());
    for (size_t i = 0; i < f->ev_used[which]; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, f->ev[which][i].first, f->ev[which][i].second));
        f->ms_total[which] += ms;
        f->launches[which]++;
    }
    f->ev_used[which] = 0;
    if (avg_ms) *avg_ms = f->launches[which] ? f->ms_total[which] / (double)f->launches[which] : 0.0;
    if (launches) *launches = f->launches[which];
    return 0;
}

#ifdef AZG_CYC_COUNTERS
extern "C" int azg_debug_prolog(unsigned long long* out /* [16] */, int reset) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prolog), 16 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_prolog), z, sizeof(z))); }
    return 0;
}
#endif
