/* ORACLE (test infrastructure).  Counter-based RNG contract shared with the HIP engine (include/azg.h):
 *   mix64 = splitmix64 finaliser;  raw(seed, stream, counter) = mix64(mix64(mix64(seed ^ GOLD) + stream) + counter)
 *   u01   = (raw >> 11) * 2^-53
 * `stream` is the global game index, `counter` counts draws of that game.  The reference itself uses NumPy's /
 * Numba's global RNG on these paths (Coach.py:289-292, MCTS.py:43,58, SplendorLogicNumba.py:311-315), which is not
 * reproducible across processes, so the stream is ours; what is pinned is how each uniform is CONSUMED. */
#include "azg_oracle.h"

static uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

uint64_t azo_rng_raw(uint64_t seed, uint64_t stream, uint64_t counter) {
    return mix64(mix64(mix64(seed ^ 0x9E3779B97F4A7C15ULL) + stream) + counter);
}

double azo_rng_u01(azo_rng* r) {
    if (r->mode == 1) {
        double v = (r->pos < r->n_injected) ? r->injected[r->pos] : 0.5;
        r->pos++;
        return v;
    }
    uint64_t x = azo_rng_raw(r->seed, r->stream, r->counter++);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
