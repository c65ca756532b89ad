// Host-side dump of the compile-time step schedule of the Santorini ResNet kernel (csrc/nn_conv5x5.hip.h C5Steps): one line per
// step "rg nt s ky k6 tile kind last" for the three row-group instances the kernel runs.  Built and read by tests/test_conv5_schedule.py.
#include "azg_common.hip.h"
#include "nn_conv5x5.hip.h"
#include <cstdio>
using namespace azg;
template <int NT, int RG> static void dump() {
    constexpr auto& S = C5StepList<NT, RG, true>::value;
    printf("G %d %d %d %d %d\n", RG, NT, S.n, S.bias_at, S.n_tail);
    for (int i = 0; i < NT; i++) printf("T %d %d %d %d\n", RG, i, c5_tile_of<true>(RG, i), S.fin[i]);
    for (int s = 0; s < S.n; s++) printf("S %d %d %d %d %d %d %d %d\n", RG, s, S.ky[s], S.k6[s], S.ti[s], S.kind[s], S.last[s] ? 1 : 0, S.epi[s]);
    for (int j = 0; j < S.n_tail; j++) printf("L %d %d\n", RG, S.tail_order[j]);
}
int main() { dump<4, 0>(); dump<4, 1>(); dump<5, 2>(); return 0; }
