// game_minivilles.hip.h -- Minivilles (Machi Koro) env step on the device plugin interface:
// minivilles/MinivillesLogicNumba.py (Board :66-372), 2-4 players.
//
// State int8 [18 + 20 n][2]: column 0 = current, column 1 = the state before the last "real" move (the radio-tower re-roll
// restores it, :270-274).  Rows (copy_state :164-175): 0 round, 1 last dice, 2 player_state, 3..17 market[15], 18.. money[n],
// 18+n.. cards[n][15], 18+16n.. monuments[n][4].  21 actions: buy card 0..14, buy monument 15..18, re-roll 19, pass 20.
//
// STOCHASTIC = true: make_move ignores random_seed and draws true randomness -- the dice of the next player (:232-242) and the
// three choices of the purple cards (:49-52) -- inside MCTS simulations too (MCTS.py:238).  The forest therefore never memoises
// an edge of this game: every descent level replays the env step with fresh uniforms of the tree's counter stream (kernels.hip.h
// k_select), exactly like the reference, whose every traversal re-rolls.  Draws: np.random.randint(1, 6) -> 1 + floor(5 u)
// (NumPy's upper bound is exclusive: a die shows 1..5); my_random_choice_and_normalize(mask) = searchsorted(cumsum(mask), u,
// 'right') with u in [0, 1) = the FIRST set index, the uniform is consumed all the same.
//
// The rules are a few dozen dependent scalar operations on ~100 bytes: make_move / init run on lane 0 over the LDS state
// (lane0_make_move), the valid-move mask is one action per lane.
#pragma once
#include "azg_common.hip.h"

namespace azg {

template <int NP>
struct MinivillesDev {
    static constexpr int P = NP;
    static constexpr int ROWS = 18 + 20 * NP, COLS = 2;
    static constexpr int S = ROWS * COLS;
    static constexpr int SP = RoundUp16<S>::value;
    static constexpr int A = 21;
    static constexpr int AW = 1;
    static constexpr bool STOCHASTIC = true;
    static constexpr bool RANDOM_SYM = false;   // get_symmetries draws no randomness
    enum { CHAMPS, FERME, BOULANGERIE, CAFE, SUPERETTE, FORET, STADE, AFFAIRES, CHAINE, FROMAGERIE, MEUBLES, MINE, RESTAURANT,
           VERGER, MARCHE };
    enum { GARE, CENTRECOM, RADIO, PARC };
    enum { R_MARKET = 3, R_MONEY = 18, R_CARDS = 18 + NP, R_MONU = 18 + 16 * NP };

    __device__ static __forceinline__ int card_cost(int c) {                 // :391  {1,1,1,2,2,3,6,8,7,5,3,6,3,3,2}, 4 bits each
        return (int)((0x233635786322111ull >> (4 * c)) & 15ull);
    }
    __device__ static __forceinline__ int monu_cost(int m) { return 4 + 6 * m; }   // :392  {4, 10, 16, 22}
    __device__ static __forceinline__ int8_t& at(int8_t* st, int row, int col) { return st[2 * row + col]; }
    __device__ static __forceinline__ int at(const int8_t* st, int row, int col) { return st[2 * row + col]; }
    __device__ static __forceinline__ int cards(const int8_t* st, int p, int c) { return st[2 * (R_CARDS + 15 * p + c)]; }
    __device__ static __forceinline__ int monu(const int8_t* st, int p, int m) { return st[2 * (R_MONU + 4 * p + m)]; }

    __device__ static int get_score(const int8_t* st, int p) {                // :78-79
        int t = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) t += monu(st, p, m) * monu_cost(m);
        return t;
    }
    __device__ static __forceinline__ int get_round(const int8_t* st) { return st[0]; }
    // clean-up age: the round never decreases along a line of play (a re-roll restores the round of its own turn)
    __device__ static __forceinline__ int gc_age(const int8_t* st) { return (int)(uint8_t)st[0]; }
    __device__ static __forceinline__ bool move_uses_seed(int) { return true; }   // (unused: STOCHASTIC edges are never memoised)

    __device__ static int wealth(const int8_t* st, int p) {                   // :81-84
        const int w = get_score(st, p) + at(st, R_MONEY + p, 0);
        return w > 127 ? 127 : w;
    }
    __device__ static void add_money(int8_t* st, int p, int amount) {         // :355-361
        int v = at((const int8_t*)st, R_MONEY + p, 0) + amount;
        v = v > 127 ? 127 : (v < 0 ? 0 : v);
        at(st, R_MONEY + p, 0) = (int8_t)v;
    }
    __device__ static int rnd_int(Rng& rng, int lo, int hi) {
        const int v = lo + (int)(rng.u01() * (double)(hi - lo));
        return v >= hi ? hi - 1 : v;
    }
    __device__ static void all_receive(int8_t* st, int c, int money) {
        for (int p = 0; p < NP; p++) add_money(st, p, money * cards(st, p, c));
    }
    __device__ static void current_receive(int8_t* st, int who, int c, int money, bool mall) {
        const int bonus = (mall && monu(st, who, CENTRECOM) > 0) ? 1 : 0;
        add_money(st, who, (money + bonus) * cards(st, who, c));
    }
    __device__ static void current_give(int8_t* st, int who, int c, int money, bool mall) {      // :259-267, as written
        for (int pl = who + NP - 1; pl > who; pl--) {
            const int p = pl % NP;
            const int bonus = (mall && monu(st, p, CENTRECOM) > 0) ? 1 : 0;
            int amount = (money + bonus) * cards(st, p, c);
            const int have = at((const int8_t*)st, R_MONEY + who, 0);
            amount = have < amount ? have : amount;
            add_money(st, p, -amount);
            add_money(st, who, amount);
        }
    }

    __device__ static void dice_effect(int8_t* st, int result, int who, Rng& rng) {             // :244-353
        switch (result) {
        case 1: all_receive(st, CHAMPS, 1); break;
        case 2: all_receive(st, FERME, 1); current_receive(st, who, BOULANGERIE, 1, true); break;
        case 3: current_give(st, who, CAFE, 1, true); current_receive(st, who, BOULANGERIE, 1, true); break;
        case 4: current_receive(st, who, SUPERETTE, 3, true); break;
        case 5: all_receive(st, FORET, 1); break;
        case 6:
            if (cards(st, who, STADE) > 0) {                                                    // _stadium :269-278
                for (int p = 0; p < NP; p++) {
                    if (p == who) continue;
                    const int m = at((const int8_t*)st, R_MONEY + p, 0);
                    const int amount = m < 2 ? m : 2;
                    add_money(st, p, -amount);
                    add_money(st, who, amount);
                }
            }
            if (cards(st, who, AFFAIRES) > 0) {                                                 // _business_center :280-303
                int best = -128, target = 0;
                (void)rng.u01();
                for (int p = 0; p < NP; p++) {                                                  // first player with the maximum
                    const int w = p == who ? 0 : (int)(int8_t)wealth(st, p);
                    if (w > best) { best = w; target = p; }
                }
                int cmx = -128, tb = 0;
                (void)rng.u01();
                for (int c = 0; c < 15; c++) {
                    const int n = cards(st, target, c);
                    int cost = (n < 1 ? n : 1) * card_cost(c);
                    if (c == STADE || c == AFFAIRES || c == CHAINE) cost = 0;
                    if (cost > cmx) { cmx = cost; tb = c; }
                }
                int mmn = 127, mb = 0;
                (void)rng.u01();
                for (int c = 0; c < 15; c++) {
                    const int n = cards(st, who, c);
                    int cost = (n < 1 ? n : 1) * card_cost(c);
                    if (cost == 0) cost = 99;
                    if (cost < mmn) { mmn = cost; mb = c; }
                }
                at(st, R_CARDS + 15 * target + tb, 0) -= 1;
                at(st, R_CARDS + 15 * who + tb, 0) += 1;
                at(st, R_CARDS + 15 * who + mb, 0) -= 1;
                at(st, R_CARDS + 15 * target + mb, 0) += 1;
            }
            if (cards(st, who, CHAINE) > 0) {                                                   // _tv_channel :305-319
                int mx = -128;
                for (int p = 0; p < NP; p++) {
                    const int m = p == who ? 0 : at((const int8_t*)st, R_MONEY + p, 0);
                    mx = m > mx ? m : mx;
                }
                mx = mx > 5 ? 5 : mx;
                int wmx = -128, target = 0;
                (void)rng.u01();
                for (int p = 0; p < NP; p++) {
                    const int m = p == who ? 0 : at((const int8_t*)st, R_MONEY + p, 0);
                    const int w = (m == mx || m >= 5) ? (int)(int8_t)wealth(st, p) : 0;
                    if (w > wmx) { wmx = w; target = p; }
                }
                const int tm = at((const int8_t*)st, R_MONEY + target, 0);
                const int amount = tm < 5 ? tm : 5;
                add_money(st, target, -amount);
                add_money(st, who, amount);
            }
            break;
        case 7: current_receive(st, who, FROMAGERIE, 3 * cards(st, who, FERME), false); break;
        case 8: current_receive(st, who, MEUBLES, 3 * (cards(st, who, FORET) + cards(st, who, MINE)), false); break;
        case 9: current_give(st, who, RESTAURANT, 2, true); all_receive(st, MINE, 5); break;
        case 10: current_give(st, who, RESTAURANT, 2, true); all_receive(st, VERGER, 3); break;
        case 11: case 12: current_receive(st, who, MARCHE, 2 * (cards(st, who, CHAMPS) + cards(st, who, VERGER)), false); break;
        default: break;
        }
    }

    // _roll_dice + _dice_effect for `who` (:232-242, 151-153); returns whether two identical dice were rolled
    __device__ static bool roll_and_apply(int8_t* st, int who, Rng& rng) {
        int d = rnd_int(rng, 1, 6);
        bool identical = false;
        if (monu(st, who, GARE) > 0) {
            const int d2 = rnd_int(rng, 1, 6);
            identical = d == d2;
            d += d2;
        }
        at(st, 1, 0) = (int8_t)d;
        dice_effect(st, d, who, rng);
        return identical;
    }

    // Board.valid_moves restricted to one action (:104-110,220-247)
    __device__ static bool valid_action(const int8_t* st, int a, int player) {
        const int money = at(st, R_MONEY + player, 0);
        if (a < 15) {
            if ((a == STADE || a == AFFAIRES || a == CHAINE) && cards(st, player, a) > 0) return false;
            return money >= card_cost(a) && at(st, R_MARKET + a, 0) > 0;
        }
        if (a < 19) return money >= monu_cost(a - 15) && monu(st, player, a - 15) == 0;
        if (a == 19) return monu(st, player, 3) != 0 && (at(st, 2, 0) % 2 == 0);     // index 3 as written (:245-247)
        return true;
    }
    __device__ static void valid_mask(const int8_t* st, int player, uint64_t* mask_lds) {
        const int l = lane_id();
        const uint64_t m = __ballot(l < A && valid_action(st, l < A ? l : 20, player));
        if (l == 0) mask_lds[0] = m;
    }

    __device__ static __forceinline__ int wave_make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        return lane0_make_move<MinivillesDev<NP>>(st, move, player, seed, rng);
    }
    // Board.make_move :112-160 -- lane 0 only
    __device__ static int make_move(int8_t* st, int move, int player, long long seed, Rng& rng) {
        (void)seed;
        if (move < 15) {
            add_money(st, player, -card_cost(move));
            at(st, R_MARKET + move, 0) -= 1;
            at(st, R_CARDS + 15 * player + move, 0) += 1;
        } else if (move < 19) {
            add_money(st, player, -monu_cost(move - 15));
            at(st, R_MONU + 4 * player + (move - 15), 0) += 1;
        } else if (move == 19) {
            for (int r = R_MARKET; r < ROWS; r++) st[2 * r] = st[2 * r + 1];
            st[0] = st[1];
        }
        int next;
        if (move == 19) next = player;
        else if (at((const int8_t*)st, 2, 0) >= 2) { st[0] = (int8_t)(st[0] + 1); next = player; }
        else { st[0] = (int8_t)(st[0] + 1); next = (player + 1) % NP; }
        if (move != 19) {
            for (int r = R_MARKET; r < ROWS; r++) st[2 * r + 1] = st[2 * r];
            st[1] = st[0];
        }
        const bool identical = roll_and_apply(st, next, rng);
        at(st, 2, 0) = (int8_t)((move == 19 ? 1 : 0) + (identical ? 2 : 0));
        return next;
    }

    // Board.check_end_game :177-185 (uniform); all lanes compute the same few scalars
    __device__ static bool game_ended(const int8_t* st, int next_player, float* out, uint64_t* mask_scratch) {
        (void)next_player; (void)mask_scratch;
        int sc[NP], mx = -128, cnt = 0;
        bool rich = false;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            sc[p] = (int)(int8_t)get_score(st, p);
            mx = sc[p] > mx ? sc[p] : mx;
            rich = rich || at(st, R_MONEY + p, 0) >= 126;
        }
        if (mx < 52 && st[0] < 126 && !rich) {
#pragma unroll
            for (int p = 0; p < NP; p++) out[p] = 0.f;
            return false;
        }
#pragma unroll
        for (int p = 0; p < NP; p++) cnt += sc[p] == mx;
#pragma unroll
        for (int p = 0; p < NP; p++) out[p] = sc[p] == mx ? (cnt == 1 ? 1.f : 0.01f) : -1.f;
        return true;
    }

    // Board.swap_players :189-198: money / cards / monuments of player i come from player (i + k) mod n, both columns
    __device__ static void swap_players(int8_t* st, int8_t* tmp, int k) {
        for (int i = lane_id(); i < S; i += 64) tmp[i] = st[i];
        wave_sync();
        for (int i = lane_id(); i < S; i += 64) {
            const int r = i >> 1, c = i & 1;
            int src_row = r;
            if (r >= R_MONU) src_row = R_MONU + ((r - R_MONU) + 4 * k) % (4 * NP);
            else if (r >= R_CARDS) src_row = R_CARDS + ((r - R_CARDS) + 15 * k) % (15 * NP);
            else if (r >= R_MONEY) src_row = R_MONEY + ((r - R_MONEY) + k) % NP;
            st[i] = tmp[2 * src_row + c];
        }
        wave_sync();
    }

    // init_game :86-102 -- lane 0; state zeroed by the caller
    __device__ static void init_board(int8_t* st, Rng& rng) {
        for (int c = 0; c < 15; c++) { const int8_t v = (c >= 6 && c < 9) ? 4 : 6; st[2 * (R_MARKET + c)] = v; st[2 * (R_MARKET + c) + 1] = v; }
        for (int p = 0; p < NP; p++) {
            st[2 * (R_MONEY + p)] = st[2 * (R_MONEY + p) + 1] = 3;
            for (int c = 0; c < 2; c++) st[2 * (R_CARDS + 15 * p + c)] = st[2 * (R_CARDS + 15 * p + c) + 1] = 1;
        }
        (void)roll_and_apply(st, 0, rng);
    }

    // get_symmetries :200-202: the identity only
    static constexpr int NSYM_CAND = 1;
    __device__ static __forceinline__ bool sym_exists(const int8_t*, int) { return true; }
    __device__ static __forceinline__ int8_t sym_state_byte(const int8_t* st, int, int i) { return st[i]; }
    __device__ static __forceinline__ int sym_action_src(const int8_t*, int, int a) { return a; }
};

}  // namespace azg
