/* ORACLE (test infrastructure).  Abalone env step: a scalar restatement of abalone/AbaloneLogicNumba.py (Board :166-440),
 * INITIAL_LAYOUT = 1 (Belgian Daisy), ENABLE_DYNAMIC_KOMI = False (the shipped constants, :5-6).
 *
 * State = int8 [9][9][4] (axial grid; plane 0 my marbles, 1 opponent's, 2 board mask (4 <= r + q <= 12), 3 misc with
 * misc[0][0..2] = score of player 0, score of player 1, move counter); byte index = (r*9 + q)*4 + plane.
 * Action = r*378 + q*42 + plane, plane 0..5: one marble in direction d; 6 + 6*axis + d: two marbles along `axis`;
 * 24 + 6*axis + d: three (:66-88).  Deterministic: make_move ignores random_seed. */
#include <string.h>
#include "azg_oracle.h"

static const int DR[6] = {0, 1, 1, 0, -1, -1}, DQ[6] = {1, 0, -1, -1, 0, 1};       /* DIRECTIONS :58-65 */
#define CELL(s, r, q, z) ((s)[(((r) * 9 + (q)) << 2) + (z)])

static int on_board(const int8_t* s, int r, int q) {                               /* is_on_board :90-94 */
    return r >= 0 && r < 9 && q >= 0 && q < 9 && CELL(s, r, q, 2) == 1;
}
static void decode(int a, int* r, int* q, int* size, int* axis, int* d) {           /* _decode_action :76-88 */
    const int plane = a % 42;
    *q = (a / 42) % 9; *r = a / 378; *d = plane % 6;
    if (plane < 6) { *size = 1; *axis = 0; }
    else if (plane < 24) { *size = 2; *axis = (plane - 6) / 6; }
    else { *size = 3; *axis = (plane - 24) / 6; }
}
static int encode(int r, int q, int size, int axis, int d) {                        /* _encode_action :67-74 */
    const int plane = size == 1 ? d : (size == 2 ? 6 + axis * 6 + d : 24 + axis * 6 + d);
    return r * 378 + q * 42 + plane;
}

/* Board.valid_moves restricted to one action (:271-356) */
static int valid_action(const int8_t* s, int a, int player) {
    int r, q, size, axis, d;
    const int opp = 1 - player;
    decode(a, &r, &q, &size, &axis, &d);
    if (CELL(s, r, q, player) == 0) return 0;
    if (size == 1) {
        const int nr = r + DR[d], nq = q + DQ[d];
        return on_board(s, nr, nq) && CELL(s, nr, nq, player) == 0 && CELL(s, nr, nq, opp) == 0;
    }
    const int r1 = r + DR[axis], q1 = q + DQ[axis];
    if (!on_board(s, r1, q1) || CELL(s, r1, q1, player) == 0) return 0;
    if (size == 3) {
        const int r2 = r1 + DR[axis], q2 = q1 + DQ[axis];
        if (!(on_board(s, r2, q2) && CELL(s, r2, q2, player) == 1)) return 0;
    }
    const int inline_mv = d == axis || d == (axis + 3) % 6;
    if (!inline_mv) {                                                              /* broadside :300-309 */
        for (int i = 0; i < size; i++) {
            const int tr = r + i * DR[axis] + DR[d], tq = q + i * DQ[axis] + DQ[d];
            if (!on_board(s, tr, tq) || CELL(s, tr, tq, player) == 1 || CELL(s, tr, tq, opp) == 1) return 0;
        }
        return 1;
    }
    const int fr = d == axis ? r + (size - 1) * DR[axis] : r, fq = d == axis ? q + (size - 1) * DQ[axis] : q;
    const int tr = fr + DR[d], tq = fq + DQ[d];
    if (!on_board(s, tr, tq)) return 0;
    if (CELL(s, tr, tq, player) == 1) return 0;
    if (CELL(s, tr, tq, opp) == 0) return 1;
    int opp_count = 0, cr = tr, cq = tq;                                            /* sumito :328-352 */
    for (;;) {
        if (!on_board(s, cr, cq)) return opp_count > 0;
        if (CELL(s, cr, cq, opp) == 1) {
            opp_count++;
            if (opp_count >= size) return 0;
            cr += DR[d]; cq += DQ[d];
        } else if (CELL(s, cr, cq, player) == 1) return 0;
        else return 1;
    }
}

void abalone_valid_moves(const azo_game* g, const int8_t* s, int player, uint8_t* out) {
    for (int a = 0; a < g->A; a++) out[a] = (uint8_t)valid_action(s, a, player);
}

int abalone_make_move(const azo_game* g, int8_t* s, int move, int player, int64_t seed, azo_rng* rng) {       /* :358-396 */
    int r, q, size, axis, d;
    const int opp = 1 - player;
    (void)g; (void)seed; (void)rng;
    decode(move, &r, &q, &size, &axis, &d);
    const int inline_mv = d == axis || d == (axis + 3) % 6;
    if (size == 1 || !inline_mv) {
        for (int i = 0; i < size; i++) {
            const int cr = size > 1 ? r + i * DR[axis] : r, cq = size > 1 ? q + i * DQ[axis] : q;
            CELL(s, cr, cq, player) = 0;
            CELL(s, cr + DR[d], cq + DQ[d], player) = 1;
        }
    } else {
        int fr, fq, br, bq;
        if (d == axis) { fr = r + (size - 1) * DR[axis]; fq = q + (size - 1) * DQ[axis]; br = r; bq = q; }
        else { fr = r; fq = q; br = r + (size - 1) * DR[axis]; bq = q + (size - 1) * DQ[axis]; }
        const int tr = fr + DR[d], tq = fq + DQ[d];
        if (on_board(s, tr, tq) && CELL(s, tr, tq, opp) == 1) {
            int cr = tr, cq = tq;
            while (on_board(s, cr, cq) && CELL(s, cr, cq, opp) == 1) { cr += DR[d]; cq += DQ[d]; }
            CELL(s, tr, tq, opp) = 0;
            if (on_board(s, cr, cq)) CELL(s, cr, cq, opp) = 1;
            else CELL(s, 0, player, 3) = (int8_t)(CELL(s, 0, player, 3) + 1);
        }
        CELL(s, br, bq, player) = 0;
        CELL(s, tr, tq, player) = 1;
    }
    CELL(s, 0, 2, 3) = (int8_t)(CELL(s, 0, 2, 3) + 1);
    return 1 - player;
}

void abalone_game_ended(const azo_game* g, const int8_t* s, int next_player, float* out) {                     /* :398-413 */
    (void)g; (void)next_player;
    const int s0 = CELL(s, 0, 0, 3), s1 = CELL(s, 0, 1, 3);
    out[0] = out[1] = 0.f;
    if (s0 >= 6) { out[0] = 1.f; out[1] = -1.f; return; }
    if (s1 >= 6) { out[0] = -1.f; out[1] = 1.f; return; }
    if (CELL(s, 0, 2, 3) >= 127) {
        if (s0 > s1) { out[0] = 1.f; out[1] = -1.f; }
        else if (s1 > s0) { out[0] = -1.f; out[1] = 1.f; }
        else { out[0] = out[1] = 0.001f; }
    }
}

void abalone_swap_players(const azo_game* g, int8_t* s, int k) {                                               /* :415-426 */
    (void)g;
    if (k % 2 != 1) return;
    for (int c = 0; c < 81; c++) { const int8_t t = s[4 * c]; s[4 * c] = s[4 * c + 1]; s[4 * c + 1] = t; }
    const int8_t t = CELL(s, 0, 0, 3); CELL(s, 0, 0, 3) = CELL(s, 0, 1, 3); CELL(s, 0, 1, 3) = t;
}

int abalone_get_round(const azo_game* g, const int8_t* s) { (void)g; return CELL(s, 0, 2, 3); }
int abalone_get_score(const azo_game* g, const int8_t* s, int p) { (void)g; return p == 0 ? CELL(s, 0, 0, 3) : CELL(s, 0, 1, 3); }

void abalone_init_board(const azo_game* g, int8_t* s, azo_rng* rng) {                                          /* :175-222, layout 1 */
    (void)rng;
    memset(s, 0, (size_t)g->S);
    for (int r = 0; r < 9; r++)
        for (int q = 0; q < 9; q++)
            if (r + q >= 4 && r + q <= 12) CELL(s, r, q, 2) = 1;
    static const int opp_rows[6][3] = {{0, 4, 6}, {1, 3, 6}, {2, 3, 5}, {6, 4, 6}, {7, 3, 6}, {8, 3, 5}};
    static const int my_rows[6][3] = {{0, 7, 9}, {1, 6, 9}, {2, 6, 8}, {6, 1, 3}, {7, 0, 3}, {8, 0, 2}};
    for (int i = 0; i < 6; i++) {
        for (int q = opp_rows[i][1]; q < opp_rows[i][2]; q++) CELL(s, opp_rows[i][0], q, 1) = 1;
        for (int q = my_rows[i][1]; q < my_rows[i][2]; q++) CELL(s, my_rows[i][0], q, 0) = 1;
    }
}

/* the symmetry `rot` (0..5 clockwise 60-degree turns) after `flip` of one cell (:112-121, 431-438) */
static void sym_cell(int rot, int flip, int r, int q, int* nr, int* nq) {
    if (flip) q = 12 - r - q;
    for (int k = 0; k < rot; k++) { const int a = q + r - 4, b = 8 - r; r = a; q = b; }
    *nr = r; *nq = q;
}
/* ACTION_SYMMETRIES[rot][flip][a] (:99-146) */
static int sym_action(int rot, int flip, int a) {
    static const int FLIPD[6] = {3, 2, 1, 0, 5, 4};
    int r, q, size, axis, d, mr[3], mq[3];
    decode(a, &r, &q, &size, &axis, &d);
    for (int i = 0; i < size; i++) sym_cell(rot, flip, r + i * DR[axis], q + i * DQ[axis], &mr[i], &mq[i]);
    int mi = 0;
    for (int i = 1; i < size; i++)
        if (mr[i] < mr[mi] || (mr[i] == mr[mi] && mq[i] < mq[mi])) mi = i;
    int new_axis = 0;
    if (size > 1) {
        const int oi = mi == 0 ? 1 : 0, dr = mr[oi] - mr[mi], dq = mq[oi] - mq[mi];
        if (dr == 0 && dq > 0) new_axis = 0;
        else if (dr > 0 && dq == 0) new_axis = 1;
        else if (dr > 0 && dq < 0) new_axis = 2;
    }
    int nd = flip ? FLIPD[d] : d;
    nd = (nd + rot) % 6;
    return encode(mr[mi], mq[mi], size, new_axis, nd);
}

int abalone_symmetries(const azo_game* g, const int8_t* s, const float* pi, const uint8_t* valids, int8_t* os, float* op,
                       uint8_t* ov, int max_sym) {                                                             /* :428-460 */
    int k = 0;
    for (int rot = 0; rot < 6 && k < max_sym; rot++)
        for (int flip = 0; flip < 2 && k < max_sym; flip++, k++) {
            int8_t* ns = os + (size_t)k * g->S;
            float* np_ = op + (size_t)k * g->A;
            uint8_t* nv = ov + (size_t)k * g->A;
            memset(ns, 0, (size_t)g->S);
            memset(np_, 0, sizeof(float) * (size_t)g->A);
            memset(nv, 0, (size_t)g->A);
            for (int r = 0; r < 9; r++)
                for (int q = 0; q < 9; q++)
                    if (CELL(s, r, q, 2) == 1) {
                        int nr, nq;
                        sym_cell(rot, flip, r, q, &nr, &nq);
                        for (int z = 0; z < 4; z++) CELL(ns, nr, nq, z) = CELL(s, r, q, z);
                    }
            for (int c = 0; c < 81; c++) ns[4 * c + 3] = s[4 * c + 3];              /* misc layer untransformed */
            for (int a = 0; a < g->A; a++)
                if (valids[a]) {
                    const int m = sym_action(rot, flip, a);
                    np_[m] = pi[a];
                    nv[m] = valids[a];
                }
        }
    return k;
}
