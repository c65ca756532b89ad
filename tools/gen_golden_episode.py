"""G5: full `Coach.executeEpisode` example lists produced by the REFERENCE's own Coach (Coach.py:37-84), with
temp_for_selfplay (:266-271), applyTemperatureAndNormalize / random_pick (:278-292), the playout-cap draw (MCTS.py:58) and the
tree clean-up (MCTS.py:86-91) -> tests/golden/episode_<variant>_<case>.npz.   Build-container only (imports /root/reference).

Every random number the reference consumes during the episode comes from the engine's counter stream (include/azg.h "RNG
contract"; harness.CounterRandom) in the REFERENCE's order of consumption:
  * `my_mcts.rng.random()` (MCTS.py:43,58; a numpy Generator in the reference) -> the stream's next uniform;
  * `np.random.choice(len(p), p=p)` (Coach.py:291) -> NumPy's own algorithm on the stream's next uniform u:
    `cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u, side='right')` (numpy/random/mtrand.pyx, legacy choice with p).
    The script ASSERTS that this formula is what the real `numpy.random.RandomState.choice` does, once per pick of every
    episode, with the episode's own p: the state of a side RandomState is saved, its next uniform peeked (`random_sample()`),
    the state restored, the real `choice(len(p), p=p)` called, and the formula evaluated on the peeked uniform must return
    the same index (`_check_choice_formula`);
  * the env step's true-random paths of `getNextState(random_seed=0)` (Coach.py:71) -> np.random.random / randint / choice
    patched as for the env fixtures (tools/gen_golden.py).
The initial board (`getInitBoard`, Numba's private RNG in the reference) is captured into the fixture and handed to the oracle /
engine as the episode's first board.

A fixture holds, per ply: canonical board, pi (the list getActionProb returned), q, is_full_search, the action picked, the
player to move; and the episode's result plus the FINAL example list exactly as executeEpisode returned it (no_compression):
(board, pi, z = roll(r, -player), valids, q) for every symmetry of every full-search ply, in order.
"""
import argparse
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402
from gen_golden import MCTS_ARGS, VARIANTS  # noqa: E402

GOLDEN = os.path.join(HERE, '..', 'tests', 'golden')

# (variant, case name, seed, stream, sims, prob_fullMCTS, temperature, tempThreshold)
CASES = [
    ('splendor2', 'full', 4242, 1000, 40, 1.0, [1.25, 0.8, 1.0], 6),
    ('splendor2', 'cap', 4242, 1001, 40, 0.5, [1.25, 0.8, 1.0], 6),
    ('santorini11', 'full', 4242, 1000, 40, 1.0, [1.25, 0.8, 1.0], 6),
    ('santorini11', 'capstep', 4242, 1003, 40, 0.5, [1.0, 0.3, 1.0], -4),      # tempThreshold < 0: the step schedule (:268-269)
    ('azul', 'full', 4242, 1000, 40, 1.0, [1.25, 0.8, 1.0], 10),
    ('azul', 'cap', 4242, 1002, 40, 0.5, [1.0, 0.1, 1.1], 10),                 # main.py's default temperatures
    # temperature == 0 after the third ply (step schedule with t_end = 0): applyTemperatureAndNormalize's np.random.choice(bests) and the
    # second np.random.choice(len(p), p=one-hot) of random_pick (Coach.py:278-292) -- two uniforms per pick
    ('splendor2', 'temp0', 4242, 1004, 40, 1.0, [1.0, 0.0, 1.0], -3),
]

_checked = [0]


def _choice_formula(p, u):
    p = np.asarray(p, dtype=np.float64)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    return int(cdf.searchsorted(u, side='right'))


def _check_choice_formula(p, rs):
    """the real numpy.random.RandomState.choice(len(p), p=p) against the formula, on the uniform the real call consumes"""
    st = rs.get_state()
    u = rs.random_sample()
    rs.set_state(st)
    real = int(rs.choice(len(p), p=p))
    assert real == _choice_formula(p, u), (real, u)
    _checked[0] += 1


class EpisodeRandom(H.CounterRandom):
    """CounterRandom + the p-weighted np.random.choice of random_pick (Coach.py:291); records every pick's uniform"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.pick_u = []
        self._rs = np.random.RandomState(12345)

    def choice(self, a, p=None):
        if p is None:
            return super().choice(a)
        assert np.isscalar(a)
        _check_choice_formula(np.asarray(p, dtype=np.float64), self._rs)
        u = self.random()
        self.pick_u.append(u)
        return _choice_formula(p, u)


class _Net:
    """hash-net behind the NeuralNet attributes Coach.__init__ touches (Coach.py:29,33)"""
    requestKnowledgeTransfer = False

    def __init__(self, game, args):
        self.args = args
        self.h = H.HashNet(game.num_players)

    def predict(self, board, valids):
        return self.h.predict(board, valids)


def gen_case(variant, case, seed, stream, sims, prob_full, temperature, temp_threshold):
    kw, modkey, cls = VARIANTS[variant]
    m = H.load_reference(**kw)
    H.enable_numba_typing(m['MCTS'])
    CoachMod = importlib.import_module('Coach')
    game = getattr(m[modkey], cls)()
    np.random.seed(777 + stream)
    init = game.getInitBoard().copy()                       # captured (Numba's private RNG in the real thing)
    margs = dict(MCTS_ARGS[variant])
    args = H.mcts_args(m['utils'], numMCTSSims=sims, prob_fullMCTS=prob_full, no_mem_optim=False, dirichletAlpha=0,
                       temperature=list(temperature), tempThreshold=temp_threshold, parallel_inferences=1, no_compression=True,
                       **margs)
    coach = CoachMod.Coach(game, _Net(game, args), args)
    mc = coach.mcts
    rec = dict(canonical=[], pi=[], q=[], full=[], action=[], player=[])

    class OneShotGame:
        """the game object executeEpisode drives: getInitBoard returns the captured board; everything else is the reference's"""

        def __init__(self, g):
            self.g = g

        def __getattr__(self, k):
            return getattr(self.g, k)

        def getInitBoard(self):
            return init.copy()

        def getNextState(self, board, player, action, random_seed=0):
            rec['action'].append(int(action))
            rec['player'].append(int(player))
            return self.g.getNextState(board, player, action, random_seed=random_seed)

    orig_gap = mc.getActionProb

    def gap(canonical, temp=1, force_full_search=False):
        probs, q, full = orig_gap(canonical, temp=temp, force_full_search=force_full_search)
        rec['canonical'].append(np.asarray(canonical, dtype=np.int8).reshape(-1).copy())
        rec['pi'].append(np.asarray(probs, dtype=np.float64))
        rec['q'].append(np.asarray(q, dtype=np.float32))
        rec['full'].append(int(bool(full)))
        return probs, q, full

    mc.getActionProb = gap
    with EpisodeRandom(seed, stream, 0) as R:
        mc.rng = R                                           # MCTS.py:43,58: `self.rng.random() < prob_fullMCTS`
        examples = coach.executeEpisode(mc, OneShotGame(game))
        n_draws, used, pick_u = R.counter, list(R.used), list(R.pick_u)
    n = len(rec['action'])
    assert len(rec['pi']) == n
    P, A = game.num_players, game.getActionSize()
    ex_board = np.array([np.asarray(e[0], dtype=np.int8).reshape(-1) for e in examples], dtype=np.int8).reshape(len(examples), -1)
    ex_pi = np.array([np.asarray(e[1], dtype=np.float64) for e in examples]).reshape(len(examples), A)
    ex_z = np.array([np.asarray(e[2], dtype=np.float32) for e in examples]).reshape(len(examples), P)
    ex_valid = np.array([np.asarray(e[3]).astype(np.uint8) for e in examples], dtype=np.uint8).reshape(len(examples), A)
    ex_q = np.array([np.asarray(e[4], dtype=np.float32) for e in examples]).reshape(len(examples), P)
    out = dict(init_board=init.reshape(-1).astype(np.int8), seed=np.array(seed, dtype=np.int64), stream=np.array(stream, dtype=np.int64),
               sims=np.array(sims), prob_full=np.array(prob_full), temperature=np.array(temperature, dtype=np.float64),
               tempThreshold=np.array(temp_threshold, dtype=np.float64),
               cpuct=np.array(margs['cpuct']), fpu=np.array(margs['fpu']), universes=np.array(margs['universes']),
               forced=np.array(int(margs['forced_playouts'])),
               canonical=np.array(rec['canonical'], dtype=np.int8), pi=np.array(rec['pi'], dtype=np.float64),
               q=np.array(rec['q'], dtype=np.float32), full=np.array(rec['full'], dtype=np.int8),
               action=np.array(rec['action'], dtype=np.int32), player=np.array(rec['player'], dtype=np.int8),
               n_draws=np.array(n_draws), uniforms=np.array(used, dtype=np.float64), pick_u=np.array(pick_u, dtype=np.float64),
               ex_board=ex_board, ex_pi=ex_pi, ex_z=ex_z, ex_valid=ex_valid, ex_q=ex_q,
               shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(P))
    H.cleanup()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    for c in CASES:
        if a.only and a.only not in (c[0], c[0] + '_' + c[1]):
            continue
        out = gen_case(*c)
        fn = os.path.join(GOLDEN, 'episode_%s_%s.npz' % (c[0], c[1]))
        np.savez_compressed(fn, **out)
        print(c[0], c[1], 'plies', len(out['action']), 'full', int(out['full'].sum()), 'examples', len(out['ex_board']),
              'draws', int(out['n_draws']), os.path.getsize(fn), 'bytes; choice formula checked', _checked[0], 'times')


if __name__ == '__main__':
    main()
