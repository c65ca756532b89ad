"""Build-container-only harness that imports the reference (/root/reference) in pure-Python mode.

Used by tools/gen_golden.py to produce the committed fixtures under tests/golden/ and by
tools/check_oracle_vs_ref.py to pin the C oracle against the live reference.  Nothing here
travels to the GPU box as an executable dependency: -m gpu tests, smoke() and bench.py never import it.

What it does (SURVEY.md §8c recipe):
  * puts the stub `numba` / `colorama` / `torchvision` packages of this directory in front of sys.path
    (identity decorators == the reference's NUMBA_DISABLE_JIT=1 debug mode, README.md:169-171);
  * variants that are source-level constants in the reference (NUMBER_PLAYERS, NB_GODS, INIT_METHOD) are
    produced by editing a TEMP COPY of the reference tree (never the repo, never /root/reference);
  * patches splendor.SplendorLogicNumba.my_packbits to wrap uint8->int8 like Numba's silent int8 store
    (NumPy 2 raises OverflowError for 255 -> int8); widens azul's np_factory_symmetries table to int64 (Numba widens
    `30*(p+1)`, NumPy 2 overflows it in int8);
  * optional "numba typing" emulation for MCTS.pick_highest_UCB: Numba promotes float32 operands to
    float64 when mixed with float64 (cpuct, fpu are float64), while NumPy-2 scalar arithmetic in
    pure-Python mode keeps `python_float * np.float32` in float32.  `numba_typing=True` calls the
    reference's own function body with Ps widened to float64 and Qs as a Python float, which is
    value-identical to what the Numba-compiled code computes (minus fastmath reassociation).
"""
import importlib
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('AZG_REFERENCE', '/root/reference')

MAGIC_SEEDS = [31416, 1, 14142, 42, 27183, 2, 16180, 7]

_tmp_roots = []


def _purge_modules():
    for name in list(sys.modules):
        root = name.split('.')[0]
        if root in ('splendor', 'santorini', 'azul', 'minivilles', 'abalone', 'thelittleprince', 'botanik', 'akropolis', 'smallworld', 'MCTS', 'Game', 'Coach', 'Arena', 'utils', 'GameSwitcher',
                    'NeuralNet'):
            del sys.modules[name]


def load_reference(splendor_players=2, santorini_gods=11, santorini_init_method=1, minivilles_players=2, tlp_players=3, smallworld_players=2, akropolis_players=2):
    """Import the reference from a temp copy with the requested source-level variants.
    Returns a dict of modules."""
    tmp = tempfile.mkdtemp(prefix='azg_ref_')
    _tmp_roots.append(tmp)
    for name in os.listdir(REFERENCE):
        src = os.path.join(REFERENCE, name)
        if os.path.isdir(src):
            if name in ('splendor', 'santorini', 'azul', 'minivilles', 'abalone', 'thelittleprince', 'botanik', 'akropolis', 'smallworld'):
                shutil.copytree(src, os.path.join(tmp, name),
                                ignore=shutil.ignore_patterns('*.pt', '*.gif', '*.jpg', '*.png', '*.mp4', '*.csv',
                                                              '__pycache__'))
        elif name.endswith('.py'):
            shutil.copy(src, os.path.join(tmp, name))

    def sub(path, old, new):
        p = os.path.join(tmp, path)
        s = open(p).read()
        assert old in s, (path, old)
        open(p, 'w').write(s.replace(old, new, 1))

    sub('splendor/SplendorGame.py', 'NUMBER_PLAYERS = 2', 'NUMBER_PLAYERS = %d' % splendor_players)
    sub('santorini/SantoriniConstants.py', 'NB_GODS = 11', 'NB_GODS = %d' % santorini_gods)
    sub('santorini/SantoriniLogicNumba.py', 'INIT_METHOD = 1', 'INIT_METHOD = %d' % santorini_init_method)
    sub('minivilles/MinivillesGame.py', 'NUMBER_PLAYERS = 2', 'NUMBER_PLAYERS = %d' % minivilles_players)
    sub('thelittleprince/TLPGame.py', 'NUMBER_PLAYERS = 3', 'NUMBER_PLAYERS = %d' % tlp_players)
    sub('smallworld/SmallworldConstants.py', 'NUMBER_PLAYERS = 2', 'NUMBER_PLAYERS = %d' % smallworld_players)
    sub('akropolis/AkropolisConstants.py', 'N_PLAYERS = 2', 'N_PLAYERS = %d' % akropolis_players)

    _purge_modules()
    sys.dont_write_bytecode = True
    for p in (tmp, HERE):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, tmp)
    sys.path.insert(0, HERE)
    importlib.invalidate_caches()

    import splendor.SplendorLogicNumba as SL
    orig_pack = SL.my_packbits
    SL.my_packbits = lambda a: np.uint8(orig_pack(a)).view(np.int8)
    mods = {'root': tmp, 'SplendorLogicNumba': SL}
    mods['SplendorGame'] = importlib.import_module('splendor.SplendorGame')
    mods['SplendorLogic'] = importlib.import_module('splendor.SplendorLogic')
    mods['SantoriniGame'] = importlib.import_module('santorini.SantoriniGame')
    mods['SantoriniLogicNumba'] = importlib.import_module('santorini.SantoriniLogicNumba')
    mods['SantoriniConstants'] = importlib.import_module('santorini.SantoriniConstants')
    try:
        mods['AzulGame'] = importlib.import_module('azul.AzulGame')
        mods['AzulLogicNumba'] = importlib.import_module('azul.AzulLogicNumba')
        # Numba widens int8 * int64 in `30*(p+1)` (AzulLogicNumba.py:320); NumPy>=2 keeps python_int*np.int8 in int8 and
        # overflows for p >= 4.  Same values, wider dtype:
        mods['AzulLogicNumba'].np_factory_symmetries = mods['AzulLogicNumba'].np_factory_symmetries.astype(np.int64)
    except Exception as e:  # pragma: no cover
        mods['AzulGame'] = None
        mods['azul_error'] = e
    try:
        mods['MinivillesGame'] = importlib.import_module('minivilles.MinivillesGame')
        mods['MinivillesLogicNumba'] = importlib.import_module('minivilles.MinivillesLogicNumba')
    except Exception as e:  # pragma: no cover
        mods['MinivillesGame'] = None
        mods['minivilles_error'] = e
    try:
        mods['AbaloneGame'] = importlib.import_module('abalone.AbaloneGame')
        mods['AbaloneLogicNumba'] = importlib.import_module('abalone.AbaloneLogicNumba')
    except Exception as e:  # pragma: no cover
        mods['AbaloneGame'] = None
        mods['abalone_error'] = e
    try:
        mods['TLPGame'] = importlib.import_module('thelittleprince.TLPGame')
        mods['TLPLogicNumba'] = importlib.import_module('thelittleprince.TLPLogicNumba')
    except Exception as e:  # pragma: no cover
        mods['TLPGame'] = None
        mods['tlp_error'] = e
    try:
        mods['BotanikGame'] = importlib.import_module('botanik.BotanikGame')
        mods['BotanikLogicNumba'] = importlib.import_module('botanik.BotanikLogicNumba')
    except Exception as e:  # pragma: no cover
        mods['BotanikGame'] = None
        mods['botanik_error'] = e
    try:
        mods['AkropolisGame'] = importlib.import_module('akropolis.AkropolisGame')
        mods['AkropolisLogicNumba'] = importlib.import_module('akropolis.AkropolisLogicNumba')
        mods['AkropolisConstants'] = importlib.import_module('akropolis.AkropolisConstants')
    except Exception as e:  # pragma: no cover
        mods['AkropolisGame'] = None
        mods['akropolis_error'] = e
    try:
        mods['SmallworldGame'] = importlib.import_module('smallworld.SmallworldGame')
        mods['SmallworldLogicNumba'] = importlib.import_module('smallworld.SmallworldLogicNumba')
        mods['SmallworldConstants'] = importlib.import_module('smallworld.SmallworldConstants')
    except Exception as e:  # pragma: no cover
        mods['SmallworldGame'] = None
        mods['smallworld_error'] = e
    mods['MCTS'] = importlib.import_module('MCTS')
    mods['utils'] = importlib.import_module('utils')
    return mods


def enable_numba_typing(mcts_mod):
    """Make MCTS.pick_highest_UCB see the operand types Numba would see (see module docstring)."""
    if getattr(mcts_mod, '_azg_typed', False):
        return
    orig = mcts_mod.pick_highest_UCB

    def typed(Es, Vs, Ps, Ns, Qsa, Nsa, Qs, cpuct, forced_playouts, n_iter, fpu):
        return orig(Es, Vs, Ps.astype(np.float64), Ns, Qsa, Nsa, float(Qs), float(cpuct), forced_playouts, n_iter,
                    float(fpu))

    mcts_mod.pick_highest_UCB = typed
    mcts_mod._azg_typed = True


class UniformStream:
    """Deterministic replacement for np.random.random()/np.random.choice inside the reference's true-random
    paths (random_seed == 0), so they can be pinned: the reference consumes u ~ U[0,1) from this stream."""

    def __init__(self, values):
        self.values = list(values)
        self.pos = 0

    def random(self):
        v = self.values[self.pos]
        self.pos += 1
        return v


def _mix64(x):
    M = (1 << 64) - 1
    x &= M
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M
    x ^= x >> 31
    return x


class CounterRandom:
    """The engine's counter-based RNG contract (include/azg.h) as a drop-in for the reference's GLOBAL np.random on the paths
    that consume it inside the env step (Minivilles: np.random.randint for the dice, np.random.random for the purple cards):
    u01 = (mix64(mix64(mix64(seed ^ GOLD) + stream) + counter) >> 11) * 2^-53; randint(lo, hi) = lo + floor(u * (hi - lo)).
    `with CounterRandom(seed, stream) as r:` patches np.random.random / randint for the duration; r.counter = draws consumed,
    r.used = the uniforms in order."""

    def __init__(self, seed=0, stream=0, counter=0, injected=None):
        self.seed, self.stream, self.counter, self.used = seed, stream, counter, []
        self.injected = None if injected is None else list(injected)

    def random(self):
        if self.injected is not None:
            u = self.injected.pop(0) if self.injected else 0.5
        else:
            M = (1 << 64) - 1
            x = _mix64((_mix64((_mix64(self.seed ^ 0x9E3779B97F4A7C15) + self.stream) & M) + self.counter) & M)
            u = (x >> 11) * (1.0 / 9007199254740992.0)
        self.counter += 1
        self.used.append(u)
        return u

    def randint(self, lo, hi=None):
        if hi is None:
            lo, hi = 0, lo
        v = lo + int(self.random() * (hi - lo))
        return min(v, hi - 1)

    def choice(self, a, p=None):
        """np.random.choice(a) without weights as the contract defines it: a[floor(u * len(a))]"""
        assert p is None
        a = np.arange(a) if np.isscalar(a) else np.asarray(a)
        return a[min(int(self.random() * len(a)), len(a) - 1)]

    def shuffle(self, arr):
        """np.random.shuffle as the contract defines it: Fisher-Yates from the top, j = randint(0, i + 1) for i = n-1 .. 1
        (nothing is drawn for fewer than two elements)"""
        for i in range(len(arr) - 1, 0, -1):
            j = self.randint(0, i + 1)
            arr[i], arr[j] = arr[j], arr[i]

    def __enter__(self):
        self._orig = (np.random.random, np.random.randint, np.random.shuffle, np.random.choice)
        np.random.random, np.random.randint, np.random.shuffle, np.random.choice = self.random, self.randint, self.shuffle, self.choice
        return self

    def __exit__(self, *a):
        np.random.random, np.random.randint, np.random.shuffle, np.random.choice = self._orig


class HashNet:
    """Integer 'hash-net' of SURVEY.md Appendix C.3: bit-reproducible on any backend."""

    def __init__(self, num_players):
        self.P = num_players
        self.calls = 0

    def predict(self, board, valids):
        self.calls += 1
        flat = board.reshape(-1).astype(np.int64)
        s = int((flat * np.arange(1, flat.size + 1, dtype=np.int64)).sum())
        h = (s * 2654435761) % (1 << 32)
        v0 = np.float32(h / 2147483648.0 - 1.0)
        v = np.array([v0] + [np.float32(-v0 / (self.P - 1))] * (self.P - 1), dtype=np.float32)
        a = np.arange(len(valids), dtype=np.int64)
        w = np.asarray(valids).astype(np.int64) * (1 + (((h >> 8) + 2654435761 * a) % (1 << 32)) % 13)
        pi = (w / w.sum()).astype(np.float32)
        return pi, v


def mcts_args(utils_mod, **kw):
    base = dict(numMCTSSims=800, cpuct=1.0, fpu=0.0, universes=1, prob_fullMCTS=1.0, ratio_fullMCTS=5,
                forced_playouts=True, no_mem_optim=True, dirichletAlpha=0, temperature=[1.0, 1.0, 1.0],
                tempThreshold=10)
    base.update(kw)
    return utils_mod.dotdict(base)


def cleanup():
    for t in _tmp_roots:
        shutil.rmtree(t, ignore_errors=True)
    _tmp_roots.clear()
