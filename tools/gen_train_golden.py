"""Trainer golden vectors from the REFERENCE's own GenericNNetWrapper.train (GenericNNetWrapper.py:44-92,179-190) and
on-disk format fixtures from the reference's own Coach.saveTrainExamples / loadTrainExamples (Coach.py:220-262).
Build-container only (needs /root/reference):  python tools/gen_train_golden.py

  tests/golden/train_splendor2_v80.npz   64 examples (board, pi, z, valids, q), the hyper-parameters, the (pi loss, v loss) of
        each of the two AdamW + OneCycleLR steps the reference took from pretrained_2players.pt (batch = all 64 examples, so the
        reference's np.random.choice(..., replace=False) only permutes the batch), and a few of the trained tensors.
  tests/golden/ref_checkpoint.examples   a `checkpoint.examples` file written by the reference's Coach.saveTrainExamples
        (two iterations, zlib-compressed 5-tuples) -- the reader side of azg_amd.formats is tested against it.
  It also checks, live, that the reference's Coach.loadTrainExamples reads a file written by azg_amd.formats.save_train_examples
  and that the reference's trainer consumes it (prints PASS / raises)."""
import os
import pickle
import sys
import tempfile
import zlib
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..')
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
HP = dict(learn_rate=1e-3, epochs=2, batch_size=64, q_weight=0.5)
KEEP = ['first_layer.linear.weight', 'first_layer.norm.running_mean', 'trunk.0.se.fc1.bias', 'output_layers_PI.4.bias',
        'output_layers_V.2.weight', 'output_layers_V.4.weight', 'output_layers_V.4.bias']


def make_examples(game, env, n, rng):
    shape = tuple(env['shape'])
    live = np.flatnonzero(~env['ended'].any(axis=1))
    sel = rng.choice(live, size=n, replace=False)
    ex = []
    for i in sel:
        b = env['canonical'][i].reshape(shape).astype(np.int8)
        va = game.getValidMoves(b, 0).astype(bool)
        pi = (rng.random(len(va)) ** 3 * va).astype(np.float32)
        pi /= pi.sum()
        z0 = float(rng.choice([-1.0, 1.0]))
        q0 = float(rng.uniform(-0.5, 0.5))
        ex.append((b, pi, np.array([z0, -z0], dtype=np.float32), va, np.array([q0, -q0], dtype=np.float32)))
    return ex


def main():
    import torch
    m = H.load_reference(splendor_players=2)
    import importlib
    NNet = importlib.import_module('splendor.NNet')
    Coach = importlib.import_module('Coach')
    game = m['SplendorGame'].SplendorGame()
    env = np.load(os.path.join(GOLDEN, 'env_splendor2.npz'))
    rng = np.random.default_rng(2024)
    examples = make_examples(game, env, HP['batch_size'], rng)

    nn_args = dict(lr=HP['learn_rate'], learn_rate=HP['learn_rate'], dropout=0., epochs=HP['epochs'], batch_size=HP['batch_size'],
                   nn_version=80, q_weight=HP['q_weight'], no_compression=True)
    w = NNet.NNetWrapper(game, nn_args)
    ck = w.load_checkpoint(os.path.join(H.REFERENCE, 'splendor'), 'pretrained_2players.pt')
    assert ck is not None and not w.requestKnowledgeTransfer
    losses = []
    lp0, lv0 = w.loss_pi, w.loss_v

    def rec_pi(t, o):
        r = lp0(t, o)
        losses.append(['pi', float(r)])
        return r

    def rec_v(tv, tq, o):
        r = lv0(tv, tq, o)
        losses.append(['v', float(r)])
        return r
    w.loss_pi, w.loss_v = rec_pi, rec_v
    np.random.seed(0)
    torch.manual_seed(0)
    w.train(examples)
    lpi = [x[1] for x in losses if x[0] == 'pi']
    lv = [x[1] for x in losses if x[0] == 'v']
    assert len(lpi) == 2 and len(lv) == 2
    sd = w.nnet.state_dict()
    out = dict(boards=np.stack([e[0] for e in examples]), pi=np.stack([e[1] for e in examples]),
               z=np.stack([e[2] for e in examples]), valids=np.stack([e[3] for e in examples]).astype(np.uint8),
               q=np.stack([e[4] for e in examples]), loss_pi=np.array(lpi), loss_v=np.array(lv),
               **{'hp/' + k: np.array(v) for k, v in HP.items()}, **{'after/' + k: sd[k].numpy() for k in KEEP})
    np.savez_compressed(os.path.join(GOLDEN, 'train_splendor2_v80.npz'), **out)
    print('reference trainer: pi losses', lpi, 'v losses', lv)

    # ---- on-disk format, reference writer -> fixture ----
    class A(dict):
        __getattr__ = dict.get
    tmp = tempfile.mkdtemp(prefix='azg_fmt_')
    c = Coach.Coach.__new__(Coach.Coach)
    c.args = A(checkpoint=tmp, load_folder_file=os.path.join(tmp, 'best.pt'), no_compression=False, numItersHistory=5,
               maxlenOfQueue=10 ** 6, useray=True)
    small = examples[:6]
    c.trainExamplesHistory = [deque([zlib.compress(pickle.dumps(e), level=1) for e in small[:4]], maxlen=10 ** 6),
                              deque([zlib.compress(pickle.dumps(e), level=1) for e in small[4:]], maxlen=10 ** 6)]
    c.saveTrainExamples()
    data = open(os.path.join(tmp, 'checkpoint.examples'), 'rb').read()
    open(os.path.join(GOLDEN, 'ref_checkpoint.examples'), 'wb').write(data)
    print('reference Coach.saveTrainExamples ->', len(data), 'bytes')

    # ---- live check: engine writer -> reference reader + reference trainer ----
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location('azg_formats', os.path.join(ROOT, 'alpha-zero-general_amd', 'formats.py'))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    cols = [np.stack([e[k].reshape(-1) for e in examples]) for k in range(5)]
    hist = [F.examples_to_iteration(cols, tuple(game.getBoardSize()), compress=True, maxlen=10 ** 6)]
    tmp2 = tempfile.mkdtemp(prefix='azg_fmt2_')
    F.save_train_examples(os.path.join(tmp2, 'checkpoint.examples'), hist)
    c2 = Coach.Coach.__new__(Coach.Coach)
    c2.args = A(checkpoint=tmp2, load_folder_file=os.path.join(tmp2, 'best.pt'), no_compression=True, numItersHistory=5,
                maxlenOfQueue=10 ** 6, useray=True)
    c2.loadTrainExamples()                                    # the reference's own reader (decompresses: no_compression=True)
    got = c2.trainExamplesHistory
    assert len(got) == 1 and len(got[0]) == len(examples) and type(got[0][0]) is tuple
    for a, b in zip(got[0], examples):
        assert all(np.array_equal(x, y) and np.asarray(x).dtype == np.asarray(y).dtype for x, y in zip(a, b))
    w2 = NNet.NNetWrapper(game, nn_args)
    w2.load_checkpoint(os.path.join(H.REFERENCE, 'splendor'), 'pretrained_2players.pt')
    np.random.seed(0)
    w2.train(list(got[0]))                                    # and the reference's trainer consumes them
    assert all(torch.equal(w2.nnet.state_dict()[k], sd[k]) for k in KEEP), 'same examples through the file -> same training'
    print('PASS: reference Coach.loadTrainExamples + GenericNNetWrapper.train read a file written by azg_amd.formats')
    H.cleanup()


if __name__ == '__main__':
    main()
