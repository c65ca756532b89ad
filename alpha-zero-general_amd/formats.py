"""The reference's on-disk formats, so that engine output feeds the reference's trainer / tools and vice versa (host-side
data plumbing only, no compute):

* `checkpoint.examples` (Coach.saveTrainExamples / loadTrainExamples, Coach.py:215-262): `pickle.dump` of a list (one
  entry per iteration) of `deque(maxlen=maxlenOfQueue)` of examples; an example is the 5-tuple
  `(board int8 ndarray of getBoardSize(), pi, z = np.roll(result, -player), valids, q)` (Coach.py:76-82), stored either
  as is (`no_compression`) or as `zlib.compress(pickle.dumps(example), level=1)` (Coach.py:84).
* Arena initial-state string (Arena.py:61-65,91-93): `base64(raw-deflate(board bytes + player u8 + turn u16 big-endian))`.
* `.pt` checkpoints (GenericNNetWrapper.save_checkpoint :192-205): `torch.save({'state_dict', 'full_model', **args})`;
  reading one needs the reference's model classes importable because `full_model` is a pickled module."""
import base64
import pickle
import zlib
from collections import deque

import numpy as np


def examples_to_iteration(examples, board_shape, compress=True, maxlen=None):
    """(boards int8[n,S], pi f32[n,A], z f32[n,P], valids u8[n,A], q f32[n,P], ...) as returned by
    SelfPlayEngine.drain_examples() -> the deque of one self-play iteration in the reference's layout."""
    boards, pi, z, valids, q = [np.asarray(x.cpu() if hasattr(x, 'cpu') else x) for x in examples[:5]]
    out = deque([], maxlen=maxlen)
    for i in range(len(boards)):
        ex = (boards[i].reshape(board_shape).astype(np.int8), pi[i].astype(np.float32), z[i].astype(np.float32),
              valids[i].astype(bool), q[i].astype(np.float32))
        out.append(zlib.compress(pickle.dumps(ex), level=1) if compress else ex)
    return out


def save_train_examples(path, history):
    """history: list of iterations (deques) as built by examples_to_iteration; the file Coach.loadTrainExamples reads"""
    with open(path, 'wb') as f:
        pickle.dump(list(history), f)


def load_train_examples(path):
    """-> list of iterations, each a list of decompressed 5-tuples (board, pi, z, valids, q)"""
    with open(path, 'rb') as f:
        history = pickle.load(f)
    return [[ex if isinstance(ex, tuple) else pickle.loads(zlib.decompress(ex)) for ex in it] for it in history]


def encode_initial_state(board, player, turn):
    """the `state = "..."` string Arena prints in verbose mode and accepts as initial_state (Arena.py:61-65,91-93)"""
    data = np.ascontiguousarray(board, dtype=np.int8).tobytes() + int(player).to_bytes(1, 'big') + int(turn).to_bytes(2, 'big')
    c = zlib.compressobj(level=9, wbits=-15)
    return str(base64.b64encode(c.compress(data) + c.flush()), 'UTF-8')


def decode_initial_state(state, board_shape):
    """-> (board int8 ndarray, player, turn)"""
    data = zlib.decompress(base64.b64decode(state), wbits=-15)
    board = np.frombuffer(data[:-3], dtype=np.int8).reshape(board_shape).copy()
    return board, int(data[-3]), int.from_bytes(data[-2:], 'big')


def load_reference_checkpoint(path):
    """-> (state_dict of numpy arrays, dict of the scalar / list args embedded by save_checkpoint).  Needs the reference's
    package on sys.path (the pickled `full_model`); azg_amd.nnet.* accept the returned state_dict directly."""
    import torch
    ck = torch.load(path, map_location='cpu', weights_only=False)
    sd = {k: v.numpy() for k, v in ck['state_dict'].items()}
    args = {k: v for k, v in ck.items() if k not in ('state_dict', 'full_model')}
    return sd, args
