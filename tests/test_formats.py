"""CPU: the reference's on-disk formats (Coach.py:215-262 examples file, Arena.py:61-65 initial-state string)."""
import os
import pickle
import zlib

import numpy as np


def test_examples_file_roundtrip_and_reference_reader(tmp_path):
    from azg_amd import formats
    rng = np.random.default_rng(0)
    n, S, A, P = 7, 392, 81, 2
    ex = (rng.integers(-3, 5, (n, S)).astype(np.int8), rng.random((n, A)).astype(np.float32), rng.random((n, P)).astype(np.float32),
          rng.integers(0, 2, (n, A)).astype(np.uint8), rng.random((n, P)).astype(np.float32))
    for compress in (True, False):
        it = formats.examples_to_iteration(ex, (56, 7), compress=compress, maxlen=100)
        path = os.path.join(tmp_path, 'checkpoint.examples')
        formats.save_train_examples(path, [it, it])
        # what Coach.loadTrainExamples does (Coach.py:241-249): pickle.load, then per example pickle.loads(zlib.decompress)
        hist = pickle.load(open(path, 'rb'))
        assert len(hist) == 2 and hist[0].maxlen == 100
        first = hist[0][0] if not compress else pickle.loads(zlib.decompress(hist[0][0]))
        assert isinstance(first, tuple) and len(first) == 5
        assert first[0].shape == (56, 7) and first[0].dtype == np.int8 and first[3].dtype == bool
        back = formats.load_train_examples(path)
        for i in range(n):
            b, p, z, v, q = back[1][i]
            assert np.array_equal(b.reshape(-1), ex[0][i]) and np.array_equal(p, ex[1][i]) and np.array_equal(z, ex[2][i])
            assert np.array_equal(v, ex[3][i].astype(bool)) and np.array_equal(q, ex[4][i])


def test_arena_initial_state_string():
    from azg_amd import formats
    board = (np.arange(392) % 11 - 3).astype(np.int8).reshape(56, 7)
    s = formats.encode_initial_state(board, 1, 300)
    # the reference's decoder, Arena.py:61-65
    import base64
    data = zlib.decompress(base64.b64decode(s), wbits=-15)
    assert np.array_equal(np.frombuffer(data[:-3], dtype=np.int8).reshape(board.shape), board)
    assert int(data[-3]) == 1 and int.from_bytes(data[-2:], 'big') == 300
    b, p, it = formats.decode_initial_state(s, (56, 7))
    assert np.array_equal(b, board) and (p, it) == (1, 300)


def test_reads_a_file_written_by_the_reference():
    """tests/golden/ref_checkpoint.examples was written by the reference's own Coach.saveTrainExamples (Coach.py:220-226,
    tools/gen_train_golden.py): two iterations of zlib-compressed 5-tuples = the first six examples of the trainer fixture.
    (The opposite direction -- the reference's Coach.loadTrainExamples + trainer reading a file written by
    formats.save_train_examples -- is executed live by tools/gen_train_golden.py in the build container and by
    tests/test_live_reference.py when /root/reference is present.)"""
    from azg_amd import formats
    here = os.path.dirname(__file__)
    hist = formats.load_train_examples(os.path.join(here, 'golden', 'ref_checkpoint.examples'))
    d = np.load(os.path.join(here, 'golden', 'train_splendor2_v80.npz'))
    assert [len(it) for it in hist] == [4, 2]
    flat = [e for it in hist for e in it]
    for i, (b, p, z, v, q) in enumerate(flat):
        assert b.dtype == np.int8 and np.array_equal(b, d['boards'][i]) and np.array_equal(p, d['pi'][i])
        assert np.array_equal(z, d['z'][i]) and np.array_equal(v, d['valids'][i].astype(bool)) and np.array_equal(q, d['q'][i])
