"""NeuralNet plugin surface (NeuralNet.py:14-50, GenericNNetWrapper.py:24-277) on the engine: the object `main.py`,
`pit.py:40-66 create_player` and `Coach.py:29-33` expect from `<game>/NNet.py`:

    nnet = NNetWrapper(game, nn_args)                  # nn_args: dict with learn_rate / lr, dropout, epochs, batch_size,
                                                       #          nn_version, q_weight  (main.py:124-156, pit.py:44)
    nnet.predict(board, valid_actions) -> (pi f32[A] probabilities, v f32[P])          (GenericNNetWrapper.py:94-120)
    nnet.train(examples[, validation_set, save_folder, every])                          (:44-92)
    nnet.save_checkpoint(folder, filename, additional_keys)                             (:192-205)
    nnet.load_checkpoint(folder, filename) -> the checkpoint dict (state_dict, full_model, embedded args) or None (:207-221)
    nnet.args, nnet.nnet (the trainable torch module), nnet.requestKnowledgeTransfer

The module is one of azg_amd.train's trainable nets (same parameter names as the reference's, so the reference's .pt files load
with strict=True); inference runs on the engine's one-launch MFMA kernels (azg_amd.nnet.*Hip) with the module's current
weights (BatchNorm folded) -- the reference exports to ONNX-runtime on one CPU core at this point (:232-277).  The evaluator is
rebuilt lazily after every weight change (train / load_checkpoint).  `predict_batch` is the batched form the engine uses."""
import copy
import os
import pickle
import zlib

import numpy as np
import torch

from . import _lib, nnet as _nn, train as _train

# nn_version -> trainable module per game (splendor/SplendorNNet.py V80, azul/AzulNNet.py V84, santorini/SantoriniNNet.py V89/V78)
_DEFAULT_VERSION = {(_lib.SPLENDOR, 2): 80, (_lib.SPLENDOR, 3): 80, (_lib.SPLENDOR, 4): 80, (_lib.AZUL, 2): 84,
                    (_lib.SANTORINI, 1): 89, (_lib.SANTORINI, 11): 78}


def _module_for(game, version, dropout):
    gid, P, A = game.GAME_ID, game.P, game.A
    if gid == _lib.SPLENDOR and version == 80:
        return _train.SplendorV80Module(P, A, dropout)
    if gid == _lib.AZUL and version == 84:
        return _train.AzulV84Module(P, A, dropout)
    if gid == _lib.SANTORINI and version == 89 and game.variant == 1:      # (88 is a different trunk in the reference, SantoriniNNet.py:167-192: not built)
        return _train.SantoriniV89Module(P, A, dropout)
    if gid == _lib.SANTORINI and version == 78 and game.variant == 11:
        return _train.SantoriniV78Module(P, A, dropout)
    raise ValueError('nn_version %r is not built for this game (engine nets: Splendor 80, Azul 84, Santorini 89 no-gods / 78 with gods)'
                     % (version,))


def evaluator_for(module, game, max_batch):
    """engine-kernel evaluator (one launch per leaf batch) of a trainable module's current weights"""
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    dev, ver = str(game.device), getattr(module, 'version', 80)
    if not isinstance(module, (_train.SplendorV80Module, _train.AzulV84Module, _train.SantoriniV89Module, _train.SantoriniV78Module)):
        return _nn.TorchModuleEvaluator(module, game, max_batch)
    if ver == 84:
        return _nn.MobileNet1dHip(_nn.AzulV84(sd, num_players=game.P, device=dev), max_batch=max_batch)
    if ver == 89:
        return _nn.SantoriniV89Hip(_nn.SantoriniV89(sd, device=dev), max_batch=max_batch)
    if ver == 78:
        return _nn.SantoriniV78Hip(_nn.SantoriniV78(sd, device=dev), max_batch=max_batch)
    if game.P == 2:
        return _nn.SplendorV80Hip(sd, num_players=2, device=dev, max_batch=max_batch)
    return _nn.MobileNet1dHip(_nn.SplendorV80(sd, num_players=game.P, device=dev), max_batch=max_batch)


def decode_examples(examples):
    """Coach's example list (5-tuples, or zlib-compressed pickles of them, Coach.py:84) -> five stacked arrays"""
    ex = [e if isinstance(e, tuple) else pickle.loads(zlib.decompress(e)) for e in examples]
    return [np.stack([np.asarray(e[k]).reshape(-1) for e in ex]) for k in range(5)]


class NNetWrapper:
    def __init__(self, game, nn_args, module=None):
        """module: a torch module with the reference's forward signature for a game / architecture without an engine net (the f4
        games with the reference's <G>NNet classes); inference then runs it on PyTorch-ROCm (nnet.TorchModuleEvaluator)"""
        self.game = game
        self.args = nn_args
        self.board_size, self.action_size, self.num_players = game.getBoardSize(), game.getActionSize(), game.num_players
        self.requestKnowledgeTransfer = False
        ver = self._arg('nn_version', -1)
        self.nnet, self._custom = module, module is not None
        if module is not None:
            pass
        elif ver is not None and ver > 0:
            self.nnet = _module_for(game, ver, float(self._arg('dropout', 0.0) or 0.0))
        else:
            # pit.py:44 builds the wrapper with nn_version = -1 and lets load_checkpoint bring the model (:258-260)
            dv = _DEFAULT_VERSION.get((game.GAME_ID, game.variant))
            self.nnet = _module_for(game, dv, float(self._arg('dropout', 0.0) or 0.0)) if dv else None
        self._eval, self._eval_batch = None, 0

    def _arg(self, k, d=None):
        a = self.args
        v = a.get(k, d) if isinstance(a, dict) else getattr(a, k, d)
        return d if v is None else v

    # ---- inference ----
    def evaluator(self, max_batch=1):
        """the engine-kernel net for leaf batches up to max_batch (what SelfPlayEngine / BatchedMCTS call per round)"""
        if self._eval is None or self._eval_batch < max_batch:
            self._eval, self._eval_batch = evaluator_for(self.nnet, self.game, max_batch), max_batch
        return self._eval

    def predict_batch(self, boards, valids):
        return self.evaluator(int(boards.shape[0])).predict_batch(boards, valids)

    def predict(self, board, valid_actions):
        """NeuralNet.predict (NeuralNet.py:32-43): one board -> (pi probabilities f32[A], v f32[P]) as numpy arrays"""
        dev = self.game.device
        b = torch.from_numpy(np.ascontiguousarray(board, dtype=np.int8)).to(dev).reshape((1,) + tuple(self.board_size))
        m = torch.from_numpy(np.ascontiguousarray(np.asarray(valid_actions).astype(np.uint8))).to(dev).reshape(1, -1)
        pi, v = self.evaluator(1).predict_batch(b, m)
        return pi[0].float().cpu().numpy(), v[0].float().cpu().numpy()

    # ---- training (GenericNNetWrapper.train :44-92) ----
    def train(self, examples, validation_set=None, save_folder=None, every=0, seed=None, log=None):
        cols = decode_examples(examples) if isinstance(examples, (list, tuple)) and len(examples) and not hasattr(examples[0], 'shape') \
            else examples
        lr = self._arg('learn_rate', self._arg('lr', 3e-3))
        hist = _train.train(self.nnet, cols, learn_rate=float(lr), batch_size=int(self._arg('batch_size', 32)),
                            epochs=int(self._arg('epochs', 1)), q_weight=float(self._arg('q_weight', 0.5)),
                            device=str(self.game.device), seed=seed, log=log, board_shape=self.board_size if self._custom else None)
        self._eval = None
        return hist

    def loss_pi(self, targets, outputs):
        return _train.loss_pi(targets, outputs)

    def loss_v(self, targets_V, targets_Q, outputs):
        return _train.loss_v(targets_V, targets_Q, outputs, float(self._arg('q_weight', 0.5)))

    # ---- checkpoints (:192-221) ----
    def save_checkpoint(self, folder='checkpoint', filename='checkpoint.pth.tar', additional_keys={}):
        os.makedirs(folder, exist_ok=True)
        cpu = copy.deepcopy(self.nnet).cpu()
        data = {'state_dict': cpu.state_dict(), 'full_model': cpu}
        data.update(additional_keys)
        torch.save(data, os.path.join(folder, filename))

    def load_checkpoint(self, folder='checkpoint', filename='checkpoint.pth.tar'):
        path = os.path.join(folder, filename)
        if not os.path.exists(path):
            print('No model in path {}'.format(path))
            return None
        try:
            ck = torch.load(path, map_location='cpu', weights_only=False)
        except Exception:
            # a reference checkpoint pickles its own module class as `full_model`; without the reference's package on
            # sys.path only the tensors and the embedded args can be read
            ck = _load_without_full_model(path)
            if ck is None:
                print("MODEL {} CAN'T BE READ but file exists".format(path))
                return None
        ver = getattr(ck.get('full_model'), 'version', ck.get('nn_version'))
        want = self._arg('nn_version', -1)
        if not self._custom and want is not None and want > 0 and ver is not None and ver != want:          # :250-253
            print('Checkpoint includes NN version', ver, ', but you ask version', want, ' so not loading it and initiate knowledge transfer')
            self.requestKnowledgeTransfer = True
            return ck
        try:
            # the replacement module is built and filled in a local: self.nnet changes only when the whole load succeeded (the
            # reference's load_network keeps its existing net when the state dict does not fit, GenericNNetWrapper.py:262-267)
            target = self.nnet
            if not self._custom and (target is None or getattr(target, 'version', None) != ver):
                target = _module_for(self.game, ver, float(self._arg('dropout', 0.0) or 0.0))
            sd = {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v for k, v in ck['state_dict'].items()}
            if target is self.nnet and target is not None:
                # load_state_dict copies tensor by tensor: check the fit first, so that a mismatch cannot leave a half-loaded net
                own = target.state_dict()
                bad = [k for k in own if k not in sd or tuple(sd[k].shape) != tuple(own[k].shape)] + [k for k in sd if k not in own]
                if bad:
                    raise RuntimeError('state dict does not fit: %s' % ', '.join(bad[:4]))
            target.load_state_dict(sd, strict=True)
        except (RuntimeError, ValueError) as ex:                         # another architecture: GenericNNetWrapper.py:262-267
            print('Could not load state dict (%s), initiate knowledge transfer' % str(ex).splitlines()[0])
            self.requestKnowledgeTransfer = True
            self._eval = None                                            # (nothing cached may outlive a failed load)
            return ck
        self.nnet = target
        self._eval = None
        return ck


class _Opaque:
    """placeholder for a pickled object whose class cannot be imported (the reference's `full_model`)"""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__['_state'] = state


def _load_without_full_model(path):
    import pickle as _p

    class U(_p.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except Exception:
                return _Opaque

    class P:                                            # the `pickle_module` interface torch.load expects
        Unpickler = U
        load = staticmethod(_p.load)
        __name__ = 'pickle'

    try:
        ck = torch.load(path, map_location='cpu', weights_only=False, pickle_module=P)
    except Exception:
        return None
    fm = ck.get('full_model')
    if isinstance(fm, _Opaque):
        ck['full_model'] = None
    return ck
