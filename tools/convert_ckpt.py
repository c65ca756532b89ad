"""Convert a reference checkpoint (.pt with pickled full_model, needs torchvision) into plain tensors + MCTS args, and
record net-forward golden vectors (G4) from the reference's own model.  Build-container only.
    python tools/convert_ckpt.py
writes tests/golden/weights_splendor2_v80.npz (state_dict tensors + embedded MCTS args, DATA only) and
       tests/golden/netfwd_splendor2_v80.npz (256 boards/masks -> the reference model's pi, v)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402

GOLDEN = os.path.join(HERE, '..', 'tests', 'golden')


def convert(name, ckpt_rel, load_kw, game_mod, game_cls, n_vec=256, forward=True):
    import torch
    m = H.load_reference(**load_kw)
    ck = torch.load(os.path.join(H.REFERENCE, ckpt_rel), map_location='cpu', weights_only=False)
    sd = {k: v.numpy() for k, v in ck['state_dict'].items()}
    meta = {k: ck[k] for k in ck if k not in ('state_dict', 'full_model')}
    out = {'sd/' + k: v for k, v in sd.items()}
    for k, v in meta.items():
        if isinstance(v, (int, float, bool)):
            out['arg/' + k] = np.array(v)
        elif isinstance(v, (list, tuple)) and all(isinstance(x, (int, float)) for x in v):
            out['arg/' + k] = np.array(v, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, 'weights_%s.npz' % name), **out)
    print(name, 'args:', {k: v for k, v in meta.items() if k in ('nn_version', 'cpuct', 'fpu', 'universes', 'numMCTSSims',
                                                                   'dirichletAlpha', 'temperature', 'tempThreshold')})
    if not forward:      # the pickled full_model needs the real torchvision (absent here): weights + args only
        print('wrote weights for', name, '(no forward vectors)')
        H.cleanup()
        return
    # G4: forward vectors from the reference's own module (GenericNNetWrapper.py:112-120 torch branch)
    model = ck['full_model'].eval()
    env = np.load(os.path.join(GOLDEN, 'env_%s.npz' % name.split('_')[0]))
    rng = np.random.default_rng(0)
    sel = rng.choice(len(env['canonical']), size=min(n_vec, len(env['canonical'])), replace=False)
    g = getattr(m[game_mod], game_cls)()
    shape = tuple(g.getBoardSize())
    boards = env['canonical'][sel].reshape((-1,) + shape)
    masks = np.array([g.getValidMoves(b, 0) for b in boards])
    with torch.no_grad():
        lp, v = model(torch.from_numpy(boards.astype(np.float32)), torch.from_numpy(masks.astype(bool)))
    np.savez_compressed(os.path.join(GOLDEN, 'netfwd_%s.npz' % name), boards=boards.astype(np.int8),
                        masks=masks.astype(np.uint8), pi=torch.exp(lp).numpy(), v=v.numpy())
    print('wrote weights +', len(sel), 'forward vectors for', name)
    H.cleanup()


def forward_f64(name, ckpt_rel, load_kw):
    """netfwd64_<name>.npz: the reference's OWN module evaluated in float64 (model.double()) on the boards / masks of
    netfwd_<name>.npz -- the rounding-free value of the reference's forward.  Evidence for the value-head tolerance: the
    reference's f32 output itself differs from it by up to ~1e-5 (printed), so the tests bound |ours - f64| <= 1e-5."""
    import torch
    H.load_reference(**load_kw)
    ck = torch.load(os.path.join(H.REFERENCE, ckpt_rel), map_location='cpu', weights_only=False)
    model = ck['full_model'].eval().double()
    d = np.load(os.path.join(GOLDEN, 'netfwd_%s.npz' % name))
    with torch.no_grad():
        lp, v = model(torch.from_numpy(d['boards'].astype(np.float64)), torch.from_numpy(d['masks'].astype(bool)))
    pi64, v64 = torch.exp(lp).numpy(), v.numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'netfwd64_%s.npz' % name), pi64=pi64, v64=v64)
    print('%-16s |ref_f32 - ref_f64|: pi %.3g  v %.3g' % (name, np.abs(d['pi'] - pi64).max(), np.abs(d['v'] - v64).max()))
    H.cleanup()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--f64':
        forward_f64('splendor2_v80', 'splendor/pretrained_2players.pt', dict(splendor_players=2))
        forward_f64('splendor4_v80', 'splendor/pretrained_4players.pt', dict(splendor_players=4))
        forward_f64('santorini1_v89', 'santorini/pretrained.pt', dict(santorini_gods=1))
        forward_f64('azul_v84', 'azul/pretrained.pt', dict())
        forward_f64('santorini11_v78', 'santorini/pretrained_withgods.pt', dict(santorini_gods=11))
        forward_f64('minivilles2_v82', 'minivilles/pretrained_2players.pt', dict(minivilles_players=2))
        forward_f64('tlp3_v83', 'thelittleprince/pretrained_3players.pt', dict(tlp_players=3))
        return
    if len(sys.argv) > 2 and sys.argv[1] == '--only':
        return convert(*{'santorini11_v78': ('santorini11_v78', 'santorini/pretrained_withgods.pt', dict(santorini_gods=11),
                                             'SantoriniGame', 'SantoriniGame', 128),
                         # the two f4 games whose shipped checkpoints are nets of the MobileNet-1d family (MinivillesNNet.py:101-123
                         # nn_version 82, TLPNNet.py:175-196 nn_version 83): engine nets through nn_mb1d.hip.h
                         'minivilles2_v82': ('minivilles2_v82', 'minivilles/pretrained_2players.pt', dict(minivilles_players=2),
                                             'MinivillesGame', 'MinivillesGame', 128),
                         'tlp3_v83': ('tlp3_v83', 'thelittleprince/pretrained_3players.pt', dict(tlp_players=3), 'TLPGame', 'TLPGame', 128)}[sys.argv[2]])
    convert('splendor2_v80', 'splendor/pretrained_2players.pt', dict(splendor_players=2), 'SplendorGame', 'SplendorGame')
    convert('splendor4_v80', 'splendor/pretrained_4players.pt', dict(splendor_players=4), 'SplendorGame', 'SplendorGame', n_vec=128)
    convert('santorini1_v89', 'santorini/pretrained.pt', dict(santorini_gods=1), 'SantoriniGame', 'SantoriniGame', n_vec=128)
    convert('azul_v84', 'azul/pretrained.pt', dict(), 'AzulGame', 'AzulGame', n_vec=128)
    # V78 is built from torchvision.models.mobilenetv3.InvertedResidual.  torchvision is not installed here; tools/refshim
    # carries a functional stand-in written from the published block algorithm, so the reference's OWN SantoriniNNet.forward
    # (SantoriniNNet.py:264-271: gods embedding, heads, masking) runs on the unpickled full_model: the block body is pinned to
    # the published algorithm, everything around it to the reference
    convert('santorini11_v78', 'santorini/pretrained_withgods.pt', dict(santorini_gods=11), 'SantoriniGame', 'SantoriniGame',
            n_vec=128)


if __name__ == '__main__':
    main()
