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


def main():
    import torch
    H.load_reference(splendor_players=2)
    ck = torch.load(os.path.join(H.REFERENCE, 'splendor', 'pretrained_2players.pt'), map_location='cpu',
                    weights_only=False)
    sd = {k: v.numpy() for k, v in ck['state_dict'].items()}
    meta = {k: ck[k] for k in ck if k not in ('state_dict', 'full_model')}
    out = {'sd/' + k: v for k, v in sd.items()}
    for k, v in meta.items():
        if isinstance(v, (int, float, bool)):
            out['arg/' + k] = np.array(v)
        elif isinstance(v, (list, tuple)) and all(isinstance(x, (int, float)) for x in v):
            out['arg/' + k] = np.array(v, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, 'weights_splendor2_v80.npz'), **out)
    print('args:', {k: v for k, v in meta.items() if not hasattr(v, 'shape')})

    # G4: forward vectors from the reference's own module (GenericNNetWrapper.py:112-120 torch branch)
    model = ck['full_model'].eval()
    env = np.load(os.path.join(GOLDEN, 'env_splendor2.npz'))
    rng = np.random.default_rng(0)
    sel = rng.choice(len(env['canonical']), size=256, replace=False)
    boards = env['canonical'][sel].reshape(-1, 56, 7)
    import splendor.SplendorGame as SG
    g = SG.SplendorGame()
    masks = np.array([g.getValidMoves(b, 0) for b in boards])
    with torch.no_grad():
        lp, v = model(torch.from_numpy(boards.astype(np.float32)), torch.from_numpy(masks.astype(bool)))
    np.savez_compressed(os.path.join(GOLDEN, 'netfwd_splendor2_v80.npz'), boards=boards.astype(np.int8),
                        masks=masks.astype(np.uint8), pi=torch.exp(lp).numpy(), v=v.numpy())
    print('wrote weights +', len(sel), 'forward vectors')
    H.cleanup()


if __name__ == '__main__':
    main()
