"""alpha-zero-general_amd -- MI355X-native batched self-play engine behind the reference's Game / NeuralNet.predict /
MCTS / Coach plugin surface (cestpasphoto/alpha-zero-general).  Importable as `azg_amd` (see ../azg_amd/__init__.py)."""
from ._lib import AzgError, SPLENDOR, SANTORINI, AZUL, game_info, lib  # noqa: F401
