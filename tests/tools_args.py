"""MCTS settings of the reference's pretrained checkpoints used across parity tests."""
MCTS_ARGS = {
    'splendor2': dict(cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True),
    'splendor3': dict(cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True),
    'splendor4': dict(cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True),
    'santorini1': dict(cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True),
    'santorini11': dict(cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True),
    'azul': dict(cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True),
    'abalone': dict(cpuct=1.0, fpu=0.0, universes=0, forced_playouts=True),
    'akropolis': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'akropolis3': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'akropolis4': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'smallworld': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'smallworld3': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'smallworld4': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
    'minivilles2': dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True),
}
