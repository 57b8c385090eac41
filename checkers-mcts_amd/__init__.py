"""checkers-mcts_amd: MI355X-native batched self-play engine for the
MCTS + Checkers hot path of AlexMGitHub/Checkers-MCTS.

Importable as `checkers_mcts_amd` (the directory name carries a hyphen; the
sibling `checkers_mcts_amd/` package is a two-line alias that loads this one).
"""
__version__ = "0.1.0"
