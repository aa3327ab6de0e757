"""MI355X-native Yin-Yang AlphaZero self-play hot path (rules + batched MCTS as HIP kernels).

Import name: `yinyang_game_alphazero_amd` (alias module at the repo root).
The reference's Python API is mirrored by: YinYangLogic / YinYangGame (rules), MCTS (search),
YinYangNeuralNetwork (evaluator), SelfPlayWorker / SelfPlayManager / generate_self_play_data.
"""
from . import _lib  # noqa: F401
from ._lib import YYError, build  # noqa: F401

__all__ = ["YYError", "build"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `build()` works in a bare interpreter
    import importlib
    table = {
        "engine": ".engine", "BatchedMCTS": ".engine",
        "YinYangLogic": ".game", "YinYangGame": ".game",
        "YinYangNeuralNetwork": ".network", "BatchedEvaluator": ".network",
        "MCTS": ".mcts", "Node": ".mcts",
        "SelfPlayWorker": ".self_play", "SelfPlayManager": ".self_play",
        "generate_self_play_data": ".self_play", "SelfPlayEngine": ".self_play", "SelfPlayLanes": ".self_play",
        "training": ".training", "AlphaZeroTrainer": ".training", "TrainingDataQueue": ".training",
        "TrainingPipeline": ".training", "run_training_pipeline": ".training", "augment_batch": ".training",
        "DataProcessor": ".training", "create_dataset_from_games": ".training",
        "arena": ".arena", "Arena": ".arena", "AlphaZero": ".arena", "AlphaZeroPlayer": ".arena",
        "RandomPlayer": ".arena", "evaluate_vs_random": ".arena",
    }
    if name in table:
        mod = importlib.import_module(table[name], __name__)
        return mod if name == table[name][1:] else getattr(mod, name)
    raise AttributeError(name)
