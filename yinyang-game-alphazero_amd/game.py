"""Rules API of the reference, backed by the HIP rules kernels.

`YinYangLogic` mirrors src/yin_yang/yin_yang_logic.py:4-134 and `YinYangGame` mirrors
src/yin_yang/yin_yang_game.py:4-206 (same method names, argument meaning, return types and the
same quirks: getNextState mutates the board it is given and returns it, illegal placements are
silently ignored, the player flips regardless).  Every rule decision is made by
csrc/yy_engine.hip through the C ABI on the current ROCm device -- there is no host
re-implementation; single-board calls are batch-of-one launches and exist for API compatibility,
the throughput path is the batched tensor API in engine.py.
"""
import numpy as np
import torch

from . import engine

DRAW_VALUE = 0.0001   # yin_yang_game.py:107


def _dev():
    if not torch.cuda.is_available():
        from ._lib import YYError
        raise YYError(-100, "the rules kernels need a ROCm device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class YinYangLogic:
    """int8 n x m board: 0 empty, 1 black, -1 white (yin_yang_logic.py:8-18)."""

    def __init__(self, n=8, m=8, rowcol_rule=False):
        self.n, self.m = n, m
        self.board = np.zeros((n, m), dtype=np.int8)
        self.rowcol_rule = rowcol_rule

    def get_board(self):
        return self.board.copy()

    def _dev_board(self):
        return torch.from_numpy(np.ascontiguousarray(self.board, dtype=np.int8)[None]).to(_dev())

    def _mask(self, piece):
        pl = torch.tensor([1 if piece == 1 else -1], dtype=torch.int8, device=_dev())
        return engine.valid_mask(self._dev_board(), pl, self.rowcol_rule)[0].cpu().numpy()

    def is_valid_move(self, x, y, piece):
        if not (0 <= x < self.n and 0 <= y < self.m):
            return False
        return bool(self._mask(piece)[x * self.m + y])

    def place_piece(self, x, y, piece):
        if self.is_valid_move(x, y, piece):
            self.board[x, y] = piece
            return True
        return False

    def get_valid_moves(self, piece):
        idx = np.flatnonzero(self._mask(piece))
        return [(int(a // self.m), int(a % self.m)) for a in idx]

    def has_valid_move(self, piece):
        return bool(self._mask(piece).any())

    def count_pieces(self):
        return int(np.sum(self.board == 1)), int(np.sum(self.board == -1))

    # the two predicates is_valid_move is made of in the reference (yin_yang_logic.py:58-109); host numpy on the current
    # board, kept for callers that probe them directly -- the kernels evaluate the fused form
    def _check_connectivity(self, piece):
        """True when colour `piece` is absent or forms one 4-connected group."""
        cells = np.argwhere(self.board == piece)
        if len(cells) == 0:
            return True
        todo, seen = [tuple(cells[0])], {tuple(cells[0])}
        while todo:
            x, y = todo.pop()
            for nx, ny in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)):
                if 0 <= nx < self.n and 0 <= ny < self.m and self.board[nx, ny] == piece and (nx, ny) not in seen:
                    seen.add((nx, ny))
                    todo.append((nx, ny))
        return len(seen) == len(cells)

    def _check_2x2_constraint(self):
        """True when no 2x2 window holds four equal stones (either colour)."""
        b = self.board
        if self.n < 2 or self.m < 2:
            return True
        same = (b[:-1, :-1] == b[1:, :-1]) & (b[:-1, :-1] == b[:-1, 1:]) & (b[:-1, :-1] == b[1:, 1:]) & (b[:-1, :-1] != 0)
        return not bool(same.any())


class YinYangGame:
    """AlphaZero Game interface (yin_yang_game.py:4-206)."""

    def __init__(self, n=8, m=8, rowcol_rule=False):
        self.n, self.m = n, m
        self.action_size = n * m
        self.rowcol_rule = rowcol_rule

    def getInitBoard(self):
        return YinYangLogic(self.n, self.m, self.rowcol_rule)

    def getBoardSize(self):
        return (self.n, self.m)

    def getActionSize(self):
        return self.action_size

    def getNextState(self, board, player, action):
        """Place iff legal IN PLACE and return the same object; next player is -player regardless
        (yin_yang_game.py:39-58)."""
        dev = _dev()
        b = board._dev_board()
        pl = torch.tensor([1 if player == 1 else -1], dtype=torch.int8, device=dev)
        act = torch.tensor([int(action)], dtype=torch.int32, device=dev)
        engine.step_(b, pl, act, self.rowcol_rule)
        board.board[...] = b[0].cpu().numpy()
        return board, -player

    def getValidMoves(self, board, player):
        """float64 0/1 vector of length action_size (yin_yang_game.py:60-78)."""
        return board._mask(1 if player == 1 else -1).astype(np.float64)

    def getGameEnded(self, board, player):
        """0 ongoing, 1 / -1 from `player`'s view, 0.0001 draw (yin_yang_game.py:80-110)."""
        pl = torch.tensor([1 if player == 1 else -1], dtype=torch.int8, device=_dev())
        v = float(engine.game_ended(board._dev_board(), pl, self.rowcol_rule)[0])
        if v == 0.0:
            return 0
        return DRAW_VALUE if v == DRAW_VALUE else int(v)

    def getCanonicalForm(self, board, player):
        return board   # identity in the reference (yin_yang_game.py:112-125)

    def getSymmetries(self, board, pi):
        """8 dihedral forms of (board, pi) (yin_yang_game.py:127-166); host numpy, not on the hot path."""
        grid = np.reshape(pi, (self.n, self.m))
        arr = board.get_board()
        out = []
        for k in range(1, 5):
            for flip in (True, False):
                nb, npi = np.rot90(arr, k), np.rot90(grid, k)
                if flip:
                    nb, npi = np.fliplr(nb), np.fliplr(npi)
                lb = YinYangLogic(self.n, self.m, self.rowcol_rule)
                lb.board = np.ascontiguousarray(nb)
                out.append((lb, npi.flatten()))
        return out

    def stringRepresentation(self, board):
        return board.get_board().tobytes()

    def _action_to_coords(self, action):
        return action // self.m, action % self.m

    def _coords_to_action(self, x, y):
        return x * self.m + y

    def display(self, board):
        b = board.get_board()
        print(" " + "".join(chr(97 + j) for j in range(self.m)))
        for i in range(self.n):
            print(str(i + 1) + "".join("B" if v == 1 else ("W" if v == -1 else ".") for v in b[i]))
