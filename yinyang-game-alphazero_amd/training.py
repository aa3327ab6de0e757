"""Replay data, augmentation and the trainer.

Mirrors src/yin_yang/ai/data_utils.py (DataProcessor.augment_sample :39-134),
ai/training_pipeline.py (TrainingDataQueue :23-106, TrainingPipeline :108-291,
run_training_pipeline :293-329) and ai/trainer.py (AlphaZeroTrainer :15-212) with the same names,
hyper-parameters (Adam lr 1e-3, weight decay 1e-4, batch 64, soft-target cross-entropy + MSE) and
checkpoint names (`checkpoint_<n>.pth.tar`, reference dict format), on tensors instead of pickled
objects: examples are `states int8 [N,R,C]`, `policies f32 [N,A]`, `values f32 [N]`.  The 5 input
planes are produced by the HIP encode kernel and the 8-fold symmetry augmentation runs batched on
the device (the reference rebuilds a whole network per sample to encode it, data_utils.py:30-32).
Training itself is stock PyTorch-ROCm (out of the hot path).
"""
import glob
import os
import pickle
import random
import time

import numpy as np
import torch
import torch.nn.functional as F

from .network import YinYangNeuralNetwork


def augment_batch(planes, policies):
    """The reference's 8 variants, in its order (data_utils.py:52-132): identity, rot90 x1/x2/x3,
    horizontal flip, vertical flip, transpose, anti-transpose -- applied to ALL 5 planes as images
    (the reference does not swap the row-/column-fill planes under a rotation; neither do we) and to
    the policy reshaped to the board.  planes [N,5,R,R], policies [N,R*R] -> [8N,...] with variant v
    of sample n at row v*N + n.  Square boards only (the reference's code also only works there)."""
    N, _, R, C = planes.shape
    assert R == C, "augmentation needs a square board"
    grid = policies.reshape(N, 1, R, C)

    def both(fn):
        return fn(planes), fn(grid).reshape(N, R * C)

    outs = [
        (planes, policies),
        both(lambda t: torch.rot90(t, 1, (2, 3))),
        both(lambda t: torch.rot90(t, 2, (2, 3))),
        both(lambda t: torch.rot90(t, 3, (2, 3))),
        both(lambda t: torch.flip(t, (3,))),
        both(lambda t: torch.flip(t, (2,))),
        both(lambda t: t.transpose(2, 3)),
        both(lambda t: torch.flip(t.transpose(2, 3), (2, 3))),
    ]
    return torch.cat([o[0] for o in outs]).contiguous(), torch.cat([o[1] for o in outs]).contiguous()


def encode_planes_host(states):
    """int8 [N,R,C] -> planes f32 [N,5,R,C] with torch ops on the tensor's own device (neural_network.py:156-196: fractions
    formed in float64, stored as float32).  The trainer uses the HIP encode kernel on a ROCm device and this elsewhere."""
    b = states.to(torch.int8)
    occ = (b != 0)
    n, m = b.shape[1:]
    rows = (occ.sum(2).double() / m).float()[:, :, None].expand(-1, n, m)
    cols = (occ.sum(1).double() / n).float()[:, None, :].expand(-1, n, m)
    return torch.stack([(b == 0).float(), (b == 1).float(), (b == -1).float(), rows, cols], dim=1)


class DataProcessor:
    """data_utils.py:7-180 with the same methods, on top of the batched routines of this module (one sample = a batch of
    one).  `augment_sample` returns the reference's 8 variants in its order (pinned by the G6 fixture)."""

    def __init__(self, game):
        self.game = game
        self.board_size = game.getBoardSize()

    def preprocess_sample(self, board, policy, player):
        """board object (get_board()) or int8 array -> (planes f32 [5,R,C], policy FloatTensor [A]); `player` is unused,
        as in the reference (:16-37)."""
        arr = board.get_board() if hasattr(board, "get_board") else np.asarray(board)
        planes = encode_planes_host(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int8))[None])[0]
        return planes, torch.as_tensor(np.asarray(policy), dtype=torch.float32)

    def augment_sample(self, board, policy):
        """(planes [5,R,R], policy [A]) -> list of 8 (planes, policy) pairs (:39-134)."""
        ap, api = augment_batch(board[None], torch.as_tensor(policy, dtype=torch.float32)[None])
        return [(ap[v], api[v]) for v in range(8)]

    def _policy_to_grid(self, policy):
        n, m = self.board_size
        if isinstance(policy, torch.Tensor):
            policy = policy.detach().cpu().numpy()
        return np.where(np.asarray(policy) > 0, policy, 0.0).reshape(n, m).astype(np.float64)

    def _grid_to_policy(self, policy_grid):
        g = np.asarray(policy_grid, dtype=np.float64)
        return torch.FloatTensor(np.where(g > 0, g, 0.0).reshape(-1))


def create_dataset_from_games(game_data, game, augment=True):
    """data_utils.py:182-220: list of (board, policy, value) -> three lists of tensors (8 entries per sample when
    augmenting, sample-major like the reference's loop)."""
    proc = DataProcessor(game)
    boards, policies, values = [], [], []
    for board, policy, value in game_data:
        bt, pt = proc.preprocess_sample(board, policy, 1)
        for ab, ap in (proc.augment_sample(bt, pt) if augment else [(bt, pt)]):
            boards.append(ab)
            policies.append(ap)
            values.append(torch.FloatTensor([value]))
    return boards, policies, values


def save_examples_reference_format(path, states, policies, values):
    """Write a self_play_data_*.npz the REFERENCE's own loader reads (training_pipeline.py:56-77: `np.load(allow_pickle=True)`,
    then `boards[i].get_board()` via neural_network.py:178): `boards` is an object array whose elements unpickle, inside the
    reference's process, as instances of ITS class `src.yin_yang.yin_yang_logic.YinYangLogic` (attributes n, m, board --
    yin_yang_logic.py:14-18).  Nothing of the reference is imported to write it: a stand-in class carrying the reference's
    module path and name is registered under that module name only while the file is pickled, so the stream holds the
    global reference `src.yin_yang.yin_yang_logic YinYangLogic` plus plain state.  `states`, `policies`, `values` are
    stored beside it as plain arrays, so this package's loader (allow_pickle=False on those keys) reads the same file."""
    import sys
    import types
    states = np.ascontiguousarray(states, dtype=np.int8)
    mod_name = "src.yin_yang.yin_yang_logic"

    class YinYangLogic:                  # state only; methods come from the reference's class when IT unpickles
        pass

    YinYangLogic.__module__, YinYangLogic.__qualname__ = mod_name, "YinYangLogic"
    boards = np.empty(states.shape[0], dtype=object)
    for i in range(states.shape[0]):
        b = YinYangLogic()
        b.n, b.m, b.board = int(states.shape[1]), int(states.shape[2]), states[i].copy()
        boards[i] = b
    # pickle resolves a class by importing its module path, parents included: stand-ins for all three names, only for
    # the duration of the write (whatever was registered under those names before is put back)
    names = ["src", "src.yin_yang", mod_name]
    saved = {n: sys.modules.get(n) for n in names}
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules[mod_name].YinYangLogic = YinYangLogic
    try:
        np.savez(path, boards=boards, states=states, policies=np.asarray(policies, dtype=np.float64),
                 values=np.asarray(values, dtype=np.float64))
    finally:
        for n in names:
            if saved[n] is None:
                del sys.modules[n]
            else:
                sys.modules[n] = saved[n]
    return path


class _BoardUnpickler(pickle.Unpickler):
    """Unpickler for the `boards` member of a reference-format data file.  It can build exactly three things: numpy arrays
    (the array reconstructor, ndarray, dtype), and a plain attribute holder standing in for the reference's
    `src.yin_yang.yin_yang_logic.YinYangLogic`.  Any other global in the stream -- i.e. anything that could run code --
    raises UnpicklingError; nothing from the file is ever imported or called."""

    class Board:
        pass

    _NUMPY = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
              ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) == ("src.yin_yang.yin_yang_logic", "YinYangLogic"):
            return _BoardUnpickler.Board
        if (module, name) in self._NUMPY:
            if name == "_reconstruct":      # numpy >= 2 keeps it in numpy._core, older releases in numpy.core
                core = getattr(np, "_core", None) or __import__("numpy.core").core
                return core.multiarray._reconstruct
            return getattr(np, name)
        raise pickle.UnpicklingError(f"refusing global {module}.{name} in a training-data file")


def _load_reference_boards(path):
    """The `boards` object array of a reference-format .npz -> int8 [N,R,C], through _BoardUnpickler only."""
    import zipfile
    with zipfile.ZipFile(path) as zf, zf.open("boards.npy") as fp:
        version = np.lib.format.read_magic(fp)
        shape, _, dtype = (np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0)(fp)
        if not dtype.hasobject:
            raise ValueError("boards is a plain array")
        arr = _BoardUnpickler(fp).load()
    boards = [np.asarray(b.board if isinstance(b, _BoardUnpickler.Board) else b, dtype=np.int8) for b in arr.reshape(-1)]
    return np.stack(boards) if boards else np.zeros((0, 0, 0), np.int8)


def load_examples(path, allow_reference_objects=False):
    """Read a self_play_data_*.npz.  Files written by this package hold plain arrays and are read with allow_pickle=False.
    A file in the reference's own format (self_play.py:374-384: `boards` = pickled board objects) is read only on request,
    and then through a restricted unpickler that can construct numpy arrays and a plain stand-in for the board class and
    nothing else (_BoardUnpickler) -- never with numpy's allow_pickle."""
    z = np.load(path, allow_pickle=False)
    if "states" in z.files:
        states = z["states"]
    else:
        try:
            states = z["boards"]
        except ValueError:                       # object array
            if not allow_reference_objects:
                raise ValueError(f"{path} stores pickled board objects (the reference's format); pass "
                                 "allow_reference_objects=True to read it through the restricted unpickler") from None
            states = _load_reference_boards(path)
    return dict(states=torch.from_numpy(states.astype(np.int8)),
                policies=torch.from_numpy(z["policies"].astype(np.float32)),
                values=torch.from_numpy(z["values"].astype(np.float32)))


class TrainingDataQueue:
    """FIFO replay buffer (training_pipeline.py:23-106): newest `max_size` examples, uniform sampling
    without replacement of `sample_size`."""

    def __init__(self, max_size=500000, sample_size=10000):
        self.max_size = max_size
        self.sample_size = min(sample_size, max_size)
        self.states = self.policies = self.values = None

    def push_examples(self, examples):
        if isinstance(examples, dict):
            s, p, v = examples["states"].cpu(), examples["policies"].cpu().float(), examples["values"].cpu().float()
        else:   # list of (board, policy, value) like the reference
            s = torch.from_numpy(np.stack([np.asarray(e[0].get_board() if hasattr(e[0], "get_board") else e[0], np.int8) for e in examples]))
            p = torch.from_numpy(np.stack([np.asarray(e[1], np.float32) for e in examples]))
            v = torch.tensor([float(e[2]) for e in examples], dtype=torch.float32)
        if self.states is None:
            self.states, self.policies, self.values = s, p, v
        else:
            self.states = torch.cat([self.states, s])
            self.policies = torch.cat([self.policies, p])
            self.values = torch.cat([self.values, v])
        if len(self) > self.max_size:
            k = len(self) - self.max_size
            self.states, self.policies, self.values = self.states[k:], self.policies[k:], self.values[k:]

    def push_file(self, file_path, allow_reference_objects=False):
        if not os.path.exists(file_path):
            return
        self.push_examples(load_examples(file_path, allow_reference_objects))

    def sample(self, sample_size=None):
        n = len(self)
        if n == 0:
            return {}
        k = min(self.sample_size if sample_size is None else sample_size, n)
        idx = torch.tensor(random.sample(range(n), k), dtype=torch.long)     # random.sample like :95
        dist = _dist()
        if dist:                                                             # every rank trains on the same sample
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            idx = idx.to(dev)
            dist.broadcast(idx, src=0)
            idx = idx.cpu()
        return dict(states=self.states[idx], policies=self.policies[idx], values=self.values[idx])

    def __len__(self):
        return 0 if self.states is None else int(self.states.shape[0])


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class AlphaZeroTrainer:
    """trainer.py:15-212.  Under a multi-rank job (torchrun, one process per GPU) the module is wrapped in
    DistributedDataParallel: every rank holds the same (all-gathered) examples, trains on its 1/world slice of each
    epoch's permutation with batch_size // world samples per step, gradients are all-reduced (RCCL), and rank 0
    writes the checkpoints.  The global batch, its order and the optimiser schedule are those of the single-process run and
    the averaged gradient is the full-batch gradient of the LOSS; BatchNorm, however, normalises with the statistics of each
    rank's own batch_size // world samples (plain DDP, no SyncBatchNorm), so activations -- and therefore weights -- are
    close to, not bit-equal to, the single-process run."""

    def __init__(self, game, model_dir="models", lr=0.001, batch_size=64, weight_decay=1e-4, device=None,
                 num_channels=128, num_res_blocks=10, graph_step=None):
        self.game, self.model_dir, self.batch_size = game, model_dir, batch_size
        os.makedirs(model_dir, exist_ok=True)
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.nnet = YinYangNeuralNetwork(game, num_channels, num_res_blocks).to(self.device)
        # capturable Adam keeps its step counter on the device so the whole step can live in a hipGraph
        on_gpu = self.device.type == "cuda"
        self.optimizer = torch.optim.Adam(self.nnet.parameters(), lr=lr, weight_decay=weight_decay,
                                          **(dict(capturable=True, fused=True) if on_gpu else {}))
        self._ddp = None
        self.graph_step = self.device.type == "cuda" if graph_step is None else bool(graph_step)
        self._graph = None

    def _encode(self, states):
        """int8 [N,R,C] -> planes f32 [N,5,R,C]: HIP encode kernel on a ROCm device, torch ops on CPU."""
        if self.device.type == "cuda":
            from . import engine
            return engine.encode_planes(states.to(self.device).contiguous())
        return encode_planes_host(states)

    def train(self, examples, epochs=10, augment=True, permutations=None, log_batches=False):
        """trainer.py:67-161.  examples: dict of tensors (or the reference's list of tuples).
        permutations: optional list (one int64 index tensor per epoch) replacing the epoch's random order -- the reference
        shuffles through a DataLoader; a test feeds the order the reference used to compare losses and weights step by step.
        log_batches: keep (policy_loss, value_loss) of every step in self.batch_log (one host read per step)."""
        self.batch_log = []
        if not isinstance(examples, dict):
            q = TrainingDataQueue(max_size=max(1, len(examples)))
            q.push_examples(examples)
            examples = dict(states=q.states, policies=q.policies, values=q.values)
        planes = self._encode(examples["states"])
        pol = examples["policies"].to(self.device).float()
        val = examples["values"].to(self.device).float()
        if augment:
            planes, pol = augment_batch(planes, pol)
            val = val.repeat(8)
        n = planes.shape[0]
        metrics = {"policy_loss": [], "value_loss": [], "total_loss": []}
        dist = _dist()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        model = self.nnet
        if dist:
            if self._ddp is None:
                from torch.nn.parallel import DistributedDataParallel as DDP
                self._ddp = DDP(self.nnet, device_ids=[self.device.index] if self.device.type == "cuda" else None)
            model = self._ddp
        self.nnet.train()
        graphed = self.graph_step and dist is None and self.device.type == "cuda" and n >= self.batch_size
        for epoch in range(epochs):
            perm = (torch.randperm(n, device=self.device) if permutations is None
                    else torch.as_tensor(permutations[epoch], dtype=torch.int64, device=self.device))
            if dist:                                  # the same permutation everywhere, then a disjoint slice per rank
                dist.broadcast(perm, src=0)
            sums = torch.zeros(4, device=self.device)
            for i in range(0, n, self.batch_size):
                gidx = perm[i:i + self.batch_size]
                if graphed and gidx.numel() == self.batch_size:
                    out = self._graphed_step(planes, pol, val, gidx)
                    if log_batches:
                        self.batch_log.append((float(out[0]), float(out[1])))
                    sums += out * self.batch_size
                    continue
                idx = gidx[rank::world]
                self.optimizer.zero_grad(set_to_none=True)
                if idx.numel() == 0:                  # ragged tail: contribute a zero gradient, keep the collective in step
                    self.nnet.eval()                  # a one-sample dummy batch must not touch the BatchNorm running statistics
                    logits, v = model(planes[gidx[:1]])
                    self.nnet.train()
                    loss = (logits.sum() + v.sum()) * 0.0
                    loss.backward()
                    self.optimizer.step()
                    continue
                logits, v = model(planes[idx])
                p_loss = F.cross_entropy(logits, pol[idx])               # soft targets, trainer.py:130
                v_loss = F.mse_loss(v.reshape(-1), val[idx])             # :131
                loss = p_loss + v_loss
                # DDP averages gradients over ranks; weight by this rank's share so the step equals the full-batch mean
                (loss * (idx.numel() * world / gidx.numel()) if dist else loss).backward()
                self.optimizer.step()
                if log_batches:
                    self.batch_log.append((float(p_loss.detach()), float(v_loss.detach())))
                sums += torch.stack([p_loss.detach(), v_loss.detach(), loss.detach(), torch.ones((), device=self.device)]) * idx.numel()
            if dist:
                dist.all_reduce(sums)
            pl, vl, tl = (sums[:3] / sums[3].clamp_min(1)).tolist()
            metrics["policy_loss"].append(pl)
            metrics["value_loss"].append(vl)
            metrics["total_loss"].append(tl)
        self.nnet.eval()
        return metrics

    def _graphed_step(self, planes, pol, val, gidx):
        """One full-batch optimiser step (forward, losses, backward, Adam) replayed from a hipGraph: the 128 x 10 network at
        batch 64 is launch-bound (a few hundred small kernels per step), so the step is captured once into static batch
        buffers and replayed.  Single-process CUDA only; ragged tail batches and DDP take the eager path.
        Returns [policy_loss, value_loss, total_loss, 1] of the step (device tensor)."""
        B = self.batch_size
        if self._graph is None or self._graph["x"].shape[1:] != planes.shape[1:]:
            st = dict(x=torch.empty((B,) + tuple(planes.shape[1:]), device=self.device),
                      p=torch.empty((B, pol.shape[1]), device=self.device), v=torch.empty(B, device=self.device),
                      out=torch.zeros(4, device=self.device))

            def step():
                logits, v = self.nnet(st["x"])
                p_loss = F.cross_entropy(logits, st["p"])
                v_loss = F.mse_loss(v.reshape(-1), st["v"])
                loss = p_loss + v_loss
                loss.backward()
                self.optimizer.step()
                return torch.stack([p_loss.detach(), v_loss.detach(), loss.detach(), torch.ones((), device=self.device)])

            for k in ("x", "p", "v"):
                torch.index_select({"x": planes, "p": pol, "v": val}[k], 0, gidx, out=st[k])
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            # MIOpen's exhaustive find picks the convolution algorithms during the eager step; the capture below hits
            # that cache (no find runs inside a capture)
            with torch.backends.cudnn.flags(enabled=True, benchmark=True):
                with torch.cuda.stream(side):                       # this batch's step runs eagerly once (creates the Adam
                    self.optimizer.zero_grad(set_to_none=True)      # state, warms the allocator and MIOpen's find) ...
                    st["out"].copy_(step())
                torch.cuda.current_stream(self.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                self.optimizer.zero_grad(set_to_none=True)
                with torch.cuda.graph(g):                           # ... then the same step is captured (not executed)
                    st["out"].copy_(step())
            st["g"] = g
            self._graph = st
            return st["out"].clone()
        st = self._graph
        torch.index_select(planes, 0, gidx, out=st["x"])
        torch.index_select(pol, 0, gidx, out=st["p"])
        torch.index_select(val, 0, gidx, out=st["v"])
        st["g"].replay()
        return st["out"].clone()

    def _path(self, filename, iteration):
        if filename is None:
            filename = f"checkpoint_{iteration}.pth.tar" if iteration is not None else "checkpoint.pth.tar"
        return os.path.join(self.model_dir, filename)

    def save_checkpoint(self, filename=None, iteration=None):
        dist = _dist()
        if dist is None or dist.get_rank() == 0:
            self.nnet.save_model(self._path(filename, iteration))
        if dist:
            dist.barrier()          # the file exists before any rank goes on to read it

    def load_checkpoint(self, filename=None, iteration=None):
        path = self._path(filename, iteration)
        if os.path.exists(path):
            self.nnet.load_model(path)
            self.nnet.to(self.device)

    def predict(self, board):
        return self.nnet.predict(board)


class TrainingPipeline:
    def __init__(self, game, model_dir="models", data_dir="data", lr=0.001, batch_size=64, weight_decay=1e-4,
                 epochs_per_iteration=10, sample_size=10000, queue_size=500000, checkpoint_interval=10,
                 device=None, num_channels=128, num_res_blocks=10, allow_reference_objects=False):
        self.game, self.model_dir, self.data_dir = game, model_dir, data_dir
        self.allow_reference_objects = bool(allow_reference_objects)      # data files in the reference's pickled-board format
        self.epochs_per_iteration, self.sample_size, self.checkpoint_interval = epochs_per_iteration, sample_size, checkpoint_interval
        for d in (model_dir, data_dir):
            os.makedirs(d, exist_ok=True)
        self.trainer = AlphaZeroTrainer(game, model_dir, lr, batch_size, weight_decay, device, num_channels, num_res_blocks)
        self.data_queue = TrainingDataQueue(queue_size, sample_size)
        self.iteration = 0
        self._load_iteration()

    def _load_iteration(self):          # training_pipeline.py:171-190: resume from the highest checkpoint_<n>
        cps = glob.glob(os.path.join(self.model_dir, "checkpoint_*.pth.tar"))
        its = [int(os.path.basename(c).split("_")[1].split(".")[0]) for c in cps]
        if its:
            self.iteration = max(its)
            self.trainer.load_checkpoint(iteration=self.iteration)

    def load_data(self):
        for f in sorted(glob.glob(os.path.join(self.data_dir, "self_play_data_*.npz"))):
            self.data_queue.push_file(f, self.allow_reference_objects)

    def train_iteration(self):
        ex = self.data_queue.sample()
        if not ex:
            return {}
        metrics = self.trainer.train(ex, epochs=self.epochs_per_iteration, augment=True)
        self.iteration += 1
        if self.iteration % self.checkpoint_interval == 0:
            self.trainer.save_checkpoint(iteration=self.iteration)
        return metrics

    def train(self, num_iterations=10):
        allm = {"policy_loss": [], "value_loss": [], "total_loss": []}
        for _ in range(num_iterations):
            m = self.train_iteration()
            for k in allm:
                allm[k].extend(m.get(k, []))
        return allm

    def get_model_performance(self, model_path=None):
        """training_pipeline.py:280-291 is a stub that returns None; here: policy / value / total loss of the model (the
        trainer's current weights, or the checkpoint at `model_path`) on a sample of the queue, no gradient.  None when the
        queue is empty."""
        ex = self.data_queue.sample()
        if not ex or len(ex["values"]) == 0:
            return None
        net = self.trainer.nnet
        if model_path is not None:
            net = YinYangNeuralNetwork(self.game, net.conv1.out_channels, len(net.res_blocks))
            net.load_model(model_path)
            net = net.to(self.trainer.device)
        was_training = net.training
        net.eval()
        with torch.no_grad():
            planes = self.trainer._encode(ex["states"])
            logits, v = net(planes)
            pl = F.cross_entropy(logits, ex["policies"].to(self.trainer.device).float())
            vl = F.mse_loss(v.reshape(-1), ex["values"].to(self.trainer.device).float())
        net.train(was_training)
        return {"policy_loss": float(pl), "value_loss": float(vl), "total_loss": float(pl + vl), "examples": int(len(ex["values"]))}

    def get_latest_model_path(self):
        name = f"checkpoint_{self.iteration}.pth.tar" if self.iteration > 0 else "checkpoint.pth.tar"
        return os.path.join(self.model_dir, name)


def run_training_pipeline(game, model_dir="models", data_dir="data", num_iterations=10, sample_size=10000,
                          checkpoint_interval=10, **kw):
    """training_pipeline.py:293-329: load every .npz in data_dir, train, return the latest checkpoint path."""
    pipe = TrainingPipeline(game, model_dir, data_dir, sample_size=sample_size, checkpoint_interval=checkpoint_interval, **kw)
    pipe.load_data()
    t0 = time.perf_counter()
    metrics = pipe.train(num_iterations)
    with torch.no_grad():       # fingerprint of THIS rank's weights after training (data-parallel ranks must agree bit for bit)
        ps = [p.detach().double() for p in pipe.trainer.nnet.parameters()]
        checksum = [float(sum(p.sum() for p in ps)), float(sum(p.abs().sum() for p in ps))]
    run_training_pipeline.last = dict(metrics=metrics, seconds=time.perf_counter() - t0, examples=len(pipe.data_queue),
                                      param_checksum=checksum)
    if not os.path.exists(pipe.get_latest_model_path()):
        pipe.trainer.save_checkpoint(iteration=pipe.iteration if pipe.iteration > 0 else None)
    return pipe.get_latest_model_path()
