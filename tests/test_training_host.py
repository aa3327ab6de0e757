"""CPU tests of the training-side widening: 8-fold augmentation against the
reference's DataProcessor.augment_sample (G6 fixture, exact), the replay queue, and the trainer on a
tiny CPU network (loss decreases, reference checkpoint format)."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_augment_batch_matches_reference():
    from yinyang_game_alphazero_amd.training import augment_batch
    z = np.load(os.path.join(GOLDEN, "augment.npz"))
    for R in (6, 8):
        boards, pi = z[f"boards_{R}"], z[f"pi_{R}"]
        planes = torch.from_numpy(O.encode_planes(boards))
        ap, api = augment_batch(planes, torch.from_numpy(pi))
        N = boards.shape[0]
        want_p, want_pi = z[f"aug_planes_{R}"], z[f"aug_pi_{R}"]      # [N, 8, ...]
        for v in range(8):
            assert np.array_equal(ap[v * N:(v + 1) * N].numpy(), want_p[:, v]), (R, v)
            assert np.array_equal(api[v * N:(v + 1) * N].numpy(), want_pi[:, v]), (R, v)


def test_training_data_queue(tmp_path):
    from yinyang_game_alphazero_amd.training import TrainingDataQueue
    q = TrainingDataQueue(max_size=10, sample_size=4)
    assert len(q) == 0 and q.sample() == {}
    ex = dict(states=torch.zeros((7, 4, 4), dtype=torch.int8), policies=torch.full((7, 16), 1 / 16.0),
              values=torch.arange(7, dtype=torch.float32))
    q.push_examples(ex)
    q.push_examples(ex)
    assert len(q) == 10 and q.values.tolist() == [4, 5, 6, 0, 1, 2, 3, 4, 5, 6]     # newest max_size kept
    s = q.sample()
    assert s["states"].shape == (4, 4, 4) and len(set(s["values"].tolist())) <= 4
    path = tmp_path / "self_play_data_1.npz"
    np.savez(path, boards=np.zeros((3, 4, 4), np.int8), states=np.zeros((3, 4, 4), np.int8),
             policies=np.full((3, 16), 1 / 16.0), values=np.ones(3))
    q2 = TrainingDataQueue()
    q2.push_file(str(path))
    q2.push_file(str(tmp_path / "missing.npz"))
    assert len(q2) == 3


def test_trainer_cpu_loss_decreases_and_checkpoints(tmp_path):
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    game = pkg.YinYangGame(4, 4)
    tr = pkg.AlphaZeroTrainer(game, model_dir=str(tmp_path), device="cpu", num_channels=8, num_res_blocks=1, batch_size=16)
    rng = np.random.default_rng(0)
    states = rng.integers(-1, 2, size=(24, 4, 4)).astype(np.int8)
    pol = rng.random((24, 16)).astype(np.float32)
    pol /= pol.sum(1, keepdims=True)
    val = np.sign(states.sum((1, 2))).astype(np.float32)
    ex = dict(states=torch.from_numpy(states), policies=torch.from_numpy(pol), values=torch.from_numpy(val))
    m = tr.train(ex, epochs=6, augment=True)
    assert len(m["total_loss"]) == 6 and m["total_loss"][-1] < m["total_loss"][0]
    assert abs(m["total_loss"][0] - (m["policy_loss"][0] + m["value_loss"][0])) < 1e-5
    tr.save_checkpoint(iteration=3)
    ck = torch.load(str(tmp_path / "checkpoint_3.pth.tar"), map_location="cpu", weights_only=True)
    assert set(ck) == {"state_dict", "board_size", "action_size"}
    # pipeline resumes from the highest checkpoint_<n> (training_pipeline.py:171-190)
    np.savez(tmp_path / "self_play_data_7.npz", boards=states, states=states, policies=pol.astype(np.float64), values=val.astype(np.float64))
    pipe = pkg.TrainingPipeline(game, model_dir=str(tmp_path), data_dir=str(tmp_path), device="cpu", num_channels=8,
                                num_res_blocks=1, epochs_per_iteration=1, checkpoint_interval=1)
    assert pipe.iteration == 3
    pipe.load_data()
    assert len(pipe.data_queue) == 24
    pipe.train_iteration()
    assert os.path.exists(tmp_path / "checkpoint_4.pth.tar")


DDP_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
import yinyang_game_alphazero_amd as pkg
r = dist.get_rank()
torch.manual_seed(0)                      # same initial weights on every rank
game = pkg.YinYangGame(4, 4)
tr = pkg.AlphaZeroTrainer(game, model_dir=sys.argv[2], device="cpu", num_channels=8, num_res_blocks=1, batch_size=16)
rng = np.random.default_rng(0)            # the already-gathered examples: identical on every rank
states = rng.integers(-1, 2, size=(21, 4, 4)).astype(np.int8)
pol = rng.random((21, 16)).astype(np.float32); pol /= pol.sum(1, keepdims=True)
val = np.sign(states.sum((1, 2))).astype(np.float32)
ex = dict(states=torch.from_numpy(states), policies=torch.from_numpy(pol), values=torch.from_numpy(val))
torch.manual_seed(100 + r)                # rank-local RNG differs: the permutation must come from rank 0
m = tr.train(ex, epochs=3, augment=True)
flat = torch.cat([p.detach().reshape(-1) for p in tr.nnet.parameters()])
ref = flat.clone(); dist.broadcast(ref, src=0)
assert torch.equal(flat, ref), "parameters diverged across ranks"
assert m["total_loss"][-1] < m["total_loss"][0]
tr.save_checkpoint(iteration=1)
assert os.path.exists(os.path.join(sys.argv[2], "checkpoint_1.pth.tar"))
q = pkg.TrainingDataQueue(sample_size=5); q.push_examples(ex)
s = q.sample(); ids = s["values"].clone(); ref2 = ids.clone(); dist.broadcast(ref2, src=0)
assert torch.equal(ids, ref2), "ranks sampled different examples"
sys.stdout.write("rank %d ok %s\n" % (r, [round(x, 4) for x in m["total_loss"]])); sys.stdout.flush()
dist.destroy_process_group()
'''


def test_trainer_ddp_gloo_world2(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ddp.py"
    script.write_text(DDP_SCRIPT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29573", str(script), root, str(tmp_path / "m")],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import re
    losses = re.findall(r"ok (\[[^\]]*\])", r.stdout)
    assert len(losses) == 2 and losses[0] == losses[1], r.stdout[-500:]      # same global loss on both ranks


def test_reference_format_writer_round_trip(tmp_path):
    """save_examples_reference_format: the pickle stream must name the reference's class by its module path (so the
    reference rebuilds ITS OWN board objects) and nothing of this package; a stand-in module with that path is enough to load
    the file the way the reference does (training_pipeline.py:67-73) and to call get_board() on every element; the plain
    keys stay readable without unpickling."""
    import pickletools
    import sys
    import types
    import zipfile
    from yinyang_game_alphazero_amd.training import load_examples, save_examples_reference_format
    rng = np.random.default_rng(0)
    states = rng.integers(-1, 2, size=(7, 6, 6)).astype(np.int8)
    pol = rng.dirichlet(np.ones(36), size=7)
    val = rng.choice([-1.0, 1.0, 1e-4], size=7)
    path = save_examples_reference_format(str(tmp_path / "self_play_data_1.npz"), states, pol, val)
    assert "src.yin_yang.yin_yang_logic" not in sys.modules                   # the shim module is gone again
    raw = zipfile.ZipFile(path).read("boards.npy")
    names = {arg for op, arg, _ in pickletools.genops(raw[raw.index(b"\x80"):]) if op.name in ("GLOBAL", "STACK_GLOBAL", "SHORT_BINUNICODE", "BINUNICODE")}
    assert "src.yin_yang.yin_yang_logic" in names and "YinYangLogic" in names
    assert not any("yinyang_game_alphazero_amd" in str(n) or "yinyang-game" in str(n) for n in names)
    ex = load_examples(path)                                                  # this package: plain keys, allow_pickle=False
    assert np.array_equal(ex["states"].numpy(), states) and np.allclose(ex["policies"].numpy(), pol.astype(np.float32))

    class YinYangLogic:                                                        # what the reference's module provides
        def get_board(self):
            return self.board.copy()

    for n in ("src", "src.yin_yang", "src.yin_yang.yin_yang_logic"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["src.yin_yang.yin_yang_logic"].YinYangLogic = YinYangLogic
    try:
        data = np.load(path, allow_pickle=True)                               # OUR OWN file, written two lines above
        boards, policies, values = data["boards"], data["policies"], data["values"]
        examples = [(boards[i], policies[i], values[i]) for i in range(len(boards))]
    finally:
        for n in ("src", "src.yin_yang", "src.yin_yang.yin_yang_logic"):
            del sys.modules[n]
    assert len(examples) == 7 and all(isinstance(b, YinYangLogic) for b, _, _ in examples)
    for i, (b, p, v) in enumerate(examples):
        assert np.array_equal(b.get_board(), states[i]) and (b.n, b.m) == (6, 6) and b.board.dtype == np.int8
        assert np.array_equal(p, pol[i]) and v == val[i]


def test_pipeline_model_performance(tmp_path):
    """TrainingPipeline.get_model_performance (a None-returning stub in the reference): losses on a queue sample, CPU."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(4, 4)
    pipe = pkg.TrainingPipeline(game, model_dir=str(tmp_path / "m"), data_dir=str(tmp_path / "d"), device="cpu",
                                num_channels=8, num_res_blocks=1, sample_size=16)
    assert pipe.get_model_performance() is None                         # empty queue
    rng = np.random.default_rng(0)
    ex = [(rng.integers(-1, 2, size=(4, 4)).astype(np.int8), rng.dirichlet(np.ones(16)), float(rng.choice([-1, 1]))) for _ in range(20)]
    pipe.data_queue.push_examples(ex)
    m = pipe.get_model_performance()
    assert set(m) == {"policy_loss", "value_loss", "total_loss", "examples"} and m["examples"] == 16
    assert abs(m["total_loss"] - m["policy_loss"] - m["value_loss"]) < 1e-5 and m["policy_loss"] > 0
    pipe.trainer.save_checkpoint("probe.pth.tar")
    m2 = pipe.get_model_performance(str(tmp_path / "m" / "probe.pth.tar"))
    assert m2["examples"] == 16 and m2["total_loss"] > 0


def test_data_processor_api_matches_reference_golden():
    """data_utils.DataProcessor / create_dataset_from_games with the reference's signatures, against the G6 fixture (the
    reference's augment_sample outputs, exact) and the oracle's plane encoding."""
    import yinyang_game_alphazero_amd as pkg
    z = np.load(os.path.join(GOLDEN, "augment.npz"))
    game = pkg.YinYangGame(6, 6)
    proc = pkg.DataProcessor(game)
    assert proc.board_size == (6, 6)
    boards, pi = z["boards_6"], z["pi_6"]
    want_p, want_pi = z["aug_planes_6"], z["aug_pi_6"]
    for n in range(3):
        lb = pkg.YinYangLogic(6, 6)
        lb.board = boards[n].copy()
        bt, pt = proc.preprocess_sample(lb, pi[n], 1)
        assert np.array_equal(bt.numpy(), O.encode_planes(boards[n:n + 1])[0]) and pt.dtype == torch.float32
        aug = proc.augment_sample(bt, pt)
        assert len(aug) == 8
        for v, (ab, ap) in enumerate(aug):
            assert np.array_equal(ab.numpy(), want_p[n, v]) and np.array_equal(ap.numpy(), want_pi[n, v].astype(np.float32))
    g = proc._policy_to_grid(pi[0])
    assert g.shape == (6, 6) and np.array_equal(proc._grid_to_policy(g).numpy(), pi[0].astype(np.float32))
    data = [(boards[n], pi[n], float(n % 2 * 2 - 1)) for n in range(4)]
    b, p, v = pkg.create_dataset_from_games(data, game, augment=True)
    assert len(b) == len(p) == len(v) == 32 and np.array_equal(b[8 + 3].numpy(), want_p[1, 3]) and float(v[8 + 3]) == 1.0
    b, p, v = pkg.create_dataset_from_games(data, game, augment=False)
    assert len(b) == 4 and np.array_equal(p[2].numpy(), pi[2].astype(np.float32))


def test_reference_format_reader_is_restricted(tmp_path):
    """Reading direction: a file in the reference's pickled-board layout (written here by our own compat writer, WITHOUT the
    plain `states` key) is refused by default, read through the restricted unpickler on request, and a stream naming any
    other global is rejected before anything runs."""
    import io
    import pickle
    import zipfile
    from yinyang_game_alphazero_amd.training import TrainingDataQueue, load_examples, save_examples_reference_format
    rng = np.random.default_rng(3)
    states = rng.integers(-1, 2, size=(9, 5, 5)).astype(np.int8)
    pol, val = rng.dirichlet(np.ones(25), size=9), rng.choice([-1.0, 1.0], size=9)
    full = save_examples_reference_format(str(tmp_path / "full.npz"), states, pol, val)
    ref_like = str(tmp_path / "ref_like.npz")              # what the reference itself writes: boards / policies / values only
    with zipfile.ZipFile(full) as zi, zipfile.ZipFile(ref_like, "w") as zo:
        for name in ("boards.npy", "policies.npy", "values.npy"):
            zo.writestr(name, zi.read(name))
    with pytest.raises(ValueError, match="restricted unpickler"):
        load_examples(ref_like)
    ex = load_examples(ref_like, allow_reference_objects=True)
    assert np.array_equal(ex["states"].numpy(), states) and np.allclose(ex["policies"].numpy(), pol.astype(np.float32))
    q = TrainingDataQueue(max_size=100, sample_size=4)
    q.push_file(ref_like, allow_reference_objects=True)
    assert len(q) == 9
    # a hostile stream: the object array's pickle names os.system -> refused, nothing executed
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    arr = np.empty(1, dtype=object)
    arr[0] = Evil()
    buf = io.BytesIO()
    np.lib.format.write_array(buf, arr, allow_pickle=True)
    hostile = str(tmp_path / "hostile.npz")
    with zipfile.ZipFile(hostile, "w") as zo, zipfile.ZipFile(full) as zi:
        zo.writestr("boards.npy", buf.getvalue())
        zo.writestr("policies.npy", zi.read("policies.npy"))
        zo.writestr("values.npy", zi.read("values.npy"))
    with pytest.raises(pickle.UnpicklingError, match="refusing global"):
        load_examples(hostile, allow_reference_objects=True)
    assert not marker.exists()


def _trainer_fixture_run(device, graph_step=None):
    """Our AlphaZeroTrainer driven exactly like the reference run that made tests/golden/trainer.npz: same seed -> same
    initial 16x1 network, same 44 examples, the batch order the reference's DataLoader used, 2 epochs, batch 8, no
    augmentation.  Returns (fixture, per-step (policy, value) losses, state_dict, epoch metrics)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    z = np.load(os.path.join(GOLDEN, "trainer.npz"))
    game = pkg.YinYangGame(6, 6)
    torch.manual_seed(5)
    tr = pkg.AlphaZeroTrainer(game, model_dir=os.path.join(GOLDEN, "..", "..", "gpurun_out", "_trainer_tmp"), lr=0.001,
                              batch_size=8, device=device, num_channels=16, num_res_blocks=1, graph_step=graph_step)
    for k, v in tr.nnet.state_dict().items():          # same seed, same module order -> the reference's initial weights
        assert np.array_equal(v.cpu().numpy(), z["init/" + k]), k
    ex = dict(states=torch.from_numpy(z["boards"]), policies=torch.from_numpy(z["policies"]).float(),
              values=torch.from_numpy(z["values"]).float())
    perms = [torch.from_numpy(np.concatenate([r[r >= 0] for r in z["order"][e]]).astype(np.int64)) for e in range(2)]
    m = tr.train(ex, epochs=2, augment=False, permutations=perms, log_batches=True)
    return z, np.asarray(tr.batch_log, np.float64), {k: v.detach().cpu().numpy() for k, v in tr.nnet.state_dict().items()}, m


def test_trainer_steps_equal_reference_trainer_cpu():
    """AlphaZeroTrainer on the CPU against the reference's trainer (trainer.py:67-161; fixture made by make_golden.py
    gen_trainer): per-step CE(soft targets) and MSE losses, the epoch means and every parameter and BatchNorm buffer after the
    12 Adam steps.  Same torch ops in the same order on the same device type: tolerance 1e-6 relative (exact in practice)."""
    import torch
    torch.set_num_threads(1)
    z, log, final, m = _trainer_fixture_run("cpu")
    assert log.shape == (12, 2)
    assert np.allclose(log[:, 0], z["policy_loss"], rtol=1e-6, atol=0) and np.allclose(log[:, 1], z["value_loss"], rtol=1e-6, atol=1e-9)
    assert np.allclose(m["policy_loss"], z["epoch_policy_loss"], rtol=1e-6) and np.allclose(m["total_loss"], z["epoch_total_loss"], rtol=1e-6)
    worst = 0.0
    for k, v in final.items():
        want = z["final/" + k]
        assert v.shape == want.shape, k
        worst = max(worst, float(np.abs(v.astype(np.float64) - want.astype(np.float64)).max()))
        assert np.allclose(v, want, rtol=1e-5, atol=1e-7), k
    print("trainer vs reference (CPU): max |dloss| %.3e, max |dparam| %.3e" % (np.abs(log[:, 0] - z["policy_loss"]).max(), worst))
