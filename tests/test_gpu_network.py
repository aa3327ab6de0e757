"""GPU: the evaluator fast path.  Floating point, so tolerance-based (written in each test):
the fused HIP epilogue against plain torch ops (bf16 rounding: 1 ulp of bf16 = 2^-8 relative), and
the bf16 tower against the fp32 module (policy abs 2e-2, value abs 5e-2: random-init 128x10 net)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bias_act_matches_torch():
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    for (N, C, H, W) in ((37, 128, 8, 8), (5, 32, 12, 12), (3, 8, 1, 1)):
        x = torch.randn(N, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        r = torch.randn(N, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device="cuda")
        for res in (None, r):
            for relu in (True, False):
                want = x.float() + b.reshape(1, C, 1, 1) + (0 if res is None else res.float())
                if relu:
                    want = want.clamp_min(0)
                got = pkg.engine.bias_act_(x.clone(memory_format=torch.channels_last), b, res, relu).float()
                # one bf16 rounding of the exact f32 result
                assert torch.equal(got, want.to(torch.bfloat16).float())


def test_bf16_tower_close_to_fp32_module():
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    game = pkg.YinYangGame(8, 8)
    net = pkg.YinYangNeuralNetwork(game).cuda().eval()
    rng = np.random.default_rng(1)
    boards = torch.from_numpy(rng.integers(-1, 2, size=(64, 8, 8)).astype(np.int8)).cuda()
    planes = pkg.engine.encode_planes(boards)
    p32, v32 = pkg.BatchedEvaluator(net, "fp32")(planes)
    for fused in (True, False):
        p16, v16 = pkg.BatchedEvaluator(net, "bf16", fused_epilogue=fused)(planes)
        assert p16.dtype == torch.float32 and v16.shape == (64,)
        assert float((p16 - p32).abs().max()) < 2e-2
        assert float((v16 - v32).abs().max()) < 5e-2
        assert torch.allclose(p16.sum(1), torch.ones(64, device="cuda"), atol=1e-4)


def test_fp32_module_matches_cpu_module():
    """same nn.Module on the GPU and on the CPU: policy/value within 1e-4 abs (torch conv/BN)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    game = pkg.YinYangGame(6, 6)
    net = pkg.YinYangNeuralNetwork(game, 32, 2).eval()
    rng = np.random.default_rng(2)
    b = rng.integers(-1, 2, size=(16, 6, 6)).astype(np.int8)
    import oracle_lib as O
    planes = torch.from_numpy(O.encode_planes(b))
    pc, vc = net.predict_batch(planes)
    pg, vg = net.cuda().predict_batch(planes.cuda())
    assert float((pg.cpu() - pc).abs().max()) < 1e-4 and float((vg.cpu() - vc).abs().max()) < 1e-4


def test_tower_kernel_matches_torch_bf16_path():
    """LDS-resident MFMA tower (csrc/yy_tower.hip) vs the same folded bf16 weights run through torch
    convolutions + the fused epilogue.  Both round activations to bf16 after every layer and accumulate
    in f32; they differ only in summation order, so: activations within 2 bf16 ulps relative (2^-7) of
    the layer-output scale, final policy within 2e-2 abs, value within 5e-2 abs (the same bounds as bf16 vs fp32:
    two bf16 evaluations of a 21-layer net differ by rounding noise of that size)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(0)
    game = pkg.YinYangGame(8, 8)
    rng = np.random.default_rng(3)
    # G picks the kernel: <= 256 one board per workgroup, <= 512 two (yy_towerq.hip), above that four (yy_tower.hip)
    for blocks, G in ((1, 7), (3, 300), (10, 130), (10, 1030)):
        net = pkg.YinYangNeuralNetwork(game, 128, blocks).cuda().eval()
        # non-trivial BatchNorm statistics and biases so that folding and the bias path are exercised
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.normal_(0, 0.1)
                if isinstance(m, torch.nn.Conv2d):
                    m.bias.normal_(0, 0.05)
        boards = torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda()
        planes = pkg.engine.encode_planes(boards)
        ref = pkg.BatchedEvaluator(net, "bf16", tower=False)
        tow = pkg.BatchedEvaluator(net, "bf16", tower=True, fused_heads=False)
        towh = pkg.BatchedEvaluator(net, "bf16", tower=True, fused_heads=True)
        assert tow.tower and not ref.tower and towh.fused_heads
        # tower activations
        x_t = pkg.engine.tower_forward(planes, tow.tower_w, tow.tower_b, tow.tower_layers).float()
        x = planes.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x = ref._conv(x, ref.stem, 1)
        for (c1, c2) in ref.blocks:
            y = ref._conv(x, c1, 1)
            x = ref._conv(y, c2, 1, residual=x)
        x_r = x.float()
        scale = float(x_r.abs().max())
        err = float((x_t - x_r).abs().max())
        assert err <= scale * 2.0 ** -7 * (1 + blocks), (blocks, err, scale)
        p_t, v_t = tow(planes)
        p_r, v_r = ref(planes)
        assert float((p_t - p_r).abs().max()) < 2e-2 and float((v_t - v_r).abs().max()) < 5e-2
        p32, v32 = pkg.BatchedEvaluator(net, "fp32")(planes)
        assert float((p_t - p32).abs().max()) < 2e-2 and float((v_t - v32).abs().max()) < 5e-2
        # fused 1x1 head convolutions: features against torch ops on the tower kernel's own output (same
        # input, so only summation order differs: 2 bf16 ulps at the feature scale), then end to end
        feats = pkg.engine.tower_heads_forward(planes, towh.towerh_w, towh.towerh_b, towh.tower_layers).float()
        xt = pkg.engine.tower_forward(planes, tow.tower_w, tow.tower_b, tow.tower_layers)
        pf = tow._conv(xt, tow.phead, 0).contiguous().flatten(1).float()
        vf = tow._conv(xt, tow.vhead, 0).contiguous().flatten(1).float()
        fs = max(float(pf.abs().max()), float(vf.abs().max()))
        ulp = 2.0 ** (int(np.floor(np.log2(fs))) - 7)      # one bf16 ulp at the feature scale; bound = 2 ulps
        assert float((feats[:, 0] - pf).abs().max()) <= 2 * ulp and float((feats[:, 1] - vf).abs().max()) <= 2 * ulp
        # head finish kernel against torch ops on the same GEMM output
        hc = torch.nn.functional.linear(pkg.engine.tower_heads_forward(planes, towh.towerh_w, towh.towerh_b, towh.tower_layers).view(G, -1),
                                        towh.fc_cat_w, towh.fc_cat_b)
        pk, vk = pkg.engine.head_finish(hc, 64, towh.fc2_w, towh.fc2_b)
        pt = torch.softmax(hc[:, :64].float(), 1)
        vt = torch.tanh(torch.relu(hc[:, 64:].float()) @ towh.fc2_w + towh.fc2_b)
        assert float((pk - pt).abs().max()) < 1e-6 and float((vk - vt).abs().max()) < 1e-5
        p_h, v_h = towh(planes)
        assert float((p_h - p_t).abs().max()) < 2e-2 and float((v_h - v_t).abs().max()) < 5e-2


def test_tower_small_batch_kernels_bit_identical_to_main():
    """8x8: the 1- and 2-boards-per-workgroup kernels (small batches) and the 4-boards-per-workgroup kernel run the same
    accumulation order and epilogue, so a board's output must not depend on the batch it is evaluated in: exact."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(1)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8), 128, 10).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "bf16")
    rng = np.random.default_rng(11)
    planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(1100, 8, 8)).astype(np.int8)).cuda())
    full_h = pkg.engine.tower_heads_forward(planes, ev.towerh_w, ev.towerh_b, ev.tower_layers)       # 4 boards / workgroup
    full_x = pkg.engine.tower_forward(planes, ev.tower_w, ev.tower_b, ev.tower_layers)
    assert float(full_h.float().abs().max()) > 0
    for g in (1, 3, 64, 255, 256, 257, 301, 512):                                                   # 1 and 2 boards / workgroup
        sub = planes[:g].contiguous()
        h = pkg.engine.tower_heads_forward(sub, ev.towerh_w, ev.towerh_b, ev.tower_layers)
        x = pkg.engine.tower_forward(sub, ev.tower_w, ev.tower_b, ev.tower_layers)
        assert torch.equal(h.view(torch.int16), full_h[:g].view(torch.int16)), g
        assert torch.equal(x.view(torch.int16), full_x[:g].view(torch.int16)), g


@pytest.mark.parametrize("R", [12, 6])
def test_tower_other_sizes_match_torch_bf16_path(R):
    """12x12 and 6x6 (csrc/yy_towerq.hip <12,2> and <6,8>) variants: same checks as the 8x8 kernel -- tower activations within 2 bf16 ulps of
    scale per layer against torch bf16 convolutions on the same folded weights, fused head features within 2 ulps,
    end-to-end policy 2e-2 / value 5e-2 abs against the torch bf16 path and the fp32 module."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(1)
    game = pkg.YinYangGame(R, R)
    rng = np.random.default_rng(4)
    for blocks, G in ((1, 5), (10, 67)):
        net = pkg.YinYangNeuralNetwork(game, 128, blocks).cuda().eval()
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.normal_(0, 0.1)
                if isinstance(m, torch.nn.Conv2d):
                    m.bias.normal_(0, 0.05)
        boards = torch.from_numpy(rng.integers(-1, 2, size=(G, R, R)).astype(np.int8)).cuda()
        planes = pkg.engine.encode_planes(boards)
        ref = pkg.BatchedEvaluator(net, "bf16", tower=False)
        tow = pkg.BatchedEvaluator(net, "bf16", tower=True, fused_heads=False)
        towh = pkg.BatchedEvaluator(net, "bf16", tower=True, fused_heads=True)
        assert tow.tower and towh.fused_heads and not ref.tower
        x_t = pkg.engine.tower_forward(planes, tow.tower_w, tow.tower_b, tow.tower_layers)
        x = planes.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x = ref._conv(x, ref.stem, 1)
        for (c1, c2) in ref.blocks:
            y = ref._conv(x, c1, 1)
            x = ref._conv(y, c2, 1, residual=x)
        scale = float(x.float().abs().max())
        assert float((x_t.float() - x.float()).abs().max()) <= scale * 2.0 ** -7 * (1 + blocks)
        feats = pkg.engine.tower_heads_forward(planes, towh.towerh_w, towh.towerh_b, towh.tower_layers).float()
        pf = tow._conv(x_t, tow.phead, 0).contiguous().flatten(1).float()
        vf = tow._conv(x_t, tow.vhead, 0).contiguous().flatten(1).float()
        fs = max(float(pf.abs().max()), float(vf.abs().max()))
        ulp = 2.0 ** (int(np.floor(np.log2(fs))) - 7)      # one bf16 ulp at the feature scale; bound = 2 ulps
        assert float((feats[:, 0] - pf).abs().max()) <= 2 * ulp and float((feats[:, 1] - vf).abs().max()) <= 2 * ulp
        p_r, v_r = ref(planes)
        p32, v32 = pkg.BatchedEvaluator(net, "fp32")(planes)
        for ev in (tow, towh):
            p, v = ev(planes)
            assert p.shape == (G, R * R) and float((p - p_r).abs().max()) < 2e-2 and float((v - v_r).abs().max()) < 5e-2
            assert float((p - p32).abs().max()) < 2e-2 and float((v - v32).abs().max()) < 5e-2


def test_f32_tower_kernel_matches_fp32_module():
    """csrc/yy_tower_f32.hip (exact float32 on v_mfma_f32_32x32x2_f32) against the fp32 nn.Module on the same device.
    Same weights (BatchNorm folded on the host in float32), so the two differ only by summation order and the
    folding: activations within 1e-4 relative to the layer scale, policy 1e-5 / value 1e-4 abs (north-star 1e-5 on pi)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.manual_seed(2)
    torch.backends.cudnn.allow_tf32 = False
    game = pkg.YinYangGame(8, 8)
    rng = np.random.default_rng(5)
    for blocks, G in ((1, 3), (10, 70)):
        net = pkg.YinYangNeuralNetwork(game, 128, blocks).cuda().eval()
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.normal_(0, 0.1)
                if isinstance(m, torch.nn.Conv2d):
                    m.bias.normal_(0, 0.05)
        boards = torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda()
        planes = pkg.engine.encode_planes(boards)
        ev = pkg.BatchedEvaluator(net, "fp32t")
        x_t = pkg.engine.tower_forward_f32(planes, ev.f32_w, ev.f32_b, ev.f32_layers)
        with torch.no_grad():
            x = torch.relu(net.bn1(net.conv1(planes)))
            for blk in net.res_blocks:
                x = blk(x)
        scale = float(x.abs().max())
        assert float((x_t - x).abs().max()) <= 1e-4 * scale, (blocks, float((x_t - x).abs().max()), scale)
        p, v = ev(planes)
        p32, v32 = pkg.BatchedEvaluator(net, "fp32")(planes)
        assert float((p - p32).abs().max()) < 1e-5 and float((v - v32).abs().max()) < 1e-4


def test_gpu_evaluators_against_reference_recorded_outputs():
    """The reference's own (board -> policy, value) pairs (search_net_8x8.npz, CPU fp32, seed 0) against the mirror on the
    GPU: fp32 module and the exact-f32 tower within 1e-5 on policy / 1e-4 on value; bf16 tower within 2e-2 / 5e-2."""
    import os
    import torch
    import yinyang_game_alphazero_amd as pkg
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_net_8x8.npz"))
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
    n = int(z["n_rec"][1])
    boards = torch.from_numpy(z["rec_boards"][1, :n]).cuda()
    rp, rv = torch.from_numpy(z["rec_policy"][1, :n]).cuda(), torch.from_numpy(z["rec_value"][1, :n]).cuda()
    planes = pkg.engine.encode_planes(boards)
    for mode, tp, tv in (("fp32", 1e-5, 1e-4), ("fp32t", 1e-5, 1e-4), ("f16x3", 1e-5, 1e-5), ("f16x3r", 1e-5, 1e-5), ("bf16", 2e-2, 5e-2)):
        p, v = pkg.BatchedEvaluator(net, mode)(planes)
        ep, ev_ = float((p - rp).abs().max()), float((v - rv).abs().max())
        print("%s vs the reference's recorded outputs: policy %.3e value %.3e" % (mode, ep, ev_))
        assert ep < tp and ev_ < tv, mode


def _randomized_net(pkg, game, blocks, seed, channels=128):
    import torch
    torch.manual_seed(seed)
    net = pkg.YinYangNeuralNetwork(game, channels, blocks).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.7, 1.3)
                m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.Conv2d):
                m.bias.normal_(0, 0.05)
    return net


def _module_f64(net, planes):
    """The module in float64 on the device (native torch convolutions): the reference point both the float32 module and
    the split-f16 kernel are measured against."""
    import copy
    import torch
    n64 = copy.deepcopy(net).double().eval()
    with torch.no_grad():
        x = torch.relu(n64.bn1(n64.conv1(planes.double())))
        for blk in n64.res_blocks:
            x = blk(x)
        logits, value = n64(planes.double())
        return x, torch.softmax(logits, 1), value.reshape(-1)


SPLIT_F16_SHAPES = [(8, 8, 128), (6, 6, 128), (12, 12, 128), (10, 10, 128), (5, 7, 96), (9, 12, 32), (7, 7, 64), (3, 3, 128), (1, 6, 64)]


@pytest.mark.parametrize("R,C,ch", SPLIT_F16_SHAPES)
def test_split_f16_tower_kernel_is_float32_grade(R, C, ch):
    """csrc/yy_tower_g.hip (activations and weights as hi + lo float16 pairs, three f16 MFMAs per product term, two f32
    accumulators) against the module evaluated in FLOAT64: 22 significant bits per operand, so the error must be that of
    float32 arithmetic itself.  Square and non-square boards, every supported width (train_alphazero.py:35-36,
    neural_network.py:39).  Bounds: tower activations within 2e-6 of the layer scale (measured ~5e-7; the float32 module
    measures ~3e-7 on the same inputs), policy and value within 2e-6 abs (north-star: 1e-5)."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    torch.backends.cudnn.allow_tf32 = False
    game = pkg.YinYangGame(R, C)
    rng = np.random.default_rng(6)
    for blocks, G in ((1, 3), (10 if ch == 128 else 4, 71 if R * C <= 64 else 23)):
        net = _randomized_net(pkg, game, blocks, 2, ch)
        boards = torch.from_numpy(rng.integers(-1, 2, size=(G, R, C)).astype(np.int8)).cuda()
        planes = pkg.engine.encode_planes(boards)
        ev = pkg.BatchedEvaluator(net)
        assert ev.mode == "f16x3"                      # what --nn auto picks for every one of these shapes
        x_t = pkg.engine.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, ev.g_big[0], ev.g_big[1])
        x64, p64, v64 = _module_f64(net, planes)
        scale = float(x64.abs().max())
        err = float((x_t.double() - x64).abs().max())
        with torch.no_grad():
            x32 = torch.relu(net.bn1(net.conv1(planes)))
            for blk in net.res_blocks:
                x32 = blk(x32)
        err32 = float((x32.double() - x64).abs().max())
        print("f16x3 tower %dx%d c%d: blocks %d max|err| %.3e (fp32 module %.3e) scale %.3e" % (R, C, ch, blocks, err, err32, scale))
        assert err <= 2e-6 * scale, (blocks, err, scale)
        p, v = ev(planes)
        ep, evv = float((p.double() - p64).abs().max()), float((v.double() - v64).abs().max())
        p32, v32 = pkg.BatchedEvaluator(net, "fp32")(planes)
        print("f16x3 evaluator: policy err %.3e value err %.3e (fp32 module: %.3e / %.3e)" % (
            ep, evv, float((p32.double() - p64).abs().max()), float((v32.double() - v64).abs().max())))
        assert ep < 2e-6 and evv < 2e-6
        assert torch.allclose(p.sum(1), torch.ones(G, device="cuda"), atol=1e-5)


def test_fc_heads_kernel_is_float32_grade_and_skips_dead_rows():
    """csrc/yy_fc_heads.hip (policy_fc + value_fc1 as one split-f16 GEMM kernel) against float64 matmuls of the same weights:
    within 2e-6 of the output scale; rows past the device-side count are neither read (NaN there must not spread) nor written."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    E = pkg.engine
    for (R, C) in ((8, 8), (6, 6), (12, 12), (5, 7)):
        torch.manual_seed(3)
        net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, C), 32, 1).cuda().eval()
        with torch.no_grad():
            net.policy_fc.bias.normal_(0, 0.1)
            net.value_fc1.bias.normal_(0, 0.1)
        ev = pkg.BatchedEvaluator(net, "f16x3")
        K, A, H = 32 * R * C, R * C, 256
        G, live = 200, 131
        feats = torch.rand((G, 2, K), device="cuda") * 3.0
        feats[feats < 1.2] = 0.0                                            # ReLU outputs: many exact zeros
        feats[live:] = float("nan")
        n = torch.tensor([live], dtype=torch.int32, device="cuda")
        logits = torch.full((G, A), -5.0, device="cuda")
        hidden = torch.full((G, H), -5.0, device="cuda")
        E.fc_heads(feats, ev.fc_w, ev.fc_b, ev.fc_jobs, A, H, ev.fc_exps, n, logits, hidden)
        f64 = feats[:live].double()
        want_l = f64[:, 0] @ net.policy_fc.weight.double().t() + net.policy_fc.bias.double()
        want_h = f64[:, 1] @ net.value_fc1.weight.double().t() + net.value_fc1.bias.double()
        el = float((logits[:live].double() - want_l).abs().max()) / float(want_l.abs().max())
        eh = float((hidden[:live].double() - want_h).abs().max()) / float(want_h.abs().max())
        print("fc heads %dx%d: logits err %.2e hidden err %.2e of scale" % (R, C, el, eh))
        assert el < 2e-6 and eh < 2e-6
        assert bool((logits[live:] == -5.0).all()) and bool((hidden[live:] == -5.0).all())


@pytest.mark.parametrize("R,C,ch", [(8, 8, 128), (6, 6, 128), (12, 12, 128), (5, 7, 96), (10, 10, 64)])
def test_split_f16_evaluator_row_compaction_is_bit_exact(R, C, ch):
    """evaluator(planes, needs_eval): the flagged rows hold exactly the bits of the full evaluation (a board's output does
    not depend on the workgroup / row it is evaluated in), the other rows are zero; every pattern incl. none and all."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    game = pkg.YinYangGame(R, C)
    net = _randomized_net(pkg, game, 2, 3, ch)
    ev = pkg.BatchedEvaluator(net, "f16x3")
    rng = np.random.default_rng(9)
    for G in (1, 2, 5, 64, 333):
        planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, C)).astype(np.int8)).cuda())
        p_all, v_all = ev(planes)
        for frac in (0.0, 0.5, 0.94, 1.0):
            flags = torch.from_numpy((rng.random(G) < frac).astype(np.uint8)).cuda()
            rows, n = pkg.engine.compact_rows(flags)
            idx = flags.nonzero(as_tuple=True)[0]
            assert int(n) == idx.numel() and torch.equal(rows[: int(n)].long(), idx)
            p, v = ev(planes, needs_eval=flags)
            keep = flags.bool()
            assert torch.equal(p[keep], p_all[keep]) and torch.equal(v[keep], v_all[keep])
            assert float(p[~keep].abs().sum()) == 0.0 and float(v[~keep].abs().sum()) == 0.0


@pytest.mark.parametrize("R,C,ch", [(8, 8, 128), (6, 6, 128), (12, 12, 128), (5, 7, 96), (4, 4, 32)])
def test_split_f16_every_kernel_form_writes_the_same_bits(R, C, ch):
    """csrc/yy_tower_g.hip: every (column blocks, boards per workgroup) form accumulates each output element in the same order,
    so tower activations and head features are bit-identical whichever form evaluates a board -- also through a row gather,
    across batch sizes, and for every number of boards per workgroup the form admits."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    E = pkg.engine
    game = pkg.YinYangGame(R, C)
    net = _randomized_net(pkg, game, 10 if ch == 128 else 3, 5, ch)
    ev = pkg.BatchedEvaluator(net, "f16x3")
    rng = np.random.default_rng(13)
    G = 117
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, C)).astype(np.int8)).cuda())
    ref_f = ref_x = None
    for nb in E.tower_g_available(ch):
        tb_max = (16 * nb) // (R * C)
        for tb in sorted({1, tb_max} - {0}):
            if tb > tb_max:
                continue
            f = E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb)
            x = E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb)
            if ref_f is None:
                ref_f, ref_x = f, x
            assert torch.equal(f, ref_f) and torch.equal(x, ref_x), (nb, tb)
    assert ref_f is not None
    flags = torch.from_numpy((rng.random(G) < 0.6).astype(np.uint8)).cuda()
    rows, n = E.compact_rows(flags)
    k = int(n)
    nb, tb = ev.g_big
    fr = E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb, rows, n)
    assert torch.equal(fr[:k], ref_f[rows[:k].long()])
    f_small = E.tower_g(planes[:40].contiguous(), ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb)
    assert torch.equal(f_small, ref_f[:40])


def test_split_f16_form_chosen_on_the_device_writes_the_same_bits():
    """BatchedEvaluator with compaction on a mid-size batch: both kernel forms of the tower are enqueued, gated on the device-side
    live row count (<= g_split rows: one 8x8 board per workgroup, more: two).  For row counts on both sides of the split, at it,
    zero and the whole batch: the evaluator returns what the dense evaluation returns, bit for bit, and the gated launches
    write exactly the rows below the count."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    E = pkg.engine
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(8, 8)).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "f16x3")
    assert ev.g_small == (4, 1) and ev.g_big == (8, 2)
    S = ev.g_split
    G = 1024
    rng = np.random.default_rng(4)
    planes = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, 8, 8)).astype(np.int8)).cuda())
    dense_p, dense_v = ev(planes)
    dense_f = E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, 8, 2, ev.g_hw, ev.g_hb)
    for n_live in (0, 1, 7, S - 1, S, S + 1, (S + G) // 2, 1024):
        flags = torch.zeros(G, dtype=torch.uint8, device="cuda")
        flags[torch.from_numpy(rng.choice(G, n_live, replace=False)).cuda()] = 1
        rows, n = E.compact_rows(flags)
        assert int(n) == n_live
        got = torch.full((G, 2, 2048), -7.0, device="cuda")
        E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, 4, 1, ev.g_hw, ev.g_hb, rows, n, got, (-1, S))
        E.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, 8, 2, ev.g_hw, ev.g_hb, rows, n, got, (S, 0x7FFFFFFF))
        assert torch.equal(got[:n_live], dense_f[rows[:n_live].long()]), n_live
        assert n_live == G or bool((got[n_live:] == -7.0).all())
        p, v = ev(planes, needs_eval=flags)
        live = flags.bool()
        assert torch.equal(p[live], dense_p[live]) and torch.equal(v[live], dense_v[live]), n_live


@pytest.mark.parametrize("R", [8, 6, 12])
def test_evaluator_rows_do_not_depend_on_the_batch(R):
    """The property the engine's evaluation reuse rests on (BatchedEvaluator.row_independent; YY_FLAG_REUSE_*, the opening book):
    a row's (policy, value) is a function of that row's planes alone, bit for bit.  The same 512 random positions of the 128 x 10
    network evaluated in batches of M = 1 ... 8192 rows (every drain tier of the engine, the book's batch, ragged sizes; dense
    and through the row compaction, static buffers or fresh ones): identical bits per row.  True by construction -- the tower
    and our own FC-head kernel accumulate every output in a fixed order; no library GEMM on the path."""
    import torch
    import yinyang_game_alphazero_amd as pkg
    E = pkg.engine
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, R)).cuda().eval()
    ev = pkg.BatchedEvaluator(net)
    assert ev.mode == "f16x3" and ev.row_independent
    rng = np.random.default_rng(21)
    B = 512
    base = E.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(B, R, R)).astype(np.int8)).cuda())
    p_ref, v_ref = (t.clone() for t in ev(base))
    assert bool(torch.isfinite(p_ref).all()) and bool(torch.isfinite(v_ref).all())
    for M in (1, 96, 256, 320, 321, 512, 513, 1024, 3072, 3928, 4096, 8192):
        idx = torch.from_numpy(rng.integers(0, B, size=M)).cuda() if M > 1 else torch.tensor([77], device="cuda")
        planes = base[idx].contiguous()
        p, v = ev(planes)
        assert torch.equal(p, p_ref[idx]) and torch.equal(v, v_ref[idx]), ("dense", M)
        flags = torch.from_numpy((rng.random(M) < 0.3).astype(np.uint8)).cuda()
        p, v = ev(planes, needs_eval=flags, static=True)
        live = flags.bool()
        assert torch.equal(p[live], p_ref[idx][live]) and torch.equal(v[live], v_ref[idx][live]), ("compacted", M)
