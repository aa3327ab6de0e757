import glob, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O
from test_gpu_mcts import HostHashEvaluator
import yinyang_game_alphazero_amd as pkg
GOLDEN = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
for path in sorted(glob.glob(os.path.join(GOLDEN, "search_[0-9]*.npz"))):
    z = np.load(path)
    n = z["counts"].shape[0]
    groups = {}
    for i in range(n):
        groups.setdefault((int(z["sims"][i]), int(z["copied"][i]), int(z["has_noise"][i])), []).append(i)
    for key, idx in sorted(groups.items()):
        idx = np.asarray(idx)
        sims, copied, hn = key
        R, C = z["root_board"].shape[1:]
        m = pkg.engine.BatchedMCTS(len(idx), R, C, sims, aliased=not copied)
        ev = HostHashEvaluator(z["pbits"][idx], z["vbits"][idx], m.needs_eval)
        noise = torch.from_numpy(z["noise"][idx]).cuda() if hn else None
        counts = m.search(torch.from_numpy(z["root_board"][idx]).cuda(), torch.from_numpy(z["root_player"][idx]).cuda(), ev, sims, noise=noise).cpu().numpy()
        c2, cw, cp = m.root_counts(with_children=True)
        bad = [int(i) for j, i in enumerate(idx) if not np.array_equal(counts[j], z["counts"][i])]
        badp = [int(i) for j, i in enumerate(idx) if not np.array_equal(cp[j].cpu().numpy(), z["child_p"][i])]
        badl = []
        for j, i in enumerate(idx):
            nl = int(z["n_leaves"][i])
            if nl:
                L = np.stack(ev.logs[j]) if ev.logs[j] else np.zeros((0, R, C), np.int8)
                if L.shape[0] != nl or not np.array_equal(L, z["leaves"][i, :nl]):
                    k = 0
                    while k < min(L.shape[0], nl) and np.array_equal(L[k], z["leaves"][i, k]): k += 1
                    badl.append((int(i), k, L.shape[0], nl))
        print(os.path.basename(path), key, "n", len(idx), "bad counts", bad, "bad priors", badp, "first leaf mismatch (case, k, got, want)", badl[:4], flush=True)
        if badl and "-v" in sys.argv:
            i, k, _, _ = badl[0]
            j = list(idx).index(i)
            print("prev leaf\n", z["leaves"][i, k-1] if k else None, "\nwant\n", z["leaves"][i, k], "\ngot\n", ev.logs[j][k] if k < len(ev.logs[j]) else None)
        m.close()
