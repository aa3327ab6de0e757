#!/usr/bin/env python3
"""Is the split-f16 tower kernel limited by the board's power cap?  Runs the dense 4096-board launch back to back for a few
seconds while a thread samples the amdgpu hwmon files of the card (power1_average / power1_input in microwatts, power1_cap,
freq1_input = shader clock in Hz) and, when present, `rocm-smi`; prints mean power, cap and clock next to the launch time.
Then the same with a sleep between launches (duty cycle ~50 %) to show the clock the chip holds when it is not power-bound.
    python tools/power_probe.py [seconds]"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import yinyang_game_alphazero_amd as pkg

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0


def read_int(path):
    try:
        return int(open(path).read().split()[0])
    except Exception:
        return None


def hwmon_dirs():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if any(os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input")):
            out.append(d)
    return out


class Sampler(threading.Thread):
    def __init__(self, dirs):
        super().__init__(daemon=True)
        self.dirs, self.stop, self.rows = dirs, False, []

    def run(self):
        while not self.stop:
            row = []
            for d in self.dirs:
                p = read_int(os.path.join(d, "power1_average"))
                if p is None:
                    p = read_int(os.path.join(d, "power1_input"))
                row.append((p, read_int(os.path.join(d, "freq1_input"))))
            self.rows.append(row)
            time.sleep(0.05)


def smi(args):
    try:
        r = subprocess.run(["rocm-smi"] + args, capture_output=True, text=True, timeout=30)
        return r.stdout.strip()[-1500:]
    except Exception as e:
        return "rocm-smi unavailable: %r" % (e,)


def main():
    G, R = 4096, 8
    torch.manual_seed(0)
    net = pkg.YinYangNeuralNetwork(pkg.YinYangGame(R, R)).cuda().eval()
    ev = pkg.BatchedEvaluator(net, "f16x3")
    rng = np.random.default_rng(0)
    planes = pkg.engine.encode_planes(torch.from_numpy(rng.integers(-1, 2, size=(G, R, R)).astype(np.int8)).cuda())
    nb, tb = ev.g_big
    feats = torch.empty((G, 2, 32 * R * R), dtype=torch.float32, device="cuda")
    launch = lambda: pkg.engine.tower_g(planes, ev.g_w, ev.g_b, ev.h3_layers, ev.g_exps, nb, tb, ev.g_hw, ev.g_hb, out=feats)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    dirs = hwmon_dirs()
    caps = [read_int(os.path.join(d, "power1_cap")) for d in dirs]
    print("hwmon:", dirs, "power caps (W):", [c / 1e6 if c else None for c in caps])
    out = {"hwmon_dirs": dirs, "power_cap_w": [c / 1e6 if c else None for c in caps]}
    for tag, gap in (("back_to_back", 0.0), ("half_duty", None), ("idle", -1.0)):
        s = Sampler(dirs)
        s.start()
        t_end = time.perf_counter() + SECONDS
        n, busy = 0, 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.perf_counter() < t_end:
            if gap == -1.0:
                time.sleep(0.1)
                continue
            e0.record()
            for _ in range(8):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            busy += ms
            n += 8
            if gap is None:
                time.sleep(ms * 1e-3)         # as long idle as busy
        s.stop = True
        s.join()
        rows = s.rows[len(s.rows) // 4:]      # the first quarter is the ramp
        rec = {"launch_ms": busy / n if n else None, "launches": n}
        for i, d in enumerate(dirs):
            pw = [r[i][0] for r in rows if r[i][0] is not None]
            fq = [r[i][1] for r in rows if r[i][1] is not None]
            rec["card%d" % i] = {"mean_power_w": sum(pw) / len(pw) / 1e6 if pw else None, "max_power_w": max(pw) / 1e6 if pw else None,
                                 "mean_sclk_mhz": sum(fq) / len(fq) / 1e6 if fq else None, "samples": len(pw)}
        out[tag] = rec
        print(tag, json.dumps(rec))
    # calibration: the vendor's dense f16 GEMM (hipBLASLt through torch.matmul) on the same box under the same cap
    for n in (8192, 4096):
        a = torch.randn(n, n, device="cuda", dtype=torch.float16)
        b = torch.randn(n, n, device="cuda", dtype=torch.float16)
        for _ in range(3):
            a @ b
        torch.cuda.synchronize()
        s = Sampler(dirs)
        s.start()
        t_end = time.perf_counter() + SECONDS
        n_mm, busy = 0, 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.perf_counter() < t_end:
            e0.record()
            for _ in range(20):
                a @ b
            e1.record()
            torch.cuda.synchronize()
            busy += e0.elapsed_time(e1)
            n_mm += 20
        s.stop = True
        s.join()
        rows = s.rows[len(s.rows) // 4:]
        rec = {"ms": busy / n_mm, "tflops": 2 * n ** 3 / (busy / n_mm * 1e-3) / 1e12}
        for i, d in enumerate(dirs):
            pw = [r[i][0] for r in rows if r[i][0] is not None]
            fq = [r[i][1] for r in rows if r[i][1] is not None]
            rec["card%d" % i] = {"mean_power_w": sum(pw) / len(pw) / 1e6 if pw else None,
                                 "mean_sclk_mhz": sum(fq) / len(fq) / 1e6 if fq else None}
        out["hipblaslt_f16_gemm_%d" % n] = rec
        print("hipblaslt f16 gemm", n, json.dumps(rec))
        del a, b
    print(smi(["--showpower", "--showclocks", "--showmaxpower"]))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/power_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
