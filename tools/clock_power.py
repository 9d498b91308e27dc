#!/usr/bin/env python3
"""Clock and power the device holds under the PairHMM stream kernel, next to a pure-FMA loop and to idle.

Samples the GPU's hwmon files (sclk in Hz, socket power in microwatts) every 50 ms from a thread while
  1. nothing runs (2 s),
  2. the fp32 stream kernel runs back to back on the resident large batch (4096 x 64, 151 / 600) for SECONDS,
  3. exp_build/fma_loop runs for SECONDS at 3 and at 8 waves per SIMD,
and prints one line per phase: mean / max sclk, mean / max power, the power cap, and what the phase achieved.
The in-kernel clock of the stream kernel (shader-clock counter against the constant 100 MHz counter) comes from
the profiling build (make -C gatk_amd/csrc phase; tools/phase_clocks.py), run after this script.
"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0


def hwmon_dirs():
    return sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self, dirs):
        super().__init__(daemon=True)
        self.dirs, self.rows, self.stop_flag = dirs, [], False

    def run(self):
        while not self.stop_flag:
            row = []
            for d in self.dirs:
                p = read_int(os.path.join(d, "power1_average"))
                if p is None:
                    p = read_int(os.path.join(d, "power1_input"))
                row.append((read_int(os.path.join(d, "freq1_input")), p))
            self.rows.append(row)
            time.sleep(0.05)

    def stop(self):
        self.stop_flag = True
        self.join()
        return self.rows


def summarise(name, rows, dirs, what):
    # the card that drew the most power in this phase is the one the work ran on
    if not rows:
        print(f"{name}: no samples"); return
    best = max(range(len(dirs)), key=lambda k: sum((r[k][1] or 0) for r in rows))
    f = [r[best][0] for r in rows if r[best][0] is not None]
    p = [r[best][1] for r in rows if r[best][1] is not None]
    cap = read_int(os.path.join(dirs[best], "power1_cap"))
    fs = f"sclk mean {sum(f) / len(f) / 1e6:.0f} MHz, max {max(f) / 1e6:.0f}" if f else "sclk n/a"
    ps = f"power mean {sum(p) / len(p) / 1e6:.0f} W, max {max(p) / 1e6:.0f} W" if p else "power n/a"
    print(f"{name:34s} {len(rows):4d} samples of {os.path.basename(os.path.dirname(os.path.dirname(dirs[best])))}: {fs}; {ps}; cap {cap / 1e6 if cap else float('nan'):.0f} W; {what}")


def main():
    dirs = hwmon_dirs()
    print(f"hwmon directories: {len(dirs)}")
    if not dirs:
        print("no hwmon files visible: clock and power cannot be sampled here")
    s = Sampler(dirs); s.start(); time.sleep(2.0)
    summarise("idle", s.stop(), dirs, "nothing running")

    from gatk_amd import synth
    from gatk_amd.pairhmm import HipLoglessPairHMM
    b = synth.config_c3()
    hmm = HipLoglessPairHMM(0)
    hmm.set_option("timing", 1)
    res = hmm.upload([b])
    for _ in range(3):
        res.compute(sync=True)
    s = Sampler(dirs); s.start()
    t0 = time.perf_counter()
    passes, fp32_ms = 0, 0.0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            res.compute(sync=False)
        res.wait()
        st = hmm.stats()
        passes += st["timed_passes"]; fp32_ms += st["fp32_kernel_ms"]
    wall = time.perf_counter() - t0
    rows = s.stop()
    summarise("PairHMM fp32 stream kernel", rows, dirs,
              f"{passes} passes in {wall:.1f} s, kernel {fp32_ms / passes:.3f} ms = {b.cells / (fp32_ms / passes) / 1e6:.0f} GCUPS = "
              f"{b.cells * 12 / (fp32_ms / passes) / 1e9:.1f} TFLOP/s at 12 FLOP per cell")
    res.free(); hmm.close()

    exe = os.path.join(ROOT, "exp_build", "fma_loop")
    for waves in (3, 8):
        s = Sampler(dirs); s.start()
        out = subprocess.run([exe, str(SECONDS), str(waves)], capture_output=True, text=True).stdout.strip()
        summarise(f"pure v_fma_f32 loop, {waves} waves/SIMD", s.stop(), dirs, out)


if __name__ == "__main__":
    main()
