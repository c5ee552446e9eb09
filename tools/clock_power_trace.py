#!/usr/bin/env python3
"""Clock and board power across a run of back-to-back eval_constraint_jacobian calls: `bench.py --steps 200` as a child process,
the amdgpu hwmon sensors of every visible card (power1_input uW, freq1_input Hz = sclk) sampled every 10 ms by this process (which
never touches the GPU); the card whose power moves is the one the child ran on.  usage: clock_power_trace.py <out.csv>"""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
cards = []
for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    if os.path.exists(os.path.join(hw, "power1_input")) and os.path.exists(os.path.join(hw, "freq1_input")):
        cards.append(hw)
print(f"{len(cards)} cards with power1_input / freq1_input", flush=True)


def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return -1


child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "5", "--no-cpu-baseline",
                          "--no-other-callbacks", "--no-bound-output", "--no-kernel-timing"], stdout=subprocess.PIPE, text=True)
rows = []
t0 = time.perf_counter()
while child.poll() is None:
    t = time.perf_counter() - t0
    rows.append([t] + [v for hw in cards for v in (read(os.path.join(hw, "power1_input")), read(os.path.join(hw, "freq1_input")))])
    time.sleep(0.01)
line = child.stdout.read().strip().splitlines()[-1] if child.stdout else ""
with open(out, "w") as f:
    f.write("t_s," + ",".join(f"card{i}_power_uW,card{i}_sclk_Hz" for i in range(len(cards))) + "\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
# the busy card: largest power swing
best, swing = -1, -1
for i in range(len(cards)):
    p = [r[1 + 2 * i] for r in rows if r[1 + 2 * i] >= 0]
    if p and max(p) - min(p) > swing:
        best, swing = i, max(p) - min(p)
if best >= 0:
    p = [r[1 + 2 * best] / 1e6 for r in rows]
    c = [r[2 + 2 * best] / 1e6 for r in rows]
    hot = [k for k in range(len(p)) if p[k] > 0.6 * max(p)]
    cap = read(os.path.join(cards[best], "power1_cap")) / 1e6
    print(f"busy card {cards[best]}: power cap {cap:.0f} W; idle {min(p):.0f} W; under load (samples above 60% of max, n={len(hot)}): "
          f"power mean {sum(p[k] for k in hot) / max(len(hot), 1):.0f} W max {max(p):.0f} W; sclk mean {sum(c[k] for k in hot) / max(len(hot), 1):.0f} MHz "
          f"min {min(c[k] for k in hot) if hot else 0:.0f} max {max(c):.0f} MHz")
print("bench line:", line[:300])
