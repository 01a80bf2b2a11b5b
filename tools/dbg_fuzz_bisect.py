"""Which earlier seed of a fuzz run makes a later one fail?  A seed that fails only after a history points at a re-used buffer:
bisect the START of the history (one child process per probe, one at a time), then confirm with the pair.
    python tools/dbg_fuzz_bisect.py FIRST FAILING [--engine]"""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
first, bad = int(sys.argv[1]), int(sys.argv[2])
extra = [a for a in sys.argv[3:] if a.startswith("--")]


def fails(seeds):
    out = subprocess.run([sys.executable, os.path.join(here, "dbg_fuzz_report.py"), *extra, *map(str, seeds)], capture_output=True, text=True, timeout=400).stdout
    line = [l for l in out.splitlines() if l.startswith(f"{bad} ")]
    verdict = bool(line) and " ok " not in line[0]
    print(f"history {seeds[0]}..{seeds[-2] if len(seeds) > 1 else ''} ({len(seeds) - 1} seeds) -> {bad} {'FAILS' if verdict else 'ok'}", flush=True)
    return verdict


lo, hi = first, bad          # invariant: starting at lo fails, starting at hi (the seed alone) passes
assert fails(list(range(lo, bad + 1))), "the full history does not fail"
while hi - lo > 1:
    mid = (lo + hi) // 2
    if fails(list(range(mid, bad + 1))):
        lo = mid
    else:
        hi = mid
print("last start that still fails:", lo)
fails([lo, bad])
fails([lo] + list(range(lo + 1, bad + 1))[-3:])
