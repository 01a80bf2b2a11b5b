"""Runs the seeded fuzz cases of tests/test_gpu_fuzz.py one by one and prints the failing ones (configuration + error)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F
dev = torch.device("cuda:0")
which = "engine" if "--engine" in sys.argv else "operator"
fn, case = ((F.test_random_engine_configuration_against_the_oracle, F._engine_case) if which == "engine"
            else (F.test_random_configuration_against_the_oracle, F._case))
seeds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or list(range(36))
for seed in seeds:
    try:
        fn(dev, seed)
        print(seed, "ok", case(seed))
    except AssertionError as e:
        a = e.args[0] if e.args else None
        print(seed, "FAIL", (str(a) if a is not None else str(e))[:500])
    except Exception as e:   # noqa: BLE001
        print(seed, "ERROR", type(e).__name__, str(e)[:300])
