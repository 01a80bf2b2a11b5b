"""Builds tests/host_harness/_build/libhh.so: a TEST-ONLY g++ build of the product's per-Gaussian
math header (splat_one_amd/csrc/splat_math.hpp) and of the counter-based RNG
(csrc/so_rng.hpp).  Nothing under splat_one_amd/ loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# SPLAT_ONE_AMD_SANITIZE=1 (tests/test_sanitizers.py): the AddressSanitizer + UBSan build of the same harness
SAN = os.environ.get("SPLAT_ONE_AMD_SANITIZE") == "1"
OUT = os.path.join(HERE, "_build", "libhh_san.so" if SAN else "libhh.so")


def build() -> str:
    src = os.path.join(HERE, "harness.cpp")
    hdrs = [os.path.join(HERE, "..", "..", "splat_one_amd", "csrc", h) for h in ("splat_math.hpp", "so_rng.hpp")]
    if (not os.path.exists(OUT)) or os.path.getmtime(OUT) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if SAN else []
        subprocess.run(["g++", "-O1", "-g", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unknown-pragmas"] + san + ["-o", OUT, src], check=True)
    return OUT


if __name__ == "__main__":
    print(build())
