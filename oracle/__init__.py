"""oracle/ -- CPU restatement of the 3DGS rasterisation path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party module
github.com/inuex35/gsplat (branch spherical_render, commit
b0e978da67fb4364611c6683c5f4e6e6c1d8d8cb; /root/reference/.gitmodules:13-16,
/root/reference/Dockerfile:81-83).  Its source is absent from /root/reference
(submodules/gsplat is an empty directory), it cannot be fetched (no network) and
none of the reference's own tests pin results at that boundary.  This oracle
therefore restates the *published* algorithm (Kerbl et al. 2023; the gsplat
mathematical supplement arXiv 2312.02121; gsplat paper arXiv 2409.06765) and is
anchored on
  * the reference's call sites       utils/gsplat_utils/gsplat_trainer.py:446-497
  * analytic known-answer tests      tests/test_oracle_kat.py
  * golden vectors produced by importing the reference's importable modules
    (utils/gsplat_utils/utils.py, utils/datasets/{normalize,traj}.py) with
    tests/golden/make_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package.  The product (splat_one_amd/) never does.
"""
