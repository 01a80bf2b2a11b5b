"""Multi-GPU layer of the path: view-sharded data parallelism (BASELINE.json north_star).

Every rank holds ALL Gaussians (parameters + Adam moments: 708 B per Gaussian, 1.4 GB at 2 M --
nothing next to 288 GB of HBM3E), renders its own view(s), and the flattened gradient SoA
((11+3K)*N floats) is summed across ranks with ONE RCCL all-reduce over xGMI; the densification
statistics (2*N floats) are all-reduced on refine steps so that every rank takes identical
duplicate / split / prune decisions.  No collective touches pixels or intersections.

`cli(fn, cfg)` mirrors `gsplat.distributed.cli` as used at
/root/reference/utils/gsplat_utils/gsplat_trainer.py:998: one process per GPU calling
fn(local_rank, world_rank, world_size, cfg); under torchrun it adopts the existing ranks.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_info():
    if is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _single_node_sockets() -> None:
    """Rendezvous on one node (MASTER_ADDR is the loopback address): pin gloo's and RCCL's bootstrap sockets to `lo` unless
    the user chose an interface.  Without it both resolve the machine's HOSTNAME, and in a container whose hostname does not
    resolve every lookup waits for the DNS timeout (measured on one MI355X box: 2.5 minutes to set up a two-rank gloo group
    that takes 3 seconds elsewhere)."""
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1") and os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from torchrun's environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    backend defaults to "nccl" (= RCCL on ROCm) when a GPU is present, else "gloo"."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _single_node_sockets()
        if backend is None:   # SPLAT_ONE_AMD_BACKEND=gloo: several ranks on one GPU (tests of the multi-GPU paths)
            backend = os.environ.get("SPLAT_ONE_AMD_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        elif torch.cuda.is_available():
            local_rank = local_rank % torch.cuda.device_count()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return local_rank, rank, world


def _worker(local_rank: int, fn: Callable, args, world_size: int, port: int, backend: Optional[str]):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_from_env(backend)
    try:
        fn(local_rank, local_rank, world_size, args)
    finally:
        if is_initialized():
            dist.destroy_process_group()


def cli(fn: Callable, args, verbose: bool = False, world_size: Optional[int] = None,
        backend: Optional[str] = None, port: int = 29517) -> None:
    """fn(local_rank, world_rank, world_size, args) on every GPU of this node."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:      # launched by torchrun
        local_rank, rank, world = init_from_env(backend)
        try:
            fn(local_rank, rank, world, args)
        finally:
            if is_initialized():
                dist.destroy_process_group()
        return
    if world_size is None:
        world_size = max(1, torch.cuda.device_count())
    if world_size == 1:
        fn(0, 0, 1, args)
        return
    if verbose:
        print(f"launching {world_size} ranks")
    torch.multiprocessing.spawn(_worker, args=(fn, args, world_size, port, backend), nprocs=world_size, join=True)


class GradientReducer:
    """Flattens the gradients of the given parameters into one contiguous buffer, all-reduces it
    once (sum), rescales by 1/world and points every `.grad` at its slice of the buffer."""

    def __init__(self, group=None):
        self.group = group
        self._flat: Optional[torch.Tensor] = None

    @torch.no_grad()
    def reduce(self, params: Iterable[torch.nn.Parameter]) -> None:
        ps: List[torch.nn.Parameter] = [p for p in params if p.grad is not None]
        if not ps or not is_initialized() or dist.get_world_size(self.group) == 1:
            return
        total = sum(p.numel() for p in ps)
        if self._flat is None or self._flat.numel() != total or self._flat.device != ps[0].device:
            self._flat = torch.empty(total, dtype=ps[0].dtype, device=ps[0].device)
        flat = self._flat
        # gradients that already live in the flat buffer (previous call re-pointed them) need no copy
        off, in_place = 0, True
        for p in ps:
            g = p.grad
            if not (g.is_contiguous() and g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                    and g.storage_offset() == off):
                in_place = False
                break
            off += p.numel()
        if not in_place:
            pieces = [p.grad.reshape(-1).clone() if p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                      else p.grad.reshape(-1) for p in ps]
            torch.cat(pieces, out=flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / dist.get_world_size(self.group))
        off = 0
        for p in ps:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n


@torch.no_grad()
def all_reduce_mean_(flat: torch.Tensor, group=None) -> None:
    """In-place mean over ranks of one flat buffer (the engine's gradient SoA): ONE collective."""
    if not is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / dist.get_world_size(group))


# Set by bench.py before the runner is built: the sharded optimisers then bracket their phases with events on the compute
# stream (four records per step) and `comm_ms()` reports the means -- so that the first real multi-GPU run explains itself.
COMM_TIMING = False


class _PhaseTimer:
    """Events on the CURRENT stream around the phases of one reduce-scatter / Adam / all-gather step.  A `work.wait()` makes
    the current stream wait for the collective, so the span between two records is the time the compute stream spent in
    (or blocked on) that phase.  Host tensors (CPU tests) are timed with the wall clock."""
    PHASES = ("reduce_scatter_wait_ms", "adam_and_issue_ms", "all_gather_wait_ms")

    def __init__(self, keep: int = 512):
        self.keep, self.marks = keep, []

    def start(self, cuda: bool):
        self.cur = [self._mark(cuda)]

    def mark(self, cuda: bool):
        self.cur.append(self._mark(cuda))

    def stop(self):
        self.marks.append(self.cur)
        del self.marks[:-self.keep]

    @staticmethod
    def _mark(cuda: bool):
        if cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def summary(self) -> Optional[dict]:
        """Mean milliseconds per step of every phase over the recorded steps (synchronises); None if nothing was timed."""
        if not self.marks:
            return None
        if not isinstance(self.marks[0][0], float):
            torch.cuda.synchronize()
        span = lambda a, b: (b - a) * 1e3 if isinstance(a, float) else a.elapsed_time(b)
        n = len(self.marks)
        out = {k: sum(span(m[i], m[i + 1]) for m in self.marks) / n for i, k in enumerate(self.PHASES)}
        out["total_ms"] = sum(span(m[0], m[-1]) for m in self.marks) / n
        out["steps_timed"] = n
        return out

    def clear(self):
        self.marks = []


def agree_on_coalescing(probe: Callable[[], None], device, group=None, trial: Optional[Callable[[], None]] = None) -> str:
    """"coalesced" if `probe()` succeeds on EVERY rank, else "per_tensor" (one async reduce_scatter_tensor /
    all_gather_into_tensor per tensor, public API).  `probe` looks at rank-LOCAL things only -- does torch's PRIVATE
    `dist._coalescing_manager` exist, can it be entered and left with nothing inside -- and must NOT issue a collective
    (a rank whose probe failed half-way would leave the others inside a grouped collective it never joins).
    The decision is collective: every rank contributes 1 / 0 to ONE all_reduce(MIN); the grouped collectives themselves
    only run after all ranks have agreed.  SPLAT_ONE_AMD_FORCE_COALESCE_FAIL = "all" | a rank number makes the probe fail
    there (tests)."""
    ok, why = 1, ""
    forced = os.environ.get("SPLAT_ONE_AMD_FORCE_COALESCE_FAIL", "")
    try:
        if forced == "all" or (forced.isdigit() and int(forced) == dist.get_rank(group)):
            raise RuntimeError("forced failure of the coalesced collectives (SPLAT_ONE_AMD_FORCE_COALESCE_FAIL)")
        probe()
    except Exception as e:       # noqa: BLE001 -- a private API: anything may have changed
        ok, why = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    mode = "coalesced" if int(flag.item()) else "per_tensor"
    if mode == "coalesced" and trial is not None:
        # Second phase: every rank is known to be here, in step.  ONE tiny grouped reduce-scatter + all-gather on
        # scratch tensors -- including the in-place `reduce_scatter_tensor(f[r], f)` form the void flags use -- so that a torch /
        # RCCL build that cannot group these calls is found out NOW and not in the middle of the first step's grouped
        # collective over the gradient buffers; then a second all_reduce(MIN).  (A build that cannot raises when the call is
        # made, on every rank alike; the forced failure of the tests raises before the rank has joined anything.)
        try:
            if forced == "trial" or (forced.startswith("trial:") and int(forced[6:]) == dist.get_rank(group)):
                raise RuntimeError("forced failure of the trial grouped collective (SPLAT_ONE_AMD_FORCE_COALESCE_FAIL)")
            trial()
        except Exception as e:   # noqa: BLE001
            ok, why = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        mode = "coalesced" if int(flag.item()) else "per_tensor"
    if mode == "per_tensor":
        import warnings
        warnings.warn("splat_one_amd: grouped collectives (dist._coalescing_manager) are not usable "
                      + (f"on this rank ({why})" if why else "on another rank")
                      + "; every rank falls back to one reduce_scatter_tensor / all_gather_into_tensor per tensor", RuntimeWarning)
    return mode


def probe_coalescing_locally(group=None) -> None:
    """Rank-local half of `agree_on_coalescing`: the private API exists and its context manager can be entered and left
    with NO collective inside.  Raises if not."""
    cm = getattr(dist, "_coalescing_manager", None)
    if cm is None:
        raise RuntimeError("torch.distributed._coalescing_manager does not exist in this torch")
    with cm(group=group, async_ops=True):
        pass


def trial_grouped_collectives(device, group=None) -> None:
    """Second half of `agree_on_coalescing` (RCCL only; called by every rank after the first vote said "coalesced"): one
    grouped reduce-scatter (two tensors, the second one in place inside its own input as the void flags are) and one grouped
    all-gather on 64-float scratch pieces, waited for and checked.  Raises if this torch / RCCL build cannot do that."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a = torch.full((world * 16,), 1.0, dtype=torch.float32, device=device)
    f = torch.full((world * 16,), 2.0, dtype=torch.float32, device=device)
    out = torch.zeros(16, dtype=torch.float32, device=device)
    with dist._coalescing_manager(group=group, async_ops=True) as cm:
        dist.reduce_scatter_tensor(out, a, op=dist.ReduceOp.SUM, group=group)
        dist.reduce_scatter_tensor(f[rank * 16:(rank + 1) * 16], f, op=dist.ReduceOp.SUM, group=group)
    cm.wait()
    g = torch.zeros(world * 16, dtype=torch.float32, device=device)
    g[rank * 16:(rank + 1) * 16] = float(rank + 1)
    with dist._coalescing_manager(group=group, async_ops=True) as cm:
        dist.all_gather_into_tensor(g, g[rank * 16:(rank + 1) * 16], group=group)
        dist.all_gather_into_tensor(a, out, group=group)
    cm.wait()
    want = torch.arange(1, world + 1, dtype=torch.float32, device=device).repeat_interleave(16)
    if not (torch.equal(g, want) and bool((out == float(world)).all()) and bool((f[rank * 16:(rank + 1) * 16] == 2.0 * world).all())
            and bool((a == float(world)).all())):
        raise RuntimeError("grouped reduce_scatter_tensor / all_gather_into_tensor returned wrong values")


class RowShardedAdam:
    """Reduce-scatter / sharded Adam / all-gather step of the replicated data-parallel scheme for a DEVICE-RESIDENT model
    (FusedEngine(device_refine=True)): parameters, moments and gradients are separate tensors of `capacity` rows of which
    the first N are live, so no flat piece layout survives a refinement.  Pieces are ROW ranges.  Round 4: the rows are cut
    into `n_chunks` CHUNKS of S rows (S a multiple of world x 64) and every chunk into `world` pieces of S / world rows --
    rank r owns rows [c S + r S/world, c S + (r+1) S/world) of every chunk c and of EVERY tensor -- so that the per-Gaussian
    backward can be launched chunk by chunk (so_train_step_bwd_rows) with the reduce-scatter of chunk c running on RCCL's
    stream under the kernel of chunk c + 1:

        for c:  backward kernel on rows of chunk c;  reduce_chunk(c): reduce-scatter of those rows, async      (overlapped)
        finish: for c:  wait RS(c);  Adam on the own rows of chunk c (1/world of the optimiser traffic, gradient
                        scaled by 1/world on the fly);  all-gather of the parameter rows of chunk c, async
                wait the all-gathers

    Every chunk moves in two groups -- shN (45 of the 59 floats of a row) and the five small tensors -- each ONE RCCL launch
    (torch's coalescing manager) so that a group's all-gather runs under the next group's Adam.  THE VOID FLAG (a view's
    binning pass overflowed: the iteration must be skipped on EVERY rank) rides in the small group of chunk 0: `flags` holds
    world x 16 floats, every rank writes its own overflow word into all of them, the reduce-scatter hands rank r the SUM in
    flags[16 r] and the Adam launches skip on it on the device -- no collective of its own, no host read (VERDICT r3 item
    3b).  Rows between N and n_chunks x S belong to nobody: they are reduced and gathered with the rest (capacity >=
    n_chunks x S is the engine's contract, `span`) and never read.  `gather` all-gathers any set of row-sharded tensors (the
    moments, before a refinement compacts them identically on every rank).  N is a host integer: replicas read it back once
    per refinement.

    gloo (tests): HIP tensors move through all_reduce only (a reduce-scatter is an all-reduce whose other rows are
    ignored, an all-gather an all-reduce of zero-padded pieces); CPU tensors through reduce / all_gather per piece."""

    GROUPS = (("shN",), ("means", "scales", "quats", "opacities", "sh0"))
    ALIGN_ROWS = 64   # piece boundaries: k_preprocess_bwd's staged write-out and the float4 Adam want 64-row starts
    FLAG_STRIDE = 16  # floats per piece of `flags` (64 bytes)

    def __init__(self, group=None, n_chunks: int = 1):
        self.group = group
        self.rank, self.world = (dist.get_rank(group), dist.get_world_size(group)) if is_initialized() else (0, 1)
        self.backend = dist.get_backend(group) if is_initialized() else "none"
        self.n_chunks = max(1, int(n_chunks))
        # "coalesced": each group of tensors is ONE RCCL launch (torch's private coalescing manager); "per_tensor": one
        # async collective per tensor (public API).  Agreed on by all ranks at the first step (agree_on_coalescing).
        self.mode: Optional[str] = None
        self.timer = _PhaseTimer() if COMM_TIMING else None
        self.flags: Optional[torch.Tensor] = None
        self._rs: List[list] = []
        self._n = 0
        self._pending_ag: list = []              # all-gathers of the last finish(defer_gather_wait=True)
        self._pending_tm = None

    def _ensure_mode(self, device) -> str:
        if self.mode is None:
            if self.world == 1:
                self.mode = "coalesced"
            else:
                probe = (lambda: probe_coalescing_locally(self.group)) if self.backend == "nccl" else (lambda: None)
                trial = (lambda: trial_grouped_collectives(device, self.group)) if self.backend == "nccl" else None
                self.mode = agree_on_coalescing(probe, device if self.backend == "nccl" else "cpu", self.group, trial)
        return self.mode

    def comm_ms(self) -> Optional[dict]:
        out = self.timer.summary() if self.timer is not None else None
        if out is not None:
            out["collectives"] = self.mode
            out["chunks"] = self.n_chunks
        return out

    # ---- layout
    def chunk_rows(self, n: int) -> int:
        """S: rows per chunk, a multiple of world x ALIGN_ROWS, n_chunks x S >= n."""
        q = self.world * self.ALIGN_ROWS
        return -(-max(int(n), 1) // (self.n_chunks * q)) * q

    def piece(self, n: int) -> int:
        return self.chunk_rows(n) // self.world

    def span(self, n: int) -> int:
        """Rows the collectives touch: the capacity of every tensor must be at least this."""
        return self.chunk_rows(n) * self.n_chunks

    def chunk_range(self, n: int, c: int):
        S = self.chunk_rows(n)
        return c * S, (c + 1) * S

    def rows(self, n: int, c: int = 0, r: Optional[int] = None):
        """Rows of chunk c that rank r owns (not clipped to n)."""
        S, r = self.chunk_rows(n), (self.rank if r is None else r)
        p = S // self.world
        return c * S + r * p, c * S + (r + 1) * p

    def owned(self, n: int, r: Optional[int] = None):
        """[(a, b)] over the chunks: the live rows rank r owns."""
        out = []
        for c in range(self.n_chunks):
            a, b = self.rows(n, c, r)
            if a < n:
                out.append((a, min(b, n)))
        return out

    def bytes_per_link_and_step(self, n: int, floats_per_row: int = 59) -> float:
        return 2.0 * (self.world - 1) / max(1, self.world) * self.span(n) * floats_per_row * 4.0

    # ---- collectives on one chunk
    def _reduce_scatter(self, tensors: List[torch.Tensor], n: int, c: int, with_flags: bool = False):
        """with_flags: the void flags join THIS group -- on RCCL in coalesced mode inside the same grouped launch (no
        collective of their own), otherwise as one more asynchronous collective behind it."""
        (lo, hi), (a, b) = self.chunk_range(n, c), self.rows(n, c)
        p = self.piece(n)
        assert all(t.shape[0] >= hi and t.is_contiguous() for t in tensors), (n, hi, [tuple(t.shape) for t in tensors])
        if self.backend == "nccl" and self.mode == "per_tensor":
            return [dist.reduce_scatter_tensor(t[a:b], t[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    for t in tensors] + (self._reduce_scatter_flags() if with_flags else [])
        if self.backend == "nccl":
            f, st, r = self.flags, self.FLAG_STRIDE, self.rank
            with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                for t in tensors:
                    dist.reduce_scatter_tensor(t[a:b], t[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
                if with_flags:
                    dist.reduce_scatter_tensor(f[r * st:(r + 1) * st], f, op=dist.ReduceOp.SUM, group=self.group)
            return [cm]
        if with_flags:
            return self._reduce_scatter(tensors, n, c) + self._reduce_scatter_flags()
        if tensors[0].is_cuda:
            return [dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for t in tensors]
        works = []
        for t in tensors:
            for j in range(self.world):
                works.append(dist.reduce(t[lo + j * p:lo + (j + 1) * p], dst=j if self.group is None else dist.get_global_rank(self.group, j),
                                         op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    def _all_gather(self, tensors: List[torch.Tensor], n: int, c: int):
        (lo, hi), (a, b) = self.chunk_range(n, c), self.rows(n, c)
        p = self.piece(n)
        assert all(t.shape[0] >= hi and t.is_contiguous() for t in tensors), (n, hi, [tuple(t.shape) for t in tensors])
        if self.backend == "nccl" and self.mode == "per_tensor":
            return [dist.all_gather_into_tensor(t[lo:hi], t[a:b], group=self.group, async_op=True) for t in tensors]
        if self.backend == "nccl":
            with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
                for t in tensors:
                    dist.all_gather_into_tensor(t[lo:hi], t[a:b], group=self.group)
            return [cm]
        if tensors[0].is_cuda:
            works = []
            for t in tensors:
                t[lo:a].zero_()
                t[b:hi].zero_()
                works.append(dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return works
        return [dist.all_gather([t[lo + j * p:lo + (j + 1) * p] for j in range(self.world)], t[a:b].clone(), group=self.group, async_op=True)
                for t in tensors]

    # ---- one optimiser step, in two halves
    @torch.no_grad()
    def begin(self, n: int, device, cuda: bool) -> None:
        """Start a step over n live rows.  (world == 1: nothing to exchange.)"""
        self.wait_gathers()                       # (a deferred gather of the previous step: the rows are about to be reused)
        self._n, self._rs, self._void_src = int(n), [None] * self.n_chunks, None
        if self.world == 1:
            return
        self._ensure_mode(device)
        if self.flags is None or self.flags.device != torch.device(device):
            self.flags = torch.zeros(self.world * self.FLAG_STRIDE, dtype=torch.float32, device=device)
        if self.timer is not None:
            self.timer.start(cuda)

    @torch.no_grad()
    def reduce_chunk(self, c: int, grads: Dict[str, torch.Tensor], void_src: Optional[torch.Tensor] = None) -> None:
        """Issue the reduce-scatter of chunk c (asynchronous: it waits for what the current stream has been given so far --
        the backward kernel of this chunk -- and runs under whatever is launched next).  void_src (chunk 0): one float, this
        rank's "my binning pass overflowed" word (non-zero = void); it is summed over the ranks with the small group."""
        if self.world == 1:
            return
        groups = [[grads[k] for k in g if k in grads] for g in self.GROUPS]
        if c == 0:
            if void_src is not None:
                self.flags.copy_(void_src.reshape(1).expand(self.flags.numel()))
            else:
                self.flags.zero_()
        works = []
        for gi, ts in enumerate(groups):
            works.append(self._reduce_scatter(ts, self._n, c, with_flags=(c == 0 and gi == len(groups) - 1)))
        self._rs[c] = works

    def _reduce_scatter_flags(self):
        f, st, r = self.flags, self.FLAG_STRIDE, self.rank
        if self.backend == "nccl":
            return [dist.reduce_scatter_tensor(f[r * st:(r + 1) * st], f, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        if f.is_cuda:
            return [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        return [dist.reduce(f[j * st:(j + 1) * st], dst=j if self.group is None else dist.get_global_rank(self.group, j),
                            op=dist.ReduceOp.SUM, group=self.group, async_op=True) for j in range(self.world)]

    def void_flag(self) -> Optional[torch.Tensor]:
        """This rank's piece of the summed flags: one device float, non-zero when ANY rank's iteration was void (valid once
        chunk 0's reduction has been waited for; identical on every rank)."""
        if self.flags is None:
            return None
        return self.flags[self.rank * self.FLAG_STRIDE:self.rank * self.FLAG_STRIDE + 1]

    @torch.no_grad()
    def wait_gathers(self) -> None:
        """The parameter all-gathers of the last `finish(defer_gather_wait=True)`: make the current stream (CPU tensors: the host)
        wait for them.  Idempotent and cheap; called before anything reads or writes the parameters (FusedEngine calls it in
        front of every launch that does, Runner before handing out `splats`)."""
        ag, self._pending_ag = getattr(self, "_pending_ag", None) or [], []
        if not ag:
            return
        for w in ag:
            w.wait()
        tm, self._pending_tm = getattr(self, "_pending_tm", None), None
        if tm is not None:
            tm[0].mark(tm[1])
            tm[0].stop()

    @torch.no_grad()
    def finish(self, grads: Dict[str, torch.Tensor], params: Dict[str, torch.Tensor],
               adam_fn: Callable[..., None], defer_gather_wait: bool = False) -> None:
        """Wait for the reductions chunk by chunk, run adam_fn(names, row_start, row_stop, skip, grad_scale) on the own live
        rows of each, all-gather the parameter rows.  skip = this rank's summed void flag (device float), grad_scale =
        1 / world (mean over the views of all ranks).  defer_gather_wait: return with the all-gathers in flight -- the caller
        stages the next iteration (so_step_inputs, a target upload: nothing there touches the parameters) and calls
        `wait_gathers()` before the first kernel that does, so the gathers' tail runs under that staging."""
        n = self._n
        groups = [tuple(k for k in g if k in grads) for g in self.GROUPS]
        if self.world == 1:
            adam_fn(tuple(k for g in groups for k in g), 0, n, None, 1.0)
            return
        tm = self.timer
        cuda = next(iter(grads.values())).is_cuda
        ag, inv, skip = [], 1.0 / self.world, self.void_flag()
        # chunk 0's small group carries the flag every Adam launch skips on: wait for it first
        for w in self._rs[0][-1]:
            w.wait()
        for c in range(self.n_chunks):
            a, b = self.rows(n, c)
            b = min(b, n)
            for gi, g in enumerate(groups):
                for w in self._rs[c][gi]:
                    w.wait()
                if tm is not None and c == 0 and gi == 0:
                    tm.mark(cuda)                                         # the first reduction has landed
                if b > a:
                    adam_fn(g, a, b, skip, inv)
                ag += self._all_gather([params[k] for k in g], n, c)
        if tm is not None:
            tm.mark(cuda)                                                 # every Adam launched, every all-gather issued
        self._pending_ag, self._pending_tm = ag, ((tm, cuda) if tm is not None else None)
        if not defer_gather_wait:
            self.wait_gathers()

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], params: Dict[str, torch.Tensor], n: int,
             adam_fn: Callable[..., None], void_src: Optional[torch.Tensor] = None) -> None:
        """The whole step on finished gradients (no overlap with the backward): begin, every chunk's reduce-scatter, finish."""
        first = next(iter(grads.values()))
        self.begin(n, first.device, first.is_cuda)
        for c in range(self.n_chunks):
            self.reduce_chunk(c, grads, void_src)
        self.finish(grads, params, adam_fn)

    @torch.no_grad()
    def gather(self, tensors: List[torch.Tensor], n: int) -> None:
        self.wait_gathers()
        if self.world == 1 or not tensors:
            return
        self._ensure_mode(tensors[0].device)
        works = []
        for c in range(self.n_chunks):
            works += self._all_gather(list(tensors), n, c)
        for w in works:
            w.wait()


class ShardedFlatAdam:
    """Optimiser step of the replicated ("allreduce") data-parallel scheme, SURVEY.md section 8(e): instead of ONE
    blocking all-reduce of the flat gradient followed by a full Adam on every rank, the flat buffer is cut into
    `n_chunks` chunks and every chunk into `world` pieces;

        reduce-scatter(chunk c)   -> rank r holds the SUM of piece (c, r)            (all chunks issued up front)
        Adam on piece (c, r)      -> as soon as reduction c has landed (1/world of the parameters per rank)
        all-gather(chunk c)       -> every rank has the updated parameters of chunk c

    so the collectives of chunk c+1 run on the RCCL stream while Adam of chunk c runs on the compute stream, the
    optimiser's HBM traffic (28 B per parameter float) shrinks by `world`, and the moments of a parameter live on one
    rank only (`gather_moments` all-gathers them before a structural edit, i.e. a refinement).  On xGMI (point-to-point,
    7 links per GPU) reduce-scatter + all-gather move (world-1)/world of the buffer per rank and phase, spread over all
    links, where a ring all-reduce pushes 2 (world-1)/world of it through one link.

    Layout contract: gradients, parameters and both moments are FLAT float32 buffers of at least `padded_total`
    elements sharing one segment layout (the engine's: tensors in a fixed order, each starting at a multiple of 64
    floats).  `adam_fn(start, stop)` updates parameters and moments of the flat range [start, stop) from the (already
    averaged) gradient of that range; it is the HIP launch on the GPU (FusedEngine.adam_on_flat_range) and a torch
    restatement in the CPU tests.

    Backends: "nccl" (= RCCL) uses reduce_scatter_tensor / all_gather_into_tensor in place; gloo, which has no
    reduce-scatter, reduces every piece to its owner (`dist.reduce`, the same bytes) and all-gathers into views."""

    ALIGN = 64   # floats: piece boundaries stay 256-byte aligned (float4 kernels)

    def __init__(self, total: int, n_chunks: int = 4, group=None):
        self.group = group
        self.rank, self.world = (dist.get_rank(group), dist.get_world_size(group)) if is_initialized() else (0, 1)
        self.n_chunks = max(1, int(n_chunks))
        q = self.world * self.ALIGN
        self.chunk = -(-int(total) // (self.n_chunks * q)) * q          # elements per chunk, a multiple of world * ALIGN
        self.piece = self.chunk // self.world
        self.total = int(total)
        self.padded_total = self.chunk * self.n_chunks
        self.backend = dist.get_backend(group) if is_initialized() else "none"
        self.timer = _PhaseTimer() if COMM_TIMING else None
        self.flags: Optional[torch.Tensor] = None    # the void flag, summed over the ranks by a reduce-scatter of its own (64 B / rank)

    FLAG_STRIDE = 16

    def comm_ms(self) -> Optional[dict]:
        out = self.timer.summary() if self.timer is not None else None
        if out is not None:
            out["collectives"] = "per_tensor"       # public API only: one collective per chunk
            out["chunks"] = self.n_chunks
        return out

    def void_flag(self) -> Optional[torch.Tensor]:
        """This rank's piece of the summed void flags (one device float; identical on every rank after a step)."""
        if self.flags is None:
            return None
        return self.flags[self.rank * self.FLAG_STRIDE:self.rank * self.FLAG_STRIDE + 1]

    def _reduce_scatter_flags(self, void_src: Optional[torch.Tensor], device):
        if self.flags is None or self.flags.device != torch.device(device):
            self.flags = torch.zeros(self.world * self.FLAG_STRIDE, dtype=torch.float32, device=device)
        f, st, r = self.flags, self.FLAG_STRIDE, self.rank
        if void_src is not None:
            f.copy_(void_src.reshape(1).expand(f.numel()))
        else:
            f.zero_()
        if self.backend == "nccl":
            return [dist.reduce_scatter_tensor(f[r * st:(r + 1) * st], f, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        if f.is_cuda:
            return [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        return [dist.reduce(f[j * st:(j + 1) * st], dst=j if self.group is None else dist.get_global_rank(self.group, j),
                            op=dist.ReduceOp.SUM, group=self.group, async_op=True) for j in range(self.world)]

    def piece_range(self, c: int, r: Optional[int] = None):
        r = self.rank if r is None else r
        a = c * self.chunk + r * self.piece
        return a, a + self.piece

    def my_ranges(self):
        return [self.piece_range(c) for c in range(self.n_chunks)]

    def bytes_per_link_and_step(self) -> float:
        """Bytes every rank sends (= receives) per optimiser step: reduce-scatter + all-gather of the padded buffer."""
        return 2.0 * (self.world - 1) / max(1, self.world) * self.padded_total * 4.0

    def _reduce_scatter(self, flat: torch.Tensor, c: int):
        chunk = flat[c * self.chunk:(c + 1) * self.chunk]
        if self.backend != "nccl" and flat.is_cuda:
            # gloo moves HIP tensors only through broadcast / all_reduce (several ranks on one GPU in the tests)
            return [dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        if self.backend == "nccl":
            a, b = self.piece_range(c)
            return [dist.reduce_scatter_tensor(flat[a:b], chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        works = []
        for j in range(self.world):      # gloo: piece j is reduced to rank j
            a, b = self.piece_range(c, j)
            works.append(dist.reduce(flat[a:b], dst=j if self.group is None else dist.get_global_rank(self.group, j),
                                     op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    def _all_gather(self, flat: torch.Tensor, c: int):
        chunk = flat[c * self.chunk:(c + 1) * self.chunk]
        a, b = self.piece_range(c)
        if self.backend != "nccl" and flat.is_cuda:      # gloo + HIP tensors: x + 0 + ... + 0 is an all-gather
            chunk[:a - c * self.chunk].zero_()
            chunk[b - c * self.chunk:].zero_()
            return dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(chunk, flat[a:b], group=self.group, async_op=True)
        outs = [chunk[j * self.piece:(j + 1) * self.piece] for j in range(self.world)]
        return dist.all_gather(outs, flat[a:b].clone(), group=self.group, async_op=True)

    @torch.no_grad()
    def step(self, grad_flat: torch.Tensor, param_flat: torch.Tensor, adam_fn: Callable[..., None],
             void_src: Optional[torch.Tensor] = None) -> None:
        """adam_fn(start, stop, skip, grad_scale): Adam on the flat range from the SUMMED gradient scaled by grad_scale
        (1 / world), skipped on the device when `skip` (one float: the void flags of all ranks summed) is non-zero.
        void_src: this rank's own flag (one float, non-zero = its binning pass overflowed)."""
        assert grad_flat.numel() >= self.padded_total and param_flat.numel() >= self.padded_total, \
            (grad_flat.numel(), param_flat.numel(), self.padded_total)
        if self.world == 1:
            adam_fn(0, self.padded_total, None, 1.0)
            return
        tm, cuda = self.timer, grad_flat.is_cuda
        if tm is not None:
            tm.start(cuda)
        fl = self._reduce_scatter_flags(void_src, grad_flat.device)
        rs = [self._reduce_scatter(grad_flat, c) for c in range(self.n_chunks)]
        ag = []
        inv, skip = 1.0 / self.world, self.void_flag()
        for w in fl:
            w.wait()
        for c in range(self.n_chunks):
            for w in rs[c]:
                w.wait()
            if tm is not None and c == 0:
                tm.mark(cuda)                        # the first chunk's reduction has landed
            a, b = self.piece_range(c)
            adam_fn(a, b, skip, inv)                 # (mean over the views of all ranks: scaled inside the launch)
            ag.append(self._all_gather(param_flat, c))
        if tm is not None:
            tm.mark(cuda)
        for w in ag:
            w.wait()
        if tm is not None:
            tm.mark(cuda)
            tm.stop()

    @torch.no_grad()
    def gather_moments(self, *flats: torch.Tensor) -> None:
        """All-gather the owner pieces of the given flat buffers (exp_avg, exp_avg_sq) so that every rank holds all of
        them -- before a refinement rewrites the Gaussian set identically on every rank."""
        if self.world == 1:
            return
        for f in flats:
            works = [self._all_gather(f, c) for c in range(self.n_chunks)]
            for w in works:
                w.wait()


def all_reduce_max_(t: torch.Tensor, group=None) -> None:
    """In-place element-wise maximum over ranks (visibility masks, overflow flags)."""
    if not is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


@torch.no_grad()
def all_reduce_strategy_state(state: Dict, group=None) -> None:
    """Sum `grad2d` / `count` over ranks (one packed all-reduce) before a refine step."""
    if not is_initialized() or dist.get_world_size(group) == 1:
        return
    keys = [k for k in ("grad2d", "count") if isinstance(state.get(k), torch.Tensor)]
    if not keys:
        return
    packed = torch.stack([state[k] for k in keys])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    for i, k in enumerate(keys):
        state[k].copy_(packed[i])
    if isinstance(state.get("radii"), torch.Tensor):
        dist.all_reduce(state["radii"], op=dist.ReduceOp.MAX, group=group)
