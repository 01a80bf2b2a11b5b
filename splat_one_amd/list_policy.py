"""The fused engine's decisions about its tile lists, as ONE pure function per event (no torch, no device).

`FusedEngine` owns buffers, graphs and the device; this module owns the POLICY: given what the engine knows (the state below) and
an event (the capacity probe of a new workspace, the list statistics the device publishes one call late, an overflow found one
step late, the caller's take-back), it returns the list of actions the engine then executes.  DESIGN.md section 4.3 prints the same
table; tests/test_list_policy.py walks every (state, event) row of it against these functions, tests/test_gpu_engine.py runs the
engine through the rows that need a device.

Actions (tuples, executed in order by FusedEngine._apply):
    ("set_kernels", raster_impl, lpt, fold)  which backward rasteriser / tile order / where the per-tile sort runs; drops every
                                        captured graph when it changes
    ("rebuild_bins", slots)             new workspace with `slots` per tile (binned layout; the list-following path: 8x the fullest tile)
    ("grow", needed)                    FusedEngine._grow: bins of >= 2 needed + 16 slots / compact buffers of 1.5 needed + 4096 entries
    ("fall_back_to_compact", fullest)   leave the binned layout for good (bins at their memory budget), warn once
    ("take_back", void, needed, compact) FusedEngine.take_back -> on_take_back below
    ("void", n)                         n iterations never happened: undo the host-side step bookkeeping (step count, Adam step, lr)
    ("defer", seen, at_limit)           replicas: remember what this rank saw; the caller decides from the summed flag
    ("raise", message)                  RuntimeError
    ("restage",)                        the probe changed the workspace: stage the view again on the new buffers
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

Action = Tuple


@dataclass
class ListState:
    binned: bool                 # per-tile bins (True) or gsplat's compact lists (False)
    bin_capacity: int            # slots per tile (binned)
    bin_limit: int               # most slots per tile the memory budget allows (binned)
    capacity: int                # intersection entries of the compact buffers
    raster_impl: int             # 0: one wave per 8x8 quadrant, 1: one wave per 16x16 tile (backward)
    lpt: bool                    # rasterisers take their tiles longest list first
    on_overflow: str             # "grow" | "raise" | "defer"
    fold: bool = False           # the per-tile sort runs in the forward rasteriser's prologue (no sort launch)
    fold_allowed: bool = True    # FusedEngine.sort_fold_ok (off by default: measured equal, profiles/r05_experiments.json)
    tile16: bool = True          # 16x16 tiles (the one-wave-per-tile kernel exists for them only)
    absgrad: bool = False
    compact_pending: bool = False      # a deferred overflow happened with the bins at their limit
    local_overflow_seen: int = 0       # replicas: entries this rank saw overflow since the last take-back
    n_tiles: int = 8160                # C x tiles of the views (default: one 1080p view); the backward kernel is chosen by it


def _round_up(x: int, m: int) -> int:
    return -(-int(x) // m) * m


MIN_TILES_FOR_TILE_WAVES = 3072     # 3 waves on each of the MI355X's 1024 SIMDs
TILES_PER_UNEVENNESS = (2000.0, 1500.0)   # one wave per tile: fullest <= tiles / 2000 x mean to enter, tiles / 1500 x to stay
TILE_WAVES_MIN_ENTRIES = (1.5e6, 1.1e6)   # ... and >= 1.5M list entries per step in all (mean x tiles) to enter, 1.1M to stay


def pick_raster_impl(now: int, mean_list: float, fullest: int, tile16: bool, absgrad: bool, first: bool = False,
                     n_tiles: int = 8160) -> int:
    """One wave per tile (1) where there is MUCH list work, spread EVENLY -- mean x tiles >= 1.5M entries (1080p: a mean of 184) and
    the fullest tile within min(6, tiles / 2000) x the mean -- else one wave per 8x8 quadrant (0); hysteresis once running (back to 0
    below 1.1M entries or beyond min(8, tiles / 1500) x).  The tile waves issue 22 % fewer instructions but walk a tile as one serial
    chain: they win once the chip is short of issue slots, not of waves -- measured crossovers (tools/gpu_r05_an.sh, uniform clouds):
    8160 tiles at a mean of ~190, 14400 tiles below 126 (-14 % there), 32400 tiles (4K) below 75 (-19 %): ~1.5M entries each time.
    Never on images of fewer than 3072 tiles: one wave per tile is then less than three waves per SIMD (tools/gpu_r05_y.sh, dense
    lists: 512 x 512 = 1024 tiles 368 us against 201 with four waves per tile; 960 x 540 346 / 305; 1440 x 720 = 4050 tiles 417 / 497)."""
    if not tile16 or absgrad or n_tiles < MIN_TILES_FOR_TILE_WAVES:
        return 0
    # How uneven the lists may be depends on how many tiles there are: with one wave per tile the kernel cannot end before its
    # fullest tile's serial chain does, and that chain outlasts the rest of the work once fullest > ~(tiles / 2000) x mean (a wave
    # alone takes ~350 ns per entry, the chip ~0.145 ns per entry and tile).  Measured (tools/gpu_r05_ah.sh): a 1440 x 720 panorama
    # from inside a 1M cloud, 4050 tiles, mean 377, fullest 1760 (4.7x): 623 us against 317 with quadrant waves; the same tile
    # count with even lists (1.9x): 417 against 497.
    enter = min(6.0, n_tiles / TILES_PER_UNEVENNESS[0])
    leave = min(8.0, n_tiles / TILES_PER_UNEVENNESS[1])
    entries = mean_list * n_tiles
    if entries >= TILE_WAVES_MIN_ENTRIES[0] and fullest <= enter * mean_list:
        return 1
    if first or entries < TILE_WAVES_MIN_ENTRIES[1] or fullest > leave * mean_list:
        return 0
    return now


def pick_tile_order(now: bool, impl: int, mean_list: float, fullest: int) -> bool:
    """Longest list first where a kernel's end is its longest tile: always with one wave per tile; with four waves per tile
    when the lists are SKEWED (fullest tile >= 512 entries and > 8x the mean; hysteresis: off below 384 / 6x) or simply LONG
    (mean >= 64 entries per tile, off below 48): the table costs one small launch (~13 us at 1080p) and the two rasterisers get it back
    from ~60 entries per tile on -- tools/gpu_r05_ak.sh: 500k at 1080p (mean 108) +1.6 %, 960 x 540 / 1M (607) +6.5 %, a 1440 x 720 panorama
    inside a 1M cloud +11 %, 512 x 512 dense +9 %; c2 (mean 32) would lose 0.8 %."""
    m = max(mean_list, 1.0)
    if impl == 1:
        return True
    if mean_list >= 64.0 or (now and mean_list >= 48.0):
        return True
    if fullest >= 512 and fullest > 8.0 * m:
        return True
    return bool(now and fullest >= 384 and fullest > 6.0 * m)


def pick_tile_order_kept(now: bool, mean_list: float) -> bool:
    """Longest list first from a table KEPT per view (so_step_desc.tile_order_ready: built once every few visits of a view, its
    ~13 us launch amortised): worth it from a mean of 24 entries per tile on (hysteresis 16) -- at c2 (mean 32) the two rasterisers
    give 7.6 us back (tools/gpu_r05_ak.sh), which the per-step table cost more than."""
    return mean_list >= (16.0 if now else 24.0)


def pick_bin_replicas(n_tiles: int, fullest: int = 0, mean_list: float = 0.0) -> int:
    """Copies of the per-tile bin counters (so_step_desc.bin_replicas), chosen when a workspace is built: the returning atomics of
    ONE counter serialise at ~230 ns, so what matters is how many entries the busiest counters take.  Few tiles: 8 copies up to
    2304 tiles (768 x 768), 4 up to 4608 (tools/gpu_r05_w.sh).  Many tiles: 1 -- the chip's atomic rate binds first -- unless one image
    region is hot (fullest list >= 1024 entries and > 16x the mean: a cloud gathered in a ninth of a 1080p image bins in 30 us
    instead of 50 with 4 copies, tools/gpu_r05_at.sh)."""
    if n_tiles <= 2304:
        return 8
    if n_tiles <= 4608:
        return 4
    if fullest >= 1024 and fullest > 16.0 * max(mean_list, 1.0):
        return 4
    return 1


def pick_bwd_segments(n_tiles: int, fullest: int, mean_list: float, now: bool = False):
    """(segments per tile, entries per segment) of the quadrant-wave backward rasteriser (so_step_desc.bwd_seg_len): eight segments where
    the kernel is bound by the serial chain of its fullest tiles -- fullest list >= 1024 entries and either few tiles (<= 2304) or one
    hot region (fullest > 16x mean; at 11x -- 400k Gaussians in a sixth of the image -- segments lose 3 %) -- else (1, 0).  Measured (tools/gpu_r05_ax.sh, backward us with 1 / 8 segments): a cloud gathered in
    a ninth of a 1080p image 123 -> 89, 512 x 512 dense 166 -> 146, 960 x 540 / 1M 209 -> 204; where lists are long EVERYWHERE on many tiles
    (a 1440 x 720 panorama inside a 1M cloud) segments only add workgroups: 237 -> 267, and on short lists the empty ones cost (c2: 70 -> 124)."""
    m = max(mean_list, 1.0)       # (hysteresis once running: down to 768 entries / 12x)
    if fullest >= (768 if now else 1024) and (n_tiles <= 2304 or fullest > (12.0 if now else 16.0) * m):
        return 8, max(256, -(-int(fullest) // 8 // 256) * 256)
    return 1, 0


def pick_sort_fold(now: bool, binned: bool, tile16: bool, fullest: int) -> bool:
    """The per-tile sort inside the forward rasteriser (one launch fewer) where lists are short EVERYWHERE: fullest tile <= 256
    entries, i.e. every workgroup sorts its list with ONE wave in registers; with hysteresis (back to the sort kernels above
    384).  Longer lists that turn up in between are still sorted correctly in there (<= 2048 by the whole workgroup; beyond
    that a slow scratch-free rank sort)."""
    if not binned or not tile16:
        return False
    return fullest <= (384 if now else 256)


def on_probe(s: ListState, fullest: int, mean_list: float, n_isects: int, headroom: int) -> List[Action]:
    """A forward-only pass on the first view of a workspace (headroom 8) or after a refinement (headroom 2) measured the lists."""
    acts: List[Action] = []
    if s.binned:
        impl = pick_raster_impl(s.raster_impl, mean_list, fullest, s.tile16, s.absgrad, first=True, n_tiles=s.n_tiles)
        lpt = pick_tile_order(False, impl, mean_list, fullest)
        fold = s.fold_allowed and pick_sort_fold(False, s.binned, s.tile16, fullest)
        if (impl, lpt, fold) != (s.raster_impl, s.lpt, s.fold):
            acts.append(("set_kernels", impl, lpt, fold))
        if headroom * fullest > s.bin_capacity:
            if 2 * fullest > s.bin_limit:
                return acts + [("fall_back_to_compact", fullest), ("restage",)]
            return acts + [("rebuild_bins", min(_round_up(8 * fullest, 256), s.bin_limit)), ("restage",)]
        return acts
    if 1.25 * n_isects > s.capacity:
        return [("grow", 2 * n_isects), ("restage",)]
    return acts


def on_lists(s: ListState, fullest: int, total: int, n_tiles: int) -> List[Action]:
    """{max, sum} of the per-tile list lengths of a training iteration two calls back (gathered on the device, published through
    host-mapped status words: no read-back).  Binned layout only."""
    if not s.binned or fullest <= 0:
        return []
    acts: List[Action] = []
    mean = total / max(n_tiles, 1)
    impl = pick_raster_impl(s.raster_impl, mean, fullest, s.tile16, s.absgrad, n_tiles=s.n_tiles)
    lpt = pick_tile_order(s.lpt, impl, mean, fullest)
    fold = s.fold_allowed and pick_sort_fold(s.fold, s.binned, s.tile16, fullest)
    if (impl, lpt, fold) != (s.raster_impl, s.lpt, s.fold):
        acts.append(("set_kernels", impl, lpt, fold))
    # bins kept at >= 2x the fullest tile, rebuilt at 8x BEFORE a tile overflows (a tile beyond the capacity HAS overflowed:
    # that is on_overflow's business)
    if 2 * fullest > s.bin_capacity and s.bin_capacity < s.bin_limit and fullest <= s.bin_capacity:
        acts.append(("rebuild_bins", min(_round_up(8 * fullest, 256), s.bin_limit)))
    return acts


def on_overflow(s: ListState, kind: str, n_prev: int, n_last: int, ov_last: bool) -> List[Action]:
    """The iteration before the last one overflowed its lists (found one step late; the device skipped its optimiser step).
    kind: what ran on the counters -- "train" or "render".  n_prev / n_last: entries needed by the two iterations in flight
    (binned: Gaussians over the fullest tile).  ov_last: the last iteration overflowed too."""
    needed = max(n_prev, n_last)
    at_limit = s.binned and s.bin_capacity >= s.bin_limit
    if kind != "train":                  # a forward-only render: no iteration to take back
        return [("fall_back_to_compact", n_last)] if at_limit else [("grow", needed)]
    if s.on_overflow == "raise":
        return [("raise", f"tile-intersection buffers overflowed ({needed} > capacity); the affected iterations were skipped on the device")]
    if s.on_overflow == "defer":
        return [("defer", max(s.local_overflow_seen, needed, s.bin_capacity if s.binned else 0), at_limit)]
    return [("take_back", 1 + (1 if ov_last else 0), needed, at_limit)]


def on_take_back(s: ListState, void: int, needed: int, grow: bool, compact: bool) -> List[Action]:
    """`void` iterations are taken back (single GPU: from on_overflow; replicas: Runner._dp_check_void from the summed flag).
    grow False: another rank's view overflowed, this rank's buffers held."""
    acts: List[Action] = [("void", void)]
    if not grow:
        return acts
    if s.binned and (compact or s.compact_pending or s.bin_capacity >= s.bin_limit):
        return acts + [("fall_back_to_compact", needed)]
    return acts + [("grow", needed)]
