"""The fused engine's list-policy state machine, row by row, on the CPU.

splat_one_amd/list_policy.py holds the decisions as pure functions; `FusedEngine._apply` executes the actions they return.
TABLE below IS the table of DESIGN.md section 4.3 (one row per (state, event) pair that behaves differently); the first test
walks it against the functions, the second drives `FusedEngine._apply` / `take_back` with a stubbed engine (no library, no device)
and checks that every action kind reaches the method that implements it, in order."""
import types

import pytest

from splat_one_amd import list_policy as LP

S = LP.ListState
BINNED = dict(binned=True, bin_capacity=1024, bin_limit=8192, capacity=8160 * 1024, raster_impl=0, lpt=False, on_overflow="grow")
SHORT = dict(BINNED, fold=True)          # short lists everywhere: the sort runs in the forward rasteriser's prologue
AT_LIMIT = dict(BINNED, bin_capacity=8192)
COMPACT = dict(binned=False, bin_capacity=0, bin_limit=0, capacity=1 << 20, raster_impl=0, lpt=False, on_overflow="grow")

# (row, state, event, arguments, expected actions)
TABLE = [
    # ---- capacity probe of a new workspace (headroom 8) / after a refinement (headroom 2)
    ("probe: short even lists -> the sort moves into the forward rasteriser", BINNED, "probe", dict(fullest=97, mean_list=31.0, n_isects=0, headroom=8),
     [("set_kernels", 0, False, True)]),
    ("probe: short even lists, already so", SHORT, "probe", dict(fullest=97, mean_list=31.0, n_isects=0, headroom=2), []),
    ("probe: short even lists, prologue sort not allowed (the default)", dict(BINNED, fold_allowed=False), "probe",
     dict(fullest=97, mean_list=31.0, n_isects=0, headroom=8), []),
    ("probe: bins too small -> 8x the fullest tile", SHORT, "probe", dict(fullest=200, mean_list=60.0, n_isects=0, headroom=8),
     [("rebuild_bins", 1792), ("restage",)]),
    ("probe: long lists everywhere -> one wave per tile, longest first", BINNED, "probe", dict(fullest=900, mean_list=520.0, n_isects=0, headroom=2),
     [("set_kernels", 1, True, False), ("rebuild_bins", 7424), ("restage",)]),
    ("probe: long lists everywhere on an image of few tiles (512 x 512) -> quadrant waves: one wave per tile would be one wave per SIMD",
     dict(BINNED, n_tiles=1024), "probe", dict(fullest=900, mean_list=520.0, n_isects=0, headroom=2),
     [("set_kernels", 0, True, False), ("rebuild_bins", 7424), ("restage",)]),
    ("probe: long but UNEVEN lists on 4050 tiles (a panorama from inside the cloud: fullest 4.7x the mean) -> quadrant waves: the fullest "
     "tile's chain would outlast everything else", dict(BINNED, n_tiles=4050, bin_capacity=16384, bin_limit=65536), "probe",
     dict(fullest=1760, mean_list=377.0, n_isects=0, headroom=2), [("set_kernels", 0, True, False)]),
    ("probe: the same tile count with even lists (1.9x) -> one wave per tile", dict(BINNED, n_tiles=4050, bin_capacity=16384, bin_limit=65536), "probe",
     dict(fullest=1088, mean_list=563.0, n_isects=0, headroom=2), [("set_kernels", 1, True, False)]),
    ("probe: skewed lists -> quadrant waves, longest first", BINNED, "probe", dict(fullest=1000, mean_list=30.0, n_isects=0, headroom=2),
     [("set_kernels", 0, True, False), ("rebuild_bins", 8192), ("restage",)]),
    ("probe: fullest tile beyond the bin budget -> compact lists", BINNED, "probe", dict(fullest=5000, mean_list=40.0, n_isects=0, headroom=8),
     [("set_kernels", 0, True, False), ("fall_back_to_compact", 5000), ("restage",)]),
    ("probe: absgrad keeps the quadrant kernel (long lists: longest first)", dict(BINNED, absgrad=True), "probe", dict(fullest=900, mean_list=520.0, n_isects=0, headroom=2),
     [("set_kernels", 0, True, False), ("rebuild_bins", 7424), ("restage",)]),
    ("probe (compact): buffers hold 1.25x the count", COMPACT, "probe", dict(fullest=0, mean_list=0.0, n_isects=800_000, headroom=8), []),
    ("probe (compact): too small -> 2x the count", COMPACT, "probe", dict(fullest=0, mean_list=0.0, n_isects=900_000, headroom=8),
     [("grow", 1_800_000), ("restage",)]),
    # ---- list statistics published by the device (no read-back), binned layout
    ("lists: nothing to do", BINNED, "lists", dict(fullest=400, total=8160 * 40, n_tiles=8160), []),
    ("lists: long lists (mean >= 64) with quadrant waves -> longest list first", BINNED, "lists", dict(fullest=400, total=8160 * 100, n_tiles=8160),
     [("set_kernels", 0, True, False)]),
    ("lists: hysteresis keeps the longest-first order down to a mean of 48", dict(BINNED, lpt=True), "lists", dict(fullest=300, total=8160 * 50, n_tiles=8160), []),
    ("lists: ... and drops it below", dict(BINNED, lpt=True), "lists", dict(fullest=300, total=8160 * 40, n_tiles=8160), [("set_kernels", 0, False, False)]),
    ("lists: hysteresis keeps the sort in the rasteriser up to 384", SHORT, "lists", dict(fullest=300, total=8160 * 40, n_tiles=8160), []),
    ("lists: lists outgrew the prologue sort -> back to the sort kernels", SHORT, "lists", dict(fullest=400, total=8160 * 40, n_tiles=8160),
     [("set_kernels", 0, False, False)]),
    ("lists: headroom below 2x -> rebuild at 8x before a tile overflows", BINNED, "lists", dict(fullest=600, total=8160 * 150, n_tiles=8160),
     [("set_kernels", 0, True, False), ("rebuild_bins", 4864)]),
    ("lists: a tile beyond the capacity is the overflow path's business (bins untouched)", BINNED, "lists", dict(fullest=1500, total=8160 * 250, n_tiles=8160),
     [("set_kernels", 0, True, False)]),
    ("lists: one hot tile -> longest list first with the quadrant waves (and roomier bins)", BINNED, "lists", dict(fullest=900, total=8160 * 60, n_tiles=8160),
     [("set_kernels", 0, True, False), ("rebuild_bins", 7424)]),
    ("lists: grown to long lists -> switch kernels (graphs dropped)", BINNED, "lists", dict(fullest=500, total=8160 * 300, n_tiles=8160),
     [("set_kernels", 1, True, False)]),
    ("lists: grown to long lists on an image of few tiles -> the quadrant waves stay", dict(BINNED, n_tiles=2040), "lists",
     dict(fullest=500, total=2040 * 300, n_tiles=2040), [("set_kernels", 0, True, False)]),
    ("lists: hysteresis keeps one wave per tile at 150 entries", dict(BINNED, raster_impl=1, lpt=True), "lists", dict(fullest=400, total=8160 * 150, n_tiles=8160), []),
    ("lists: back to quadrant waves below 1.1M entries (1080p: a mean of 135)", dict(BINNED, raster_impl=1, lpt=True), "lists", dict(fullest=400, total=8160 * 120, n_tiles=8160),
     [("set_kernels", 0, True, False)]),
    ("lists: a 4K view (32400 tiles) moves to one wave per tile at a mean of 75", dict(BINNED, n_tiles=32400, lpt=True), "lists",
     dict(fullest=300, total=32400 * 75, n_tiles=32400), [("set_kernels", 1, True, False)]),
    ("lists: bins at their limit are left alone", AT_LIMIT, "lists", dict(fullest=6000, total=8160 * 700, n_tiles=8160), [("set_kernels", 0, True, False)]),
    ("lists (compact): ignored", COMPACT, "lists", dict(fullest=600, total=8160 * 150, n_tiles=8160), []),
    # ---- an overflow found one step late
    ("overflow, grow: one void iteration", BINNED, "overflow", dict(kind="train", n_prev=1500, n_last=1500, ov_last=False), [("take_back", 1, 1500, False)]),
    ("overflow, grow: the last iteration overflowed too", BINNED, "overflow", dict(kind="train", n_prev=1500, n_last=1600, ov_last=True),
     [("take_back", 2, 1600, False)]),
    ("overflow, grow, bins at their limit -> take back, then compact lists", AT_LIMIT, "overflow", dict(kind="train", n_prev=9000, n_last=9000, ov_last=True),
     [("take_back", 2, 9000, True)]),
    ("overflow, raise", dict(BINNED, on_overflow="raise"), "overflow", dict(kind="train", n_prev=1500, n_last=1500, ov_last=False), "raise"),
    ("overflow, raise, bins at their limit: raises too (ADVICE r4)", dict(AT_LIMIT, on_overflow="raise"), "overflow",
     dict(kind="train", n_prev=9000, n_last=9000, ov_last=False), "raise"),
    ("overflow, defer: remembered, nothing changes", dict(BINNED, on_overflow="defer"), "overflow", dict(kind="train", n_prev=1500, n_last=1500, ov_last=False),
     [("defer", 1500, False)]),
    ("overflow, defer, bins at their limit: the fall-back is left to take_back", dict(AT_LIMIT, on_overflow="defer"), "overflow",
     dict(kind="train", n_prev=9000, n_last=9000, ov_last=False), [("defer", 9000, True)]),
    ("overflow of a forward-only render: grow, nothing to take back", dict(BINNED, on_overflow="raise"), "overflow",
     dict(kind="render", n_prev=1500, n_last=1700, ov_last=True), [("grow", 1700)]),
    ("overflow of a render with the bins at their limit -> compact lists", AT_LIMIT, "overflow", dict(kind="render", n_prev=9000, n_last=9100, ov_last=False),
     [("fall_back_to_compact", 9100)]),
    ("overflow (compact), grow", COMPACT, "overflow", dict(kind="train", n_prev=2_000_000, n_last=1_900_000, ov_last=False), [("take_back", 1, 2_000_000, False)]),
    # ---- take-back (single GPU: from the overflow row; replicas: Runner._dp_check_void, from the flag summed over the ranks)
    ("take back: grow the bins", BINNED, "take_back", dict(void=2, needed=1500, grow=True, compact=False), [("void", 2), ("grow", 1500)]),
    ("take back on a replica whose own buffers held", dict(BINNED, on_overflow="defer"), "take_back", dict(void=2, needed=0, grow=False, compact=False), [("void", 2)]),
    ("take back with the bins at their limit -> compact lists", AT_LIMIT, "take_back", dict(void=1, needed=9000, grow=True, compact=False),
     [("void", 1), ("fall_back_to_compact", 9000)]),
    ("take back after a deferred overflow at the limit", dict(BINNED, on_overflow="defer", compact_pending=True), "take_back",
     dict(void=2, needed=9000, grow=True, compact=False), [("void", 2), ("fall_back_to_compact", 9000)]),
    ("take back (compact)", COMPACT, "take_back", dict(void=1, needed=2_000_000, grow=True, compact=False), [("void", 1), ("grow", 2_000_000)]),
]
EVENTS = {"probe": LP.on_probe, "lists": LP.on_lists, "overflow": LP.on_overflow, "take_back": LP.on_take_back}


@pytest.mark.parametrize("row", TABLE, ids=[r[0] for r in TABLE])
def test_list_policy_row(row):
    name, state, event, args, want = row
    got = EVENTS[event](S(**state), **args)
    if want == "raise":
        assert len(got) == 1 and got[0][0] == "raise" and "overflowed" in got[0][1], got
    else:
        assert got == want, (name, got)


def test_every_event_and_action_kind_is_in_the_table():
    kinds = set()
    for _, state, event, args, want in TABLE:
        kinds.update(a[0] for a in EVENTS[event](S(**state), **args))
    assert kinds == {"set_kernels", "rebuild_bins", "grow", "fall_back_to_compact", "take_back", "void", "defer", "raise", "restage"}
    assert {r[2] for r in TABLE} == set(EVENTS)


def test_engine_executes_the_actions_in_order_with_a_stubbed_engine():
    """FusedEngine._apply / take_back on a stand-in object: no library, no device -- which method each action reaches."""
    import warnings
    from splat_one_amd.engine import FusedEngine
    log = []
    eng = types.SimpleNamespace(
        cfg={"raster_impl": 0, "tile_size": 16, "absgrad": False}, _lpt=False, _fold=True, sort_fold_ok=True, binned=True, bin_capacity=8192, _bin_limit=8192, capacity=1,
        on_overflow="grow", M=8160, bin_replicas=1, _compact_pending=False, _local_overflow_seen=0, _graph=1, _graph_fb=1, _graph_opt=1, _graphs={1: 1}, _graphs_fb={1: 1},
        _graphs_head={1: 1}, _rows_desc=1, _bin_hint=None, _probe_capacity=True,
        _build_workspace=lambda: log.append("build"), _grow=lambda n: log.append(("grow", n)),
        _fall_back_to_compact_lists=lambda n: log.append(("compact", n)), _void=lambda n: log.append(("void", n)))
    eng._list_state = lambda: FusedEngine._list_state(eng)
    eng._apply = lambda acts: FusedEngine._apply(eng, acts)
    eng.take_back = lambda *a, **k: FusedEngine.take_back(eng, *a, **k)
    assert FusedEngine._apply(eng, [("set_kernels", 1, True, False), ("rebuild_bins", 2048), ("restage",)]) is True
    assert eng.cfg["raster_impl"] == 1 and eng._lpt and not eng._fold and eng._graphs == {} and eng._graph is None and eng._rows_desc is None
    assert eng._bin_hint == 2048 and eng._probe_capacity is False and log == ["build"]
    del log[:]
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        # an overflow with the bins at their limit, policy "grow": void iterations taken back FIRST, then the compact lists
        assert FusedEngine._apply(eng, LP.on_overflow(eng._list_state(), "train", 9000, 9000, True)) is False
    assert log == [("void", 2), ("compact", 9000)] and any("memory budget" in str(w.message) for w in caught)
    del log[:]
    eng.on_overflow = "defer"
    FusedEngine._apply(eng, LP.on_overflow(eng._list_state(), "train", 9000, 9000, False))
    assert log == [] and eng._compact_pending and eng._local_overflow_seen == 9000          # nothing changes until the caller takes back
    eng.bin_capacity = 1024                                                                  # (even if the bins could grow: the pending flag wins)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng.take_back(2, 9000, grow=True)
    assert log == [("void", 2), ("compact", 9000)] and not eng._compact_pending
    eng.on_overflow = "raise"
    with pytest.raises(RuntimeError, match="overflowed"):
        FusedEngine._apply(eng, LP.on_overflow(eng._list_state(), "train", 1500, 1500, False))


@pytest.mark.parametrize("n_tiles,fullest,mean,expect", [(1024, 0, 0.0, 8), (2304, 900, 500.0, 8), (2305, 0, 0.0, 4), (4608, 0, 0.0, 4), (8160, 0, 0.0, 1),
                                                          (8160, 900, 512.0, 1), (8160, 1376, 33.0, 4), (8160, 1000, 33.0, 1), (8160, 2000, 200.0, 1),
                                                          (32400, 5000, 75.0, 4)])
def test_bin_counter_copies_by_tile_count_and_hot_regions(n_tiles, fullest, mean, expect):
    """so_step_desc.bin_replicas: by the tile count, and on large images only where one region is hot."""
    assert LP.pick_bin_replicas(n_tiles, fullest, mean) == expect


@pytest.mark.parametrize("n_tiles,fullest,mean,expect", [(1024, 1568, 520.0, (8, 256)), (1024, 900, 500.0, (1, 0)), (8160, 1376, 33.0, (8, 256)),
                                                          (8160, 900, 512.0, (1, 0)), (4050, 1760, 377.0, (1, 0)), (8160, 5000, 40.0, (8, 768)), (8160, 1100, 100.0, (1, 0)),
                                                          (2040, 1568, 607.0, (8, 256))])
def test_backward_segments_where_the_fullest_tiles_chain_binds(n_tiles, fullest, mean, expect):
    """so_step_desc.bwd_seg_len: few tiles or one hot region, and lists of >= 1024 entries."""
    assert LP.pick_bwd_segments(n_tiles, fullest, mean) == expect


def test_tile_tables_kept_per_view_follow_visits_age_and_size_with_a_stubbed_engine(monkeypatch):
    """FusedEngine._pick_order_mode / _keep_order_table on a stand-in object (no library, no device): a view's table is built at its
    first visit, handed back at the next `order_refresh - 1`, rebuilt at the visit after those, when it is more than `order_max_age`
    iterations old, and when the tile count it was built for is not this workspace's; without a view key nothing is kept."""
    import torch
    from splat_one_amd import engine as E
    monkeypatch.setattr(E._lib, "ptr", lambda t: 1000 + t.numel())
    eng = types.SimpleNamespace(_lpt=False, tile_order_lpt=True, _order_key=("img", 3), _list_stats=(0, 32.0), binned=True, cfg={"tile_size": 16},
                                _lpt_kept=False, _order_cache={}, _order_mode="none", M=12, order_refresh=4, order_max_age=2000, steps_done=0,
                                ws={"tile_order": torch.arange(12, dtype=torch.int32)}, device="cpu")
    pick = lambda: E.FusedEngine._pick_order_mode(eng)
    keep = lambda: E.FusedEngine._keep_order_table(eng)
    modes = []
    for it in range(9):                                   # nine visits of one view, refresh every 4th
        eng.steps_done = it
        src = pick()
        modes.append(eng._order_mode)
        assert (src != 0) == (eng._order_mode == "kept")
        keep()
    assert modes == ["build", "kept", "kept", "kept", "build", "kept", "kept", "kept", "build"], modes
    assert torch.equal(eng._order_cache[("img", 3)][0], eng.ws["tile_order"]) and eng._order_cache[("img", 3)][2] == 8
    eng.steps_done = 8 + 2001                             # too old: rebuilt although only one visit has used it
    assert pick() == 0 and eng._order_mode == "build"
    keep()
    eng.M, eng.ws = 20, {"tile_order": torch.arange(20, dtype=torch.int32)}   # another workspace size: the kept table is not handed out
    assert pick() == 0 and eng._order_mode == "build"
    keep()
    assert eng._order_cache[("img", 3)][0].numel() == 20
    assert pick() == 1020 and eng._order_mode == "kept"
    eng._order_key = None                                 # no key (a caller that names no view): nothing kept, and short lists need no table
    assert pick() == 0 and eng._order_mode == "none"
    eng._lpt = True                                       # ... long lists: a table every step
    assert pick() == 0 and eng._order_mode == "each"
    eng._order_key, eng._lpt, eng._list_stats = ("img", 4), False, (0, 10.0)   # lists too short for a kept table to pay (hysteresis: on from 24, off below 16)
    assert pick() == 0 and eng._order_mode == "none"
    assert E.FusedEngine._order_variants(eng, (1, 2, 0)) == [("none", (1, 2, 0))]
    eng._order_mode = "build"
    assert E.FusedEngine._order_variants(eng, (1, 2, 1)) == [("build", (1, 2, 1)), ("kept", (1, 2, 2))]
