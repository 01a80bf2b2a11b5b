"""Host logic of the operator path that needs no device: the registry of per-grid bins (splat_one_amd/raster_op.py)."""
import torch

from splat_one_amd import raster_op


class _FakeBins:
    def __init__(self, device, M):
        self.device, self.M, self.last_used = device, M, 0


def test_bins_are_kept_per_grid_and_the_registry_is_bounded(monkeypatch):
    monkeypatch.setattr(raster_op, "_Bins", _FakeBins)
    monkeypatch.setattr(raster_op._lib, "stream", lambda: 0)
    monkeypatch.setattr(raster_op, "_BINS", {})
    dev = torch.device("cuda", 0)
    a = raster_op._bins_for(dev, 30, (1, 10, 3))
    b = raster_op._bins_for(dev, 30, (2, 5, 3))          # the same tile count, another grid (fuzz seeds 2222 / 2344)
    assert a is not b and a.M == b.M == 30
    assert raster_op._bins_for(dev, 30, (1, 10, 3)) is a
    for i in range(raster_op._MAX_BINS - 2):             # fill the registry ...
        raster_op._bins_for(dev, 100 + i, (1, 100 + i, 1))
    assert len(raster_op._BINS) == raster_op._MAX_BINS
    raster_op._bins_for(dev, 30, (1, 10, 3))             # ... `a` is used again, `b` is now the grid used longest ago
    raster_op._bins_for(dev, 7, (1, 7, 1))
    assert len(raster_op._BINS) == raster_op._MAX_BINS
    grids = {k[3] for k in raster_op._BINS}
    assert (1, 10, 3) in grids and (2, 5, 3) not in grids and (1, 7, 1) in grids
    assert raster_op._bins_for(dev, 30, (2, 5, 3)) is not b   # an evicted grid starts again (and is measured again)
