"""Import-site stand-in for `gsplat.compression` (the reference imports `PngCompression` at
/root/reference/utils/gsplat_utils/gsplat_trainer.py:45 and instantiates it only when
`cfg.compression == "png"`, :355-360; `run_compression`, :903-914, calls `.compress(dir, splats)` /
`.decompress(dir)`).

The PNG / k-means export is storage, outside the rasterisation path this package replaces
(DESIGN.md "out of scope").  On an AMD box `gsplat` does not exist, so
without this module the reference's trainer would fail at import after the swap of INTEGRATION.md
section 2; with it the import resolves and the class refuses to be USED, loudly, at construction --
`cfg.compression = None` (the reference's default, :106) never reaches it.
"""
from __future__ import annotations

from typing import Any, Dict

_MSG = ("splat_one_amd.compression.PngCompression: the PNG / k-means export of the reference "
        "(gsplat_trainer.py:903-914) is storage, not part of the MI355X rasterisation path -- "
        "leave Config.compression = None (its default)")


class PngCompression:
    """Same constructor keywords as gsplat's class (`use_sort`, `verbose`); raises on construction."""

    def __init__(self, use_sort: bool = True, verbose: bool = True, **kwargs: Any) -> None:
        raise NotImplementedError(_MSG)

    def compress(self, compress_dir: str, splats: Dict[str, Any]) -> None:      # pragma: no cover (unreachable)
        raise NotImplementedError(_MSG)

    def decompress(self, compress_dir: str) -> Dict[str, Any]:                  # pragma: no cover (unreachable)
        raise NotImplementedError(_MSG)
