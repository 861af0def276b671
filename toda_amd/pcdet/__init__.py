"""Host-side mirror of the reference's `pcdet` operator surface (config, build_network,
Detector3DTemplate, module registries, dataset contract) for the voxel/pillar detection path.
`toda_amd.pcdet.install_as_pcdet()` registers it under the names `pcdet` and `spconv` so the
reference's entry points (`from pcdet.models import build_network`, ...) import it unchanged."""
import sys

__version__ = "0.5.2+toda_amd"


def install_as_pcdet():
    """Alias toda_amd.pcdet -> `pcdet` and toda_amd.spconv -> `spconv` / `spconv.pytorch`."""
    import importlib

    from .. import spconv as _sp

    me = sys.modules[__name__]
    sys.modules.setdefault("pcdet", me)
    for sub in ("config", "models", "datasets", "utils"):
        sys.modules.setdefault(f"pcdet.{sub}", importlib.import_module(f"{__name__}.{sub}"))
    sys.modules.setdefault("spconv", _sp)
    sys.modules.setdefault("spconv.pytorch", importlib.import_module("toda_amd.spconv.pytorch"))
    return me
