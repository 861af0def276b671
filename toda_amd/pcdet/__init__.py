"""Host-side mirror of the reference's `pcdet` operator surface (config, build_network,
Detector3DTemplate, module registries, dataset contract) for the voxel/pillar detection path.
`toda_amd.pcdet.install_as_pcdet()` registers it under the names `pcdet` and `spconv` so the
reference's entry points (`from pcdet.models import build_network`, ...) import it unchanged."""
import importlib
import importlib.machinery
import importlib.util
import sys

__version__ = "0.5.2+toda_amd"


class _AliasFinder:
    """Meta-path finder + loader: `import <alias>.x.y` yields the module object of `<real>.x.y` (one object under two
    names, so relative imports inside the package keep resolving against the real package)."""

    def __init__(self, alias, real):
        self.alias, self.real = alias, real

    def _real_name(self, fullname):
        if fullname == self.alias or fullname.startswith(self.alias + "."):
            return self.real + fullname[len(self.alias):]
        return None

    def find_spec(self, fullname, path=None, target=None):
        real = self._real_name(fullname)
        if real is None:
            return None
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if real_spec is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=real_spec.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(self._real_name(spec.name))

    def exec_module(self, module):
        return None


def install_as_pcdet():
    """Alias toda_amd.pcdet -> `pcdet` and toda_amd.spconv -> `spconv` (every submodule, lazily, as the SAME module
    objects): `from pcdet.models import build_network`, `import spconv.pytorch as spconv`, `from pcdet.ops.iou3d_nms
    import iou3d_nms_utils`, ... then import this package."""
    me = sys.modules[__name__]
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder("pcdet", __name__))
        sys.meta_path.insert(0, _AliasFinder("spconv", __name__.rsplit(".", 1)[0] + ".spconv"))
    sys.modules.setdefault("pcdet", me)
    sys.modules.setdefault("spconv", importlib.import_module(__name__.rsplit(".", 1)[0] + ".spconv"))
    return me
