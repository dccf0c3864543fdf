"""``tritonclient`` compatibility namespace.

Code written against the Triton client libraries keeps working unchanged:
``tritonclient.http``, ``tritonclient.grpc`` (and their ``aio`` / ``auth``
sub-packages), ``tritonclient.utils``, ``tritonclient.utils.shared_memory`` and
``tritonclient.utils.cuda_shared_memory`` resolve to the modules of
:mod:`triton_client_amd`.  The last one maps to ``hip_shared_memory`` (hipshm):
same function names, HIP IPC handles underneath.

Implementation: a meta-path finder aliases every ``tritonclient.X`` import to
``triton_client_amd.X`` so both names share one module object (class
identities, module level registries and ``isinstance`` checks stay coherent).
"""

import importlib
import importlib.abc
import importlib.util
import sys

_PREFIX = "tritonclient"
_TARGET = "triton_client_amd"
_RENAMES = {
    "tritonclient.utils.cuda_shared_memory": "triton_client_amd.utils.hip_shared_memory",
}


def _resolve(name):
    for old, new in _RENAMES.items():
        if name == old or name.startswith(old + "."):
            return new + name[len(old):]
    return _TARGET + name[len(_PREFIX):]


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self._target = target

    def create_module(self, spec):
        return importlib.import_module(self._target)

    def exec_module(self, module):
        pass  # already initialised under its real name


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX + "."):
            return None
        real = _resolve(fullname)
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real),
                                               is_package=real_spec.submodule_search_locations is not None)
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

from triton_client_amd import __version__  # noqa: E402,F401
