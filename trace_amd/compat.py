"""Import-name shim: the reference's drivers import `Trace.trace.<module>` (repo directory named `Trace` with its
parent on sys.path: scripts/inference/inference.py:5-12, trace/eval/evaluate.py:15-24).  `install()` maps those
names onto this package so the drivers run unchanged."""
import importlib
import sys
import types

_MAP = {
    "Trace.trace": "trace_amd",
    "Trace.trace.constants": "trace_amd.constants",
    "Trace.trace.conversation": "trace_amd.conversation",
    "Trace.trace.mm_utils": "trace_amd.mm_utils",
    "Trace.trace.model": "trace_amd.model",
    "Trace.trace.model.builder": "trace_amd.model.builder",
}


def install():
    pkg = types.ModuleType("Trace")
    pkg.__path__ = []
    sys.modules.setdefault("Trace", pkg)
    for alias, real in _MAP.items():
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
    sys.modules["Trace"].trace = sys.modules["Trace.trace"]
    return sorted(_MAP)
