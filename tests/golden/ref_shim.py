"""Import shim for RUNNING THE REFERENCE in the build container: ``s3prl.hub`` star-imports every hubconf, some of
which import ``torchaudio`` / ``omegaconf`` (un-installed, un-vendored dependencies that the wav2vec2 / HuBERT / WavLM /
DistilHuBERT forwards never touch).  ``install()`` registers a meta-path finder that fabricates empty placeholder
modules for those roots so the imports succeed.  Used only by ``tests/golden/make_golden.py`` and the
reference-dependent tests (skipped where ``/root/reference`` is absent, e.g. on the GPU box)."""
import sys, types, importlib.abc, importlib.machinery

class _Fake(types.ModuleType):
    __path__ = []
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        # class-like placeholder usable as base class / callable
        obj = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, obj)
        return obj

class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, roots): self.roots = roots
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
    def create_module(self, spec):
        return _Fake(spec.name)
    def exec_module(self, module): pass

def install(roots=("torchaudio",)):
    sys.meta_path.insert(0, _Finder(set(roots)))


REFERENCE = "/root/reference"


def import_reference(roots=("torchaudio", "omegaconf")):
    """Make ``import s3prl`` resolve to the reference tree (with the placeholder dependencies installed)."""
    import os

    if not os.path.isdir(REFERENCE):
        raise ImportError(f"{REFERENCE} is not present (the reference only exists in the build container)")
    install(roots)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
