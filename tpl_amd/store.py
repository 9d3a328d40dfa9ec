"""The two pieces of tpl's plumbing a caller of the accelerated engines needs:

* ``StructStoreRegistry`` -- the registry tpl's applications use to find their shared-memory stores
  (library/tpl/util.py:310-332).  With the ``structstore`` package present it hands out the same
  ``StructStoreShared`` segments, so planner / controller plugins built on tpl_amd publish into tpl's IPC unchanged;
  without it (this image) ``get`` returns an in-process store with the surface the plugins use -- attribute
  access, ``lock()``, ``revalidate()``, ``deepcopy()`` -- enough to run a planner or controller stand-alone.
* ``load_state`` -- a reader of the ``state.json`` files objtoolbox writes under ``data/params/**`` and
  ``data/scenarios/**`` (nested objects tagged with ``"__class__"``; ``numpy.ndarray`` as {dtype, data};
  ``__extern__`` as a path to ``extern/<path>.npy``), into plain attribute objects.
"""
import copy
import json
import os
import threading
from contextlib import contextmanager
from types import SimpleNamespace

import numpy as np

try:                                    # optional: tpl's shared-memory IPC
    import structstore as _sts
except ImportError:                     # pragma: no cover - depends on the image
    _sts = None


class LocalStore(SimpleNamespace):
    """In-process stand-in for ``structstore.StructStoreShared`` (single process, re-entrant lock)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        object.__setattr__(self, "_lock", threading.RLock())

    @contextmanager
    def lock(self):
        with self._lock:
            yield self

    def revalidate(self):
        return True

    def deepcopy(self):
        return copy.deepcopy(SimpleNamespace(**{k: v for k, v in vars(self).items() if k != "_lock"}))

    def __deepcopy__(self, memo):
        return LocalStore(**{k: copy.deepcopy(v, memo) for k, v in vars(self).items() if k != "_lock"})


class StructStoreRegistry:
    """util.py:310-332: one store per path, created on first use."""

    REGISTRY = {}
    SHM_DIR = f"/tmp/shm-{os.getuid()}/"

    @staticmethod
    def get(path, *args, **kwargs):
        if path.startswith("/"):
            path = path[1:]
        path = os.path.join(StructStoreRegistry.SHM_DIR, path)
        try:
            return StructStoreRegistry.REGISTRY[path]
        except KeyError:
            if _sts is not None:
                os.makedirs(StructStoreRegistry.SHM_DIR, exist_ok=True)
                store = _sts.StructStoreShared(path, *args, use_file=True, **kwargs)
            else:
                store = LocalStore()
            StructStoreRegistry.REGISTRY[path] = store
        return store

    @staticmethod
    def shared_memory_available():
        return _sts is not None


class Bundle(SimpleNamespace):
    """objtoolbox.utils.bundle / types.SimpleNamespace / any tagged class: attributes, plus the tag it came with."""

    def __init__(self, _class=None, **kw):
        super().__init__(**kw)
        object.__setattr__(self, "__class_tag__", _class)

    def __deepcopy__(self, memo):
        twin = type(self).__new__(type(self))
        memo[id(self)] = twin
        twin.__dict__.update((k, copy.deepcopy(v, memo)) for k, v in self.__dict__.items())
        return twin


def _decode(obj, base_dir):
    if isinstance(obj, list):
        return [_decode(v, base_dir) for v in obj]
    if not isinstance(obj, dict):
        return obj
    tag = obj.get("__class__")
    if tag == "numpy.ndarray":
        return np.array(obj.get("data", []), dtype=np.dtype(obj.get("dtype", "float64")))
    if tag == "__extern__":
        f = os.path.join(base_dir, "extern", obj["path"] + ".npy")
        return np.load(f, allow_pickle=False) if os.path.exists(f) else None
    fields = {k: _decode(v, base_dir) for k, v in obj.items() if k != "__class__"}
    if tag is None:
        return fields                                          # a plain dict stays a dict
    return Bundle(_class=tag, **fields)


def load_state(path):
    """``path``: a directory holding ``state.json`` (objtoolbox's layout) or the json file itself."""
    f = os.path.join(path, "state.json") if os.path.isdir(path) else path
    with open(f) as fd:
        raw = json.load(fd)
    return _decode(raw, os.path.dirname(os.path.abspath(f)))


def merge(dst, src):
    """objtoolbox.merge, recursively: values of ``src`` onto the attributes ``dst`` already has."""
    names = getattr(dst, "__slots__", None) or [k for k in vars(dst) if not k.startswith("_")]
    for k in names:
        if not hasattr(src, k):
            continue
        v = getattr(src, k)
        cur = getattr(dst, k)
        if isinstance(v, SimpleNamespace) and not isinstance(cur, (int, float, str, np.ndarray, list, type(None))):
            merge(cur, v)
        else:
            setattr(dst, k, v)
    return dst
