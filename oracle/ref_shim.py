"""Import shim for the read-only reference (TEST INFRASTRUCTURE, this container only).

The reference (``/root/reference``, package ``kinematics`` 0.4.1) needs Python >= 3.12
for exactly two things: ``enum.StrEnum`` and five PEP-695 ``type X = ...`` statements.
This container has Python 3.10, so before any ``kinematics`` import we

* install a ``StrEnum`` backport on the ``enum`` module, and
* register a ``sys.meta_path`` finder whose loader rewrites ``type X = ...`` into
  ``X = ...`` when compiling ``kinematics*`` modules in memory.

Nothing is written to ``/root/reference`` (``sys.dont_write_bytecode``), and nothing of
the reference is copied into this repository: this module only exists so that
``oracle/gen_golden.py`` can *run* the reference here and dump input/output vectors.
It is never imported by the product package, by ``bench.py``'s GPU leg or by ``-m gpu``
tests, and ``/root/reference`` does not exist on the GPU box.
"""

from __future__ import annotations

import enum
import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys

REFERENCE_ROOT = os.environ.get("OKX_REFERENCE_ROOT", "/root/reference")
REFERENCE_SRC = os.path.join(REFERENCE_ROOT, "src")

_TYPE_STMT = re.compile(r"^type (\w+) = ", flags=re.M)


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "kinematics"))


class _Loader(importlib.machinery.SourceFileLoader):
    def source_to_code(self, data, path, *, _optimize=-1):  # type: ignore[override]
        text = data.decode("utf-8") if isinstance(data, (bytes, bytearray)) else data
        text = _TYPE_STMT.sub(r"\1 = ", text)
        return compile(text, path, "exec", dont_inherit=True, optimize=_optimize)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "kinematics" and not fullname.startswith("kinematics."):
            return None
        spec = importlib.machinery.PathFinder.find_spec(fullname, path)
        if spec is None or spec.origin is None or not spec.origin.endswith(".py"):
            return spec
        loader = _Loader(fullname, spec.origin)
        return importlib.util.spec_from_file_location(
            fullname,
            spec.origin,
            loader=loader,
            submodule_search_locations=spec.submodule_search_locations,
        )


_installed = False


def install() -> None:
    """Make ``import kinematics`` work against the read-only reference tree."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]
    sys.meta_path.insert(0, _Finder())
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    _installed = True
