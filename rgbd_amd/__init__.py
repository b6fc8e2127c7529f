"""Importable alias of the `learning-based-rgb-d-image-compression_amd` package (its name is not an identifier).

`import rgbd_amd` and `from rgbd_amd import x` / `import rgbd_amd.x` resolve to the very same module objects as the real
package (no second copy of any submodule is created)."""
import importlib
import os
import sys

_REAL = "learning-based-rgb-d-image-compression_amd"
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
