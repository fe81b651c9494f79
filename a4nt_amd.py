"""Import alias: the package directory is named `a4nt-author-masking_amd` (not a valid Python
identifier); `import a4nt_amd` loads it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module('a4nt-author-masking_amd')
