"""
Import shim: the reference is imported both as `src.gcn_grabcut` (inference.py:20-22)
and, with <repo>/src on sys.path, as `gcn_grabcut` (tests/test.py:12).  The real
package lives in gcn-grabcut_amd/gcn_grabcut; this module re-points its search
path there and runs that package's __init__ under this name.
"""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                                        "..", "..", "gcn-grabcut_amd", "gcn_grabcut"))
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"), globals())
del _os, _f
