"""Import shim: the implementation lives in ``path-space-pde-solver_amd/`` (a directory name
that is not a Python identifier).  This package re-points its search path there."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          'path-space-pde-solver_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _fh:
    exec(compile(_fh.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
