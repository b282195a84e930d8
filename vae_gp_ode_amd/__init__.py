"""Importable alias of the ``vae-gp-ode_amd/`` package directory.

The product directory name carries a hyphen (it is named after the reference repository), which
Python cannot import by name; this stub makes ``import vae_gp_ode_amd`` resolve to it.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'vae-gp-ode_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f, _real
