"""Import shim: the package directory is named ``drone-oa-iree-vit-accelerator_amd`` (with
hyphens, as the build contract names it), which Python cannot import directly.  This module
turns itself into that package: ``import drone_oa_iree_vit_accelerator_amd as ita`` and
``from drone_oa_iree_vit_accelerator_amd import host, params, synth`` both work."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "drone-oa-iree-vit-accelerator_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
