"""Import shim: the package directory is named `unet-studio_amd/` (not a valid Python identifier), so
`import unet_studio_amd` resolves here and runs the package's __init__ with the right __path__."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "unet-studio_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
