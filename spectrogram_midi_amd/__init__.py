"""Import alias: the package directory is named `spectrogram-midi_amd/` (not a valid
Python identifier), so `import spectrogram_midi_amd` resolves to it through this shim."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "spectrogram-midi_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
