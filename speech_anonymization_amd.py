"""Import alias: the package directory is ``speech-anonymization_amd/`` (a hyphen is not a
valid Python identifier), so ``import speech_anonymization_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "speech-anonymization_amd")
_spec = importlib.util.spec_from_file_location(
    "speech_anonymization_amd", os.path.join(_dir, "__init__.py"),
    submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["speech_anonymization_amd"] = _mod
_spec.loader.exec_module(_mod)
