"""`import surgical_world_model` for the reference's unchanged scripts.

The training scripts of HananKibria/VIT import the model file by its bare module name
(stage1_surgvu_pretraining.py:26, stage1_vjepa2_finetuning.py:40, load_vjepa2_pretrained.py:23).  Putting this
directory in front of the script's own on `sys.path` / `PYTHONPATH` makes that import resolve to the MI355X
implementation in the `vit_amd` package; nothing else about the scripts changes.

    PYTHONPATH=/path/to/repo/dropin python stage1_surgvu_pretraining.py ...
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:           # the package lives next to this directory
    sys.path.insert(1, _REPO)

from vit_amd.surgical_world_model import *  # noqa: E402,F401,F403
from vit_amd.surgical_world_model import __all__  # noqa: E402,F401
