"""`import surgical_world_model` for the reference's unchanged scripts.

The training scripts of HananKibria/VIT import the model file by its bare module name
(stage1_surgvu_pretraining.py:26, stage1_vjepa2_finetuning.py:40, load_vjepa2_pretrained.py:23).  With this directory
on `sys.path` / `PYTHONPATH` that import resolves to the MI355X implementation in the `vit_amd` package wherever no
other `surgical_world_model.py` is found first: `python -c`, `python -m`, a notebook, a script kept outside the
reference's directory.

    PYTHONPATH=/path/to/repo/dropin python -c "from surgical_world_model import VideoViTEncoder"

A script launched as a file from INSIDE the reference's directory finds the reference's own model file first (a
script's directory leads `sys.path`, ahead of `PYTHONPATH`); run such a script through the launcher instead:

    PYTHONPATH=/path/to/repo python -m vit_amd.run /path/to/reference/stage1_vjepa2_finetuning.py ...
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:           # the package lives next to this directory
    sys.path.insert(1, _REPO)

from vit_amd.surgical_world_model import *  # noqa: E402,F401,F403
from vit_amd.surgical_world_model import __all__  # noqa: E402,F401
