"""Run one of the reference's scripts, unchanged and from where it lies, on this package's model.

    python -m vit_amd.run /path/to/stage1_vjepa2_finetuning.py --data_root ... --pretrained_checkpoint adapted.pth

The scripts import the model by its bare module name (`from surgical_world_model import
SurgicalActionConditionedWorldModel`, stage1_surgvu_pretraining.py:26, stage1_vjepa2_finetuning.py:40;
`... import SurgicalActionConditionedWorldModel, VideoViTEncoder`, load_vjepa2_pretrained.py:23) and live in the same
directory as the reference's own `surgical_world_model.py`.  Python puts a script's directory in front of everything
`PYTHONPATH` names, so a directory on `PYTHONPATH` can never win that import.  This launcher therefore registers
`vit_amd.surgical_world_model` in `sys.modules` under the bare name first (an entry there is found before any path is
searched, also by the modules the script imports in turn, e.g. stage1_vjepa2_finetuning.py importing
stage1_surgvu_pretraining.py), then runs the script as `__main__` in this same interpreter with the `sys.argv` and
`sys.path[0]` it would have had under `python script.py`.  Nothing is re-executed and no GPU call is made before the
script's own code runs.
"""
import os
import runpy
import sys

MODULE_NAME = "surgical_world_model"


def install():
    """Make `import surgical_world_model` resolve to this package's module; returns that module."""
    from vit_amd import surgical_world_model as module
    sys.modules[MODULE_NAME] = module
    return module


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        sys.stderr.write("usage: python -m vit_amd.run SCRIPT.py [script arguments ...]\n")
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        sys.stderr.write(f"vit_amd.run: no such script: {argv[0]}\n")
        return 2
    install()
    sys.argv = [script] + argv[1:]
    # what `python script.py` does: the script's directory leads the search path (its sibling modules stay importable)
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
