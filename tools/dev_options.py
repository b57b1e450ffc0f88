"""Development aid for the scripts in this directory: scene options from CRT_* environment
variables, so that one script can be run with different settings from the shell.  The library
itself never reads the environment; this only fills Scene.set_options()."""
import os

_ACCEL = {"": 0, "auto": 0, "reference": 1, "ordered": 2}


def from_env(env=os.environ):
    opt = {}
    if "CRT_ACCEL" in env:
        opt["accel"] = _ACCEL[env["CRT_ACCEL"]]
    if "CRT_MODE" in env:
        opt["mode"] = int(env["CRT_MODE"])
    if env.get("CRT_FAST_RCP") == "0":
        opt["ieee_division"] = 1
    for var, name in (("CRT_TARGET_WAVES", "work_items"), ("CRT_BLOCKS", "blocks"), ("CRT_READY_PERCENT", "ready_percent"),
                      ("CRT_PRIM_BATCH", "prim_batch"), ("CRT_DETERMINISTIC", "deterministic"), ("CRT_WAVEFRONT", "wavefront"),
                      ("CRT_WAVEFRONT_PATHS", "wavefront_paths"), ("CRT_FLAT_CUBOIDS", "flat_cuboids"), ("CRT_GENERAL_KERNEL", "general_kernel"), ("CRT_EXACT_TIES", "exact_ties"), ("CRT_ITEM_ORDER", "item_order"), ("CRT_CHIP_PARTS", "chip_parts")):
        if var in env:
            opt[name] = int(env[var])
    return opt
