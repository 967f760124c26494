"""Generate golden vectors for the heatmap peak decode by running the REFERENCE
implementation (`/root/reference/inference/utils.py:29-52`, `get_keypoints`).

The reference module is imported by file path (it only needs numpy + PIL); nothing
from it is copied into this repo - only inputs and the outputs it produced are
stored in `tests/golden/decode_goldens.npz`.

Run (in the build container, where /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_decode_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

REF = os.environ.get("MPN_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_goldens.npz")


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "_ref_inference_utils", os.path.join(REF, "inference", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.get_keypoints


from decode_cases import cases  # noqa: E402


def main():
    get_keypoints = load_reference()
    out = {}
    names = []
    for name, hm, box, thr in cases():
        kp = get_keypoints(hm, box, thr)
        assert kp.dtype == np.int32 and kp.shape == (17, 3)
        out[f"{name}/keypoints"] = kp
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(names), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
