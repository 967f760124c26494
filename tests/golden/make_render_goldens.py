"""Generate golden vectors for target-heatmap rendering by running the REFERENCE implementation
(`/root/reference/detector/input_pipeline/heatmap_creation.py:6-72`, `get_heatmaps`).

The reference module is imported by file path (it needs only math, numpy, scipy.signal; the package `__init__` that
pulls in TensorFlow is bypassed); nothing from it is copied into this repo - only the outputs it produced are stored,
as sparse (index, value) pairs + shape, in `tests/golden/render_goldens.npz`.  numpy/scipy versions used: see the
`versions` entry of the file (promotion rules of numpy >= 2 apply to the float32 scalar arithmetic).

Run (in the build container, where /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_render_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

REF = os.environ.get("MPN_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_goldens.npz")


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "_ref_heatmap_creation", os.path.join(REF, "detector", "input_pipeline", "heatmap_creation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.get_heatmaps


from render_cases import cases  # noqa: E402


def main():
    get_heatmaps = load_reference()
    out = {}
    names = []
    for name, kp, boxes, width, height, ds in cases():
        hm = get_heatmaps(kp.copy(), boxes.copy(), width, height, ds)
        assert hm.dtype == np.float32 and hm.ndim == 3 and hm.shape[2] == 17
        flat = hm.ravel()
        nz = np.flatnonzero(flat)
        out[f"{name}/shape"] = np.array(hm.shape, np.int64)
        out[f"{name}/index"] = nz.astype(np.int32)
        out[f"{name}/value"] = flat[nz]
        names.append(name)
    out["names"] = np.array(names)
    out["versions"] = np.array([np.__version__, scipy.__version__])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(names), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
