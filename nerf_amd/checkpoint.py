"""Reader/writer for the reference's weight checkpoints (run_nerf.py:838-846 save, :244-261 load).

Format: `np.save(path, net.get_weights())` -- a .npy file holding a 1-D object array whose elements
are the fp32 arrays in get_weights() order, i.e. an .npy header followed by a pickle stream.  The
file names are `{model,model_fine,optimizer}_{step:06d}.npy`.

The reference loads them with np.load(allow_pickle=True), which executes whatever the pickle says.
This reader walks the same stream with an unpickler that can build NumPy arrays and nothing else,
so a checkpoint from an untrusted source cannot run code.
"""
import io
import os
import pickle

import numpy as np
from numpy.lib import format as npformat

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"),
    ("numpy", "dtype"),
}


class _ArraysOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}; only NumPy arrays are allowed")


def load_weights(path):
    """Returns the list of arrays stored in a reference-format checkpoint."""
    with open(path, "rb") as f:
        version = npformat.read_magic(f)
        if version == (1, 0):
            shape, _, dtype = npformat.read_array_header_1_0(f)
        elif version == (2, 0):
            shape, _, dtype = npformat.read_array_header_2_0(f)
        else:
            raise ValueError(f"{path}: unsupported .npy version {version}")
        if dtype.hasobject:
            arr = _ArraysOnlyUnpickler(io.BytesIO(f.read())).load()
            items = list(arr)
        else:
            # equal-shaped arrays would have been stacked into one plain array
            f.seek(0)
            items = list(np.load(f, allow_pickle=False))
    return [np.asarray(a, dtype=np.float32) for a in items]


def save_weights(net, prefix, i, basedir, expname):
    """run_nerf.py:838-842: np.save('{basedir}/{expname}/{prefix}_{i:06d}.npy', net.get_weights())."""
    path = os.path.join(basedir, expname, '{}_{:06d}.npy'.format(prefix, i))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    ws = net.get_weights()
    obj = np.empty((len(ws),), dtype=object)
    for k, w in enumerate(ws):
        obj[k] = np.asarray(w)
    np.save(path, obj, allow_pickle=True)
    print('saved weights at', path)
    return path
