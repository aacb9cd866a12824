"""Deterministic .npz writer for the golden fixtures (test infrastructure): np.savez stamps every zip member with the current
time, so regenerating a fixture changed its bytes even when every array was identical.  Members here carry a fixed date and are
written in sorted key order: `python oracle/gen_golden*.py` leaves `git status` clean unless an array really changed."""
import io
import zipfile

import numpy as np


def savez_deterministic(path: str, **arrays) -> None:
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for key in sorted(arrays):
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(arrays[key]), allow_pickle=False)
            info = zipfile.ZipInfo(key + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            zf.writestr(info, buf.getvalue())
