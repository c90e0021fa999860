#!/usr/bin/env python
"""Generates tests/golden/scene_libhdf5*.h5 with the REAL HDF5 library (h5import / h5repack of HDF5 1.10, found under
/opt/conda/bin in the build container) in the layout prepare_scenes.save_scene writes (prepare_scenes.py:16-47):
  /map/<seq>/pose/<jjj> f64[4,4], /map/<seq>/point_cloud/<jjj> f32[n, 3+C], /scans/<seq>/{pose, point_cloud}
so that vfmreg.h5lite's reader is pinned against files it did not write.  Three variants of the same content:
  scene_libhdf5.h5          contiguous datasets, symbol-table groups (what h5py's create_dataset(data=...) yields)
  scene_libhdf5_gzip.h5     chunked + shuffle + gzip (h5repack -f SHUF -f GZIP=4 -l CHUNK=...)
  scene_libhdf5_latest.h5   --high=2 (1.10 "latest" format: superblock 3, version-2 object headers, compact link groups)
The expected arrays are stored next to them in scene_libhdf5_expected.npz.   Run:  python tests/golden/make_h5_fixture.py
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BIN = Path("/opt/conda/bin")


def main():
    rng = np.random.default_rng(2024)
    C = 6
    tree = {}
    for j in range(11):  # > 8 entries: the group index needs more than one symbol node
        tree[f"map/seqA/pose/{j:03}"] = rng.standard_normal((4, 4))
        tree[f"map/seqA/point_cloud/{j:03}"] = rng.standard_normal((5 + j, 3 + C)).astype(np.float32)
    for s in ("2012-02-04", "seqB"):
        tree[f"scans/{s}/pose"] = rng.standard_normal((4, 4))
        tree[f"scans/{s}/point_cloud"] = rng.standard_normal((9, 3 + C)).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        args = []
        for k, (path, a) in enumerate(tree.items()):
            raw, cfg = td / f"d{k}.bin", td / f"d{k}.cfg"
            a.tofile(raw)
            bits = 8 * a.dtype.itemsize
            cfg.write_text(f"PATH {path}\nINPUT-CLASS FP\nINPUT-SIZE {bits}\nRANK 2\nDIMENSION-SIZES {a.shape[0]} {a.shape[1]}\n"
                           f"OUTPUT-CLASS FP\nOUTPUT-SIZE {bits}\nOUTPUT-ARCHITECTURE IEEE\nOUTPUT-BYTE-ORDER LE\n")
            args += [str(raw), "-c", str(cfg)]
        out = HERE / "scene_libhdf5.h5"
        for old in HERE.glob("scene_libhdf5*.h5"):  # h5import appends to an existing file
            old.unlink()
        subprocess.run([str(BIN / "h5import")] + args + ["-o", str(out)], check=True)
        subprocess.run([str(BIN / "h5repack"), "-f", "SHUF", "-f", "GZIP=4", "-l", "CHUNK=8x9", str(out),
                        str(HERE / "scene_libhdf5_gzip.h5")], check=True)
        subprocess.run([str(BIN / "h5repack"), "--low=2", "--high=2", str(out), str(HERE / "scene_libhdf5_latest.h5")], check=True)
    np.savez(HERE / "scene_libhdf5_expected.npz", **tree)
    print("wrote", [p.name for p in HERE.glob("scene_libhdf5*")])


if __name__ == "__main__":
    sys.exit(main())
