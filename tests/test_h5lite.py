"""Row F3: the processed-scene HDF5 files (prepare_scenes.py:16-47 writer, vfm_reg/read_h5.py:17-49 reader).
``vfmreg.h5lite`` is pinned against the REAL HDF5 library in both directions:
  * files libhdf5 1.10 wrote (h5import / h5repack; tests/golden/make_h5_fixture.py, committed as binary fixtures)
    parse to the expected arrays -- contiguous (what h5py's create_dataset(data=...) yields) and chunked+shuffle+gzip;
  * files h5lite writes are read back by libhdf5's own h5dump / h5ls when those tools exist on the box."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from vfmreg import h5lite
from vfmreg.evaluation import read_scenes, save_scene

GOLDEN = Path(__file__).resolve().parent / "golden"


def _flat(t, pre=""):
    for k, v in t.items():
        if isinstance(v, dict):
            yield from _flat(v, pre + k + "/")
        else:
            yield pre + k, v


def _tool(name):
    for cand in (shutil.which(name), f"/opt/conda/bin/{name}"):
        if cand and Path(cand).exists():
            return cand
    return None


@pytest.mark.parametrize("fname", ["scene_libhdf5.h5", "scene_libhdf5_gzip.h5"])
def test_reads_files_written_by_libhdf5(fname):
    exp = dict(np.load(GOLDEN / "scene_libhdf5_expected.npz"))
    got = dict(_flat(h5lite.read_h5(GOLDEN / fname)))
    assert set(got) == set(exp)
    for k in exp:
        assert got[k].dtype == exp[k].dtype and got[k].shape == exp[k].shape
        np.testing.assert_array_equal(got[k], exp[k])
    # group iteration order = the file's B-tree order = sorted names (what h5py's .values() yields, read_h5.py:25)
    keys = list(h5lite.read_h5(GOLDEN / fname)["map"]["seqA"]["pose"])
    assert keys == sorted(keys) and len(keys) == 11
    scene = read_scenes(GOLDEN / fname)
    assert len(scene["map_poses"]) == 11 and scene["scene_sequences"] == ["2012-02-04", "seqB"]
    np.testing.assert_array_equal(scene["map_point_clouds"][3], exp["map/seqA/point_cloud/003"])
    np.testing.assert_array_equal(scene["scene_poses"][1], exp["scans/seqB/pose"])


def test_unsupported_structures_fail_loudly():
    """1.10 'latest' format with > 8 links per group uses dense link storage (fractal heap): not silently mis-read."""
    with pytest.raises(NotImplementedError, match="dense link storage"):
        h5lite.read_h5(GOLDEN / "scene_libhdf5_latest.h5")
    with pytest.raises(ValueError, match="not an HDF5 file"):
        bad = GOLDEN / "kabsch_dsc.npz"
        h5lite.read_h5(bad)


def _scene(rng, n_map=13, C=384):
    seqs = ["2012-01-08", "2012-02-04", "2012-03-17", "2012-05-26"]
    map_poses = [np.linalg.qr(rng.standard_normal((4, 4)))[0] for _ in range(n_map)]
    map_clouds = [rng.standard_normal((40 + j, 3 + C)).astype(np.float32) for j in range(n_map)]
    seq_poses = [rng.standard_normal((4, 4)), None, rng.standard_normal((4, 4))]   # the middle sequence has no hit (PS:41-42)
    seq_clouds = [rng.standard_normal((25, 3 + C)).astype(np.float32), None, rng.standard_normal((31, 3 + C)).astype(np.float32)]
    return seqs, map_poses, map_clouds, seq_poses, seq_clouds


@pytest.mark.parametrize("suffix", [".h5", ".npz"])
def test_save_scene_read_scenes_round_trip(tmp_path, suffix):
    rng = np.random.default_rng(0)
    seqs, map_poses, map_clouds, seq_poses, seq_clouds = _scene(rng)
    f = tmp_path / ("scene_007" + suffix)
    save_scene(f, seqs, map_poses, map_clouds, seq_poses, seq_clouds)
    s = read_scenes(f)
    assert s["scene_sequences"] == ["2012-02-04", "2012-05-26"] and s["map_clip"] == []
    for a, b in zip(s["map_poses"], map_poses):
        assert a.dtype == np.float64
        np.testing.assert_array_equal(a, b)
    for a, b in zip(s["map_point_clouds"], map_clouds):
        assert a.dtype == np.float32
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(s["scene_poses"][1], seq_poses[2])
    np.testing.assert_array_equal(s["scene_point_clouds"][0], seq_clouds[0])


def test_many_entries_need_a_two_level_group_index(tmp_path):
    """> 256 map clouds: more symbol nodes than one B-tree node holds (32) -> a level-1 node; plus int / scalar datasets"""
    rng = np.random.default_rng(1)
    tree = {"g": {f"{j:04}": rng.standard_normal((2, 3)).astype(np.float32) for j in range(700)},
            "ints": np.arange(12, dtype=np.int64).reshape(3, 4), "u8": np.arange(5, dtype=np.uint8), "empty": {},
            "scalar": np.float64(3.5), "zero_rows": np.zeros((0, 7), np.float32)}
    f = tmp_path / "big.h5"
    h5lite.write_h5(f, tree)
    back = h5lite.read_h5(f)
    assert list(back["g"]) == sorted(tree["g"]) and back["empty"] == {}
    for k, v in tree["g"].items():
        np.testing.assert_array_equal(back["g"][k], v)
    np.testing.assert_array_equal(back["ints"], tree["ints"])
    assert back["ints"].dtype == np.int64 and back["u8"].dtype == np.uint8 and back["scalar"] == 3.5
    assert back["zero_rows"].shape == (0, 7)
    h5ls = _tool("h5ls")
    if h5ls:
        out = subprocess.run([h5ls, "-r", str(f)], capture_output=True, text=True, check=True).stdout
        assert out.count("Dataset") == 704 and "/g/0699" in out
    h5dump = _tool("h5dump")
    if h5dump:   # look-ups BY NAME walk the B-tree's keys (a node's left key = its left sibling's largest name)
        for name in ("0000", "0063", "0064", "0300", "0650", "0699"):
            raw = tmp_path / f"{name}.bin"
            subprocess.run([h5dump, "-d", f"/g/{name}", "-b", "LE", "-o", str(raw), str(f)], capture_output=True, text=True, check=True)
            np.testing.assert_array_equal(np.fromfile(raw, dtype="<f4").reshape(2, 3), tree["g"][name])


@pytest.mark.skipif(_tool("h5dump") is None, reason="HDF5 command-line tools not on this box")
def test_libhdf5_reads_what_save_scene_writes(tmp_path):
    """the independent reader: libhdf5's h5dump must see the same groups, types, shapes and VALUES"""
    rng = np.random.default_rng(2)
    seqs, map_poses, map_clouds, seq_poses, seq_clouds = _scene(rng, n_map=10, C=5)
    f = tmp_path / "scene_000.h5"
    save_scene(f, seqs, map_poses, map_clouds, seq_poses, seq_clouds)
    h5dump = _tool("h5dump")
    listing = subprocess.run([h5dump, "-n", str(f)], capture_output=True, text=True, check=True).stdout
    for j in range(10):
        assert f"/map/2012-01-08/pose/{j:03}" in listing and f"/map/2012-01-08/point_cloud/{j:03}" in listing
    assert "/scans/2012-02-04/point_cloud" in listing and "/scans/2012-05-26/pose" in listing and "2012-03-17" not in listing
    for path, arr in (("/map/2012-01-08/pose/007", map_poses[7]), ("/map/2012-01-08/point_cloud/009", map_clouds[9]),
                      ("/scans/2012-05-26/point_cloud", seq_clouds[2])):
        raw = tmp_path / "dump.bin"
        r = subprocess.run([h5dump, "-d", path, "-b", "LE", "-o", str(raw), str(f)], capture_output=True, text=True, check=True)
        assert ("H5T_IEEE_F64LE" if arr.dtype == np.float64 else "H5T_IEEE_F32LE") in r.stdout
        assert f"( {arr.shape[0]}, {arr.shape[1]} )" in r.stdout
        np.testing.assert_array_equal(np.fromfile(raw, dtype=arr.dtype).reshape(arr.shape), arr)
