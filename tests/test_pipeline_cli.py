"""The command-line pipeline: the reference's YAML schema and validation without a GPU, and -- on the
GPU -- a small three-cycle run from .npy stacks, compared with the same pipeline composed from the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

from microaligner_amd import pipeline as P, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = dict(NumberPyramidLevels=2, NumberIterationsPerLevel=3, TileSize=150, Overlap=30, NumberOfWorkers=0,
           UseFullResImage=True, UseDOG=False)


def config(tmp_path, paths, **over):
    cfg = {"Input": {"InputImagePaths": paths, "ReferenceCycle": 1, "ReferenceChannel": "0"},
           "Output": {"OutputDir": str(tmp_path / "out"), "OutputPrefix": "exp_", "SaveOutputToCycleStack": True},
           "RegistrationParameters": {"OptFlowReg": dict(REG)}}
    for k, v in over.items():
        sec, key = k.split("__")
        if v is None:
            cfg[sec].pop(key)
        else:
            cfg[sec][key] = v
    path = tmp_path / "config.yaml"
    path.write_text(yaml.safe_dump(cfg))
    return path


def test_config_schema_and_validation(tmp_path):
    """config_reader.py:76-97,154-260: same fields, same bounds, same messages."""
    paths = {"Cycle 1": "a.npy", "Cycle 2": "b.npy"}
    cfg = P.read_config(config(tmp_path, paths))
    assert cfg.paths == {1: P.Path("a.npy"), 2: P.Path("b.npy")} and cfg.ref_cycle == 1 and cfg.ref_channel == "0"
    assert cfg.feature is None and cfg.optflow.optflow_kwargs() == dict(num_pyr_lvl=2, num_iterations=3, tile_size=150,
                                                                         use_full_res_img=True, use_dog=False, overlap=30)
    # the shipped example of the reference parses (config_examples/config_1.yaml keys)
    example = {"Input": {"InputImagePaths": {"Cycle 1": "img_path", "Cycle 2": "img_path2"}, "ReferenceCycle": 1,
                         "ReferenceChannel": "DAPI"},
               "Output": {"OutputDir": "/tmp/x", "OutputPrefix": "experiment_001_", "SaveOutputToCycleStack": True},
               "RegistrationParameters": {"FeatureReg": dict(REG, TileSize=1000, Overlap=100, NumberPyramidLevels=3, UseDOG=True,
                                                             UseFullResImage=False),
                                          "OptFlowReg": dict(REG, TileSize=1000, Overlap=100, NumberPyramidLevels=3)}}
    c = P.PipelineConfig(example)
    assert c.feature.feature_kwargs()["use_dog"] is True and c.optflow.Overlap == 100 and c.to_stack
    with pytest.raises(ValueError, match="These fields are absent"):
        P.PipelineConfig({"Input": {}})
    with pytest.raises(KeyError, match="Field ReferenceChannel is absent"):
        P.read_config(config(tmp_path, paths, Input__ReferenceChannel=None))
    with pytest.raises(ValueError, match="smaller than minimum: 1"):
        P.read_config(config(tmp_path, paths, Input__ReferenceCycle=0))
    with pytest.raises(ValueError, match="Cycle names in config file should follow pattern Cycle N"):
        P.read_config(config(tmp_path, {"First": "a.npy"}))
    for key, val, exc, msg in (("TileSize", 10, ValueError, "smaller than minimum: 20"),
                               ("Overlap", 500, ValueError, "greater than maximum: 150"),
                               ("NumberPyramidLevels", 9, ValueError, "greater than maximum: 8"),
                               ("UseDOG", "yes", TypeError, "Field UseDOG has wrong data type"),
                               ("NumberIterationsPerLevel", 0, ValueError, "smaller than minimum: 1")):
        with pytest.raises(exc, match=msg):
            P.read_config(config(tmp_path, paths, RegistrationParameters__OptFlowReg=dict(REG, **{key: val})))
    with pytest.raises(ValueError, match="At least one of the registration methods"):
        P.read_config(config(tmp_path, paths, RegistrationParameters__OptFlowReg=None))
    cb = P.read_config(config(tmp_path, {"Cycle 1": {"DAPI": "a.npy", "CD3": "b.npy"}, "Cycle 2": {"DAPI": "c.npy"}}))
    assert cb.input_type == "CycleBuilder" and cb.paths[1] == {"DAPI": P.Path("a.npy"), "CD3": P.Path("b.npy")}
    assert P.read_config(config(tmp_path, {"CycleStack": "all.npy"})).input_type == "CycleStack"


def test_stack_io_and_channel_lookup(tmp_path):
    a = np.arange(2 * 3 * 4 * 5, dtype=np.uint16).reshape(2, 3, 4, 5)
    np.save(tmp_path / "a.npy", a)
    arr, names = P.read_stack(tmp_path / "a.npy")
    assert arr.shape == (2, 3, 4, 5) and names == ["0", "1"] and np.array_equal(arr, a)
    (tmp_path / "a.channels.json").write_text('["DAPI", "CD3"]')
    assert P.read_stack(tmp_path / "a.npy")[1] == ["DAPI", "CD3"]
    assert P.channel_index(["DAPI", "CD3"], "CD3", "x") == 1 and P.channel_index(["DAPI", "CD3"], "0", "x") == 0
    with pytest.raises(ValueError, match="is not among the channels"):
        P.channel_index(["DAPI"], "CD8", "cycle 2")
    np.save(tmp_path / "b.npy", a[0, 0])
    assert P.read_stack(tmp_path / "b.npy")[0].shape == (1, 1, 4, 5)
    # the CycleBuilder form (one file per channel) and the single stack of all cycles load into the same structure
    np.save(tmp_path / "dapi.npy", a[0])
    np.save(tmp_path / "cd3.npy", a[1, :2])
    cb = P.read_config(config(tmp_path, {"Cycle 1": {"DAPI": str(tmp_path / "dapi.npy"), "CD3": str(tmp_path / "cd3.npy")}}))
    (cyc, stack, names), = P._load_cycles(cb)
    assert cyc == 1 and sorted(names) == ["CD3", "DAPI"] and stack.shape == (2, 3, 4, 5)
    d, c = names.index("DAPI"), names.index("CD3")
    assert np.array_equal(stack[d], a[0]) and np.array_equal(stack[c, :2], a[1, :2]) and not stack[c, 2].any()
    np.save(tmp_path / "all.npy", np.stack([a, a + 1]))
    cs = P._load_cycles(P.read_config(config(tmp_path, {"CycleStack": str(tmp_path / "all.npy")})))
    assert [c[0] for c in cs] == [1, 2] and np.array_equal(cs[1][1], a + 1)
    mm, path = P.create_output(tmp_path / "o.tif", (1, 2, 3, 4, 5), np.uint16, "npy")
    mm[0] = a
    mm.flush()
    assert path.suffix == ".npy" and np.array_equal(np.load(path)[0], a)
    if P._tifffile() is None:
        with pytest.raises(RuntimeError, match="tifffile"):
            P.read_stack(tmp_path / "img.tif")
        with pytest.raises(RuntimeError, match="tifffile"):
            P.create_output(tmp_path / "o.tif", (1, 1, 1, 4, 5), np.uint16, "tif")
    r = subprocess.run([sys.executable, "-m", "microaligner_amd"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_cli_runs_the_cycle_chain_like_the_oracle_pipeline(tmp_path):
    """Three cycles of (2 channels x 2 z planes) uint16 pages through `python -m microaligner_amd config.yaml`: the
    written stack equals the reference pipeline composed from the oracle (max projection + normalisation, chained
    registration, every page warped with its cycle's flow)."""
    from oracle import oracle as O
    from oracle import register_oracle as RO
    H, W = 420, 380
    cycles = []
    for k in range(3):
        ref, mov = synthetic.make_pair(H, W, seed=40 + k, dtype=np.uint16)
        base = ref if k == 0 else mov
        stack = np.stack([np.stack([base, (base // 2).astype(np.uint16)]), np.stack([(base // 3).astype(np.uint16), base[::-1].copy()])])
        cycles.append(stack)
        np.save(tmp_path / f"cyc{k + 1}.npy", stack)
    cfg = config(tmp_path, {f"Cycle {k + 1}": str(tmp_path / f"cyc{k + 1}.npy") for k in range(3)})
    r = subprocess.run([sys.executable, "-m", "microaligner_amd", str(cfg)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = np.load(tmp_path / "out" / "exp_optflow_reg_result_stack.npy")
    assert out.shape == (1, 6, 2, H, W) and out.dtype == np.uint16
    params = dict(num_pyr_lvl=2, num_iterations=3, tile_size=150, overlap=30, use_full_res_img=True, use_dog=False)
    prep = lambda st: O.normalize_minmax_u8(np.maximum.reduce(list(st)).astype(np.float32))   # noqa: E731  utils.py:92-94
    ref_img = prep(cycles[0][0])
    assert np.array_equal(out[0, 0:2], cycles[0])
    for k in (1, 2):
        mov_img = prep(cycles[k][0])
        flow, _ = RO.register(ref_img, mov_img, **params)
        ref_img = RO.warp(mov_img, flow, 150, 30)
        for c in range(2):
            for z in range(2):
                assert np.array_equal(out[0, 2 * k + c, z], RO.warp(cycles[k][c, z], flow, 150, 30)), (k, c, z)


def _python_with_tifffile():
    """An interpreter that imports tifffile and numpy: this one, or the image's conda python (tifffile 2021.7.2)."""
    import shutil
    for exe in (sys.executable, "/opt/conda/bin/python3.9", shutil.which("python3.9")):
        if exe and os.path.exists(exe):
            r = subprocess.run([exe, "-c", "import tifffile, numpy, yaml"], capture_output=True)
            if r.returncode == 0:
                return exe
    return None


def test_tiff_branch_round_trips_real_files(tmp_path):
    """The TIFF branch of the pipeline under an interpreter that has tifffile: OME-TIFF in through
    read_stack, BigTIFF memory map out as create_memmap_for_saving makes it (__main__.py:116-132) with the input's OME-XML
    passed through (sizes and channel list patched), a TIFF CycleStack cut into cycles where the reference channel recurs
    (metadata_handling.py:100-132), the writer of run().  tests/_tiff_roundtrip.py does the work and prints one JSON line."""
    exe = _python_with_tifffile()
    if exe is None:
        pytest.skip("no interpreter with tifffile in this image")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "_tiff_roundtrip.py"), str(tmp_path)], capture_output=True, text=True,
                       cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = __import__("json").loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["ok"] is True and "create_output BigTIFF memmap + OME passthrough" in res["checked"]


def test_ome_passthrough_patches_only_what_registration_changes():
    xml = ('<?xml version="1.0" encoding="UTF-8"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06">'
           '<Instrument ID="Instrument:0"/><Image ID="Image:0" Name="x"><Pixels ID="Pixels:0" DimensionOrder="XYCZT" '
           'Type="uint16" SizeX="10" SizeY="20" SizeZ="1" SizeC="2" SizeT="1" PhysicalSizeX="0.325" PhysicalSizeXUnit="µm">'
           '<Channel ID="Channel:0:0" Name="DAPI" Fluor="Hoechst" SamplesPerPixel="1"><LightPath/></Channel>'
           '<Channel ID="Channel:0:1" Name="CD3" SamplesPerPixel="1"/><TiffData IFD="0" PlaneCount="1"/>'
           '<TiffData IFD="1" PlaneCount="1"/><Plane TheC="0" TheZ="0" TheT="0"/></Pixels></Image></OME>')
    out = P.ome_passthrough(xml, (1, 4, 3, 50, 60), np.uint16, ["c01 DAPI", "c01 CD3", "DAPI", "new"])
    for frag in ('SizeX="60"', 'SizeY="50"', 'SizeZ="3"', 'SizeC="4"', 'SizeT="1"', 'DimensionOrder="XYZCT"', 'PhysicalSizeX="0.325"',
                 '<Instrument ID="Instrument:0"/>', '&#181;m'):
        assert frag in out, frag
    assert out.count("<TiffData") == 1 and "<Plane" not in out and out.isascii()
    assert P.channel_names_of(out) == ["c01 DAPI", "c01 CD3", "DAPI", "new"]
    assert 'ID="Channel:0:2" Name="DAPI" Fluor="Hoechst"' in out and "<LightPath/>" in out     # a known channel keeps its attributes
    assert [P.strip_cycle_info(n) for n in ("c01 DAPI", "cyc2_CD3-2", "cycle3-CD8_1", "DAPI")] == ["DAPI", "CD3", "CD8", "DAPI"]
    mini = P.ome_passthrough(None, (1, 2, 1, 5, 6), np.float32, ["a", "b"])
    assert 'Type="float"' in mini and 'SizeC="2"' in mini and P.channel_names_of(mini) == ["a", "b"]


@pytest.mark.gpu
def test_cli_with_tiff_inputs_and_outputs_equals_the_npy_run(tmp_path):
    """`python -m microaligner_amd config.yaml` with OME-TIFF cycles in and a BigTIFF stack out (under the interpreter that
    has tifffile) writes the very pixels of the .npy run of the same data -- which
    test_cli_runs_the_cycle_chain_like_the_oracle_pipeline holds against the oracle pipeline -- and an OME-XML with the
    stack's sizes and cycle-decorated channel names."""
    exe = _python_with_tifffile()
    if exe is None:
        pytest.skip("no interpreter with tifffile in this image")
    H, W = 420, 380
    names = ["DAPI", "CD3"]
    paths_npy, paths_tif = {}, {}
    for k in range(3):
        ref, mov = synthetic.make_pair(H, W, seed=40 + k, dtype=np.uint16)
        base = ref if k == 0 else mov
        stack = np.stack([np.stack([base, (base // 2).astype(np.uint16)]), np.stack([(base // 3).astype(np.uint16), base[::-1].copy()])])
        np.save(tmp_path / f"cyc{k + 1}.npy", stack)
        (tmp_path / f"cyc{k + 1}.channels.json").write_text('["DAPI", "CD3"]')
        paths_npy[f"Cycle {k + 1}"] = str(tmp_path / f"cyc{k + 1}.npy")
        paths_tif[f"Cycle {k + 1}"] = str(tmp_path / f"cyc{k + 1}.ome.tif")
    conv = ("import sys, numpy as np, tifffile\n"
            "for k in (1, 2, 3):\n"
            f"    a = np.load(r'{tmp_path}/cyc%d.npy' % k)\n"
            f"    tifffile.imwrite(r'{tmp_path}/cyc%d.ome.tif' % k, a[None], bigtiff=True, photometric='minisblack',\n"
            "                     metadata={'axes': 'TCZYX', 'Channel': {'Name': ['DAPI', 'CD3']}})\n")
    assert subprocess.run([exe, "-c", conv], capture_output=True, text=True).returncode == 0
    (tmp_path / "npy").mkdir()
    (tmp_path / "tif").mkdir()
    cfg_npy = config(tmp_path / "npy", paths_npy, Input__ReferenceChannel="DAPI")
    cfg_tif = config(tmp_path / "tif", paths_tif, Input__ReferenceChannel="DAPI")
    for cfg in (cfg_npy, cfg_tif):
        r = subprocess.run([exe, "-m", "microaligner_amd", str(cfg)], capture_output=True, text=True, cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    check = ("import sys, json, numpy as np, tifffile\n"
             f"a = np.load(r'{tmp_path}/npy/out/exp_optflow_reg_result_stack.npy')\n"
             f"tf = tifffile.TiffFile(r'{tmp_path}/tif/out/exp_optflow_reg_result_stack.tif')\n"
             "b = tf.series[0].asarray().reshape(a.shape)\n"
             "print(json.dumps({'equal': bool(np.array_equal(a, b)), 'bigtiff': bool(tf.is_bigtiff), 'ome': tf.ome_metadata}))\n")
    r = subprocess.run([exe, "-c", check], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res = __import__("json").loads(r.stdout.strip().splitlines()[-1])
    assert res["equal"] and res["bigtiff"]
    assert P.channel_names_of(res["ome"]) == [f"c{c:02d} {n}" for c in (1, 2, 3) for n in names]
    assert 'SizeC="6"' in res["ome"] and 'SizeZ="2"' in res["ome"]
