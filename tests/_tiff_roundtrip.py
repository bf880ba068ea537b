"""Run by tests/test_pipeline_cli.py under an interpreter that has `tifffile` (here: /opt/conda/bin/python3.9): the TIFF
branch of microaligner_amd/pipeline.py -- read_stack on OME-TIFF, create_output as a BigTIFF memory map with the
passed-through OME-XML (create_memmap_for_saving, __main__.py:116-132), TIFF CycleStack splitting, and the writer of
run() -- round trips real files.  No GPU: everything here is the control plane around the hot path.
Prints one JSON line."""
import json
import os
import sys

import numpy as np
import tifffile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microaligner_amd import pipeline as P  # noqa: E402


def main(tmp):
    rng = np.random.default_rng(0)
    C, Z, H, W = 3, 2, 70, 90
    names = ["DAPI", "CD3", "CD8"]
    stack = rng.integers(0, 65535, (C, Z, H, W), dtype=np.uint16)
    src = os.path.join(tmp, "cycle1.ome.tif")
    tifffile.imwrite(src, stack[None], bigtiff=True, photometric="minisblack",
                     metadata={"axes": "TCZYX", "Channel": {"Name": names}, "PhysicalSizeX": 0.325, "PhysicalSizeXUnit": "µm"})
    arr, got_names, xml = P.read_stack_meta(src)
    assert arr.shape == (C, Z, H, W) and np.array_equal(arr, stack) and got_names == names, (arr.shape, got_names)
    assert xml and "PhysicalSizeX" in xml

    # output of one cycle: BigTIFF memory map carrying the input's OME-XML, sizes and channels patched
    out_shape = (1, C, Z + 1, H, W)                        # zmax of the data set may exceed this cycle's planes
    desc = P.ome_passthrough(xml, out_shape, stack.dtype, names)
    mm, path = P.create_output(os.path.join(tmp, "out_cyc001.tif"), out_shape, stack.dtype, "tif", desc)
    mm[...] = 0
    mm[0, :, :Z] = stack
    mm.flush()
    del mm
    with tifffile.TiffFile(str(path)) as tf:
        assert tf.is_bigtiff and tf.is_ome
        back = tf.series[0].asarray()
        axes = tf.series[0].axes
        oxml = tf.ome_metadata
    assert P.channel_names_of(oxml) == names and 'SizeZ="3"' in oxml and 'SizeC="3"' in oxml and 'SizeT="1"' in oxml
    assert "PhysicalSizeX" in oxml and oxml.count("<TiffData") == 1 and "<Plane" not in oxml
    assert set("CZYX") <= set(axes), axes
    arr2, names2 = P.read_stack(path)
    assert names2 == names and arr2.shape == (C, Z + 1, H, W) and np.array_equal(arr2[:, :Z], stack)
    assert not arr2[:, Z].any()

    # a stack of all cycles: channels named per cycle, cut back into cycles where the reference channel recurs
    cyc2 = (stack // 2).astype(np.uint16)
    all_names = [f"c{c:02d} {n}" for c in (1, 2) for n in names]
    sdesc = P.ome_passthrough(xml, (1, 2 * C, Z, H, W), stack.dtype, all_names)
    mm, spath = P.create_output(os.path.join(tmp, "stack.tif"), (1, 2 * C, Z, H, W), stack.dtype, "tif", sdesc)
    mm[0, :C], mm[0, C:] = stack, cyc2
    mm.flush()
    del mm
    sarr, snames, sxml = P.read_stack_meta(spath)
    assert snames == all_names and 'SizeC="6"' in sxml
    cycles = P.split_cycle_stack(sarr, snames, "DAPI")
    assert [c for c, _, _ in cycles] == [1, 2] and [n for _, _, n in cycles] == [names, names]
    assert np.array_equal(cycles[0][1], stack) and np.array_equal(cycles[1][1], cyc2)
    try:
        P.split_cycle_stack(sarr, snames, "CD45")
        raise AssertionError("an unknown reference channel must be rejected")
    except ValueError as e:
        assert "Incorrect reference channel" in str(e)

    # no input description (.npy inputs, TIFF output): the minimal document
    mdesc = P.ome_passthrough(None, out_shape, np.float32, ["a&b", "c"])
    mm, mpath = P.create_output(os.path.join(tmp, "min.tif"), (1, 2, 1, H, W), np.float32, "tif",
                                P.ome_passthrough(None, (1, 2, 1, H, W), np.float32, ["a&b", "c"]))
    mm[...] = 1.5
    mm.flush()
    del mm
    with tifffile.TiffFile(str(mpath)) as tf:
        assert tf.is_ome and tf.series[0].asarray().squeeze().shape == (2, H, W)
        assert P.channel_names_of(tf.ome_metadata) == ["a&amp;b", "c"]
    assert "a&amp;b" in mdesc

    # the config layer: a TIFF CycleStack is read and cut into cycles; the writer names and describes the outputs
    import yaml
    cfg_path = os.path.join(tmp, "config.yaml")
    reg = dict(NumberPyramidLevels=1, NumberIterationsPerLevel=1, TileSize=100, Overlap=20, NumberOfWorkers=0,
               UseFullResImage=True, UseDOG=False)
    yaml.safe_dump({"Input": {"InputImagePaths": {"CycleStack": str(spath)}, "ReferenceCycle": 1, "ReferenceChannel": "DAPI"},
                    "Output": {"OutputDir": os.path.join(tmp, "out"), "OutputPrefix": "t_", "SaveOutputToCycleStack": True},
                    "RegistrationParameters": {"OptFlowReg": reg}}, open(cfg_path, "w"))
    cfg = P.read_config(cfg_path)
    cfg.out_dir.mkdir(parents=True, exist_ok=True)
    loaded = P._load_cycles(cfg)
    assert [c for c, _, _ in loaded] == [1, 2] and cfg.input_ome
    dst, written, state = P._writer(cfg, loaded, "optflow_reg", "tif")
    for n, (cyc, a, _) in enumerate(loaded):
        dst(n, cyc, a)[...] = a
    for m_ in state.values():
        m_.flush()
    del state, dst
    res, res_names, res_xml = P.read_stack_meta(written[0])
    assert res.shape == (2 * C, Z, H, W) and np.array_equal(res[:C], stack) and np.array_equal(res[C:], cyc2)
    assert res_names == all_names and 'SizeC="6"' in res_xml and "PhysicalSizeX" in res_xml
    print(json.dumps({"ok": True, "tifffile": tifffile.__version__, "python": sys.version.split()[0],
                      "checked": ["read_stack OME-TIFF", "create_output BigTIFF memmap + OME passthrough", "TIFF CycleStack split",
                                  "minimal OME", "writer of run()"]}))


if __name__ == "__main__":
    main(sys.argv[1])
