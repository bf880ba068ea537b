"""`python -m microaligner_amd config.yaml` -- the command-line pipeline around the hot path, minimal.

Counterpart of microaligner/__main__.py:624-642 (main), :440-516 (run_feature_reg), :532-609 (run_opt_flow_reg),
:320-437 (register_and_save_ofreg_imgs) and pipeline_modules/config_reader.py:76-97,154-260: the reference's YAML keys
with the reference's validation (same messages), per-cycle image stacks in, registered stacks out.  What it drives is
this package's own machinery -- FeatureRegistrator, parallel.register_cycle_chain (device max-projection and
normalisation, chained optical-flow registration, the overlapped page-warp driver writing straight into the output
memory map).

Deliberately small (the control plane is not the product): one file per cycle ("Cycle N": path), one file per channel
("Cycle N": {channel: path}, the reference's CycleBuilder form) or one stack of all cycles ("CycleStack": path -- a
(cycles, C, Z, Y, X) .npy, or an OME-TIFF whose channel names repeat from cycle to cycle: the cycles are cut where the
reference channel recurs, metadata_handling.py:100-132); images are TIFF when `tifffile` imports (read through
TiffFile.series, written to a BigTIFF memory map exactly as create_memmap_for_saving does, __main__.py:116-132) and
`.npy` arrays otherwise or on request -- (C, Z, Y, X), (C, Y, X) or (Y, X) per cycle -- so the pipeline runs in images
without tifffile too.  OME-XML: the description of the input is passed through to the output with its sizes and channel
list patched (ome_passthrough) -- not the reference's full rebuild (ome_meta_processing.py: per-plane metadata, physical
size conversion), which is control plane.
"""
import json
import os
import re
import sys
from pathlib import Path

import numpy as np


# ---- config (pipeline_modules/config_reader.py) -----------------------------------------------------------------
def _check_dtype(name, types, d):
    """check_field_dtype (config_reader.py:34-49): KeyError when absent, TypeError on the wrong type (a bool is an int
    to isinstance, as in the reference)."""
    types = list(types) if isinstance(types, (tuple, list)) else [types]
    if name not in d:
        raise KeyError(f"Field {name} is absent")
    if not any(isinstance(d[name], t) for t in types):
        raise TypeError(f"Field {name} has wrong data type {type(d[name])}, expected {types}")


def _check_min_max(name, lo, hi, d):
    v = d[name]
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        if lo is not None and v < lo:
            raise ValueError(f"Field {name} value is smaller than minimum: {lo}")
        if hi is not None and v > hi:
            raise ValueError(f"Field {name} value is greater than maximum: {hi}")


class RegParam:
    """config_reader.py:76-97."""
    FIELDS = (("NumberPyramidLevels", int), ("NumberIterationsPerLevel", int), ("TileSize", int), ("Overlap", int),
              ("NumberOfWorkers", int), ("UseFullResImage", bool), ("UseDOG", bool))

    def __init__(self, d):
        if not isinstance(d, dict):
            raise TypeError("registration parameters must be a mapping")
        for name, t in self.FIELDS:
            _check_dtype(name, t, d)
        _check_min_max("NumberPyramidLevels", 0, 8, d)
        _check_min_max("NumberIterationsPerLevel", 1, None, d)
        _check_min_max("TileSize", 20, None, d)
        _check_min_max("Overlap", 10, d["TileSize"], d)
        _check_min_max("NumberOfWorkers", 0, None, d)
        for name, _ in self.FIELDS:
            setattr(self, name, d[name])

    def feature_kwargs(self):
        return dict(num_pyr_lvl=self.NumberPyramidLevels, num_iterations=self.NumberIterationsPerLevel,
                    tile_size=self.TileSize, use_full_res_img=self.UseFullResImage, use_dog=self.UseDOG)

    def optflow_kwargs(self):
        return dict(self.feature_kwargs(), overlap=self.Overlap)

    def __repr__(self):
        return str(self.__dict__)


class PipelineConfig:
    def __init__(self, cfg):
        missing = [f for f in ("Input", "Output", "RegistrationParameters") if f not in cfg]
        if missing:
            raise ValueError("Incorrectly formatted config file.These fields are absent: " + str(missing))
        inp, out, reg = cfg["Input"], cfg["Output"], cfg["RegistrationParameters"]
        if not isinstance(inp, dict):
            raise ValueError("Input field is incorrect")
        _check_dtype("InputImagePaths", (dict, list), inp)
        _check_dtype("ReferenceCycle", int, inp)
        _check_dtype("ReferenceChannel", str, inp)
        _check_min_max("ReferenceCycle", 1, None, inp)
        paths = inp["InputImagePaths"]
        if isinstance(paths, dict) and "CycleStack" in paths:
            self.input_type, self.paths = "CycleStack", {0: Path(paths["CycleStack"])}
        elif isinstance(paths, dict) and paths and all(isinstance(v, dict) for v in paths.values()):
            # CycleBuilder (config_reader.py:203-218): "Cycle N": {channel name: path of that channel's (Z, Y, X) stack}
            self.input_type, self.paths = "CycleBuilder", {}
            for name, chans in paths.items():
                if not re.match(r"Cycle \d+", name):
                    raise ValueError("Cycle names in config file should follow pattern Cycle N")
                cyc = int(re.search(r"(\d+)", name).group(1))
                if len(chans) > len(set(chans)):
                    raise ValueError(f"Channel names are repeated in the Cycle {cyc}: {list(chans)}")
                self.paths[cyc] = {ch: Path(p) for ch, p in chans.items()}
        else:
            self.input_type, self.paths = "CycleList", {}
            for name, p in (paths.items() if isinstance(paths, dict) else enumerate(paths, 1)):
                name = name if isinstance(name, str) else f"Cycle {name}"
                if not re.match(r"Cycle \d+", name):
                    raise ValueError("Cycle names in config file should follow pattern Cycle N")
                self.paths[int(re.search(r"(\d+)", name).group(1))] = Path(p)
        self.ref_cycle, self.ref_channel = inp["ReferenceCycle"], inp["ReferenceChannel"]
        _check_dtype("OutputDir", str, out)
        _check_dtype("OutputPrefix", str, out)
        _check_dtype("SaveOutputToCycleStack", bool, out)
        self.out_dir, self.out_prefix, self.to_stack = Path(out["OutputDir"]), out["OutputPrefix"], out["SaveOutputToCycleStack"]
        self.output_format = out.get("OutputFormat")        # addition: "tif" | "npy"; default follows the input
        if "FeatureReg" not in reg and "OptFlowReg" not in reg:
            raise ValueError("Parameters for hte registration methods are absent. At least one of the registration methods: "
                             "FeatureReg or OptFlowReg must be present.")
        self.input_ome = None               # OME-XML description of the (first) input, passed through to the outputs
        self.feature = RegParam(reg["FeatureReg"]) if "FeatureReg" in reg else None
        self.optflow = RegParam(reg["OptFlowReg"]) if "OptFlowReg" in reg else None


def read_config(path):
    import yaml
    with open(path) as f:
        return PipelineConfig(yaml.safe_load(f))


# ---- image stacks: TIFF through tifffile when it imports, .npy always -----------------------------------------------
def _tifffile():
    try:
        import tifffile
        return tifffile
    except ImportError:
        return None


def _as_czyx(arr, what):
    arr = np.asarray(arr)
    if arr.ndim == 5 and arr.shape[0] == 1:
        arr = arr[0]                                  # TCZYX with one time point
    if arr.ndim == 2:
        arr = arr[None, None]
    elif arr.ndim == 3:
        arr = arr[:, None]
    if arr.ndim != 4:
        raise ValueError(f"{what}: expected (C, Z, Y, X), (C, Y, X) or (Y, X), got shape {arr.shape}")
    return arr


def read_stack_meta(path):
    """((C, Z, Y, X) array or memory map, channel names, OME-XML description or None)."""
    path = Path(path)
    if path.suffix.lower() == ".npy":
        arr = _as_czyx(np.load(path, mmap_mode="r"), str(path))
        side = path.with_suffix(".channels.json")
        names = json.load(open(side)) if side.exists() else [str(i) for i in range(arr.shape[0])]
        return arr, names, None
    tif = _tifffile()
    if tif is None:
        raise RuntimeError(f"{path}: reading TIFF needs the `tifffile` package, which does not import here; .npy stacks "
                           "(C, Z, Y, X) work without it")
    with tif.TiffFile(str(path)) as tf:
        series = tf.series[0]
        arr = series.asarray()
        axes = series.axes
        xml = tf.ome_metadata or None
    names = channel_names_of(xml)
    order = [axes.index(a) for a in "CZYX" if a in axes]
    arr = np.transpose(arr.squeeze() if arr.ndim > len(order) else arr, order) if len(order) == arr.ndim else arr
    arr = _as_czyx(arr, str(path))
    return arr, (names if len(names) == arr.shape[0] else [str(i) for i in range(arr.shape[0])]), xml


def read_stack(path):
    """((C, Z, Y, X) array or memory map, channel names)."""
    return read_stack_meta(path)[:2]


# ---- OME-XML: pass the input's description through, patched (no rebuild of ome_meta_processing.py) -------------------
def channel_names_of(xml):
    return re.findall(r'<Channel\b[^>]*?\bName="([^"]*)"', xml or "")


def _xml_escape(text):
    return (str(text).replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;").replace('"', "&quot;"))


def minimal_ome(shape_tczyx, dtype, names):
    """The smallest OME-XML a reader needs to get shape, axis order and channel names back (used when the input carried
    no description: .npy inputs written to TIFF)."""
    T, C, Z, Y, X = (int(v) for v in shape_tczyx)
    ptype = {"uint8": "uint8", "uint16": "uint16", "float32": "float", "int16": "int16", "uint32": "uint32"}.get(
        np.dtype(dtype).name, np.dtype(dtype).name)
    chans = "".join(f'<Channel ID="Channel:0:{i}" Name="{_xml_escape(n)}" SamplesPerPixel="1"/>' for i, n in enumerate(names))
    return ('<?xml version="1.0" encoding="UTF-8"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06">'
            f'<Image ID="Image:0" Name="microaligner_amd"><Pixels ID="Pixels:0" DimensionOrder="XYZCT" Type="{ptype}" '
            f'SizeX="{X}" SizeY="{Y}" SizeZ="{Z}" SizeC="{C}" SizeT="{T}">{chans}<TiffData/></Pixels></Image></OME>'
            ).encode("ascii", "xmlcharrefreplace").decode("ascii")


def ome_passthrough(xml, shape_tczyx, dtype, names):
    """The input's OME-XML description for an output of `shape_tczyx` holding the channels `names` (the reference
    regenerates the whole document, ome_meta_processing.py:455-473; here the input's document is kept -- instrument,
    physical sizes, whatever else it carries -- and only what the registration changes is patched: SizeX/Y/Z/C/T,
    DimensionOrder, the channel list, and the TiffData / Plane entries, which describe the input's page layout and are
    replaced by one <TiffData/> = "pages in dimension order").  Falls back to minimal_ome without an input document."""
    if not xml or "<Pixels" not in xml:
        return minimal_ome(shape_tczyx, dtype, names)
    T, C, Z, Y, X = (int(v) for v in shape_tczyx)
    m = re.search(r"<Pixels\b[^>]*>", xml)
    head = m.group(0)
    self_closing = head.endswith("/>")
    for key, val in (("SizeX", X), ("SizeY", Y), ("SizeZ", Z), ("SizeC", C), ("SizeT", T), ("DimensionOrder", "XYZCT")):
        if re.search(rf'\b{key}="[^"]*"', head):
            head = re.sub(rf'\b{key}="[^"]*"', f'{key}="{val}"', head)
        else:
            head = head[:-2 if self_closing else -1] + f' {key}="{val}"' + ("/>" if self_closing else ">")
    # channel elements of the input, reused by name where they exist (their attributes -- Fluor, Color, wavelengths --
    # survive), plain ones for names the input does not know
    known = {}
    for el in re.findall(r"<Channel\b[^>]*?(?:/>|>.*?</Channel>)", xml, flags=re.S):
        nm = re.search(r'\bName="([^"]*)"', el)
        if nm:
            known.setdefault(nm.group(1), el)
    chans = []
    for i, n in enumerate(names):
        el = known.get(n) or f'<Channel ID="Channel:0:{i}" Name="{_xml_escape(n)}" SamplesPerPixel="1"/>'
        chans.append(re.sub(r'\bID="[^"]*"', f'ID="Channel:0:{i}"', el, count=1))
    if self_closing:
        body_start = body_end = m.end()
        head = head[:-2] + ">"
        tail = "</Pixels>"
    else:
        body_start, body_end = m.end(), xml.index("</Pixels>", m.end())
        tail = ""
    body = xml[body_start:body_end]
    body = re.sub(r"<Channel\b[^>]*?(?:/>|>.*?</Channel>)", "", body, flags=re.S)
    body = re.sub(r"<TiffData\b[^>]*?(?:/>|>.*?</TiffData>)", "", body, flags=re.S)
    body = re.sub(r"<Plane\b[^>]*?(?:/>|>.*?</Plane>)", "", body, flags=re.S)
    doc = xml[:m.start()] + head + "".join(chans) + "<TiffData/>" + body.strip() + tail + xml[body_end:]
    # TIFF ImageDescription strings are 7-bit ASCII: anything else (a micro sign in a unit) becomes a character reference
    return doc.encode("ascii", "xmlcharrefreplace").decode("ascii")


def strip_cycle_info(name):
    """Channel name without the cycle decoration a CycleStack carries ("c01 DAPI", "cyc2_CD3-2", ome_meta_processing.py:71-74)."""
    name = re.sub(r"^(c|cyc|cycle)\d+(\s+|_|-)?", "", name)
    return re.sub(r"(-\d+)?(_\d+)?$", "", name)


def split_cycle_stack(arr, names, ref_channel):
    """One (C_total, Z, Y, X) stack of all cycles -> [(cycle id, (C, Z, Y, X) view, channel names)]: the cycles are cut
    where the reference channel recurs among the undecorated channel names (metadata_handling.py:100-132: channels per
    cycle = distance between its first two occurrences)."""
    clean = [strip_cycle_info(n) for n in names]
    ids = [i for i, n in enumerate(clean) if re.match(ref_channel, n, re.IGNORECASE)]
    if not ids:
        raise ValueError(f"Incorrect reference channel {ref_channel}. Available channel names: {set(clean)}")
    per = ids[1] - ids[0] if len(ids) > 1 else len(names)
    if len(names) % per:
        raise ValueError(f"CycleStack: {len(names)} channels do not divide into cycles of {per}")
    return [(k + 1, arr[k * per:(k + 1) * per], clean[k * per:(k + 1) * per]) for k in range(len(names) // per)]


def create_output(path, shape, dtype, fmt, description=None):
    """Writable (1, C, Z, Y, X) memory map: BigTIFF as create_memmap_for_saving makes it (tifffile.memmap with
    photometric minisblack, contiguous pages and the OME-XML as the description, __main__.py:116-132) or .npy."""
    path = Path(path)
    if fmt == "tif":
        tif = _tifffile()
        if tif is None:
            raise RuntimeError("writing TIFF output needs the `tifffile` package, which does not import here; set "
                               "Output: OutputFormat: npy")
        kw = dict(description=description, metadata=None) if description else dict(metadata={"axes": "TCZYX"})
        return tif.memmap(str(path), shape=shape, dtype=dtype, photometric="minisblack", bigtiff=True, contiguous=True,
                          **kw), path
    path = path.with_suffix(".npy")
    return np.lib.format.open_memmap(str(path), mode="w+", dtype=dtype, shape=shape), path


def channel_index(names, ref_channel, what):
    """Index of the reference channel: by name, or -- when the stack carries no names -- by its decimal index."""
    if ref_channel in names:
        return names.index(ref_channel)
    if re.fullmatch(r"\d+", ref_channel) and int(ref_channel) < len(names):
        return int(ref_channel)
    raise ValueError(f"Reference channel {ref_channel!r} is not among the channels of {what}: {names}")


# ---- the two stages ---------------------------------------------------------------------------------------------------
def _load_cycles(cfg):
    """[(cycle id, (C, Z, Y, X) array, channel names)] in cycle order."""
    if cfg.input_type == "CycleStack":
        # one file holding every cycle: a (cycles, C, Z, Y, X) .npy array; the TIFF form needs the per-cycle channel
        # layout of the OME-XML, which is not rebuilt
        path = cfg.paths[0]
        if path.suffix.lower() != ".npy":
            arr, names, xml = read_stack_meta(path)
            cfg.input_ome = xml
            return split_cycle_stack(arr, names, cfg.ref_channel)
        arr = np.load(path, mmap_mode="r")
        if arr.ndim != 5:
            raise ValueError(f"{path}: a CycleStack .npy must be (cycles, C, Z, Y, X), got shape {arr.shape}")
        names = [str(i) for i in range(arr.shape[1])]
        return [(k + 1, arr[k], names) for k in range(arr.shape[0])]
    if cfg.input_type == "CycleBuilder":
        out = []
        for cyc, chans in sorted(cfg.paths.items()):
            planes = []
            for ch, p in chans.items():
                a = np.load(p, mmap_mode="r") if p.suffix.lower() == ".npy" else read_stack(p)[0]
                a = a.reshape((-1,) + a.shape[-2:])          # one channel per file: whatever leads Y, X is Z
                planes.append(a)
            zmax = max(pl.shape[0] for pl in planes)
            stack = np.zeros((len(planes), zmax) + planes[0].shape[1:], planes[0].dtype)
            for c, pl in enumerate(planes):
                stack[c, :pl.shape[0]] = pl
            out.append((cyc, stack, list(chans)))
        return out
    out = []
    for cyc, p in sorted(cfg.paths.items()):
        arr, names, xml = read_stack_meta(p)
        if getattr(cfg, "input_ome", None) is None:
            cfg.input_ome = xml              # the first cycle's description is the one passed through to the outputs
        out.append((cyc, arr, names))
    return out


def run_feature_reg(cfg, cycles, log=print):
    """run_feature_reg / do_feature_reg / transform_and_save_freg_imgs (__main__.py:226-286,440-516): every cycle's
    max-projected reference channel against the reference cycle's -> one 2x3 matrix per cycle -> every page padded to the
    common shape and transformed.  Returns the transformed cycles (in memory) and the matrices."""
    from . import FeatureRegistrator, pad_to_shape, transform_img_with_tmat
    from .shared_modules.utils import max_project_and_normalize
    target = (max(a.shape[2] for _, a, _ in cycles), max(a.shape[3] for _, a, _ in cycles))
    freg = FeatureRegistrator()
    for k, v in cfg.feature.feature_kwargs().items():
        setattr(freg, k, v)
    by_id = {cyc: (arr, names) for cyc, arr, names in cycles}
    if cfg.ref_cycle not in by_id:
        raise ValueError(f"ReferenceCycle {cfg.ref_cycle} is not among the input cycles {sorted(by_id)}")
    ref_arr, ref_names = by_id[cfg.ref_cycle]
    ref_img, _ = pad_to_shape(max_project_and_normalize(ref_arr[channel_index(ref_names, cfg.ref_channel, "the reference cycle")]),
                              target)
    freg.ref_img = ref_img
    tmats, out = {}, []
    for n, (cyc, arr, names) in enumerate(cycles):
        log(f"Processing Cycle {cyc} [{n + 1}/{len(cycles)}]")
        if cyc == cfg.ref_cycle:
            log("Skipping as it is a reference cycle")
            tmats[cyc] = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
        else:
            mov, _ = pad_to_shape(max_project_and_normalize(arr[channel_index(names, cfg.ref_channel, f"cycle {cyc}")]), target)
            freg.mov_img = mov
            tmats[cyc] = freg.register(reuse_ref_img=True)
        res = np.empty(arr.shape[:2] + target, arr.dtype)
        for c in range(arr.shape[0]):
            for z in range(arr.shape[1]):
                res[c, z] = transform_img_with_tmat(np.ascontiguousarray(arr[c, z]), target, tmats[cyc])
        out.append((cyc, res, names))
    return out, tmats


def run_optflow_reg(cfg, cycles, writer, log=print):
    """run_opt_flow_reg / register_and_save_ofreg_imgs (__main__.py:320-437,532-609): cycle k+1 is registered against the
    warped reference channel of cycle k (the chain), every page of the cycle is warped with that one flow and written."""
    from . import OptFlowRegistrator, Warper
    from .shared_modules.utils import max_project_and_normalize
    reg = OptFlowRegistrator()
    for k, v in cfg.optflow.optflow_kwargs().items():
        setattr(reg, k, v)
    warper = Warper()
    warper.tile_size, warper.overlap = reg.tile_size, reg.overlap
    ref_img = None
    for n, (cyc, arr, names) in enumerate(cycles):
        log(f"Processing Cycle {cyc} [{n + 1}/{len(cycles)}]")
        ch = channel_index(names, cfg.ref_channel, f"cycle {cyc}")
        mov_img = max_project_and_normalize(arr[ch], on_device=True)
        dst = writer(n, cyc, arr)
        if n == 0:
            log("Skipping as it is a reference image")
            ref_img = mov_img
            dst[...] = arr
            continue
        reg.ref_img, reg.mov_img = ref_img, mov_img
        flow = reg.register()
        warper.image, warper.flow = mov_img, flow
        ref_img = warper.warp()                      # will be used in the next cycle (:424)
        log(f"Saving Cycle {cyc} [{n + 1}/{len(cycles)}]")
        warper.flow = flow
        pages = [np.ascontiguousarray(arr[c, z]) for c in range(arr.shape[0]) for z in range(arr.shape[1])]
        rows = [dst[c, z] for c in range(arr.shape[0]) for z in range(arr.shape[1])]
        if all(isinstance(r, np.ndarray) and r.flags.c_contiguous and r.flags.writeable for r in rows):
            warper.warp_pages(pages, out=rows)       # straight into the output memory map
        else:
            for r, w in zip(rows, warper.warp_pages(pages)):
                r[...] = w


def _writer(cfg, cycles, stage, fmt):
    """dst(n, cyc, arr) -> writable (C, Z, Y, X) view for cycle n: one stack of all cycles or one file per cycle
    (__main__.py:376-409), named like the reference's outputs."""
    C0, zmax = cycles[0][1].shape[0], max(a.shape[1] for _, a, _ in cycles)
    H, W, dtype = cycles[0][1].shape[2], cycles[0][1].shape[3], cycles[0][1].dtype
    written, state = [], {}
    xml = getattr(cfg, "input_ome", None)
    names_of = {cyc: names for cyc, _, names in cycles}
    if cfg.to_stack:
        total_c = sum(a.shape[0] for _, a, _ in cycles)
        shape = (1, total_c, zmax, H, W)
        # channel names of the stack carry their cycle, as the reference's do ("c01 DAPI", stack_builder.py:124-134)
        all_names = [f"c{cyc:02d} {strip_cycle_info(n)}" for cyc, _, names in cycles for n in names]
        desc = ome_passthrough(xml, shape, dtype, all_names) if fmt == "tif" else None
        state["mm"], p = create_output(cfg.out_dir / f"{cfg.out_prefix}{stage}_result_stack.tif", shape, dtype, fmt, desc)
        written.append(p)

    def dst(n, cyc, arr):
        if cfg.to_stack:
            c0 = n * C0                               # cross-cycle channel id as the reference computes it (:415,429)
            return state["mm"][0, c0:c0 + arr.shape[0], :arr.shape[1]]
        shape = (1, arr.shape[0], zmax, H, W)
        desc = ome_passthrough(xml, shape, dtype, names_of[cyc]) if fmt == "tif" else None
        mm, p = create_output(cfg.out_dir / f"{cfg.out_prefix}{stage}_result_cyc{cyc:03d}.tif", shape, dtype, fmt, desc)
        written.append(p)
        state[cyc] = mm
        return mm[0, :, :arr.shape[1]]

    return dst, written, state


def run(config_path, log=print):
    log("Started\n")
    cfg = read_config(config_path)
    cfg.out_dir.mkdir(parents=True, exist_ok=True)
    cycles = _load_cycles(cfg)
    all_paths = [q for p in cfg.paths.values() for q in (p.values() if isinstance(p, dict) else [p])]
    fmt = cfg.output_format or ("npy" if all(Path(p).suffix.lower() == ".npy" for p in all_paths) else "tif")
    if fmt not in ("tif", "npy"):
        raise ValueError("Output: OutputFormat must be tif or npy")
    outputs = []
    if cfg.feature is not None:
        log("Performing linear feature based image registration")
        cycles, tmats = run_feature_reg(cfg, cycles, log)
        dst, written, state = _writer(cfg, cycles, "feature_reg", fmt)
        for n, (cyc, arr, _) in enumerate(cycles):
            dst(n, cyc, arr)[...] = arr
        for mm in state.values():
            mm.flush()
        with open(cfg.out_dir / f"{cfg.out_prefix}feature_reg_parameters.json", "w") as f:
            json.dump({str(c): m.tolist() for c, m in tmats.items()}, f, indent=1)
        outputs += written
        log("Finished\n")
    if cfg.optflow is not None:
        shapes = {a.shape[2:] for _, a, _ in cycles}
        if len(shapes) > 1:
            raise ValueError("Image dimensions do not match: run FeatureReg first (add a FeatureReg section)")
        log("Performing non-linear optical flow based image registration")
        dst, written, state = _writer(cfg, cycles, "optflow_reg", fmt)
        run_optflow_reg(cfg, cycles, dst, log)
        for mm in state.values():
            mm.flush()
        outputs += written
        log("Finished\n")
    return outputs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1 or argv[0] in ("-h", "--help"):
        print("usage: python -m microaligner_amd config.yaml", file=sys.stderr)
        return 2
    if not os.path.exists(argv[0]):
        print(f"config file {argv[0]} does not exist", file=sys.stderr)
        return 2
    for p in run(argv[0]):
        print("wrote", p)
    return 0
