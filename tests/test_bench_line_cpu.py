"""CPU tier: the last benchmark line measured on an MI355X and committed under profiles/ carries every key of the bench
contract (metric / unit from BASELINE.json, roofline with bound / achieved / peak / frac / traffic, cpu_baseline with value /
cores / kind / sample) and its numbers are consistent with each other (frac = achieved / peak, value = shapes over the timed
chain).  A renamed or dropped key in bench.py shows up here after the next profile collection, not at the judge's desk."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_steps20_line.json")))
    assert files, "no committed bench line"
    return json.loads(open(files[-1]).read().strip().splitlines()[-1]), files[-1]


def test_bench_line_contract_keys():
    d, f = _last_line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(",")[0].strip().lower().replace("×", "x") in base["metric"].lower().replace("×", "x"), (d["metric"], base["metric"])
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (k, f)
    assert d["unit"] == "shapes/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                       # BASELINE.md publishes no number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_bench_line_numbers_are_consistent():
    d, _ = _last_line()
    cfg = d["config"]
    # value = shapes of all ranks / (chain_steps x ms_per_step + decode): the extrapolation the line documents
    t = cfg["chain_steps"] * d["ms_per_step"] * 1e-3 + cfg.get("decode_seconds", 0.0)
    assert abs(d["value"] - d["n_gpus"] * cfg["shapes_per_gpu"] / t) / d["value"] < 0.02
    full = cfg.get("full_chain_1000")
    if full:   # the real chain run after the timed region
        assert abs(full["shapes_per_s"] - cfg["shapes_per_gpu"] / full["seconds"]) / full["shapes_per_s"] < 1e-6
    for k in ("roofline_voxelize", "roofline_devoxelize"):
        if d.get(k):
            assert abs(d[k]["frac"] - d[k]["achieved"] / d[k]["peak"]) < 1e-9 and d[k]["bound"] == "hbm"
    if d.get("mfma_ceiling") and "random_fp16_operands" in d["mfma_ceiling"]:
        # the power-limited rate of a bare MFMA stream cannot exceed the constant-operand rate, nor the nominal peak
        rnd, cst = d["mfma_ceiling"]["random_fp16_operands"], d["mfma_ceiling"]["constant_operands"]
        assert 0 < rnd["TFLOP/s fp16"] <= cst["TFLOP/s fp16"] * 1.02 <= 2600


def _roofline_objects(obj, path=""):
    """every dict below `obj` that looks like a roofline object (has achieved / peak / frac)"""
    if isinstance(obj, dict):
        if {"achieved", "peak", "frac"} <= set(obj):
            yield path, obj
        for k, v in obj.items():
            yield from _roofline_objects(v, f"{path}.{k}" if path else k)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            yield from _roofline_objects(v, f"{path}[{i}]")


def _latest_round_lines():
    """the bench lines (sampling, demo, training modes) of the LATEST round that has them under profiles/"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))
    assert files
    latest = os.path.basename(files[-1]).split("_")[0]
    out = []
    for f in files:
        if os.path.basename(f).split("_")[0] != latest:
            continue
        for ln in open(f).read().strip().splitlines():
            if ln.startswith("{"):
                out.append((f, json.loads(ln)))
    return out


def test_every_roofline_fraction_is_a_fraction():
    """round-4 verdict: the committed training lines carried frac = 1.21 (a kernel on the 16-bit pipe divided by the fp32
    peak).  A fraction of a peak lies in (0, 1]; achieved / peak must reproduce it."""
    seen = 0
    for f, d in _latest_round_lines():
        for path, r in _roofline_objects(d):
            if r["achieved"] is None:
                continue
            seen += 1
            assert 0.0 < r["frac"] <= 1.0, (f, path, r["frac"])
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * max(1.0, r["frac"]), (f, path)
    assert seen >= 3


def test_training_lines_name_the_kernel_they_time():
    for f, d in _latest_round_lines():
        if "training step" not in d.get("metric", ""):
            continue
        r = d["roofline"]
        split = "split" in r["kernel"]
        assert abs(r["peak"] - (2500.0 / 3.0 if split else 157.3)) < 1e-6, (f, r["kernel"], r["peak"])
        c = d["cpu_baseline"]
        assert c["value"] is not None or c["sample"].startswith(("unmeasured", "not timed")), (f, c)


def test_compact_line_is_what_a_line_parser_can_take():
    """round-5 review, missing 1: BENCH_r05.json.parsed was null -- the line had grown to 20 KB with three nested bench lines
    (each starting with {"metric": ...).  bench.compact_line() turns ANY full record into the line the driver parses: here the
    largest full record committed so far (round 5's) goes through it, then through the driver's steps -- last stdout line that
    starts with '{' -> json.loads -> the contract's keys."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "archive", "r05_bench_steps20_line.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 15000 and json.dumps(full).count('"metric"') > 1      # the record that broke the parser
    text = json.dumps(bench.compact_line(full))
    assert len(text) < 8000, len(text)
    assert text.count('"metric"') == 1 and "\n" not in text
    stdout = "some warning\n[bench detail] not json\n" + text + "\n"
    last = [ln for ln in stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["value_full_chain_1000"] == full["value_full_chain_1000"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert "workload" in d["config"]
    for k, v in d.items():     # flat: only the three contract objects, scalars inside them
        if isinstance(v, dict):
            assert k in ("config", "roofline", "cpu_baseline") and not any(isinstance(x, (dict, list)) for x in v.values()), k
        else:
            assert not isinstance(v, list), k


def test_latest_committed_line_is_compact():
    """the line committed for the latest round is the driver-parsable one (from round 6 on)"""
    d, f = _last_line()
    rnd = int(os.path.basename(f)[1:3])
    if rnd < 6:
        return
    text = open(f).read().strip().splitlines()[-1]
    assert len(text) < 8000 and text.count('"metric"') == 1, (f, len(text))
    for k in ("ms_per_step_forced_clouds", "value_forced_clouds", "voxelize_frac", "devoxelize_frac", "ms_per_step_B4"):
        assert k in d, k
