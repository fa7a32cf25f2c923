"""-m gpu: bench.py itself -- the contract the driver depends on (one JSON line, the metric's keys, the roofline and
cpu_baseline objects) at a reduced size, and the N > 1 launch path: `--gpus 2` starts its own two ranks, which here share
the one device of the box and rendezvous over gloo (LION_BENCH_BACKEND=gloo; RCCL needs one device per rank).  What this
covers of the 8-GPU run without 8 GPUs: rank discovery, per-rank seeding, the barrier + MAX-over-ranks timing, the
bucketed gradient averager with its overlap hooks armed at world size 2, the whole-job aggregation of `value`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None, timeout=600, detail=False):
    """runs bench.py the way the driver does and parses its stdout the way the driver does: the LAST line that starts with
    '{' is the record; it must be one compact object (round 5's 20-KB line with nested bench lines was unparseable).
    detail=True also returns the full record bench.py wrote to --detail-file."""
    import tempfile
    e = dict(os.environ, **(env or {}))
    dfile = os.path.join(tempfile.mkdtemp(prefix="lion_bench_"), "detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", dfile] + list(args),
                         capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 8000 and lines[0].count('"metric"') == 1, len(lines[0])
    assert out.stdout.rstrip().splitlines()[-1] == lines[0]          # nothing after the record
    line = json.loads(lines[0])
    for k, v in line.items():                                         # flat: contract objects one level deep, scalars inside
        if isinstance(v, dict):
            assert k in ("config", "roofline", "cpu_baseline"), k
            assert not any(isinstance(x, (dict, list)) for x in v.values()), (k, v)
    if not detail:
        return line
    mode = [a for a in args if a.startswith("train_")]
    if mode:
        dfile = dfile.replace(".json", f"_{mode[0]}.json")
    return line, json.load(open(dfile))


def test_sample_line_contract_single_rank():
    line, d = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2", "--repeats", "2", "--side-lines",
                     "--side-steps", "2", "--forced-steps", "8", "--small-batches", "1", detail=True)
    # the compact line: the contract's keys + the round-6 scalars (forced clouds, small batch, fractions, census)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_full_chain_1000",
              "ms_per_step_full_chain", "ms_per_step_forced_clouds", "value_forced_clouds", "voxelize_frac", "devoxelize_frac"):
        assert k in line, k
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    lc = line["config"]
    assert "workload" in lc and lc["forced_clouds_steps"] == 8 and lc["forced_clouds_ms_per_step"] > 0
    assert 0.0 <= lc["forced_clouds_conv1_empty_tile_frac"] <= 1.0 and 0.0 <= lc["chain_conv1_empty_tile_frac"] <= 1.0
    assert lc["vendor_library_fallbacks_in_step"] == 0
    assert "launches_per_step" in lc and "aten_kernels_in_step" in lc    # (from profiles/r*_step_census_B32.json: null at B = 2)
    assert d["config"]["small_batches"]["1"]["ms_per_step"] > 0
    assert d["config"]["forced_clouds"]["tiles"]["steps"] == [0, 2, 4, 6, 7]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert len(d["config"]["ms_per_step_all_runs"]) == 2 and "workload" in d["config"]
    # round 4: the in-step conv instantiation is THE roofline, the plain kernel a side key; the real 1000-step chain
    # rides along with a short timed region; the chain structure that was replayed is on the line
    assert "PRO=true" in d["roofline"]["kernel"] and d["roofline_plain_kernel"]["frac"] > 0
    fc = d["config"]["full_chain_1000"]
    assert fc["seconds"] > 0 and abs(fc["shapes_per_s"] - 2 / fc["seconds"]) < 1e-9
    chains = d["config"]["streams"]["captured_chains [global prior, local prior]"]
    assert len(chains) == 2 and chains[0]["streams"] == 1 and chains[1]["geometry_graphs"] >= 1
    for k in ("roofline_chamfer", "roofline_emd", "latency_bound_operators", "roofline_voxelize", "roofline_devoxelize"):
        assert k in d, k
    assert d["cpu_baseline"]["ms_per_step_B32"] > 0 and len(d["cpu_baseline"]["per_kernel_B32"]) == 3
    # round 5: the real chain as top-level / config scalars (the driver's record keeps scalars), the aggregate voxelize /
    # devoxelize rooflines over the 14 + 14 calls of a forward, no vendor-library fallback inside the step, and the three
    # training configurations as side lines with the weight-gradient roofline of the kernel that runs (frac <= 1)
    assert abs(d["value_full_chain_1000"] - 2 / fc["seconds"]) < 1e-9 and d["ms_per_step_full_chain"] == fc["ms_per_step"]
    assert d["config"]["full_chain_1000_shapes_per_s"] == d["value_full_chain_1000"]
    for k in ("roofline_voxelize_forward_total", "roofline_devoxelize_forward_total"):
        assert d[k]["bound"] == "hbm" and 0 < d[k]["frac"] <= 1 and d[k]["us_per_forward"] > 0
    assert d["config"]["vendor_library_fallbacks_in_step"] == 0
    for mode in ("train_vae", "train_prior", "train_prior_clip"):
        ln = d["training_side_runs"][mode]
        assert "error" not in ln, ln
        assert ln["value"] > 0 and 0 < ln["roofline"]["frac"] <= 1 and "split" in ln["roofline"]["kernel"]
        assert abs(ln["roofline"]["peak"] - 2500.0 / 3.0) < 1e-6
        assert ln["cpu_baseline"]["value"] > 0 and ln["cpu_baseline"]["kind"] == "port"
        assert d["config"][f"{mode}_samples_per_s"] == ln["value"]


def test_sample_strong_scaling_two_ranks_over_gloo():
    """--shapes-total: a fixed job split over the ranks (SURVEY.md 8e read as 32 shapes in total): 5 shapes over 2 ranks =
    3 + 2, `value` = the whole job's 5 shapes over the slower rank's time, scaling = strong."""
    d = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--shapes-total", "5", "--repeats", "1", "--no-dense-check",
               "--no-full-chain", "--forced-steps", "0", "--small-batches", "", env={"LION_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shapes_total"] == 5
    assert d["config"]["shapes_per_gpu"] == 3      # rank 0's share (remainder to the low ranks)
    per_step_s = d["ms_per_step"] / 1e3
    assert abs(d["value"] - 5 / (1000.0 * per_step_s + d["config"]["decode_seconds"])) < 1e-6 * d["value"] + 1e-9


def test_sample_two_ranks_share_the_device_over_gloo():
    d = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--repeats", "1", "--no-dense-check",
               "--no-full-chain", "--forced-steps", "0", "--small-batches", "", env={"LION_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["shapes_per_gpu"] == 2
    assert "2 independent rank(s)" in d["config"]["parallelism"]
    # value is the whole job's: both ranks' shapes over the slower rank's time
    per_step_s = d["ms_per_step"] / 1e3
    assert abs(d["value"] - 2 * 2 / (1000.0 * per_step_s + d["config"]["decode_seconds"])) < 1e-6 * d["value"] + 1e-9


def test_train_vae_two_ranks_bucketed_averaging_over_gloo():
    d = _bench("--gpus", "2", "--mode", "train_vae", "--steps", "2", "--warmup", "1", "--batch", "2",
               env={"LION_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["unit"] == "samples/s" and d["value"] > 0
    # the captured step runs at world size 2 as well (round 4): gloo's collectives cannot be stream-captured, so the
    # product's GraphedTrainStep replays [forward + backward], averages the buckets eagerly, replays [optimizer step]
    launch = d["config"]["launch"]
    assert "world 2" in d["config"]["gradient_averaging"]
    assert not launch.startswith("eager") and "eager bucket all-reduce (gloo)" in launch and "hipGraph" in launch, launch
    assert d["config"]["final_loss"] == d["config"]["final_loss"]  # not NaN


def test_train_line_counts_the_gpu_steps_fallbacks_only():
    """round-5 review, weak 6: the host baseline's own vendor-library notes must not appear on the training line"""
    line, d = _bench("--gpus", "1", "--mode", "train_prior", "--steps", "2", "--warmup", "1", "--batch", "2", detail=True)
    c = d["config"]
    assert c["vendor_library_fallbacks_total"] == sum(c["vendor_library_fallbacks"].values())
    assert not any(k.startswith("conv_ops.conv3d_module") for k in c["vendor_library_fallbacks"]), c["vendor_library_fallbacks"]
    assert line["config"]["vendor_library_fallbacks_total"] == c["vendor_library_fallbacks_total"]
    assert d["cpu_baseline"]["value"] > 0


def test_train_prior_clip_line():
    d = _bench("--gpus", "1", "--mode", "train_prior_clip", "--steps", "2", "--warmup", "1", "--batch", "2")
    assert "configs[4]" in d["config"]["workload"] and d["value"] > 0
    assert d["config"]["final_loss"] == d["config"]["final_loss"]


def test_train_prior_single_rank_whole_step_graph():
    d = _bench("--gpus", "1", "--mode", "train_prior", "--steps", "2", "--warmup", "1", "--batch", "2")
    assert d["config"]["launch"].startswith("hipGraph replay of the whole step"), d["config"]["launch"]
    assert d["config"]["final_loss"] == d["config"]["final_loss"]
