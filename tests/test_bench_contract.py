"""CPU checks of the committed measurement evidence (no GPU, no reference): the bench lines under profiles/ keep bench.py's
contract, and what they say mechanically agrees with the rocprofv3 summaries committed beside them.

* the kernel names a line carries (from the library's own `scl_*_kernel_names`, C ABI 6) are kernels of the kernel-trace
  summary of the same family -- the line can be matched against the profile without a human in between;
* `roofline.frac` is algorithmic bytes / average launch time / 8 TB/s, and the trace summary's average duration of the
  same kernel agrees with the line's HIP-event average (two runs on two boxes: within 12 %);
* the default line ends with `summary` (<= 1 KB), one entry per BASELINE.json single-GPU configuration;
* a quoted PMC traffic figure belongs to the same kernel sources (`csrc_sha`) as the line.
"""
import glob
import json
import os
import re

import pytest

from conftest import ROOT

PROF = os.path.join(ROOT, "profiles")
ROUND = "r06"


def _line(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError(f"no JSON line in {path}")


def _trace(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(?:void )?(\S.*?)\s+calls=\s*(\d+)\s+avg_us=\s*([0-9.]+)", ln)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


FAMILIES = sorted(os.path.basename(p)[len(ROUND) + 1:-len("_kernel_trace_summary.txt")]
                  for p in glob.glob(os.path.join(PROF, f"{ROUND}_*_kernel_trace_summary.txt")))


@pytest.mark.skipif(not FAMILIES, reason="no round-5 evidence committed yet")
@pytest.mark.parametrize("family", FAMILIES)
def test_line_matches_its_kernel_trace(family):
    line = _line(os.path.join(PROF, f"{ROUND}_bench_{family}.json"))
    trace = _trace(os.path.join(PROF, f"{ROUND}_{family}_kernel_trace_summary.txt"))
    assert line["round_trip_verified"] is True and line["data"] == "synthetic" and line["scaling"] == "weak"
    for side in ("roofline_encode", "roofline_decode"):
        r = line[side]
        frac = r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0
        assert abs(frac - r["frac"]) < 2e-4 and r["peak"] == 8000.0 and r["bound"] == "hbm"
        if line["config"]["coder"] in ("rans", "tans"):
            # the library named the kernel: it must be a kernel of the committed trace, at a duration that agrees
            assert r["kernel"] in trace, f"{family}: {r['kernel']} not among {sorted(trace)[:6]}"
            calls, avg_us = trace[r["kernel"]]
            assert calls >= 10
            assert abs(avg_us * 1e-3 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.12, (family, side, avg_us, r["avg_launch_ms"])
        if r["traffic"] is not None:
            entries = json.load(open(os.path.join(PROF, "traffic.json")))["entries"]
            mine = [e for e in entries if e["key"] == line["traffic_key"]]
            assert mine and mine[0]["csrc_sha"] == line["csrc_sha"]


@pytest.mark.skipif(not os.path.exists(os.path.join(PROF, f"{ROUND}_bench_headline_full.json")),
                    reason="no round-5 default line committed yet")
@pytest.mark.parametrize("name", ["headline_full", "driver_style"])
def test_default_line_contract(name):
    line = _line(os.path.join(PROF, f"{ROUND}_bench_{name}.json"))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "MB/s" and line["n_gpus"] == 1
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "u32"
    assert line["config"]["chunks_per_gpu"] == 262144 and line["config"]["chunk_len"] == 4096
    assert line["config"]["NUM_BITS_OUT"] == 1 and line["config"]["RANGE_FACTOR"] == 1 << 16
    assert abs(line["value"] - 2 ** 30 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) / line["value"] < 1e-3
    for k in ("roofline", "cpu_baseline", "cpu_baseline_restatement", "roofline_dense", "other_configs", "summary"):
        assert k in line, k
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert list(line)[-1] == "summary", "the driver keeps the tail of the line: summary must be the last key"
    s = line["summary"]
    assert len(json.dumps(s)) <= 1024
    assert set(s) == {"headline", "c[1]", "c[2]", "c[3]", "c[3]b"}
    for tag, e in s.items():
        assert "error" not in e, (tag, e)
        assert e["enc"] > 0 and e["dec"] > 0 and e["dec_dd"] > 0 and 0 < e["f_enc"] < 1 and 0 < e["f_dec"] < 1
        assert e["cpu_c"] > 0 and e["cpu_py"] > 0 and e["MBps"] > 0
    # the summary repeats the headline's own numbers
    assert abs(s["headline"]["enc"] - line["roofline_encode"]["avg_launch_ms"]) < 1e-3
    assert abs(s["headline"]["f_dense"] - line["roofline_dense"]["frac"]) < 1e-3
    d = line["roofline_dense"]
    assert d["traffic"] is None or d["traffic"] > 1.8 * d["algorithmic_bytes_per_launch"]  # two transfers more than algorithmic


# ---- round 6: the line the driver records (a 21.6 KB line left BENCH_r05.json unparsed) -------------------
def _import_bench():
    import importlib
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


FULL_RECORDS = sorted(glob.glob(os.path.join(PROF, "r0[5-9]_bench_driver_style*.json")) +
                      glob.glob(os.path.join(PROF, "r0[5-9]_bench_headline_full.json")) +
                      glob.glob(os.path.join(PROF, "r0[6-9]_bench_full*.json")))


@pytest.mark.parametrize("path", FULL_RECORDS, ids=[os.path.basename(p) for p in FULL_RECORDS])
def test_compact_line_is_small_and_carries_the_contract(path):
    bench = _import_bench()
    full = _line(path)
    if "other_configs" not in full and "full_record" in full:
        pytest.skip("already a compact line")
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 6144, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in back, k
    assert list(back)[-1] == "summary"
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    c = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert "other_configs" not in back and "stream_file" not in back
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]


COMPACT_LINES = sorted(glob.glob(os.path.join(PROF, "r0[6-9]_bench_line*.json")))


@pytest.mark.parametrize("path", COMPACT_LINES, ids=[os.path.basename(p) for p in COMPACT_LINES])
def test_committed_driver_line_parses_and_is_small(path):
    """the bytes bench.py printed on the GPU box, as the driver sees them"""
    raw = [ln for ln in open(path) if ln.startswith("{")]
    assert len(raw) == 1
    assert len(raw[0]) < 6144
    line = json.loads(raw[0])
    assert line["roofline"]["frac"] > 0 and list(line)[-1] == "summary"
    if line["n_gpus"] == 1:  # (the multi-rank line, taken with --no-cpu-baseline on ranks that share one GPU, has none)
        assert line["cpu_baseline"]["value"] > 0
    else:
        assert line["multi_gpu"]["ranks"] == line["n_gpus"] and line["gather"]["verified"] is True


def test_compact_line_survives_an_oversized_record():
    """whatever the full record grows to, the printed line sheds optional objects until it fits"""
    bench = _import_bench()
    full = _line(FULL_RECORDS[0])
    full["cpu_baseline"]["sample"] = "x" * 5000
    full["config"]["workload"] = "w" * 3000
    for k in ("roofline_encode", "roofline_decode", "roofline_dense"):
        full[k]["kernel"] = "k" * 800
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and "summary" in line
