"""The driver reads ONE JSON line from bench.py's stdout; round 4's grew to 20 KB and was not parsed.  bench.compact_line() must keep the
printed line under 6 KB whatever the attached measurements carry, with the contract's keys, `roofline` and `cpu_baseline` intact."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _canned():
    # a real full-size line: round 4's 20 KB one
    return json.loads(open(os.path.join(ROOT, "profiles", "r04_bench.json")).read().strip().splitlines()[-1])


def test_compact_line_is_small_and_round_trips():
    full = _canned()
    assert len(json.dumps(full)) > 16000                      # the input really is the line that broke the reader
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT == 6144
    assert "\n" not in s
    back = json.loads(s)
    assert back == line
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float(f"{full['value']:.6g}") and back["n_gpus"] == 1 and back["scaling"] == "weak"
    assert back["config"]["workload"]
    r = back["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["all_cores"]["cores"] == 256 and c["reference_check"]["oracle_equals_reference"] is True
    assert back["verified"]["ok"] is True and back["exact_mode"]["value"] > 0
    assert back["e2e10"]["value"] > 0 and back["strong_scan"]["value"] > 0 and back["ingest"]["y_plane"]["pipelined_fps"] > 0
    assert all(len(v) <= 160 for v in back["config"].values() if isinstance(v, str))


def test_compact_line_sheds_optional_parts_before_it_outgrows_the_reader():
    full = _canned()
    full["kernels"] = {f"kernel_with_a_long_name_{i:04d}": {"avg_ms": 1.0 / 3, "frac": 0.1} for i in range(200)}
    full["boundary"] = {f"path_{i}_fps": i for i in range(300)}
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    for k in CONTRACT + ("roofline", "cpu_baseline", "verified"):
        assert k in line, k


def test_compact_line_of_the_sharded_variants():
    # --scaling strong / --workload e2e10 lines carry no roofline of their own; they must still be small and parse
    full = _canned()
    for key in ("strong_scan", "e2e10"):
        line = bench.compact_line({**{k: full[k] for k in CONTRACT}, "collectives": full["collectives"], key: full[key]})
        assert len(json.dumps(line)) < 2048 and line[key]["value"] > 0


def test_emit_prints_the_compact_line_last(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _canned()
    bench.emit(full)
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    assert len(last) < bench.LINE_LIMIT and json.loads(last)["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert json.loads(cap.err.strip().splitlines()[-1])["bench_detail"]["kernels"] == full["kernels"]      # the detail goes to stderr ...
    assert json.load(open(tmp_path / bench.DETAIL_NAME))["ingest"] == full["ingest"]                      # ... and next to the script


def test_round5_detail_compacts_to_the_committed_line():
    """profiles/r05_bench_detail.json is what bench.py wrote next to the line profiles/r05_bench.json in the same run: the line builder must
    reproduce the printed line from the detail, under the limit, with the full-batch verification in it"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_detail.json")))
    printed = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench.json")).read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT

    def within(a, b, path=""):
        """everything round 5 printed is still in the line, unchanged (round 6 added keys -- verified frame counts of the attached
        configurations, the quota flag -- which a round-5 detail file leaves null); strings may only have been cut at a different length"""
        if isinstance(a, dict):
            assert isinstance(b, dict), path
            for k in a:
                assert k in b, path + "/" + k
                within(a[k], b[k], path + "/" + k)
        elif isinstance(a, str) and a.endswith("..."):
            assert b.startswith(a[:-3]), path
        else:
            assert a == b, path
    within(printed, line)
    assert line["verified"]["ok"] is True and line["verified"]["frames"] == line["config"]["frames_per_gpu"] == 10000
    assert line["roofline"]["bound"] in ("hbm", "fp32-valu") and 0 < line["roofline"]["frac"] < 1
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1


def test_round6_detail_compacts_to_the_committed_line():
    """profiles/r06_bench.json is the line bench.py printed, profiles/r06_bench_detail.json the detail of the same run: the builder reproduces
    the line exactly, and the line carries what round 6 was about -- the mode the value is in, and full-size verification of configs[3] / [4]"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    printed = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert line == printed
    cfg = line["config"]
    assert cfg["analysis_mode"] == "linear" and cfg["library_default_mode"] == "exact" and "exact_mode" in cfg["value_is_in_mode"]
    assert not cfg["workload"].endswith("...") and "configs[1]" in cfg["workload"]
    assert line["exact_mode"]["value"] < line["value"]
    # the erase writes a resident copy of the frames; rounds 1-5's step (in place + rectangles put back) is reported next to it
    assert "writable copy" in cfg["erase"] and line["in_place_erase_step"]["value"] < line["value"]
    assert line["verified"]["ok"] is True and line["verified"]["frames"] == cfg["frames_per_gpu"] == 10000
    ss = line["strong_scan"]
    assert ss["verified_ok"] is True and ss["verified_frames"] == ss["frames_total"] == 107892
    assert ss["lgd_equals_cpu_oracle"] is True and ss["scanlogo_quota_hit"] is True
    sl = line["configs"]["scanlogo_60min"]
    assert sl["verified_frames"] == sl["frames"] == 107892 and sl["quota_hit"] is True
    e2e = line["e2e10"]
    assert e2e["verified_ok"] is True and e2e["verified_whole_stream"] is True and e2e["verified_frames"] == e2e["frames_total"]
    assert line["roofline"]["traffic_source"].startswith("profiles/r06_") and 0 < line["roofline"]["frac"] < 1
