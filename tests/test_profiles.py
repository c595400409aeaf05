"""Evidence hygiene (CPU): the committed ``profiles/rNN_pmc_chain.json`` -- what ``bench.py`` reports as
``roofline.traffic`` -- must be exactly what ``tools/pmc_chain_json.py`` derives from the two committed PMC summaries of
the same round, and the bench line committed next to it must carry the contract fields."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rounds():
    tags = sorted(re.match(r"(r\d+)_pmc_chain\.json", os.path.basename(p)).group(1)
                  for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_chain.json")))
    return tags


@pytest.mark.parametrize("tag", _rounds())
def test_pmc_chain_json_regenerates_from_the_summaries(tag):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_chain_json.py"), tag, "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    doc = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_chain.json")))
    # the text tower's chain launch: self_chain_groups_kernel<5> once it became the fp32 default (round 5), the fused kernel before
    text = doc.get("self_chain_groups_kernel<5>") or doc["self_chain_fused_kernel<5, 0>"]
    assert text["algorithmic_bytes"] == 2 * 12 * 64 * 8 * 77 * 77 * 4 + 64 * 77 * 77 * 4
    # the kernel reads every slab once: measured traffic within 15 % above the algorithmic bytes, never below them
    # (round 6 on the launch skips the chunks above the diagonal of the causal text tower: the floor is what it asks for)
    measured = text["fetch_bytes"] + text["write_bytes"]
    assert text.get("causal_requested_bytes", text["algorithmic_bytes"]) <= measured <= 1.15 * text["algorithmic_bytes"]


@pytest.mark.parametrize("tag", _rounds())
def test_committed_bench_line_has_the_contract_fields(tag):
    path = os.path.join(ROOT, "profiles", tag + "_bench.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["roofline"]["bound"] in ("hbm", "mfma") and 0 < line["roofline"]["frac"] <= 1
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0


def test_newest_round_is_the_one_bench_reads():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    src = open(spec.origin).read()
    assert "_pmc_chain.json" in src and "glob" in src      # bench.py picks the newest round's file, not a fixed name
