"""profiles/pmc_traffic.json carries the HBM-traffic figures bench.py prints as roofline.traffic — and bench.py withholds a figure whose
stamp (sha256 of the kernel's source file at the time of the PMC passes) does not match the source it runs.  A refactor that lands after the
evidence visit would therefore null the driver's roofline.traffic silently (it did in round 4): this test makes that a red CPU suite
instead.  Re-stamp with `gpurun -- 'bash tools/gpu_visit.sh evidence TAG'` + `python tools/collect_profiles.py TAG`."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_stamps_match_the_sources():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert {"csv_parse_regular", "mask_hmac_sha256"} <= set(t), sorted(t)
    stale = []
    for kernel, e in t.items():
        with open(os.path.join(ROOT, "transferia_amd", "csrc", e["source_file"]), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest() != e["source_sha256"]:
                stale.append((kernel, e["source_file"]))
    assert not stale, "PMC evidence is of another build of %s: re-run the evidence visit (tools/gpu_visit.sh evidence) after the last change to those files" % stale


def test_pmc_traffic_is_plausible():
    """traffic is bytes per launch of the kernel at the bench's launch shape: at least the bytes the kernel must move, not absurdly more"""
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    c = t["csv_parse_regular"]
    assert c["rows_per_launch"] == 1 << 20 and 0.8e9 < c["bytes_per_launch"] < 3e9, c
