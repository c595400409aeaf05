"""Regenerate ``profiles/<tag>_pmc_chain.json`` (what ``bench.py`` reports as ``roofline.traffic``) from the two PMC
summaries ``profiles/<tag>_pmc_fetch.txt`` / ``<tag>_pmc_write.txt`` written by ``tools/pmc_summary.py`` (separate
``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` passes over ``bench.py``: ``tools/gpu_round.sh``).

    python tools/pmc_chain_json.py r02            # writes profiles/r02_pmc_chain.json
    python tools/pmc_chain_json.py r02 --check    # exit 1 if the committed json differs from what the txt files give

Per kernel the MAXIMUM over dispatches is taken (= the all-layer launch of the headline step; the bench process also
launches the same kernel on fewer layers).  FETCH_SIZE is doubled (gfx950: 128-B requests of wide coalesced reads are
tallied at 64 B, MI355X_MICROARCH.md HBM section) -- the ``max_bytes_corrected`` column of the summaries.
``tests/test_profiles.py`` runs the check on CPU.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = 64
TOWERS = {"5": ("text", 12, 8, 77), "4": ("image", 12, 12, 50)}       # NT -> (tower, layers, heads, tokens) of cfg 2


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(?:void )?(mmx::\S+?)(<[^>]*>)?\s+(%s)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$" % counter, line.rstrip())
        if m:
            out[m.group(1) + (m.group(2) or "")] = int(m.group(7))
    return out


def build(tag):
    fetch = parse(os.path.join(ROOT, "profiles", tag + "_pmc_fetch.txt"), "FETCH_SIZE")
    write = parse(os.path.join(ROOT, "profiles", tag + "_pmc_write.txt"), "WRITE_SIZE")
    doc = {"source": "profiles/%s_pmc_fetch.txt + profiles/%s_pmc_write.txt (tools/pmc_summary.py over separate rocprofv3 "
                     "--pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py; per-kernel maxima = the all-layer launches)" % (tag, tag),
           "correction": "FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB = 1024 B",
           "generated_by": "tools/pmc_chain_json.py %s" % tag}
    for name in sorted(fetch):
        m = re.match(r"mmx::(self_chain\w*kernel)<(\d+)(?:, (\d+))?", name)
        if not m or name not in write:
            continue
        # self_chain_fused_kernel<NT, dtype, ...> (rounds 1-5) / self_chain_groups_kernel<NT> (round 5 on: the fp32 default)
        key = "%s<%s, %s>" % (m.group(1), m.group(2), m.group(3)) if m.group(3) is not None else "%s<%s>" % (m.group(1), m.group(2))
        entry = {"fetch_bytes": fetch[name], "write_bytes": write[name]}
        if m.group(2) in TOWERS and m.group(1) in ("self_chain_fused_kernel", "self_chain_groups_kernel"):
            _, L, H, N = TOWERS[m.group(2)]
            entry["algorithmic_bytes"] = 2 * L * BATCH * H * N * N * 4 + BATCH * N * N * 4
            if m.group(2) == "5" and int(re.sub(r"\D", "", tag) or 0) >= 6:
                # round 6 on: the text tower's launch carries MMX_CHAIN_CAUSAL (csrc/chain_stream.h) -- 16-byte chunks entirely above
                # the diagonal are not requested, so the counters may read BELOW the algorithmic bytes, never below the requested ones
                live = sum(1 for c in range((N * N + 3) // 4) if not ((4 * c) % N > (4 * c) // N and (4 * c) % N + 3 < N))
                entry["causal_requested_bytes"] = 2 * L * BATCH * H * live * 16 + BATCH * N * N * 4
        doc[key] = entry
    return doc


def main():
    tag = sys.argv[1]
    path = os.path.join(ROOT, "profiles", tag + "_pmc_chain.json")
    doc = build(tag)
    if "--check" in sys.argv:
        have = json.load(open(path))
        if have != doc:
            print("MISMATCH: %s is not what the PMC summaries give" % path)
            sys.exit(1)
        print("ok", path)
        return
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
