"""Every ``path/to/file.py:LINE`` citation of the reference (header, DESIGN.md, INTEGRATION.md, package and oracle
docstrings) points at a file that exists in the reference checkout and is at least that long.  Skipped where the
reference is not mounted (e.g. on the GPU box)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MMX_REFERENCE", "/root/reference")
CITE = re.compile(r"((?:[A-Za-z_.][\w.-]*/)*[A-Za-z_][\w.-]*\.(?:py|rst|yaml|yml)):(\d+)")


def _line_count(path, cache={}):
    if path not in cache:
        with open(path, errors="replace") as f:
            cache[path] = sum(1 for _ in f)
    return cache[path]


def _resolve(rel):
    """Citations are written relative to the reference root, often with the leading directories elided ('.../x.py',
    bare 'lxmert_lrp.py').  Returns every reference file the citation can mean."""
    if "/.../" in rel:                                              # 'lxmert/.../ExplanationGenerator.py'
        head, tail = rel.split("/.../", 1)
        return glob.glob(os.path.join(REF, head, "**", tail), recursive=True)
    rel = rel.lstrip("./")
    cand = os.path.join(REF, rel)
    if os.path.exists(cand):
        return [cand]
    return [p for p in glob.glob(os.path.join(REF, "**", os.path.basename(rel)), recursive=True)
            if p.endswith("/" + rel)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted")
def test_reference_citations_exist():
    files = [os.path.join(ROOT, "include", "mmx_relevancy.h"), os.path.join(ROOT, "DESIGN.md"),
             os.path.join(ROOT, "INTEGRATION.md")]
    files += glob.glob(os.path.join(ROOT, "transformer-mm-explainability_amd", "*.py"))
    files += glob.glob(os.path.join(ROOT, "transformer-mm-explainability_amd", "csrc", "*.h*"))
    files += glob.glob(os.path.join(ROOT, "oracle", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.c"))
    bad, checked = [], 0
    for path in files:
        text = open(path, errors="replace").read()
        for rel, line in CITE.findall(text):
            if rel.startswith(("transformer-mm-explainability_amd/", "tests/", "tools/", "oracle/", "profiles/", "examples/")):
                continue                                             # citations of this repository's own files
            if os.path.exists(os.path.join(ROOT, rel)) or os.path.exists(os.path.join(ROOT, "transformer-mm-explainability_amd", rel)) \
                    or os.path.exists(os.path.join(ROOT, "tests", rel)):
                continue                                             # one of this repository's own files
            targets = _resolve(rel)
            if not targets:
                bad.append("%s: %s:%s (no such reference file)" % (os.path.relpath(path, ROOT), rel, line))
                continue
            checked += 1
            if all(int(line) > _line_count(t) for t in targets):
                bad.append("%s: %s:%s (file has %d lines)" % (os.path.relpath(path, ROOT), rel, line,
                                                              max(_line_count(t) for t in targets)))
    assert checked > 100, "expected a few hundred resolvable citations, found %d" % checked
    assert not bad, "\n".join(bad)
