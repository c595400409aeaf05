"""Copy the outputs of ``tools/gpu_round.sh <tag>`` (+ ``tools/gpu_extras.sh <tag>``) from the scratch directory
``gpurun_out/<tag>/`` into the tracked ``profiles/<tag>_*`` files and regenerate ``profiles/<tag>_pmc_chain.json``.

    python tools/collect_profiles.py r03

Only files that exist are copied; the hand-annotated probe records (``*_cfg5_probe.txt``, ``*_detr_probe.txt`` ...) are
NOT overwritten -- their fresh raw numbers are left next to them as ``profiles/<tag>_<name>.raw.txt`` for merging.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
DIRECT = {"bench.json": "bench.json", "kernel_stats.txt": "bench_kernel_stats.txt", "kernel_split.json": "bench_kernel_split.json",
          "pmc_fetch.txt": "pmc_fetch.txt", "pmc_write.txt": "pmc_write.txt", "cpu_threads.txt": "cpu_threads.txt",
          "chain_kernel_trace.txt": "chain_kernel_trace.txt"}
ANNOTATED = {"cfg5_probe.txt": "cfg5_probe", "attn_bf16_probe.txt": "attn_bf16_probe", "detr_probe.txt": "detr_probe",
             "lxmert_probe.txt": "lxmert_probe"}
for name, out in DIRECT.items():
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, out)))
        print("copied", name)
for name, out in ANNOTATED.items():
    p = os.path.join(src, name)
    if os.path.exists(p):
        target = os.path.join(dst, "%s_%s.txt" % (tag, out))
        if os.path.exists(target):
            target = os.path.join(dst, "%s_%s.raw.txt" % (tag, out))
        shutil.copy(p, target)
        print("copied", name, "->", os.path.basename(target))
if all(os.path.exists(os.path.join(dst, "%s_%s" % (tag, f))) for f in ("pmc_fetch.txt", "pmc_write.txt")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_chain_json.py"), tag], check=True)
