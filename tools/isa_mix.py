"""Static instruction mix of a HIP kernel's hottest basic blocks (those that hold MFMAs), from the gfx950 ISA hipcc emits.
Runs on the build container (no GPU):   python tools/isa_mix.py csrc/attention_bf16_v3.hip kv_v3_kernelILi8ELb1 q_v3_kernelILi8ELi2
Why: on CDNA a wave64 VALU instruction occupies its SIMD for 4 cycles and a 16x16x32 bf16 MFMA for 16, so a loop body with V VALU and
M MFMA instructions is issue-bound on the VALU side once V > 4 M per wave (with W waves per SIMD both sides scale by W)."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer-mm-explainability_amd")


def group(c):
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_mfma"): g["mfma"] += v
        elif k.startswith("v_pk_"): g["valu_packed"] += v
        elif k.startswith("v_"): g["valu"] += v
        elif k.startswith("ds_"): g["lds"] += v
        elif k.startswith(("global_", "buffer_", "scratch_")): g["vmem"] += v
        elif k.startswith("s_"): g["salu+ctl"] += v
        else: g["other"] += v
    return dict(g)


def main():
    src = os.path.join(CSRC, sys.argv[1])
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src,
                        "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    for pat in sys.argv[2:]:
        start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), ln))
        ops, blocks, cur = collections.Counter(), [], None
        for ln in lines[start + 1:]:
            t = ln.strip()
            if "s_endpgm" in t:
                break
            if not t or t.startswith(";"):
                continue
            if t.startswith(".LBB") and t.endswith(":"):
                cur = [t[:-1], collections.Counter()]
                blocks.append(cur)
                continue
            if t.startswith("."):
                continue
            op = t.split()[0]
            ops[op] += 1
            if cur:
                cur[1][op] += 1
        print("%s: %d instructions %s" % (pat, sum(ops.values()), group(ops)))
        for name, c in blocks:
            m = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            if m:
                g = group(c)
                valu = g.get("valu", 0) + g.get("valu_packed", 0)
                print("  block %-10s %4d instructions %s  VALU per MFMA %.1f" % (name, sum(c.values()), g, valu / m))
                print("     top: " + ", ".join("%s %d" % kv for kv in c.most_common(12)))


if __name__ == "__main__":
    main()
