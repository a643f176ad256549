"""Instruction mix of the loops of a kernel, counted from the compiler's assembly (no GPU needed): how many non-MFMA instructions a
loop body issues per MFMA, which is first-order on this chip.  Found the Winograd address arithmetic (219 -> 167
instructions per chunk) and showed that the same count does not bound heal_conv1x1 or heal_linear.

    python scripts/isa_mix.py heal_amd/csrc/conv3x3.hip k_conv3x3_wino          # substring of the (mangled) kernel name
"""
import os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                        f"-I{ROOT}/heal_amd/csrc", "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, src)], check=True,
                       stderr=subprocess.DEVNULL, cwd=td)
        lines = open(asm).read().split("\n")
    names = [(i, l) for i, l in enumerate(lines) if l.startswith("_Z") and (": ;" in l or l.rstrip().endswith(":"))]
    for start, nm in names:
        if pat not in nm:
            continue
        end = next(i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l)
        body = lines[start:end]
        meta = {k: v for l in lines[end:end + 80] for k, v in [l.strip().split(" ")[:2] if " " in l.strip() else (None, None)]
                if k in (".amdhsa_next_free_vgpr", ".amdhsa_private_segment_fixed_size", ".amdhsa_group_segment_fixed_size")}
        print(nm.split(":")[0][:90], {k.replace(".amdhsa_", ""): v for k, v in meta.items()})
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        seen = set()
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            a = labels[m.group(1)]
            seg = [x.strip() for x in body[a:i + 1] if x.strip() and not x.strip().startswith((".", ";"))]
            c = Counter(x.split()[0] for x in seg)
            mf = sum(v for k, v in c.items() if "mfma" in k)
            if mf and (a, mf) not in seen:
                seen.add((a, mf))
                print(f"  loop @{a}: {len(seg)} instructions, {mf} MFMA, {(len(seg) - mf) / mf:.2f} others per MFMA")
                print("    ", dict(c.most_common(16)))


if __name__ == "__main__":
    main()
