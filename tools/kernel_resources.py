"""Registers, spills, scratch, LDS and waves per SIMD of every kernel of libmdfri_hip.so, from the compiler's own remarks
(`-Rpass-analysis=kernel-resource-usage`): each csrc/*.hip is compiled once more with the Makefile's flags (no GPU needed; gcn.hip takes a
minute or two) and the remarks are folded into one table.  Exit code 1 if any kernel spills vector registers or uses scratch (a spilled
SGPR lives in a lane of a VGPR: no memory traffic; counted in its own column).
    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "metagenomic-deepfri_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"]     # csrc/Makefile: CXXFLAGS
EXTRA = {"cmap.hip": ["-ffp-contract=off"]}                                                     # csrc/Makefile: the contact kernels' rule
KEYS = (("VGPRs", r"\bVGPRs: (\d+)"), ("AGPRs", r"\bAGPRs: (\d+)"), ("SGPRs", r"TotalSGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("LDS", r"LDS Size \[bytes/block\]: (\d+)"), ("waves/SIMD", r"Occupancy \[waves/SIMD\]: (\d+)"))


def resources(src: str) -> dict:
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *EXTRA.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "o.o")]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        if cur is not None:
            for key, pat in KEYS:
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
    return rows


def main() -> int:
    bad = 0
    print("%-92s %s" % ("kernel", " ".join("%10s" % k for k, _ in KEYS)))
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        rows = resources(src)
        if not rows:
            continue
        names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
        print("# %s: %d kernels" % (src, len(rows)))
        for mangled, name in sorted(zip(rows, names), key=lambda t: t[1]):
            r = rows[mangled]
            short = re.sub(r"\(.*", "", name.replace("void ", "").replace("mdf::", ""))
            print("%-92s %s" % (short[:92], " ".join("%10d" % r.get(k, -1) for k, _ in KEYS)))
            bad += bool(r.get("vspill") or r.get("scratch"))
    print("# kernels that spill vector registers or use scratch: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
