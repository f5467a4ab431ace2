#!/usr/bin/env python3
"""kernel_resources.py [object or library ...]: per-kernel register / spill / scratch / LDS figures of the gfx950 code
objects inside hipcc outputs (.o, .so), read from the code object's own metadata: objcopy takes the .hip_fatbin section,
clang-offload-bundler the gfx950 bundle, llvm-readelf --notes the kernel descriptors.  Default: every object under
ac_tsr_amd/csrc.  Used by tests/test_abi_cpu.py (which streaming-forward instantiations spill) and for
profiles/r04_kernel_resources.txt.  Measurement / build helper, not product."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def code_object_kernels(path):
    """[{name, vgpr_count, ...}] for every kernel of the gfx950 bundle inside `path`."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        # a library holds one bundle per translation unit, back to back: split at the magic strings
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__|CCOB", blob)]
        starts = [s for k, s in enumerate(starts) if k == 0 or blob[starts[k - 1]:starts[k - 1] + 4] != b"CCOB" or True]
        out = []
        seen = set()
        for k, s in enumerate(starts):
            e = starts[k + 1] if k + 1 < len(starts) else len(blob)
            piece = os.path.join(d, "piece.bin")
            open(piece, "wb").write(blob[s:e])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + piece, "--output=" + co],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            os.remove(co)
            cur = None
            for ln in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key == "agpr_count":  # first field of a kernel entry (alphabetical YAML)
                    cur = {"agpr_count": int(val)}
                    out.append(cur)
                elif cur is not None and key in FIELDS:
                    cur[key] = int(val)
                elif cur is not None and key == "name":
                    cur["name"] = val
            out = [k_ for k_ in out if "name" in k_]
        uniq = []
        for k_ in out:
            if k_["name"] not in seen:
                seen.add(k_["name"])
                uniq.append(k_)
        return uniq


def main():
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "ac_tsr_amd", "csrc", "*.o")))
    for p in paths:
        ks = code_object_kernels(p)
        if not ks:
            continue
        names = demangle([k["name"] for k in ks])
        print(f"== {os.path.relpath(p, ROOT)}: {len(ks)} kernels")
        print(f"{'vgpr':>5} {'agpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}  kernel")
        for k in ks:
            n = names[k["name"]].replace("(anonymous namespace)::", "").split("(")[0]
            print(f"{k.get('vgpr_count', 0):5d} {k.get('agpr_count', 0):5d} {k.get('vgpr_spill_count', 0):6d} "
                  f"{k.get('sgpr_spill_count', 0):6d} {k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):6d}  {n}")


if __name__ == "__main__":
    main()
