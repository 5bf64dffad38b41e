"""Occupancy table of every hand-written kernel, from the code objects themselves: the gfx950 device code of each
csrc/*.o (llvm-objcopy --dump-section .hip_fatbin -> clang-offload-bundler --unbundle -> llvm-readelf --notes).
-> {demangled kernel: vgpr, agpr, sgpr, lds_bytes, scratch_bytes, wg_size, waves_per_simd by registers / LDS / both}
(MI355X_MICROARCH.md: 512 registers per lane per SIMD in granules of 8, 160 KiB LDS per CU, 8 waves per SIMD max)."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.o")
        r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "x.o")],
                           capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []                     # a host-only object (csrc/van_block.hip: C++ orchestration, no device code)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(2), m.group(3).strip()
        if key == "agpr_count" and m.group(1):
            cur = {}
            out.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size",
                                       "private_segment_fixed_size", "max_flat_workgroup_size", "name"):
            cur[key] = val if key == "name" else int(val)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(.*$", "", n).replace("void ", "") for n in p.stdout.splitlines()]


def table():
    rows = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "rs_detection_amd", "csrc", "*.o"))):
        ks = kernels_of(obj)
        for k, name in zip(ks, demangle([k["name"] for k in ks])):
            regs = k["vgpr_count"] + k["agpr_count"]
            alloc = max(8, -(-regs // 8) * 8)
            by_reg = min(8, 512 // alloc)
            wg = k.get("max_flat_workgroup_size", 256)
            waves_wg = -(-wg // 64)
            lds = k["group_segment_fixed_size"]
            wgs_lds = (160 * 1024) // lds if lds else 10 ** 9
            by_lds = min(8, (min(wgs_lds, 32) * waves_wg) // 4) if lds else 8
            rows[name] = dict(file=os.path.basename(obj)[:-2] + ".hip", vgpr=k["vgpr_count"], agpr=k["agpr_count"],
                              sgpr=k["sgpr_count"], lds_bytes=lds, scratch_bytes=k["private_segment_fixed_size"],
                              wg_size=wg, waves_per_simd_by_registers=by_reg, waves_per_simd_by_lds=by_lds,
                              waves_per_simd=min(by_reg, by_lds))
    return rows


if __name__ == "__main__":
    rows = table()
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for n, r in sorted(rows.items(), key=lambda kv: (kv[1]["file"], kv[0])):
        print("%-22s %-58s vgpr %3d sgpr %3d lds %6d scratch %3d  waves/SIMD %d (regs %d, lds %d)" % (
            r["file"], n[-58:], r["vgpr"], r["sgpr"], r["lds_bytes"], r["scratch_bytes"], r["waves_per_simd"],
            r["waves_per_simd_by_registers"], r["waves_per_simd_by_lds"]))
