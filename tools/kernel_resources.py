"""Resource usage of every kernel in robosuite_amd/librsim_hip.so, read from the code objects embedded in the library (no GPU, no rebuild):
LDS bytes per workgroup, private-segment (scratch) bytes, VGPR / AGPR / SGPR counts.  The occupancy of the fused kernel is LDS-bound
(DESIGN.md section 5), so these numbers are the budget the kernel configurations are designed to: tests/test_kernel_resources.py pins them.
Usage: python tools/kernel_resources.py [path/to/librsim_hip.so]"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LDS_GRANULE = 1280   # bytes; 160 KB per CU = 128 granules


def code_objects(lib):
    """gfx950 ELF code objects of the clang offload bundles in the library's .hip_fatbin section (one bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "copy.so")])
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[base + off:base + off + size])
    return out


# mangled template arguments of k_step<NB, NJ, NV, NG, NS, NCON, NEFC, NPAIR> of the configuration that serves each bench.py --config
CONFIG_TAG = {"lift": "ILi32ELi16ELi16ELi24ELi16ELi16ELi64ELi192E", "stack": "ILi32ELi16ELi32ELi24ELi16ELi32ELi64ELi192E",
              "peg": "ILi64ELi16ELi16ELi32ELi32ELi32ELi64ELi320E", "pickplace": "ILi64ELi32ELi48ELi64ELi32ELi32ELi128ELi640E"}


def config_code_sha16(lib, config):
    """sha256[:16] of the machine code (.text: every kernel of the translation unit; .rodata: their kernel descriptors) of the gfx950 code object that holds the fused
    kernel of a bench configuration.  PMC evidence under profiles/ carries it next to the library's sha: a build whose machine code for the configuration is
    bit-identical runs the very kernel that was measured, whatever changed in the other configurations of the library (bench.py pmc_evidence).  Not the whole code
    object: it also carries a compilation-unit id hashed from the compiler's command line (output path included), which differs between `make` and a variant build of
    the same source and flags (tools/sessions/r05_s13_prep.sh)."""
    import hashlib
    tag = ("_Z6k_step" + CONFIG_TAG[config]).encode()
    hits = [co for co in code_objects(lib) if tag in co]
    if len(hits) != 1:
        return None
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "a.co")
        open(f, "wb").write(hits[0])
        code = b""
        for sec in (".text", ".rodata"):
            o = os.path.join(td, "sec.bin")
            subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", f"--only-section={sec}", f, o])
            code += open(o, "rb").read()
    return hashlib.sha256(code).hexdigest()[:16]


# the configuration whose own kernel (k_step_list) steps the envs of a bench configuration that outgrow its capacity; lift: the tier is a second body inside the
# native code object (round 6), so the native sha covers it
WIDE_TAG = {"stack": "ILi32ELi16ELi32ELi24ELi16ELi32ELi128ELi192E", "peg": "ILi64ELi32ELi48ELi64ELi32ELi32ELi128ELi640E", "pickplace": "ILi64ELi32ELi48ELi64ELi32ELi64ELi256ELi640E"}


def wide_code_sha16(lib, config):
    """sha256[:16] of the machine code of the code object that holds the capacity tier's kernel of a bench configuration ("same-object" for lift)."""
    import hashlib
    if config not in WIDE_TAG:
        return "same-object"
    tag = ("_Z11k_step_list" + WIDE_TAG[config]).encode()
    hits = [co for co in code_objects(lib) if tag in co]
    if len(hits) != 1:
        return None
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "a.co")
        open(f, "wb").write(hits[0])
        code = b""
        for sec in (".text", ".rodata"):
            o = os.path.join(td, "sec.bin")
            subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", f"--only-section={sec}", f, o])
            code += open(o, "rb").read()
    return hashlib.sha256(code).hexdigest()[:16]


def kernels(lib=None):
    """{kernel name: {lds, scratch, vgpr, agpr, sgpr}} over all code objects of the library."""
    lib = lib or os.path.join(ROOT, "robosuite_amd", "librsim_hip.so")
    res = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], text=True)
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(agpr_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_count|vgpr_count|name):\s*(\S+)", line)
            if not m:
                continue
            key, val = m.groups()
            if key == "name" and not val.startswith("_Z") and not val.startswith("k_"):
                continue   # argument names
            cur[key] = val
            if all(k in cur for k in ("name", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count", "sgpr_count")):
                res[cur["name"]] = dict(lds=int(cur["group_segment_fixed_size"]), scratch=int(cur["private_segment_fixed_size"]),
                                        vgpr=int(cur["vgpr_count"]), agpr=int(cur.get("agpr_count", 0)), sgpr=int(cur["sgpr_count"]))
                cur = {}
    return res


def residency(r):
    """One-wavefront workgroups per CU: (by LDS, by registers, the smaller of the two and the 32-wave cap).  Registers: the unified VGPR + AGPR file
    is allocated in granules of 8 per lane, 512 per SIMD, four SIMDs per CU (MI355X_MICROARCH.md, register files); `vgpr` of the notes is the total."""
    by_lds = 128 // (-(-r["lds"] // LDS_GRANULE)) if r["lds"] else 32   # LDS is allocated in granules of 1280 B, 128 of them per CU (measured, round 6: a 32 084-B workgroup is admitted four times, a 31 892-B one five times)
    alloc = -(-max(1, r["vgpr"]) // 8) * 8
    by_reg = 4 * min(8, 512 // alloc)
    return by_lds, by_reg, min(by_lds, by_reg, 32)


if __name__ == "__main__":
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else None)
    for name, r in sorted(ks.items()):
        bl, br, n = residency(r)
        print(f"{name[:70]:70s} LDS {r['lds']:6d} B  scratch {r['scratch']:4d} B  VGPR+AGPR {r['vgpr']:3d} (AGPR {r['agpr']:3d}) SGPR {r['sgpr']:3d}"
              f"  -> workgroups / CU: {bl} by LDS, {br} by registers = {n}")
