"""Per-field byte budget of the per-environment LDS object (struct Smem) of one kernel configuration, from clang's record-layout dump of the
device compilation.  Occupancy is LDS-bound: this is the table to look at before adding a field.
Usage: python tools/lds_budget.py [cfg 0..7]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
src = os.path.join(ROOT, "robosuite_amd", "csrc", "rsim_step.hip")
# the -D flags the Makefile builds this configuration with (J / M / contact block in global memory, hull pool) decide the layout
var = {0: "CFG0FLAGS", 1: "CFG1FLAGS", 2: "CFG2FLAGS", 3: "CFG3FLAGS", 5: "CFG3FLAGS", 7: "CFG7FLAGS"}.get(cfg)
flags = subprocess.run(["make", "-s", f"print-{var}"], capture_output=True, text=True, cwd=os.path.dirname(src)).stdout.split() if var else []
defs = [f for f in flags if f.startswith("-D")]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O0", "-std=c++17", "--offload-arch=gfx950", f"-DRSIM_CFG={cfg}", *defs, "--cuda-device-only", "-fsyntax-only",
                      "-Xclang", "-fdump-record-layouts", src], capture_output=True, text=True, cwd=os.path.dirname(src)).stdout
for block in out.split("*** Dumping AST Record Layout"):
    lines = block.strip().splitlines()
    if len(lines) < 3 or not re.match(r"\s*0 \| struct Smem<[\d, ]+>$", lines[0]) or "qpos" not in lines[1]:
        continue
    print(lines[0].split("|")[1].strip())
    top = [(int(m.group(1)), m.group(2).strip()) for l in lines[1:] if (m := re.match(r"\s*(\d+) \|   (\S.*)$", l))]
    total = int(re.search(r"sizeof=(\d+)", block).group(1))
    rows = [(nxt - off, off, name) for (off, name), (nxt, _) in zip(top, top[1:] + [(total, "")])]
    for size, off, name in sorted(rows, reverse=True):
        if size >= 256:
            print(f"  {size:7d} B  @{off:6d}  {name}")
    print(f"  {sum(s for s, _, _ in rows if s < 256):7d} B  (fields below 256 B)")
    print(f"  {total:7d} B  total -> {163840 // total} environments per CU (160 KB)")
    break
