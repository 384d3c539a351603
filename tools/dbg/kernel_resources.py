"""VGPR / SGPR / LDS / scratch of every kernel of libmp2gpu.so, read from the gfx950 code objects' metadata (no GPU needed):
    python tools/dbg/kernel_resources.py > profiles/r03/kernel_resources.txt"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "mapreduce-plonky2_amd", "libmp2gpu.so")  # or a variant library
llvm = "/opt/rocm/lib/llvm/bin"
rows = []
with tempfile.TemporaryDirectory() as d:
    shutil.copy(lib, os.path.join(d, "libmp2gpu.so"))
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "libmp2gpu.so"], cwd=d, capture_output=True)
    for f in sorted(x for x in os.listdir(d) if "gfx950" in x):
        txt = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name.replace("mp2g::", "")).split("(")[0].replace("void ", "")
            rows.append((name[:64], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
print(f"{'kernel':64s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'lds':>7s} {'scratch':>7s} {'max_wg':>6s}")
for r in sorted(rows):
    print(f"{r[0]:64s} {r[1]:>5s} {r[2]:>5s} {r[3]:>6s} {r[4]:>6s} {r[5]:>7s} {r[6]:>7s} {r[7]:>6s}")
