"""Per-kernel VGPR / SGPR / LDS / scratch / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage (the remarks the last
`make -C gl-radix-sort_amd/csrc` left in gl-radix-sort_amd/lib/kernel_resources.log).
usage: python tools/kernel_resources.py [substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = open(os.path.join(ROOT, "gl-radix-sort_amd", "lib", "kernel_resources.log")).read()
sub = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-6s %-6s %-8s %-8s %-5s  %s" % ("VGPR", "SGPR", "scratch", "LDS", "occ", "kernel"))
for r in rows:
    if sub and sub not in r["name"]:
        continue
    name = re.sub(r"\(.*", "", r["name"]).replace("glu_hip::", "").replace("unsigned int", "u32").replace("unsigned long", "u64")
    print("%-6s %-6s %-8s %-8s %-5s  %s" % (r.get("VGPRs", "?"), r.get("SGPRs", "?"), r.get("ScratchSize", "?"),
                                          r.get("LDS Size", "?"), r.get("Occupancy", "?"), name[:110]))
