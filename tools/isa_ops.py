"""Print the memory / MFMA / wait skeleton of one kernel from a hipcc --save-temps .s file (development aid).

    python tools/isa_ops.py file.s <substring of the mangled name>
"""
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [l.split(":")[0] for l in s.splitlines() if pat in l and ":" in l and not l.startswith((".", "\t", " ", ";"))]
name = names[0]
i = s.index("\n" + name + ":")
j = s.index("s_endpgm", i)
n = 0
for l in s[i:j].splitlines():
    t = l.strip()
    if not l.startswith("\t") or not t or t.startswith((".", ";")):
        continue
    n += 1
    op = t.split()[0]
    if op.startswith(("s_waitcnt", "global_load", "ds_", "v_mfma", "s_barrier", "s_cbranch", "buffer_", "global_store", "s_branch", "scratch")):
        print(t.split(";")[0].strip())
print("instructions:", n, "kernel:", name, file=sys.stderr)
