"""Instruction mix per basic block of one kernel in a hipcc --save-temps .s file.
Usage: python tools/isa_mix.py <file.s> <kernel-name-substring>"""
import collections, re, sys

s = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\w+):", s, re.M)
name = next(n for n in names if sys.argv[2] in n)
start = s.index(name + ":")
end = s.index(".Lfunc_end", start)
blocks, cur = [], ["<entry>"]
blocks.append(cur)
for ln in s[start:end].splitlines()[1:]:
    t = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = [t.split(":")[0]]
        blocks.append(cur)
    elif t and not t.startswith((".", ";")):
        cur.append(t)
tot = collections.Counter()
for b in blocks:
    c = collections.Counter()
    for i in b[1:]:
        op = i.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "trans" if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")) else
             "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
             "wait" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else "salu" if op.startswith("s_") else "other")
        c[k] += 1
    tot.update(c)
    print(f"{b[0]:12s} {len(b) - 1:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", dict(tot))
