"""Static VALU-instruction profile of a kernel by source phase, from `hipcc -S -gline-tables-only` output.
Instructions located in headers (inlined math) are attributed to the most recent line of the kernel's own
source files.  usage: isa_profile.py file.s kernel_substring file_id:lo:hi:label ..."""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
buckets = []
for b in sys.argv[3:]:
    fid, lo, hi, label = b.split(":")
    buckets.append((int(fid), int(lo), int(hi), label))
own = {b[0] for b in buckets}
inside = False
cur = (0, 0)
counts = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for line in open(path):
    if not inside:
        if re.match(r"^_Z\S*%s\S*:" % kern, line):
            inside = True
        continue
    if line.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        f, l = int(m.group(1)), int(m.group(2))
        if f in own and l > 0:
            cur = (f, l)
        continue
    m = re.match(r"\s+(v_\w+|ds_\w+|global_\w+|buffer_\w+|s_waitcnt|s_\w+)", line)
    if not m:
        continue
    op = m.group(1)
    label = "other"
    for fid, lo, hi, lab in buckets:
        if cur[0] == fid and lo <= cur[1] <= hi:
            label = lab
            break
    cls = "valu" if op.startswith("v_") else ("lds" if op.startswith("ds_") else ("vmem" if op.startswith(("global_", "buffer_")) else "salu"))
    counts[(label, cls)] += 1
    if cls == "valu":
        kinds[label]["f64" if "f64" in op else ("f32" if "f32" in op else "int/other")] += 1
labels = [b[3] for b in buckets] + ["other"]
seen = []
for lab in labels:
    if lab in seen:
        continue
    seen.append(lab)
    print("%-28s valu %5d (f64 %4d f32 %4d other %4d)  salu %5d  lds %4d  vmem %4d" % (
        lab, counts[(lab, "valu")], kinds[lab]["f64"], kinds[lab]["f32"], kinds[lab]["int/other"],
        counts[(lab, "salu")], counts[(lab, "lds")], counts[(lab, "vmem")]))
print("total valu", sum(v for (l, c), v in counts.items() if c == "valu"))
