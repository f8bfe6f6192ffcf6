#!/usr/bin/env python
"""Static instruction histogram of one kernel in a hipcc -S listing.  Usage: tools/isa_histogram.py <file.s> <kernel-name-substring> [top]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sub in l and ": " in l and l.split(":")[0].startswith("_Z"))][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end")][0]
cnt = collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    cnt[re.split(r"\s+", t)[0]] += 1
total = sum(cnt.values())
print("total", total)
groups = collections.Counter()
for k, v in cnt.items():
    g = k.split("_")[0] if not k.startswith(("global_", "flat_", "scratch_", "buffer_", "ds_")) else k.split("_")[0] + "_" + k.split("_")[1]
    groups[g] += v
print("groups", dict(groups.most_common()))
for k, v in cnt.most_common(top):
    print(f"{v:6d} {k}")
