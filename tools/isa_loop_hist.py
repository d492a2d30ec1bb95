"""Histogram of the instructions inside the main loop of a kernel (dev tool).
usage: python tools/isa_loop_hist.py <file.s> <kernel-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0] and ':' in l][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith('s_endpgm')][0]
body = lines[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
print("kernel lines", len(body), "loop span", best)
loop = body[best[0]:best[1]]
mb = [i for i, l in enumerate(body) if 'QC_ITER_BEGIN' in l]
me = [i for i, l in enumerate(body) if 'QC_ITER_END' in l]
if mb and me:
    loop = body[mb[0]:me[-1]]
    print("using QC_ITER markers", mb[0], me[-1])
ops = [l.split()[0] for l in loop if re.match(r'^\s+[a-z]', l)]
c = collections.Counter(ops)
print(sum(c.values()), "instructions in loop;", sum(v for k, v in c.items() if k.endswith('_f64') or '_f64_' in k), "f64")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"{v:5d} {k}")
