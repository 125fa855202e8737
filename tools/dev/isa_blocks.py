"""dev: basic blocks of one kernel in the device assembly (instruction mix per block):
   hipcc ... --cuda-device-only -S -o /tmp/dis/siftmi.s sift_pyocl_amd/csrc/siftmi.hip ; python tools/dev/isa_blocks.py /tmp/dis/siftmi.s extrema_kernel [min_len]"""
import sys, collections
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]; min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 25
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks = []; cur = ["entry", []]; total = collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";"): continue
    if t.startswith(".LBB") and t.split()[0].endswith(":"):
        blocks.append(cur); cur = [t.split()[0], []]; continue
    if t.startswith("."): continue
    op = t.split()[0]; cur[1].append(op); total[op] += 1
blocks.append(cur)
print("total", sum(total.values()))
for name, ins in blocks:
    if len(ins) >= min_len: print(name, len(ins), collections.Counter(ins).most_common(10))
