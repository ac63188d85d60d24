"""tools/asm_hotloop.py FILE.s KERNEL_SUBSTRING -- per basic block of one kernel: LDS 16-byte reads, packed adds, scratch (spill) ops.
Used to check that register spills stay out of the walk's slice loop.  FILE.s from
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o FILE.s csrc/lsq_icmq.hip"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = [l.split(":")[0], []]
    else:
        cur[1].append(l)
blocks.append(cur)
tot = 0
for name, ins in blocks:
    nds = sum("ds_read_b128" in x for x in ins)
    nsc = sum("scratch_" in x for x in ins)
    npk = sum("v_pk_add_u16" in x for x in ins)
    tot += nsc
    if nds >= 4 or nsc:
        print("%-12s %5d instr  ds_read_b128 %3d  v_pk_add_u16 %3d  scratch %3d" % (name, len(ins), nds, npk, nsc))
print("total scratch ops", tot)
